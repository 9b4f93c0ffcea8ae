#!/usr/bin/env python3
"""bench.py -- residues/s of the RoseTTAFold forward path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|1|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one forward of the full model (config 2 = BASELINE.json configs[1]: B=4 MSAs of N=128 x L=256 per GPU,
d_msa=384, d_pair=288, 8 two-track + 4 three-track + final block, 4 encoder layers) on synthetic token inputs already
resident in HBM, random-init weights.  Independent MSAs shard across ranks (one process per GPU); the only collective is
the gather of the results to rank 0 (RCCL), inside the timed region.  Rank 0 prints ONE JSON line.  Extra legs on rank 0
at N=1:
  * roofline: the dominant kernel family timed live with HIP events around every rf_gemm launch of one extra profiled
    step (the library reports which kernel family each launch took: rf_gemm_last_family); achieved = algorithmic FLOPs of
    those launches / their time.  `traffic` comes from the PMC file of THIS tree (profiles/r04_traffic_pmc.json carries a
    hash of csrc/); a file collected on another tree is reported as stale and `traffic` stays null.
  * parity: the SAME inputs in every compute mode of the library -- the timed bf16 path, the fp16-operand build of the
    same kernels (librfmi_f16.so) and the exact-fp32 mode (pinned to the CPU oracle at depth, tests/test_depth_gpu.py) --
    each with its step time; agreement with the fp32 mode: distogram argmax (all pairs / clear-margin pairs), relative L2
    of the four logit maps / xyz / plddt.
  * cpu_baseline: the CPU oracle (oracle/rf_oracle.py, pinned to the reference's golden vectors) on the host cores: one
    layer of each kind at the bench shapes (B=1), warm-up + best of 3, scaled by the layer counts.
Other workloads: --config 4 (BASELINE.json configs[3]: B=1, N=64, L=1024) and --config 5 (configs[4]: the SE(3) structure
module alone, B=8, L=256, k=128) print a JSON line of the same form with their own `roofline` (dominant rf_gemm family of
that workload, PMC files profiles/r04_config{4,5}_*); no parity / cpu legs.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL = dict(d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=8, n_three_track_blocks=5,
            n_encoder_layers=4, n_neighbors=[128, 128, 64, 64, 64])
CONFIGS = {
    # BASELINE.json configs[0] with the constructible d_msa: plumbing case
    1: dict(B=1, N=8, L=64, model=dict(d_msa=96, d_pair=64, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1,
                                       n_three_track_blocks=2, n_encoder_layers=1, max_len=64,
                                       n_neighbors=[128, 128])),
    # configs[1]: the configuration the metric is quoted on (README hyper-parameters of the reference)
    2: dict(B=4, N=128, L=256, model=dict(FULL, max_len=260)),
    # configs[3]: long-sequence stress
    4: dict(B=1, N=64, L=1024, model=dict(FULL, max_len=1030)),
    # configs[4]: SE(3) structure-module-only microbench (CoordUpdateWithMsaAndPair, k = 128)
    5: dict(B=8, N=16, L=256, model=dict(FULL, max_len=260)),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
FAMILY = {0: "gemm_f32_kernel (fp32 MFMA 16x16x4)", 1: "gemm_bf16_kernel (MFMA 16x16x32)",
          2: "conv3x3_c288_kernel (3x3 convolution: halo-tile kernel; other channel counts take the implicit-GEMM tile kernel)", 3: "gemm_fast_kernel (persistent tiles, MFMA 16x16x32)",
          4: "gemm_wreg_kernel (register-resident weights, MFMA 16x16x32)",
          5: "ffn_fused_kernel (one-launch feed-forward, MFMA 16x16x32)"}


def tree_hash():
    """Identifies the kernel sources a PMC collection belongs to."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "rosettafold-pytorch_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def make_inputs(B, N, L, seed, device):
    g = torch.Generator().manual_seed(seed)
    msa = torch.randint(0, 21, (B, N, L), generator=g)
    seq = msa[:, 0].clone()
    aa_idx = torch.arange(L).unsqueeze(0).repeat(B, 1)
    return msa.to(device), seq.to(device), aa_idx.to(device)


def profile_gemms(run):
    """One extra forward with HIP events around every rf_gemm launch (same stream the kernels run on)."""
    from rosettafold_pytorch_amd import ops
    recs = []
    orig = ops.lib.rf_gemm
    fam_of = ops.lib.rf_gemm_last_family

    def wrapped(desc, stream):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = orig(desc, stream)
        e.record()
        d = desc._obj
        nb = max(d.nb0, 1) * max(d.nb1, 1) * max(d.nb2, 1)
        esz = 2 if d.ab_dtype == 1 else 4
        k_eff = d.K // 9 if d.a_mode == 1 else d.K  # conv: the image is read once algorithmically
        nbytes = nb * (d.M * k_eff * esz + d.N * d.K * esz + d.M * d.N * (4 if d.c_dtype == 0 else 2)
                       + (d.M * d.N * 4 if d.residual else 0) + (d.M * d.N * 2 if d.ln_out else 0))  # + the fused LayerNorm copy
        recs.append((s, e, 2.0 * d.M * d.N * d.K * nb, int(fam_of()), d.M, d.N, d.K, nb, nbytes))
        return rc

    # the one-launch feed-forward (rf_ffn_fused, csrc/ffn.hip) is the path's other dense-contraction entry point: family 5.
    # Algorithmic bytes: input rows + both weight matrices once + fp32 residual read and write + the fused LayerNorm copy.
    orig_ffn = ops.lib.rf_ffn_fused

    def wrapped_ffn(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = orig_ffn(*a)
        e.record()
        M, D, hidden = int(a[14]), int(a[15]), int(a[16])
        nbytes = M * D * 2 + 2 * D * hidden * 2 + 2 * M * D * 4 + (M * D * 2 if a[9] else 0)
        recs.append((s, e, 4.0 * M * D * hidden, 5, M, D, hidden, 1, nbytes))
        return rc

    ops.lib.rf_gemm = wrapped
    ops.lib.rf_ffn_fused = wrapped_ffn
    try:
        run()
        torch.cuda.synchronize()
    finally:
        ops.lib.rf_gemm = orig
        ops.lib.rf_ffn_fused = orig_ffn
    tab = {}
    for s, e, fl, fam, M, N, K, nb, nbytes in recs:
        t = tab.setdefault((fam, M, N, K, nb), [0.0, 0.0, 0, 0.0])
        t[0] += s.elapsed_time(e) * 1e-3
        t[1] += fl
        t[2] += 1
        t[3] += nbytes
    shapes = []
    for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])[:12]:
        shapes.append({"kernel": FAMILY.get(k[0], str(k[0])).split(" ")[0], "M": k[1], "N": k[2], "K": k[3], "batch": k[4],
                       "launches": v[2], "avg_launch_ms": 1e3 * v[0] / v[2], **bound_of(v[1], v[3], v[0])})
    if os.environ.get("RF_GEMM_TABLE"):
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])[:40]:
            log("gemm family=%d M=%d N=%d K=%d batch=%d : %d calls %.2f ms total, %.0f TF/s" % (*k, v[2], 1e3 * v[0], v[1] / v[0] / 1e12))
    fams = {}
    for s, e, fl, fam, M, N, K, nb, nbytes in recs:
        f = fams.setdefault(FAMILY.get(fam, f"family {fam}"), [0.0, 0.0, 0, 0.0])
        f[0] += s.elapsed_time(e) * 1e-3
        f[1] += fl
        f[2] += 1
        f[3] += nbytes
    return fams, shapes


HBM_PEAK_GBPS = 8000.0


def bound_of(flops, nbytes, secs):
    """Which roof an (algorithmic flops, algorithmic bytes) pair sits under on MI355X: the ridge is peak_flops / peak_bw
    = 2.5e15 / 8e12 = 312 FLOP/byte; below it the HBM roof is the lower one."""
    tf, gbps = flops / secs / 1e12, nbytes / secs / 1e9
    if flops / max(nbytes, 1.0) < 1e3 * MFMA_BF16_PEAK_TFLOPS / HBM_PEAK_GBPS:
        return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "flop_per_byte": flops / max(nbytes, 1.0), "tflops": tf}
    return {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
            "flop_per_byte": flops / max(nbytes, 1.0), "algorithmic_GBps": gbps}


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _agreement(lb, xb, pb, lg, xyz, pl):
    agree = {k: (lb[k].argmax(-1) == lg[k].argmax(-1)).float().mean().item() for k in lg}
    # margin-aware view: bins whose top-2 fp32 logits are further apart than 2 % of the map's range cannot flip by rounding
    d = lg["dist"]
    top2 = d.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.02 * (d.max() - d.min())
    agree_clear = (lb["dist"].argmax(-1) == d.argmax(-1))[clear].float().mean().item() if clear.any() else None
    return {"dist_argmax_agreement": agree["dist"], "argmax_agreement": agree,
            "dist_argmax_agreement_clear_margin": agree_clear, "clear_margin_fraction": clear.float().mean().item(),
            "rel_l2": {**{k: rel_l2(lb[k], lg[k]) for k in lg}, "xyz": rel_l2(xb, xyz), "plddt": rel_l2(pb, pl)}}


def _time_mode(model, inputs, R, dtype, steps, graph=False):
    """One compute mode timed like the headline number: eager launches, or (graph=True) replays of one hipGraph."""
    R.set_compute_dtype(dtype)
    try:
        out = model(*inputs)  # warm-up (weight copies of this mode)
        torch.cuda.synchronize()
        run = lambda: model(*inputs)  # noqa: E731
        if graph:
            try:
                g = R.GraphedForward(model, *inputs)
                run = lambda: g(*inputs)  # noqa: E731
            except Exception as e:  # noqa: BLE001
                log(f"hipGraph capture failed in {dtype} mode ({type(e).__name__}: {e}); eager launches")
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        return (dict(out[0]), out[1].clone(), out[2].clone()), ms
    finally:
        R.set_compute_dtype(torch.bfloat16)


def parity_block(model, inputs, out_bf16, R, ms_bf16, steps=2, graph=False):
    """Every compute mode of the library on the SAME inputs and weights, against the exact-fp32 mode (the mode that
    tests/test_depth_gpu.py and tests/test_config2_gpu.py pin to the CPU oracle at 5e-4 / 2e-5): the timed bf16 path, and
    the fp16-operand build of the same kernels (librfmi_f16.so: same MFMA rate and bytes, 8x smaller operand rounding),
    each with its own step time.  Top-level fields describe the timed (bf16) path."""
    (lg, xyz, pl), ms32 = _time_mode(model, inputs, R, torch.float32, steps)
    (l16, x16, p16), ms16 = _time_mode(model, inputs, R, torch.float16, max(steps, 3), graph)
    lb, xb, pb = out_bf16
    bf = _agreement(lb, xb, pb, lg, xyz, pl)
    fp = _agreement(l16, x16, p16, lg, xyz, pl)
    # yardstick (CPU only, tools/oracle_sensitivity.py): the ORACLE against itself at this depth with nothing changed but its weights
    # rounded to the 16-bit type -- what no 16-bit-operand mode can beat; and the fp32 mode against the oracle at the full depth
    yard = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r04_oracle_sensitivity.json")) as fh:
            sj = json.load(fh)
        for name in ("fp16", "bf16"):
            r = sj[f"weights_rounded_{name}"]
            yard[name] = {"dist_argmax_agreement": r["dist_argmax_agreement"], "rel_l2": r["rel_l2"]}
        yard["source"] = "profiles/r04_oracle_sensitivity.json (B=1, same weights / depth; oracle vs oracle with weights rounded to the type)"
        with open(os.path.join(ROOT, "profiles", "r04_depth_parity_oracle_full.json")) as fh:
            dj = json.load(fh)
        yard["fp32_mode_vs_oracle_full_depth"] = {"dist_argmax_agreement": dj["fp32"]["dist_argmax_agreement"], "rel_l2": dj["fp32"]["rel_l2"],
                                                  "source": "profiles/r04_depth_parity_oracle_full.json (tools/depth_parity.py --oracle --full)"}
    except (OSError, KeyError, ValueError):
        pass
    return {"reference_mode": "exact fp32 kernels of the same library (pinned to the CPU oracle at depth: tests/test_depth_gpu.py)",
            "oracle_weight_rounding_yardstick": yard,
            **bf, "fp32_mode_ms_per_step": ms32,
            "modes": {"bf16": {**bf, "ms_per_step": ms_bf16, "library": "librfmi.so (v_mfma_f32_16x16x32_bf16)"},
                      "fp16": {**fp, "ms_per_step": ms16, "vs_bf16_step_time": ms16 / ms_bf16,
                               "library": "librfmi_f16.so (v_mfma_f32_16x16x32_f16)"},
                      "fp32": {"ms_per_step": ms32, "library": "librfmi.so (v_mfma_f32_16x16x4_f32)"}}}


def cpu_baseline(cfg):
    """Oracle timed on the host cores, bounded: ONE layer of each kind at the bench shapes with B=1 (every block repeats
    the same layers), warm-up + best of 3, scaled by the layer counts of the full model."""
    from oracle import rf_oracle as O
    import rosettafold_pytorch_amd as R
    mc, N, L = cfg["model"], cfg["N"], cfg["L"]
    # the GPU box grants a 16-CPU share per GPU: more threads than that only oversubscribe
    ncores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1),
                 int(os.environ.get("RF_CPU_THREADS", "16")))
    torch.set_num_threads(ncores)
    torch.manual_seed(1234)
    small = dict(mc, n_two_track_blocks=1, n_three_track_blocks=2, n_encoder_layers=1, p_dropout=0.0)
    model = R.RoseTTAFold(**small)
    P = {k: v.detach().float() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    msa = torch.randn(1, N, L, mc["d_msa"], generator=g)
    pair = torch.randn(1, L, L, mc["d_pair"], generator=g)
    att = torch.rand(1, L, L, 12, generator=g)
    seq = torch.randint(0, 21, (1, L), generator=g)
    onehot = torch.nn.functional.one_hot(seq, 21).float()
    aa = torch.arange(L).unsqueeze(0)
    steps = torch.randn(1, L, 3, generator=g)  # protein-like CA trace: non-degenerate kNN graph / distance masks
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
    xyz = ca[:, :, None, :] + 0.5 * torch.randn(1, L, 3, 3, generator=g)
    state = torch.randn(1, L, mc["d_state"], generator=g)
    tb, t3 = "two_track_blocks.0", "three_track_blocks.0"
    pieces = {
        "msa_row_layer": lambda: O.encoder_layer_tied(P, tb + ".msa_update_using_self_att.residue_wise_encoder_layers.0", msa, 12),
        "msa_col_layer": lambda: O.encoder_layer_performer(P, tb + ".msa_update_using_self_att.sequence_wise_encoder_layers.0", msa.transpose(1, 2), 12),
        "pair_update_with_msa": lambda: O.pair_update_with_msa(P, tb + ".pair_update_with_msa", msa, pair, att),
        "pair_axial_layer": lambda: O.pair_axial_layer(P, tb + ".pair_update_with_axial_attention.layers.0", pair),
        "msa_with_pair_layer": lambda: O.msa_update_with_pair_layer(P, tb + ".msa_update_with_pair.encoder_layers.0", msa, pair, 4),
        "init_coord": lambda: O.initial_coord_generation(P, "initial_coord_generation_with_msa_and_pair", msa, pair, onehot, aa),
        "coord_update": lambda: O.coord_update(P, t3 + ".coord_update_with_msa_and_pair", xyz, msa, pair, aa, onehot, mc["n_neighbors"][0], mc["d_state"]),
        "msa_with_coord": lambda: O.msa_update_with_pair_and_coord(P, t3 + ".msa_update_with_pair_and_coord", xyz, state, msa),
        "head": lambda: O.prediction_head(P, "prediction_head", pair),
    }
    t = {}
    with torch.no_grad():
        for name, fn in pieces.items():
            fn()  # warm-up (allocator, MKL thread pools)
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                best = min(best, time.perf_counter() - t0)
            t[name] = best
            log(f"cpu oracle {name}: best of 3 {t[name]:.2f}s")
    n2, n3, ne = mc["n_two_track_blocks"], mc["n_three_track_blocks"], mc["n_encoder_layers"]
    nb = n2 + n3
    full = (nb * (ne * (t["msa_row_layer"] + t["msa_col_layer"] + t["pair_axial_layer"] + t["msa_with_pair_layer"])
                  + t["pair_update_with_msa"]) + t["init_coord"] + n3 * t["coord_update"] + (n3 - 1) * t["msa_with_coord"]
            + t["head"])
    return {"value": L / full, "unit": "residues/s", "cores": ncores, "kind": "port",
            "sample": "B=1,N=%d,L=%d, one layer of each kind, warm-up + best of 3 (%s) = %.1fs of CPU work per pass; scaled by "
                      "the layer counts of the %d+%d-block model -> %.0fs/sample" % (
                          N, L, ", ".join(f"{k} {v:.2f}s" for k, v in t.items()), sum(t.values()), n2, n3, full)}


def se3_inputs(cfg, dev, seed):
    B, N, L, mc = cfg["B"], cfg["N"], cfg["L"], cfg["model"]
    g = torch.Generator().manual_seed(seed)
    steps = torch.randn(B, L, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
    xyz = ca[:, :, None, :] + 0.5 * torch.randn(B, L, 3, 3, generator=g)
    xyz[:, :, 1] = ca
    msa = torch.randn(B, N, L, mc["d_msa"], generator=g)
    pair = torch.randn(B, L, L, mc["d_pair"], generator=g)
    seq = torch.randint(0, 21, (B, L), generator=g)
    oh = torch.nn.functional.one_hot(seq, 21).float()
    aa = torch.arange(L).unsqueeze(0).repeat(B, 1)
    return [t.to(dev) for t in (xyz, msa, pair, aa, oh)]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process (never an exec: this process
    may not be replaced once a GPU library is loaded) and return its exit code.  The children's stdout (rank 0's one JSON
    line) and stderr pass straight through."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *argv]
    log("starting %d ranks: %s" % (n, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--no-graph", action="store_true", help="issue the forward's launches eagerly instead of replaying one hipGraph")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="one-GPU rehearsal of the multi-GPU code path: with WORLD_SIZE=1 still create the process group and run the "
                         "barrier / gather / all-reduce of every step through it (RCCL one-rank group; collectives forced)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU call): every rank joins the group, rank 0 prints {n_gpus, ranks_seen}; "
                         "the CPU test of the --gpus N launcher (tests/test_bench_launcher.py)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: nothing has touched the GPU yet, so start the N ranks as a child job (one process
        # per GPU, torch.distributed.run -> RCCL), relay rank 0's JSON line and leave with the child's exit code
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        ap.error(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or unset WORLD_SIZE and "
                 f"let bench.py start the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29743")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.zeros(world, dtype=torch.int64)
        seen[rank] = 1
        dist.all_reduce(seen)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(seen.sum()), "local_rank": local}), flush=True)
        dist.destroy_process_group()
        return
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)  # before the process group: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local)
    dist_on = world > 1 or args.rehearse_dist
    if args.rehearse_dist:
        os.environ["RF_SHARD_FORCE_COLLECTIVES"] = "1"  # (read when rosettafold_pytorch_amd.shard is imported, below)
        os.environ.setdefault("MASTER_PORT", "29741")
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    import rosettafold_pytorch_amd as R
    cfg = CONFIGS[args.config]
    R.set_compute_dtype({"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype])
    torch.manual_seed(1234)  # identical weights on every rank
    B, N, L = cfg["B"], cfg["N"], cfg["L"]
    used_graph = False
    if args.config == 5:
        mc = cfg["model"]
        model = R.CoordUpdateWithMsaAndPair(mc["d_msa"], mc["d_pair"], mc["d_node"], mc["d_edge"], mc["d_state"],
                                            n_neighbors=128, p_dropout=0.0).to(dev)
        inputs = se3_inputs(cfg, dev, seed=rank)
        run = lambda: model.run(*inputs)  # noqa: E731
    else:
        model = R.RoseTTAFold(p_dropout=0.0, **cfg["model"]).to(dev)
        inputs = make_inputs(B, N, L, seed=rank, device=dev)  # independent MSAs per rank
        run = lambda: model(*inputs)  # noqa: E731
        eager = run
        if not args.no_graph:
            # the whole forward as ONE hipGraph (rosettafold-pytorch_amd/graph.py): every step still validates its inputs and
            # copies them into the graph's static tensors before the replay.  Falls back to eager launches if capture fails.
            try:
                graphed = R.GraphedForward(model, *inputs)
                run = lambda: graphed(*inputs)  # noqa: E731
                used_graph = True
            except Exception as e:  # noqa: BLE001
                log(f"hipGraph capture failed ({type(e).__name__}: {e}); eager launches")

    def step():
        out = run()
        if dist_on and args.config != 5:  # the one collective of the path: gather the results on rank 0 (RCCL over xGMI)
            from rosettafold_pytorch_amd import shard
            shard.gather_results(out[0], out[1], out[2], dst=0)
        return out

    def fence():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            log(f"warmup step {i} done")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if rank == 0:
        log(f"{args.steps} timed steps: {dt:.3f}s")
    if dist_on:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    if rank == 0:
        mc = cfg["model"]
        if args.config == 5:
            workload = (f"BASELINE.json configs[4]: SE(3) structure module only (CoordUpdateWithMsaAndPair), bsz={B}/GPU, L={L}, "
                        f"n_neighbors=128, d_node=d_edge=d_state=32")
        else:
            workload = (f"BASELINE.json configs[{args.config - 1}]: bsz={B}/GPU, n_seq={N}, L={L}, d_msa={mc['d_msa']}, "
                        f"d_pair={mc['d_pair']}, {mc['n_two_track_blocks']}+{mc['n_three_track_blocks']} blocks, "
                        f"{mc['n_encoder_layers']} encoder layers, random-init weights")
        res = {
            "metric": f"residues/sec forward (L={L}, N={N})", "value": world * B * L * args.steps / dt,
            "unit": "residues/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * B,
                       "parallelism": f"batch-sharded x{world} (replicated weights)",
                       "launch": "one hipGraph replay per step (validation + input copy outside the graph)" if used_graph
                       else "eager launches through the C ABI"},
        }
        full = args.config == 2 and args.dtype == "bf16"
        pmc_tag = "r04_" if args.config == 2 else f"r04_config{args.config}_"
        if world == 1 and args.dtype == "bf16" and args.config in (2, 4, 5) and not args.no_roofline:
            fams, shapes = profile_gemms(eager if args.config != 5 else run)
            name, (secs, flops, n, nbytes) = max(fams.items(), key=lambda kv: kv[1][0])
            traffic, tnote = None, "no PMC collection for this tree (tools/pmc_traffic.py)"
            try:
                with open(os.path.join(ROOT, "profiles", pmc_tag + "traffic_pmc.json")) as fh:
                    pm = json.load(fh)
                if pm.get("tree") == tree_hash():
                    traffic = pm["families"][name.split(" ")[0]]["hbm_bytes_per_launch"]
                    tnote = f"rocprofv3 PMC passes (FETCH_SIZE x2, WRITE_SIZE) on this tree ({pm['tree']})"
                else:
                    tnote = f"stale: profiles/{pmc_tag}traffic_pmc.json was collected on tree {pm.get('tree')}, this is {tree_hash()}"
            except (OSError, KeyError, ValueError):
                pass
            mfma_util = None  # matrix-pipe busy fraction from the hardware counters (tools/pmc_mfma.py), same staleness rule
            try:
                with open(os.path.join(ROOT, "profiles", pmc_tag + "mfma_pmc.json")) as fh:
                    pm = json.load(fh)
                if pm.get("tree") == tree_hash():
                    mfma_util = {k: v["mfma_util"] for k, v in pm["families"].items() if v.get("mfma_util")}
            except (OSError, KeyError, ValueError):
                pass
            # the roof the dominant kernel sits under is chosen by its arithmetic intensity (algorithmic FLOP per algorithmic
            # byte against the 312 FLOP/byte ridge); the other roof's fraction is reported beside it
            b = bound_of(flops, nbytes, secs)
            res["roofline"] = {"bound": b["bound"], "kernel": name, "achieved": b["achieved"], "peak": b["peak"],
                               "unit": b["unit"], "frac": b["frac"], "traffic": traffic,
                               "traffic_source": tnote, "launches": n, "avg_launch_ms": 1e3 * secs / n,
                               "flop_per_byte": b["flop_per_byte"],
                               "algorithmic_gflop_per_launch": flops / n / 1e9,
                               "algorithmic_bytes_per_launch": nbytes / n,
                               "mfma_TFLOPs": flops / secs / 1e12, "mfma_frac": flops / secs / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                               "algorithmic_hbm_GBps": nbytes / secs / 1e9, "hbm_frac": nbytes / secs / 1e9 / HBM_PEAK_GBPS,
                               "share_of_step": secs / (dt / args.steps), "shapes": shapes,
                               "mfma_util_pmc": mfma_util,
                               "families": {k: {"s": v[0], "tflops": v[1] / max(v[0], 1e-12) / 1e12, "n": v[2],
                                                "algorithmic_GBps": v[3] / max(v[0], 1e-12) / 1e9}
                                            for k, v in fams.items()}}
        if world == 1 and full and not args.no_parity:
            res["parity"] = parity_block(model, inputs, out, R, 1e3 * dt / args.steps, graph=used_graph)
        if world == 1 and full and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(res), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
