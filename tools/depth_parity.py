"""Error growth through the depth of the model (GPU box): where does the whole-model gap of a 16-bit mode come from?

    python tools/depth_parity.py                 # config-2 dims, B=1, full 8+5 depth: every 16-bit mode against the
                                                 # exact-fp32 mode of the same library, block by block
    python tools/depth_parity.py --oracle        # 2 two-track + 2 three-track(+final) blocks, n_enc=4: every mode
                                                 # against the CPU oracle (oracle/rf_oracle.py), block by block
    python tools/depth_parity.py --oracle --full --modes fp32,bf16,fp16   # the benchmarked 8+5 depth against the oracle
    python tools/depth_parity.py --struct-lowp   # attribution: structure-track node input in the 16-bit type (round-2 policy)

Prints one JSON object (relative L2 of msa / pair / xyz after every block, logits / xyz / plddt at the end, distogram
argmax agreement over all pairs and over the pairs with a clear top-2 margin in the reference).  tests/test_depth_gpu.py
asserts the --oracle numbers; profiles/r03_depth_parity*.json keep both outputs.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rosettafold_pytorch_amd as R  # noqa: E402
from rosettafold_pytorch_amd import ops  # noqa: E402
from rosettafold_pytorch_amd import model as M  # noqa: E402

MODES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def rel2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def hip_trace(model, msa, seq, aa):
    """RoseTTAFold.forward (structure.py) with a snapshot after every block."""
    snaps = []
    with torch.no_grad(), torch.cuda.device(msa.device):
        mono = M.check_index_range(msa, seq, aa, 21, model.msa_emb.pos_enc.max_len)
        m = model.msa_emb.run(msa, aa)
        p = model.pair_emb.run(seq, aa)
        onehot = ops.onehot(seq, 21)
        for i, blk in enumerate(model.two_track_blocks):
            p = blk.run(m, p)
            snaps.append((f"two_track.{i}", m.float().cpu(), p.float().cpu(), None))
        xyz = model.initial_coord_generation_with_msa_and_pair.run(m, p, onehot, aa)
        snaps.append(("initial_coords", None, None, xyz.cpu()))
        for i, blk in enumerate(model.three_track_blocks):
            m, p, xyz = blk.run3(m, p, xyz, onehot, aa, mono)
            snaps.append((f"three_track.{i}", m.float().cpu(), p.float().cpu(), xyz.cpu()))
        m, p, xyz, plddt = model.final_block.run3(m, p, xyz, onehot, aa, mono)
        snaps.append(("final_block", m.float().cpu(), p.float().cpu(), xyz.cpu()))
        logits = model.prediction_head.run(p)
        torch.cuda.synchronize()
    return snaps, {k: v.float().cpu() for k, v in logits.items()}, xyz.cpu(), plddt.cpu()


def oracle_trace(P, msa, seq, aa, cfg):
    from oracle import rf_oracle as O
    import torch.nn.functional as F
    snaps = []
    ne = cfg["n_encoder_layers"]
    with torch.no_grad():
        m = O.msa_embedding(P, "msa_emb", msa, aa, cfg["max_len"])
        p = O.pair_embedding(P, "pair_emb", seq, aa, cfg["max_len"])
        onehot = F.one_hot(seq, 21).float()
        for i in range(cfg["n_two_track_blocks"]):
            m, p = O.two_track_block(P, f"two_track_blocks.{i}", m, p, ne)
            snaps.append((f"two_track.{i}", m, p, None))
        xyz = O.initial_coord_generation(P, "initial_coord_generation_with_msa_and_pair", m, p, onehot, aa)
        snaps.append(("initial_coords", None, None, xyz))
        for i in range(cfg["n_three_track_blocks"] - 1):
            m, p, xyz = O.three_track_block(P, f"three_track_blocks.{i}", m, p, xyz, onehot, aa, ne, cfg["n_neighbors"][i],
                                            cfg["d_state"])
            snaps.append((f"three_track.{i}", m, p, xyz))
        m, p, xyz, plddt = O.three_track_block(P, "final_block", m, p, xyz, onehot, aa, ne, 32, cfg["d_state"], final=True)
        snaps.append(("final_block", m, p, xyz))
        logits = O.prediction_head(P, "prediction_head", p)
    return snaps, logits, xyz, plddt


def compare(got, ref):
    gs, gl, gx, gp = got
    rs, rl, rx, rp = ref
    curve = []
    for (name, m, p, x), (_, rm_, rp_, rx_) in zip(gs, rs):
        row = {"after": name}
        if m is not None:
            row["msa"], row["pair"] = rel2(m, rm_), rel2(p, rp_)
        if x is not None:
            row["xyz"] = rel2(x, rx_)
        curve.append(row)
    d = rl["dist"]
    top2 = d.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.02 * (d.max() - d.min())
    same = gl["dist"].argmax(-1) == d.argmax(-1)
    return {"curve": curve,
            "rel_l2": {**{k: rel2(gl[k], rl[k]) for k in rl}, "xyz": rel2(gx, rx), "plddt": rel2(gp, rp)},
            "argmax_agreement": {k: (gl[k].argmax(-1) == rl[k].argmax(-1)).float().mean().item() for k in rl},
            "dist_argmax_agreement": same.float().mean().item(),
            "dist_argmax_agreement_clear_margin": same[clear].float().mean().item() if clear.any() else None,
            "clear_margin_fraction": clear.float().mean().item()}


def run(args):
    dev = torch.device("cuda", 0)
    n2, n3 = (2, 2) if args.oracle and not getattr(args, 'full', False) else (args.n_two, args.n_three)
    cfg = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=n2,
               n_three_track_blocks=n3, n_encoder_layers=4, max_len=args.L + 4, n_neighbors=[128, 128, 64, 64, 64],
               p_dropout=0.0)
    torch.manual_seed(1234)
    model = R.RoseTTAFold(**cfg).to(dev)
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (args.B, args.N, args.L), generator=g)
    seq = msa[:, 0].clone()
    aa = torch.arange(args.L).unsqueeze(0).repeat(args.B, 1)
    dmsa, dseq, daa = msa.to(dev), seq.to(dev), aa.to(dev)
    M.RT.struct_inputs_fp32 = not args.struct_lowp
    modes = [m for m in args.modes.split(",") if m]
    res = {"config": {"B": args.B, "N": args.N, "L": args.L, "blocks": f"{n2}+{n3}", "n_enc": 4, "struct_inputs_fp32": not args.struct_lowp,
                      "reference": "CPU oracle (oracle/rf_oracle.py)" if args.oracle else "exact-fp32 mode of the library"}}
    traces = {}
    for name in (["fp32"] if "fp32" not in modes else []) + modes:
        R.set_compute_dtype(MODES[name])
        hip_trace(model, dmsa, dseq, daa)  # warm-up: weight copies of this mode
        t0 = time.perf_counter()
        traces[name] = hip_trace(model, dmsa, dseq, daa)
        res.setdefault("ms_forward_with_snapshots", {})[name] = 1e3 * (time.perf_counter() - t0)
    R.set_compute_dtype(torch.bfloat16)
    if args.oracle:
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        P = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        t0 = time.perf_counter()
        ref = oracle_trace(P, msa, seq, aa, cfg)
        res["oracle_seconds"] = time.perf_counter() - t0
    else:
        ref = traces["fp32"]
    for name in traces:
        if name == "fp32" and not args.oracle:
            continue
        res[name] = compare(traces[name], ref)
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--full", action="store_true", help="with --oracle: the whole --n-two + --n-three depth (8+5: ~4 min of oracle on 16 cores)")
    ap.add_argument("--struct-lowp", action="store_true")
    ap.add_argument("--modes", default="bf16,fp16")
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--L", type=int, default=256)
    ap.add_argument("--n-two", type=int, default=8)
    ap.add_argument("--n-three", type=int, default=5)
    run(ap.parse_args())
