"""Which part of the forward carries the whole-model gap of the 16-bit modes?  (GPU box; config-2 dims, B=1, full 8+5 depth.)

    python tools/precision_probe.py [--n-two 8 --n-three 5] [--variants ...]

Every variant runs the SAME weights and inputs with a per-module choice of compute dtype and is compared with the
exact-fp32 mode of the library (pinned to the CPU oracle: tests/test_depth_gpu.py, tools/depth_parity.py --oracle --full):
  fp16 / bf16            the shipped modes
  X+head32               body in X, PredictionHead (rf.py:1130-1172) in exact fp32
  fp32+head16            body exact, head in fp16: the head's OWN operand rounding
  fp16+struct32          two-track part and the head in fp16, every three-track / final block in fp32
  fp16+coordmsa32        fp16, MsaUpdateWithPairAndCoord (rf.py:865-920) in fp32
Also a stage-by-stage error trace through the prediction head (fp16 head on the fp32 body's pair tensor).
Prints one JSON object.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rosettafold_pytorch_amd as R  # noqa: E402
from rosettafold_pytorch_amd import model as M, ops  # noqa: E402

DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def rel2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@contextlib.contextmanager
def module_dtype(mods, dtype):
    """Run the .run / .run3 of every module in `mods` under `dtype` (the weight caches are keyed by dtype)."""
    saved = []
    for m in mods:
        for name in ("run", "run3"):
            fn = getattr(m, name, None)
            if fn is None:
                continue

            def wrapped(*a, _fn=fn, **k):
                prev = M.T()
                R.set_compute_dtype(dtype)
                try:
                    return _fn(*a, **k)
                finally:
                    R.set_compute_dtype(prev)
            object.__setattr__(m, name, wrapped)
            saved.append((m, name))
    try:
        yield
    finally:
        for m, name in saved:
            object.__delattr__(m, name)


def metrics(out, ref):
    (lg, x, p), (rl, rx, rp) = out, ref
    d = rl["dist"]
    top2 = d.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.02 * (d.max() - d.min())
    same = lg["dist"].argmax(-1) == d.argmax(-1)
    return {"dist_argmax": same.float().mean().item(), "dist_argmax_clear": same[clear].float().mean().item(),
            "argmax": {k: (lg[k].argmax(-1) == rl[k].argmax(-1)).float().mean().item() for k in rl},
            "rel_l2": {**{k: rel2(lg[k], rl[k]) for k in rl}, "xyz": rel2(x, rx), "plddt": rel2(p, rp)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-two", type=int, default=8)
    ap.add_argument("--n-three", type=int, default=5)
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--L", type=int, default=256)
    ap.add_argument("--variants", default="fp16,bf16,fp16+head32,bf16+head32,fp32+head16,fp32+ln16white,fp32+ln16centred")
    ap.add_argument("--ln-sites", default="", choices=["", "fp16", "bf16"],
                    help="sweep: round the outputs of one group of LayerNorm sites at a time (exact-fp32 mode otherwise)")
    ap.add_argument("--module-sweep", default="", choices=["", "fp16", "bf16"],
                    help="sweep: one module class of every block in the 16-bit type at a time (exact-fp32 mode otherwise)")
    ap.add_argument("--pum-sweep", default="", choices=["", "fp16", "bf16"],
                    help="sweep: one 16-bit intermediate of PairUpdateWithMsa at a time (exact-fp32 mode otherwise)")
    ap.add_argument("--gemm-sweep", default="", help="<block child>:<fp16|bf16>, e.g. msa_update_using_self_att:fp16 -- round the "
                    "activation operands of that module's GEMMs one call site at a time (exact-fp32 mode otherwise)")
    ap.add_argument("--gemm-sites", default="", help="with --gemm-sweep: only these sites (comma separated; suffix -c / -cn / -cl: centred rounding)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=args.n_two,
               n_three_track_blocks=args.n_three, n_encoder_layers=4, max_len=args.L + 4, n_neighbors=[128, 128, 64, 64, 64],
               p_dropout=0.0)
    torch.manual_seed(1234)
    model = R.RoseTTAFold(**cfg).to(dev)
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (1, args.N, args.L), generator=g)
    inp = (msa.to(dev), msa[:, 0].clone().to(dev), torch.arange(args.L).unsqueeze(0).to(dev))

    def fwd(body, overrides=()):
        R.set_compute_dtype(DT[body])
        with contextlib.ExitStack() as st:
            for mods, dt in overrides:
                st.enter_context(module_dtype(mods, DT[dt]))
            t0 = time.perf_counter()
            lg, x, p = model(*inp)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
        R.set_compute_dtype(torch.bfloat16)
        return ({k: v.float().cpu() for k, v in lg.items()}, x.cpu(), p.cpu()), ms

    # Emulation in the exact-fp32 mode: round ONLY the LayerNorm outputs that are GEMM operands (M.ln with the default output
    # type) to fp16 -- "white": as the fp16 mode does; "centred": after removing their per-(sample, channel) mean over all
    # positions (what a bias-folded centring of every LayerNorm -> Linear pair would achieve).  Everything else stays exact, so
    # the two numbers bound what operand conditioning of the LayerNorm outputs alone could buy in the body.
    from rosettafold_pytorch_amd import structure as S

    blocks_all = list(model.two_track_blocks) + list(model.three_track_blocks) + [model.final_block]
    site_sets = {
        "projmsa": {id(b.pair_update_with_msa.proj_msa[0]) for b in blocks_all},
        "proj2": {id(b.pair_update_with_msa.proj_msa[2]) for b in blocks_all},
        "msa2value": {id(l.msa2value[0]) for b in blocks_all for l in b.msa_update_with_pair.encoder_layers},
        "msaenc": {id(m_) for b in blocks_all for lay in list(b.msa_update_using_self_att.residue_wise_encoder_layers)
                   + list(b.msa_update_using_self_att.sequence_wise_encoder_layers) for m_ in (lay.ln, lay.ff.fn[0])},
    }
    exact_sites = set()
    only_sites = None      # when a set: round ONLY at these modules (the --ln-sites sweep)
    import re
    ln_groups = {}
    for mname, mod in model.named_modules():
        if type(mod).__name__ == "LayerNorm":
            ln_groups.setdefault(re.sub(r"\.\d+\.", ".*.", re.sub(r"^(two|three)_track_blocks\.\d+\.|^final_block\.", "block.", mname)), set()).add(id(mod))

    @contextlib.contextmanager
    def ln_rounding(kind, dt=torch.float16):
        orig = M.ln

        def patched(mod, x, out_dtype=None, **kw):
            y = orig(mod, x, out_dtype=out_dtype, **kw)
            if id(mod) in exact_sites or (only_sites is not None and id(mod) not in only_sites):
                return y
            if out_dtype is None and y.dtype == torch.float32 and y.dim() == 4:
                centre = kind == "centred" or (kind == "centred-pair" and y.shape[-1] == cfg["d_pair"] and y.shape[1] == y.shape[2])
                if not centre:
                    y.copy_(y.to(dt).float())
                else:
                    mu = y.mean(dim=(1, 2), keepdim=True)
                    y.copy_((y - mu).to(dt).float() + mu)
            return y
        M.ln = S.ln = patched
        try:
            yield
        finally:
            M.ln = S.ln = orig

    head = [model.prediction_head]
    struct = list(model.three_track_blocks) + [model.final_block]
    coordmsa = [b.msa_update_with_pair_and_coord for b in model.three_track_blocks]
    ref, ms32 = fwd("fp32")
    res = {"config": {"B": 1, "N": args.N, "L": args.L, "blocks": f"{args.n_two}+{args.n_three}",
                      "reference": "exact-fp32 mode of the library"}, "fp32_ms": ms32}
    table = {
        "fp16": ("fp16", []), "bf16": ("bf16", []),
        "fp16+head32": ("fp16", [(head, "fp32")]), "bf16+head32": ("bf16", [(head, "fp32")]),
        "fp32+head16": ("fp32", [(head, "fp16")]),
        "fp16+struct32": ("fp16", [(struct, "fp32")]),
        "fp16+coordmsa32": ("fp16", [(coordmsa, "fp32")]),
    }
    if args.ln_sites:
        # one forward per group of LayerNorm call sites: fp16 (or bf16) rounding of THAT group's outputs alone, every other value exact
        dt = torch.bfloat16 if args.ln_sites == "bf16" else torch.float16
        sweep = {}
        for gname, ids in sorted(ln_groups.items()):
            only_sites = ids
            with ln_rounding("white", dt):
                out, _ = fwd("fp32")
            m = metrics(out, ref)
            sweep[gname] = {"modules": len(ids), "dist_rel_l2": m["rel_l2"]["dist"], "xyz_rel_l2": m["rel_l2"]["xyz"], "dist_argmax": m["dist_argmax"]}
            print(f"[sites] {gname:70s} n={len(ids):3d} dist {m['rel_l2']['dist']:.3e} argmax {m['dist_argmax']:.4f} xyz {m['rel_l2']['xyz']:.3e}",
                  file=sys.stderr, flush=True)
        only_sites = None
        res["ln_site_sweep_" + args.ln_sites] = sweep
    if args.module_sweep:
        # one forward per module class: THAT class of every block in the 16-bit type, everything else exact
        groups = {}
        for b in blocks_all:
            for cname, child in b.named_children():
                if hasattr(child, "run") or hasattr(child, "run3"):
                    groups.setdefault("block." + cname, []).append(child)
        groups["initial_coord_generation_with_msa_and_pair"] = [model.initial_coord_generation_with_msa_and_pair]
        groups["prediction_head"] = [model.prediction_head]
        sweep = {}
        for gname, mods in groups.items():
            fwd("fp32", [(mods, args.module_sweep)])
            out, _ = fwd("fp32", [(mods, args.module_sweep)])
            m = metrics(out, ref)
            sweep[gname] = {"modules": len(mods), "dist_rel_l2": m["rel_l2"]["dist"], "xyz_rel_l2": m["rel_l2"]["xyz"], "dist_argmax": m["dist_argmax"]}
            print(f"[modules] {gname:60s} n={len(mods):3d} dist {m['rel_l2']['dist']:.3e} argmax {m['dist_argmax']:.4f} xyz {m['rel_l2']['xyz']:.3e}",
                  file=sys.stderr, flush=True)
        res["module_sweep_" + args.module_sweep] = sweep
    if args.pum_sweep:
        # PairUpdateWithMsa (rf.py:430-498) alone carries the 16-bit modes' logits gap (--module-sweep): which of ITS 16-bit
        # intermediates?  Exact-fp32 mode, one intermediate of the 13 modules rounded at a time ("-c": after removing the
        # per-(sample, channel) mean over the picture, what an exact bias-folded centring would leave).
        dt = DT[args.pum_sweep]
        pums = [b.pair_update_with_msa for b in blocks_all]
        Dp, P = cfg["d_pair"], pums[0].d_proj
        cols = {"coevol": (0, Dp), "1d": (Dp, Dp + 4 * P), "pair": (Dp + 4 * P, 2 * Dp + 4 * P), "att": (2 * Dp + 4 * P, pums[0].d_feat)}
        state = {"site": None, "inside": False, "in_calls": 0}

        def rnd(t, centred=False):
            if centred:
                mu = t.mean(dim=tuple(range(1, t.dim() - 1)), keepdim=True)
                return (t - mu).to(dt).float() + mu
            return t.to(dt).float()

        o_linear, o_cast, o_conv, o_inst = ops.linear, ops.cast, M.conv3x3, ops.instnorm

        def w_linear(x, w, *a, **k):
            site = state["site"]
            if state["inside"] and site.startswith("feat:") and x.dim() == 4 and x.shape[-1] >= pums[0].d_feat and x.shape[-1] - pums[0].d_feat < 8:
                key = site[5:]
                c0, c1 = cols[key[:-2] if key.endswith("-c") else key]
                x[..., c0:c1] = rnd(x[..., c0:c1], key.endswith("-c"))
            return o_linear(x, w, *a, **k)

        def w_cast(x, dtype):
            r = o_cast(x, dtype)
            if state["inside"] and state["site"] in ("x_cast", "x_cast-c"):
                return rnd(r, state["site"].endswith("-c")).contiguous()
            return r

        def w_conv(mod, key, conv, x, dil, out_dtype=None):
            r = o_conv(mod, key, conv, x, dil, out_dtype)
            if state["inside"] and state["site"] in ("conv_" + key, "conv_" + key + "-c"):
                r.copy_(rnd(r, state["site"].endswith("-c")))
            return r

        def w_inst(x, *a, **k):
            r = o_inst(x, *a, **k)
            if state["inside"] and k.get("residual") is None and state["site"] == "instnorm1":
                r[0].copy_(rnd(r[0]))
            return r

        saved = []
        for m_ in pums:
            def run(msa_, pair_, att_, _fn=m_.run):
                state["inside"] = True
                try:
                    return _fn(msa_, pair_, att_)
                finally:
                    state["inside"] = False

            def operands(msa_, _fn=m_._msa_operands):
                msa1d, xt, yt, Np = _fn(msa_)
                site = state["site"]
                if site in ("msa1d", "msa1d-c"):
                    msa1d = rnd(msa1d, site.endswith("-c"))
                if site == "outer_operands":
                    xt, yt = rnd(xt), rnd(yt)
                return msa1d, xt, yt, Np
            object.__setattr__(m_, "run", run)
            object.__setattr__(m_, "_msa_operands", operands)
            saved.append(m_)
        ops.linear, ops.cast, M.conv3x3, ops.instnorm = w_linear, w_cast, w_conv, w_inst
        sweep = {}
        try:
            for site in ("feat:coevol", "feat:1d", "feat:1d-c", "feat:pair", "feat:pair-c", "feat:att", "feat:att-c", "msa1d", "msa1d-c",
                         "outer_operands", "x_cast", "x_cast-c", "conv_c1", "conv_c1-c", "instnorm1", "conv_c2", "weights"):
                state["site"] = site
                if site == "weights":
                    keep_w = [(p_, p_.data.clone()) for m_ in pums for p_ in m_.parameters()]
                    for p_, _ in keep_w:
                        p_.data.copy_(p_.data.to(dt).float())
                    M.invalidate_weight_caches(model)
                out, _ = fwd("fp32")
                if site == "weights":
                    for p_, d_ in keep_w:
                        p_.data.copy_(d_)
                    M.invalidate_weight_caches(model)
                m = metrics(out, ref)
                sweep[site] = {"dist_rel_l2": m["rel_l2"]["dist"], "xyz_rel_l2": m["rel_l2"]["xyz"], "dist_argmax": m["dist_argmax"]}
                print(f"[pum] {site:20s} dist {m['rel_l2']['dist']:.3e} argmax {m['dist_argmax']:.4f} xyz {m['rel_l2']['xyz']:.3e}", file=sys.stderr, flush=True)
        finally:
            ops.linear, ops.cast, M.conv3x3, ops.instnorm = o_linear, o_cast, o_conv, o_inst
            for m_ in saved:
                object.__delattr__(m_, "run")
                object.__delattr__(m_, "_msa_operands")
        res["pair_update_with_msa_sweep_" + args.pum_sweep] = sweep
    if args.gemm_sweep:
        # Inside one module class of every block (exact-fp32 mode): the activation operands of its GEMMs rounded to the 16-bit type
        # IN PLACE, one call site at a time (what the 16-bit mode stores there; weights are left alone).  Sites = source lines of
        # the calls.  Only for modules whose operands are private tensors (not the fp32 residual streams).
        child, dts = args.gemm_sweep.split(":")
        dt = DT[dts]
        targets = [getattr(b, child) for b in blocks_all if hasattr(b, child)]
        weights = set()
        state = {"inside": False, "site": None, "seen": {}}
        o_gemm = ops.gemm

        def collect_weights():
            weights.clear()
            for m_ in model.modules():
                c_ = getattr(m_, "_rfc", None)
                if isinstance(c_, dict):
                    for v in c_.values():
                        for t_ in (v if isinstance(v, (tuple, list)) else (v,)):
                            if torch.is_tensor(t_):
                                weights.add(t_.data_ptr())

        def w_gemm(A, B_, Cout, M_, N_, K_, **k):
            if state["inside"]:
                f = sys._getframe(1)
                while f.f_code.co_filename.endswith("ops.py"):
                    f = f.f_back
                for tag, t_ in (("A", A), ("B", B_)):
                    if t_.data_ptr() in weights or (t_._base is not None and t_._base.data_ptr() in weights):
                        continue
                    label = f"{f.f_code.co_name}:{f.f_lineno}:{tag}"
                    state["seen"][label] = state["seen"].get(label, 0) + 1
                    if state["site"] in (label, "ALL"):
                        t_.copy_(t_.to(dt).float())
                    elif state["site"] == label + "-c":      # centred: minus the per-channel mean over ALL rows (one constant vector)
                        mu = t_.reshape(-1, t_.shape[-1]).mean(0)
                        t_.copy_((t_ - mu).to(dt).float() + mu)
                    elif state["site"] == label + "-cn" and t_.dim() == 4:   # minus the mean over dim 2 (per MSA row / per outer index)
                        mu = t_.mean(2, keepdim=True)
                        t_.copy_((t_ - mu).to(dt).float() + mu)
                    elif state["site"] == label + "-cl" and t_.dim() == 4:   # minus the mean over dim 1
                        mu = t_.mean(1, keepdim=True)
                        t_.copy_((t_ - mu).to(dt).float() + mu)
            return o_gemm(A, B_, Cout, M_, N_, K_, **k)

        saved = []
        for m_ in targets:
            for meth in ("run", "run3"):
                fn = getattr(m_, meth, None)
                if fn is None:
                    continue

                def wrapped(*a, _fn=fn, **k):
                    state["inside"] = True
                    try:
                        return _fn(*a, **k)
                    finally:
                        state["inside"] = False
                object.__setattr__(m_, meth, wrapped)
                saved.append((m_, meth))
        ops.gemm = w_gemm
        sweep = {}
        try:
            fwd("fp32")
            collect_weights()
            state["seen"].clear()
            fwd("fp32")   # dry pass: the call sites
            sites = sorted(state["seen"])
            print("[gemm] sites: " + ", ".join(f"{k} x{v}" for k, v in sorted(state["seen"].items())), file=sys.stderr, flush=True)
            if args.gemm_sites:
                sites = args.gemm_sites.split(",")
            for site in (["ALL"] if not args.gemm_sites else []) + sites:
                state["site"] = site
                out, _ = fwd("fp32")
                m = metrics(out, ref)
                sweep[site] = {"dist_rel_l2": m["rel_l2"]["dist"], "xyz_rel_l2": m["rel_l2"]["xyz"], "dist_argmax": m["dist_argmax"]}
                print(f"[gemm] {site:40s} dist {m['rel_l2']['dist']:.3e} argmax {m['dist_argmax']:.4f} xyz {m['rel_l2']['xyz']:.3e}", file=sys.stderr, flush=True)
        finally:
            ops.gemm = o_gemm
            for m_, meth in saved:
                object.__delattr__(m_, meth)
        res["gemm_operand_sweep_" + args.gemm_sweep] = sweep
    for name in [v for v in args.variants.split(",") if v]:
        if name.startswith("fp32+ln"):   # fp32+ln16white | fp32+ln16centred | fp32+lnbf16white | ... | fp32+ln16white-exact:projmsa
            exact_sites.clear()
            if "-exact:" in name:
                for key in name.split("-exact:")[1].split("+"):
                    exact_sites.update(site_sets[key])
                name_kind = name.split("-exact:")[0]
            else:
                name_kind = name
            kind = "white" if name_kind.endswith("white") else ("centred-pair" if name_kind.endswith("centredpair") else "centred")
            with ln_rounding(kind, torch.bfloat16 if "bf16" in name else torch.float16):
                out, ms = fwd("fp32")
            res[name] = {**metrics(out, ref), "ms_eager": ms}
            print(f"[probe] {name}: {json.dumps(res[name])}", file=sys.stderr, flush=True)
            continue
        body, ov = table[name]
        fwd(body, ov)  # warm-up (weight copies)
        out, ms = fwd(body, ov)
        res[name] = {**metrics(out, ref), "ms_eager": ms}
        print(f"[probe] {name}: {json.dumps(res[name])}", file=sys.stderr, flush=True)

    # stage-by-stage trace through the head: record the first output of every instnorm / linear launched inside head.run
    def head_trace(dt, pair):
        rec = []
        o_in, o_lin = ops.instnorm, ops.linear

        def w_in(*a, **k):
            r = o_in(*a, **k)
            rec.append(("instnorm", r[0].float().cpu()))
            return r

        def w_lin(*a, **k):
            r = o_lin(*a, **k)
            rec.append(("linear", r.float().cpu()))
            return r
        ops.instnorm, ops.linear = w_in, w_lin
        R.set_compute_dtype(DT[dt])
        try:
            with torch.no_grad():
                model.prediction_head.run(pair)
            torch.cuda.synchronize()
        finally:
            ops.instnorm, ops.linear = o_in, o_lin
            R.set_compute_dtype(torch.bfloat16)
        return rec

    # the pair tensor the head sees in the fp32 forward
    keep = {}
    orig = model.prediction_head.run
    object.__setattr__(model.prediction_head, "run", lambda p, *a, **k: (keep.__setitem__("pair", p.clone()), orig(p, *a, **k))[1])
    fwd("fp32")
    object.__delattr__(model.prediction_head, "run")
    t32, t16 = head_trace("fp32", keep["pair"]), head_trace("fp16", keep["pair"])
    res["head_trace_fp16_on_exact_input"] = [{"stage": i, "op": a[0], "shape": list(a[1].shape), "rel_l2": rel2(b[1], a[1])}
                                             for i, (a, b) in enumerate(zip(t32, t16))][:16]
    # ... and the exact head on a pair tensor perturbed at the size of the fp16 body's stream error (8e-4 relative, white)
    gp = torch.Generator(device=dev).manual_seed(5)
    noise = torch.randn(keep["pair"].shape, generator=gp, device=dev)
    pp = keep["pair"] + 8e-4 * keep["pair"].norm() / noise.norm() * noise
    tpp = head_trace("fp32", pp)
    res["head_trace_exact_on_input_noise_8e-4"] = [{"stage": i, "op": a[0], "rel_l2": rel2(b[1], a[1])}
                                                   for i, (a, b) in enumerate(zip(t32, tpp))][:16]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
