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

    @contextlib.contextmanager
    def ln_rounding(kind, dt=torch.float16):
        orig = M.ln

        def patched(mod, x, out_dtype=None, **kw):
            y = orig(mod, x, out_dtype=out_dtype, **kw)
            if out_dtype is None and y.dtype == torch.float32 and y.dim() == 4:
                if kind == "white":
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
    for name in [v for v in args.variants.split(",") if v]:
        if name.startswith("fp32+ln"):   # fp32+ln16white | fp32+ln16centred | fp32+lnbf16white | fp32+lnbf16centred
            kind = "white" if name.endswith("white") else "centred"
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
