#!/usr/bin/env python3
"""Fused feed-forward kernel (csrc/ffn.hip) against the two-GEMM path on the forward's two shapes (GPU box):
parity vs an fp32 formula, fused vs unfused, time per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

QUICK = bool(os.environ.get("FFN_QUICK"))
for dt in ((torch.bfloat16,) if QUICK else (torch.bfloat16, torch.float16)):
    R.set_compute_dtype(dt)
    for name, M, D in (("msa", 131072, 384), ("pair", 262144, 288)):
        torch.manual_seed(0)
        ff = R.FeedForward(D, 4 * D, 0.0).cuda()
        ln = torch.nn.LayerNorm(D).cuda()
        with torch.no_grad():
            ln.weight.normal_(1.0, 0.2); ln.bias.normal_(0.0, 0.2)
        xn = torch.randn(M, D, device="cuda").to(dt)
        x0 = torch.randn(M, D, device="cuda")
        def run(fused, with_ln=True):
            ops.FUSE_FFN = fused
            x = x0.clone()
            o = ff.apply_residual(xn, x, ln if with_ln else None)
            return x, o
        if QUICK:
            xw = x0.clone()
            print(f"{name}: fused {timeit(lambda: ff.apply_residual(xn, xw, ln)):.0f} us with LN, {timeit(lambda: ff.apply_residual(xn, xw, None)):.0f} us without", flush=True)
            continue
        xa, la = run(True)
        xb, lb = run(False)
        n = 8192   # fp32 formula on a slice (same 16-bit rounding of the hidden activations)
        w1, w2 = ff.net[0].weight.to(dt).float(), ff.net[3].weight.to(dt).float()
        h = torch.relu(xn[:n].float() @ w1.t() + ff.net[0].bias).to(dt).float()
        ref = x0[:n] + h @ w2.t() + ff.net[3].bias
        refln = torch.nn.functional.layer_norm(ref, (D,), ln.weight, ln.bias, ln.eps)
        e = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
        print(f"{str(dt).split('.')[-1]} {name} M={M} D={D}: fused vs fp32 formula x {e(xa[:n], ref):.2e} ln {e(la[:n], refln):.2e} | "
              f"two-GEMM vs formula x {e(xb[:n], ref):.2e} ln {e(lb[:n], refln) if lb is not None else float('nan'):.2e} | "
              f"fused vs two-GEMM x {e(xa, xb):.2e}", flush=True)
        xw = x0.clone()
        for fused in (True, False):
            ops.FUSE_FFN = fused
            t = timeit(lambda: ff.apply_residual(xn, xw, ln))
            t2 = timeit(lambda: ff.apply_residual(xn, xw, None))
            fl = 4.0 * M * D * 4 * D
            print(f"   {'fused' if fused else 'two GEMMs'}: {t:.0f} us with LN ({fl / t / 1e6:.0f} TF/s), {t2:.0f} us without", flush=True)
        ops.FUSE_FFN = True
R.set_compute_dtype(torch.bfloat16)
