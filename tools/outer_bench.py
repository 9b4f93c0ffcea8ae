#!/usr/bin/env python3
"""OuterProductMean at the bench shape (run on the GPU box): fused kernel vs the round-1 two-GEMM path."""
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
    return s.elapsed_time(e) / iters
B, N, L, P, Dout = 4, 128, 256, 32, 288
torch.manual_seed(0)
m = R.OuterProductMean(P, Dout).cuda()
xt = torch.randn(B, L, P, N, device="cuda").bfloat16()
yt = (torch.randn(B, L, P, N, device="cuda") * 0.05).bfloat16()
fl = 2.0 * B * N * (P * L) ** 2 + 2.0 * B * L * L * 1024 * Dout
for fused in (True, False):
    R.RT.fused_outer = fused
    t = timeit(lambda: m.run(xt, yt, N))
    print(f"OuterProductMean.run fused={fused}: {t*1e3:.1f} us = {fl/t/1e9:.0f} TF/s (223 GF)")
R.RT.fused_outer = True
a = m.run(xt, yt, N)
R.RT.fused_outer = False
b = m.run(xt, yt, N)
R.RT.fused_outer = True
print("fused vs two-GEMM path: max-rel", ((a - b).abs().max() / b.abs().max()).item())
# with PairUpdateWithMsa.ln_coevol_feat in the epilogue (bf16 rows of the 720-wide feature tensor)
ln2 = R.LayerNorm(Dout).cuda()
feat = torch.empty(B, L, L, 720, device="cuda", dtype=torch.bfloat16)
t = timeit(lambda: m.run_into(xt, yt, ln2, feat, 720))
t_ln = timeit(lambda: R.model.ln(ln2, a, out=feat, out_ld=720, out_off=0))
print(f"fused + LayerNorm(288) epilogue -> feat: {t*1e3:.1f} us (separate LayerNorm launch it replaces: {t_ln*1e3:.1f} us)")
# the two fused kernels side by side (csrc/outer_pairs.hip = default, csrc/outer.hip = RF_OUTER_CHUNKS=1)
for pairs in (True, False):
    ops.OUTER_PAIRS = pairs
    R.invalidate_weight_caches(m)
    t = timeit(lambda: m.run(xt, yt, N))
    t2 = timeit(lambda: m.run_into(xt, yt, ln2, feat, 720))
    o = m.run(xt, yt, N)
    print(f"{'pairs-in-registers' if pairs else 'column-split (round 2/3a)'} kernel: {t*1e3:.1f} us fp32 rows ({fl/t/1e9:.0f} TF/s), {t2*1e3:.1f} us with LayerNorm(288)"
          f"  | vs two-GEMM path max-rel {((o - b).abs().max() / b.abs().max()).item():.2e}")
ops.OUTER_PAIRS = True
