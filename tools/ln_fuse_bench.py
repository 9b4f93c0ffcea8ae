#!/usr/bin/env python3
"""Residual GEMM + next LayerNorm at the forward's shapes (GPU box): fused epilogue of the persistent GEMM against GEMM + rf_layernorm."""
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
torch.manual_seed(0)
for M, N, K in ((262144, 288, 512), (262144, 288, 1152), (131072, 384, 384), (131072, 384, 768), (131072, 384, 1536)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16(); b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda"); ln = torch.nn.LayerNorm(N).cuda()
    ops.FUSE_LN = True
    tf = timeit(lambda: ops.linear_residual_ln(x, w, b, res, ln))
    ops.FUSE_LN = False
    tg = timeit(lambda: ops.linear_residual_ln(x, w, b, res, ln))
    tl = timeit(lambda: ops.layernorm(res, ln.weight.detach(), ln.bias.detach(), eps=ln.eps, out_dtype=torch.bfloat16))
    ops.FUSE_LN = True
    print(f"M={M} N={N} K={K}: fused {tf:.1f} us | GEMM {tg:.1f} + LayerNorm {tl:.1f} = {tg + tl:.1f} us")
