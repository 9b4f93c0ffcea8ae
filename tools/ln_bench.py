#!/usr/bin/env python3
"""LayerNorm bandwidth on the two residual-stream shapes (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosettafold_pytorch_amd import ops
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for rows, D in [(262144, 288), (131072, 384), (262144, 1024)]:
    x = torch.randn(rows, D, device="cuda")
    g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.layernorm(x, g, b, out=y))
    print(f"rows {rows} D {D}: {t*1e3:.1f} us, {rows*D*6/t/1e6:.0f} GB/s", flush=True)
# reference points on the same device: fp32 -> bf16 cast (same bytes as LayerNorm) and an fp32 copy, both PyTorch kernels
x = torch.randn(262144, 288, device="cuda"); y = torch.empty(262144, 288, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print(f"torch cast fp32->bf16 (262144 x 288): {t*1e3:.1f} us, {262144*288*6/t/1e6:.0f} GB/s")
t = timeit(lambda: z.copy_(x)); print(f"torch copy fp32 (262144 x 288): {t*1e3:.1f} us, {262144*288*8/t/1e6:.0f} GB/s")
