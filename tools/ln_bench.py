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
