#!/usr/bin/env python3
"""Host cost of one launch through the PyTorch dispatcher (torch.ops.rfmi.*) against the direct C-ABI call (ops.py), on
tensors small enough that the kernel itself is a few microseconds (run on the GPU box).  This is the number behind the
choice documented in custom_ops.py / DESIGN.md: the model's forward calls the C ABI directly."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
import rosettafold_pytorch_amd.custom_ops  # noqa: F401


def host_us(fn, iters=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t1 = time.perf_counter()  # launches only: the queue is drained after the clock stops
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / iters


x = torch.randn(64, 384, device="cuda")
g, b = torch.ones(384, device="cuda"), torch.zeros(384, device="cuda")
out = torch.empty(64, 384, device="cuda", dtype=torch.bfloat16)
xb = x.bfloat16()
w = torch.randn(384, 384, device="cuda").bfloat16()
bias = torch.zeros(384, device="cuda")
rows = [
    ("layernorm, direct C ABI (preallocated out)", lambda: ops.layernorm(x, g, b, eps=1e-5, out=out)),
    ("layernorm, direct C ABI (allocating)", lambda: ops.layernorm(x, g, b, eps=1e-5, out_dtype=torch.bfloat16)),
    ("layernorm, torch.ops.rfmi.layernorm", lambda: torch.ops.rfmi.layernorm(x, g, b, 1e-5, False)),
    ("linear, direct C ABI", lambda: ops.linear(xb, w, bias)),
    ("linear, torch.ops.rfmi.linear", lambda: torch.ops.rfmi.linear(xb, w, bias, 0, False)),
    ("torch.nn.functional.layer_norm (ATen, for scale)", lambda: torch.nn.functional.layer_norm(x, (384,), g, b, 1e-5)),
]
for name, fn in rows:
    print(f"{host_us(fn):7.1f} us/launch (host)  {name}")
