#!/usr/bin/env python3
"""Practical HBM rates of the box next to the 8 TB/s spec figure the roofline fractions are quoted against (GPU box):
write-only (rf_fill), read-dominant (LayerNorm rows: 4 B in, 2 B out) and copy (rf_axpby: 4 B in, 4 B out) over 1 GiB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosettafold_pytorch_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    n = 384 * 699008  # fp32 elements, ~1 GiB, whole rows of 384
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    yb = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    g, b = torch.ones(384, device="cuda"), torch.zeros(384, device="cuda")
    for _ in range(2):  # (second pass: warm clocks)
        t = timeit(lambda: ops.fill(y, 1.0))
        print(f"write only  (rf_fill, 1 GiB):                 {4 * n / t / 1e12:.2f} TB/s")
        t = timeit(lambda: ops.axpby(x, 1.0, None, 0.0, y))
        print(f"copy        (rf_axpby fp32 -> fp32, 2 GiB):   {8 * n / t / 1e12:.2f} TB/s")
        t = timeit(lambda: ops.axpby(x, 1.0, None, 0.0, yb))
        print(f"read-heavy  (rf_axpby fp32 -> bf16, 1.5 GiB): {6 * n / t / 1e12:.2f} TB/s")
        t = timeit(lambda: ops.layernorm(x.view(-1, 384), g, b, out=yb.view(-1, 384)))
        print(f"LayerNorm   (fp32 rows of 384 -> bf16):        {6 * n / t / 1e12:.2f} TB/s")
        t = timeit(lambda: x.sum())
        print(f"read only   (torch sum, 1 GiB):               {4 * n / t / 1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
