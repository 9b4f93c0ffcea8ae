#!/usr/bin/env python3
"""Tied-attention logits + softmax kernel at the bench shape (run on the GPU box)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosettafold_pytorch_amd import ops
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, H, N, L, dh = 4, 12, 128, 256, 32
D = H * dh
qkp = (torch.randn(B, N, L, 3 * D, device="cuda") * (0.6 / math.sqrt(N))).bfloat16()
att = torch.empty(B, H, L, L, device="cuda", dtype=torch.bfloat16)
sym = torch.empty(B, L, L, H, device="cuda", dtype=torch.float32)
t = timeit(lambda: ops.tied_logits_softmax(qkp, qkp[..., D:], N * L * 3 * D, L * 3 * D, 3 * D, att, None, B, H, N, L, dh))
fl = 2.0 * B * H * L * L * N * dh
print(f"logits+softmax: {t*1e3:.1f} us, {fl/t/1e9:.0f} TF/s")
t2 = timeit(lambda: ops.tied_logits_softmax(qkp, qkp[..., D:], N * L * 3 * D, L * 3 * D, 3 * D, att, sym, B, H, N, L, dh))
print(f"with symmetrised map: {t2*1e3:.1f} us")
logits = torch.empty(B, H, L, L, device="cuda", dtype=torch.float32)
W3 = 3 * D
def old():
    ops.gemm(qkp, qkp, logits, L, L, N * dh, batch=(B, H, 1), b_off=D, a_bs=(N * L * W3, dh, 0), a_row=(0, 0, W3), a_ko=L * W3,
             b_bs=(N * L * W3, dh, 0), b_row=(0, 0, W3), b_ko=L * W3, kc=dh, c_bs=(H * L * L, L * L, 0), c_row=(0, 0, L))
    ops.tied_softmax(logits, att, sym, H)
print(f"rf_gemm + rf_tied_softmax: {timeit(old)*1e3:.1f} us")
