#!/usr/bin/env python3
"""Does chunking [q|k|v projection -> FAVOR attention] so that the q|k|v chunk stays in the 256 MB Infinity Cache pay?
(timing experiment; run on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R  # noqa: E402
from rosettafold_pytorch_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    torch.manual_seed(0)
    B, L1, L2, D, H = 4, 256, 256, 288, 8
    inner, W3 = 64 * H, 3 * 64 * H
    m = R.PerformerSelfAttention(dim=D, heads=H, generalized_attention=True).cuda()
    pc = m.proj_scaled(log2e=False)
    R_ = B * L1 * L2
    xn = torch.randn(R_, D, device="cuda").bfloat16()
    w = torch.randn(W3, D, device="cuda").bfloat16() * 0.05
    o = torch.empty(R_, inner, device="cuda", dtype=torch.bfloat16)
    qkv = torch.empty(R_, W3, device="cuda", dtype=torch.bfloat16)
    # axis = 2: item o = (b, i), sequence = the 256 consecutive rows j
    Ls, NO = L2, B * L1

    def full():
        ops.linear(xn, w, None, out=qkv)
        ops.favor_attention(qkv, pc, o, (NO * Ls * W3, Ls * W3, W3, 64), (NO * Ls * inner, Ls * inner, inner), 0, inner, 2 * inner,
                            1, NO, H, Ls, 64, 266, False, 1e-3)

    print(f"full: {timeit(full) * 1e3:.0f} us", flush=True)
    for nch in (2, 4, 8, 16):
        co = NO // nch
        rows = co * Ls
        scratch = torch.empty(rows, W3, device="cuda", dtype=torch.bfloat16)  # one q|k|v chunk, reused

        def chunked():
            for c in range(nch):
                ops.linear(xn[c * rows:(c + 1) * rows], w, None, out=scratch)
                ops.favor_attention(scratch, pc, o[c * rows:(c + 1) * rows], (co * Ls * W3, Ls * W3, W3, 64),
                                    (co * Ls * inner, Ls * inner, inner), 0, inner, 2 * inner, 1, co, H, Ls, 64, 266, False, 1e-3)

        print(f"{nch} chunks ({rows * W3 * 2 / 1e6:.0f} MB q|k|v each): {timeit(chunked) * 1e3:.0f} us", flush=True)


if __name__ == "__main__":
    main()
