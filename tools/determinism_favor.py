#!/usr/bin/env python3
"""Run-to-run bitwise comparison of the fused FAVOR+ kernel over its variants (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
torch.manual_seed(0)
def same(name, fn, n=6):
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    ok = all(torch.equal(outs[0], o) for o in outs[1:])
    md = max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:])
    nbad = max(((outs[0] != o).any(-1)).sum().item() for o in outs[1:])
    print(f"{name:34s} bitwise identical: {ok}   max |diff| {md:.3e}  rows differing {nbad}", flush=True)
for gen in (True, False):
    for Ls, Lo, H, D in ((64, 1024, 8, 288), (128, 1024, 12, 384), (256, 1024, 8, 288)):
        m = R.PerformerSelfAttention(dim=D, heads=H, generalized_attention=gen).cuda()
        inner = 64 * H
        W3 = 3 * inner
        qkv = torch.randn(Lo * Ls, W3, device="cuda").bfloat16()
        pc = m.proj_scaled(log2e=not gen)
        def f():
            o = torch.empty(Lo * Ls, inner, device="cuda", dtype=torch.bfloat16)
            ops.favor_attention(qkv, pc, o, (Lo * Ls * W3, Ls * W3, W3, 64), (Lo * Ls * inner, Ls * inner, inner), 0, inner, 2 * inner, 1, Lo, H, Ls, 64, 266, not gen, 1e-3 if gen else 1e-4)
            return o
        same(f"favor {'relu' if gen else 'softmax'} Ls={Ls}", f)
