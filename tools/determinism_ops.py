#!/usr/bin/env python3
"""Bitwise run-to-run comparison of the main kernels at bench shapes (a race shows up as a difference)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops, _lib as L
torch.manual_seed(0)
def same(name, fn, n=4):
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    def flat(o): return [t for t in (o if isinstance(o, (tuple, list)) else [o]) if t is not None]
    ok = all(all(torch.equal(a, b) for a, b in zip(flat(outs[0]), flat(o))) for o in outs[1:])
    md = max((a.float() - b.float()).abs().max().item() for o in outs[1:] for a, b in zip(flat(outs[0]), flat(o)))
    print(f"{name:34s} bitwise identical: {ok}   max |diff| {md:.3e}", flush=True)
M = 262144
x288 = torch.randn(M, 288, device="cuda").bfloat16(); w1536 = (torch.randn(1536, 288, device="cuda") * 0.05).bfloat16()
same("gemm_fast 256 (qkv)", lambda: ops.linear(x288, w1536, None))
x512 = torch.randn(M, 512, device="cuda").bfloat16(); w288 = (torch.randn(288, 512, device="cuda") * 0.05).bfloat16()
res = torch.randn(M, 288, device="cuda"); b288 = torch.randn(288, device="cuda")
same("gemm_fast 288 f32+res", lambda: ops.linear(x512, w288, b288, out_dtype=torch.float32, residual=res))
x384 = torch.randn(131072, 384, device="cuda").bfloat16(); w384 = (torch.randn(384, 384, device="cuda") * 0.05).bfloat16()
res2 = torch.randn(131072, 384, device="cuda")
same("gemm_fast 192 f32+res", lambda: ops.linear(x384, w384, None, out_dtype=torch.float32, residual=res2))
xf = torch.randn(M, 288, device="cuda"); g = torch.randn(288, device="cuda"); b = torch.randn(288, device="cuda")
same("layernorm rows8", lambda: ops.layernorm(xf, g, b, out_dtype=torch.bfloat16))
# FAVOR (pair, relu) and (msa col, softmax)
for name, gen, B, L1, L2, D, H, axis in [("favor pair relu", True, 4, 256, 256, 288, 8, 2), ("favor msa softmax", False, 4, 128, 256, 384, 12, 1)]:
    m = R.PerformerSelfAttention(dim=D, heads=H, generalized_attention=gen).cuda()
    inner = 64 * H
    qkv = torch.randn(B * L1 * L2, 3 * inner, device="cuda").bfloat16()
    Ls, Lo = (L1, L2) if axis == 1 else (L2, L1)
    ss, so = (L2, 1) if axis == 1 else (1, L2)
    RB, W3 = L1 * L2, 3 * inner
    pc = m.proj_scaled(log2e=not gen)
    def f():
        o = torch.empty(B * L1 * L2, inner, device="cuda", dtype=torch.bfloat16)
        ops.favor_attention(qkv, pc, o, (RB * W3, so * W3, ss * W3, 64), (RB * inner, so * inner, ss * inner), 0, inner, 2 * inner, B, Lo, H, Ls, 64, 266, not gen, 1e-3 if gen else 1e-4)
        return o
    same(name, f)
Bt, Ht, Nt, Lt = 4, 12, 128, 256
qkp = (torch.randn(Bt, Nt, Lt, 3 * 384, device="cuda") * (0.6 / math.sqrt(Nt))).bfloat16()
def tied():
    att = torch.empty(Bt, Ht, Lt, Lt, device="cuda", dtype=torch.bfloat16)
    ops.tied_logits_softmax(qkp, qkp[..., 384:], Nt * Lt * 1152, Lt * 1152, 1152, att, None, Bt, Ht, Nt, Lt, 32)
    return att
same("tied logits+softmax", tied)
img = torch.randn(4, 256, 256, 288, device="cuda").bfloat16()
gi, bi = torch.randn(288, device="cuda"), torch.randn(288, device="cuda")
same("instnorm (fp64 atomics)", lambda: ops.instnorm(img, gi, bi, act=L.ACT_ELU, out_dtype=torch.bfloat16)[0])
wk = (torch.randn(288, 9 * 288, device="cuda") * 0.02).bfloat16()
def conv():
    out = torch.empty(4, 256, 256, 288, device="cuda", dtype=torch.bfloat16)
    ops.gemm(img, wk, out, 4 * 256 * 256, 288, 9 * 288, conv=(4, 256, 256, 288, 1))
    return out
same("conv3x3", conv)
# round-2 kernels
x384b = torch.randn(131072, 384, device="cuda").bfloat16()
for n_out in (1152, 1536, 2304):
    wn = (torch.randn(n_out, 384, device="cuda") * 0.05).bfloat16()
    same(f"gemm_wreg K=384 N={n_out}", lambda: ops.linear(x384b, wn, None))
for n_out in (1152, 1536):
    wn = (torch.randn(n_out, 288, device="cuda") * 0.05).bfloat16()
    same(f"gemm_wreg K=288 N={n_out}", lambda: ops.linear(x288, wn, None))
x768 = torch.randn(131072, 768, device="cuda").bfloat16(); w768 = (torch.randn(384, 768, device="cuda") * 0.05).bfloat16()
same("gemm_fast K=768 N=384 f32+res", lambda: ops.linear(x768, w768, None, out_dtype=torch.float32, residual=res2))
x1536 = torch.randn(131072, 1536, device="cuda").bfloat16(); w1536b = (torch.randn(384, 1536, device="cuda") * 0.05).bfloat16()
same("gemm_fast K=1536 N=384 f32+res", lambda: ops.linear(x1536, w1536b, None, out_dtype=torch.float32, residual=res2))
