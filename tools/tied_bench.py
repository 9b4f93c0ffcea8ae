#!/usr/bin/env python3
"""Tied MSA-row attention core at the bench shape (run on the GPU box): round-2 head-major kernels vs the round-1 path."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
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
fl = 2.0 * B * H * L * L * N * dh
qkv = (torch.randn(B, N, 3 * H, L, dh, device="cuda") * (0.9 / math.sqrt(math.sqrt(N)))).bfloat16()
q, k, v = qkv[:, :, :H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:]
w = torch.rand(B, H, N, L, device="cuda").softmax(2).contiguous()
att = torch.empty(B, H, L, L, device="cuda", dtype=torch.bfloat16)
sym = torch.empty(B, L, L, H, device="cuda", dtype=torch.float32)
out = torch.empty(B, N, L, D, device="cuda", dtype=torch.bfloat16)
o5 = out.view(B, N, L, H, dh).permute(0, 1, 3, 2, 4)
import ctypes as C
from rosettafold_pytorch_amd._lib import lib, I64x4, I64x3
t_all = timeit(lambda: ops.tied_attention(q, k, v, o5, att, w=w, qscale=0.17))
print(f"v2 core (logits+softmax with weights, A.V): {t_all*1e3:.1f} us = {2*fl/t_all/1e9:.0f} TF/s (51.5 GF)")
vs, os_ = I64x4(*v.stride()[:4]), I64x4(*o5.stride()[:4])
t_av = timeit(lambda: lib.rf_tied_av(ops.ptr(att), ops.ptr(v), C.byref(vs), ops.ptr(out), C.byref(os_), B, H, N, L, dh, ops.stream()))
print(f"   A.V kernel alone: {t_av*1e3:.1f} us = {fl/t_av/1e9:.0f} TF/s; logits+softmax: {(t_all-t_av)*1e3:.1f} us = {fl/(t_all-t_av)/1e9:.0f} TF/s")
t_one = timeit(lambda: ops.tied_attention(q, k, v, o5, att, w=w, qscale=0.17, partial_ws=False))
print(f"   one-pass logits kernel instead of the contraction-split one: {t_one*1e3:.1f} us -> logits+softmax {(t_one-t_av)*1e3:.1f} us")
t_nw = timeit(lambda: ops.tied_attention(q, k, v, o5, att, w=None))
print(f"   without position weights (q pre-scaled): {t_nw*1e3:.1f} us -> logits+softmax {(t_nw-t_av)*1e3:.1f} us")
t_sym = timeit(lambda: ops.tied_attention(q, k, v, o5, att, w=w, qscale=0.17, att_sym=sym))
print(f"   with the symmetrised map: {t_sym*1e3:.1f} us")
xn = torch.randn(B, N, L, D, device="cuda").bfloat16()
u = (torch.randn(B, L, H, D, device="cuda") * 0.1).bfloat16()
print(f"position weights (collapsed, MFMA): {timeit(lambda: ops.poswise_collapsed(xn, u, 0.17))*1e3:.1f} us")
wq = (torch.randn(3 * D, D, device="cuda") * 0.05).bfloat16()
bq = torch.randn(3 * D, device="cuda")
hm = torch.empty(B, N, 3 * H, L, dh, device="cuda", dtype=torch.bfloat16)
t_p = timeit(lambda: ops.gemm(xn, wq, hm, B * N * L, 3 * D, D, bias=bq, c_row=(L, 3 * H * L * dh, dh), c_col=(dh, L * dh)))
plain = torch.empty(B, N, L, 3 * D, device="cuda", dtype=torch.bfloat16)
t_pp = timeit(lambda: ops.linear(xn, wq, bq, out=plain))
print(f"q|k|v projection head-major: {t_p*1e3:.1f} us; plain layout: {t_pp*1e3:.1f} us")
# round-1 path for comparison
qkp = (torch.randn(B, N, L, 3 * D, device="cuda") * (0.6 / math.sqrt(N))).bfloat16()
t1 = timeit(lambda: ops.tied_logits_softmax(qkp, qkp[..., D:], N * L * 3 * D, L * 3 * D, 3 * D, att, None, B, H, N, L, dh))
print(f"round-1 logits+softmax on [B,N,L,3D]: {t1*1e3:.1f} us = {fl/t1/1e9:.0f} TF/s")
# whole layer through the model classes
m = R.EncoderLayer(d_msa=D, d_ff=4 * D, n_heads=H, p_dropout=0.0, tied=True, return_att=True).cuda()
x = torch.randn(B, N, L, D, device="cuda")
for v2 in (True, False):
    R.RT.tied_v2 = v2
    print(f"tied encoder layer, tied_v2={v2}: {timeit(lambda: m.run(x, want_att=False), 10)*1e3:.1f} us")
R.RT.tied_v2 = True
