#!/usr/bin/env python3
"""Stress test of the fused FAVOR+ kernel (GPU box): every shipped variant (feature map x sequence length x 16-bit build),
N launches each (default 200) on the same inputs, every launch compared bit for bit with the first AND every row checked
against an independent fp32 formula of the same attention (host = PyTorch on the GPU, fp32 operands from the 16-bit inputs):
a launch whose rows are merely "reproducibly wrong" would pass the first check and fail the second.

    python tools/determinism_favor.py [launches]
"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)

def reference(qkv, pc, Lo, Ls, H, gen):
    """D^-1 q'(k'^T v) with the kernel's feature maps in fp32 (pc carries d^-1/4, and log2 e for the softmax features)."""
    x = qkv.float().view(Lo, Ls, 3, H, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [Lo, H, Ls, 64]
    P = pc.float()[:266]
    if gen:
        fq, fk = torch.relu(q @ P.t()) + 1e-3, torch.relu(k @ P.t()) + 1e-3
    else:
        c = 0.0625 * 1.4426950408889634                                    # |x|^2 / (2 sqrt d) in log2 units
        dq, dk = q @ P.t(), k @ P.t()
        fq = torch.exp2(dq - (q * q).sum(-1, keepdim=True) * c - dq.amax(-1, keepdim=True)) + 1e-4
        fk = torch.exp2(dk - (k * k).sum(-1, keepdim=True) * c - dk.amax((-1, -2), keepdim=True)) + 1e-4
    ctx = fk.transpose(-1, -2) @ v
    den = fq @ fk.sum(-2, keepdim=True).transpose(-1, -2)
    return ((fq @ ctx) / den).permute(0, 2, 1, 3).reshape(Lo * Ls, H * 64)

bad = 0
for dt in (torch.bfloat16, torch.float16):
    R.set_compute_dtype(dt)
    for gen in (True, False):
        for Ls, Lo, H, D in ((64, 1024, 8, 288), (128, 1024, 12, 384), (256, 1024, 8, 288)):
            m = R.PerformerSelfAttention(dim=D, heads=H, generalized_attention=gen).cuda()
            inner, W3 = 64 * H, 3 * 64 * H
            qkv = torch.randn(Lo * Ls, W3, device="cuda").to(dt)
            pc = m.proj_scaled(log2e=not gen)
            def f():
                o = torch.empty(Lo * Ls, inner, device="cuda", dtype=dt)
                ops.favor_attention(qkv, pc, o, (Lo * Ls * W3, Ls * W3, W3, 64), (Lo * Ls * inner, Ls * inner, inner), 0, inner, 2 * inner, 1,
                                    Lo, H, Ls, 64, 266, not gen, 1e-3 if gen else 1e-4)
                return o
            first = f()
            ref = reference(qkv, pc, Lo, Ls, H, gen)
            scale = ref.abs().max()
            tol = (4e-2 if dt == torch.bfloat16 else 6e-3) * scale
            row_err = (first.float() - ref).abs().amax(1)
            ndiff = 0
            for _ in range(N - 1):
                o = f()
                ndiff += int(not torch.equal(o, first))
            torch.cuda.synchronize()
            nwrong = int((row_err > tol).sum())
            bad += ndiff + nwrong
            print(f"{str(dt).split('.')[-1]:8s} {'relu' if gen else 'softmax':7s} Ls={Ls:3d} ({Lo * H} items): {N} launches, {ndiff} differ from the first; "
                  f"rows off the fp32 formula by more than {tol.item() / scale.item():.0e} of the range: {nwrong} (max {row_err.max().item() / scale.item():.2e})", flush=True)
R.set_compute_dtype(torch.bfloat16)
print("RESULT:", "clean" if bad == 0 else f"{bad} problems")
sys.exit(1 if bad else 0)
