#!/usr/bin/env python3
"""Where do two runs of the fused FAVOR+ kernel differ?  (run on the GPU box)"""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
torch.manual_seed(0)
gen = os.environ.get("FV_GEN", "1") == "1"   # 1: ReLU features, 0: softmax features
Ls = int(os.environ.get("FV_LS", "128"))
Lo, H, D = 1024, (12 if Ls == 128 else 8), (384 if Ls == 128 else 288)
NRUN = int(os.environ.get("FV_RUNS", "24"))
m = R.PerformerSelfAttention(dim=D, heads=H, generalized_attention=gen).cuda()
inner = 64 * H; W3 = 3 * inner
qkv = torch.randn(Lo * Ls, W3, device="cuda").bfloat16()
pc = m.proj_scaled(log2e=not gen)
def f():
    o = torch.empty(Lo * Ls, inner, device="cuda", dtype=torch.bfloat16)
    ops.favor_attention(qkv, pc, o, (Lo * Ls * W3, Ls * W3, W3, 64), (Lo * Ls * inner, Ls * inner, inner), 0, inner, 2 * inner, 1, Lo, H, Ls, 64, 266, not gen, 1e-3 if gen else 1e-4)
    return o
outs = [f() for _ in range(NRUN)]
torch.cuda.synchronize()
a = outs[0].view(Lo, Ls, H, 64)
for k, o in enumerate(outs[1:]):
    b = o.view(Lo, Ls, H, 64)
    d = (a != b)
    print(f"run {k+1}: differing elements {int(d.sum())}")
    if not d.any():
        continue
    items = d.any(-1).any(1)          # [Lo, H]: items with a difference
    idx = items.nonzero()
    print("  items differing:", idx.shape[0], "of", Lo * H, " first:", idx[:12].tolist())
    item_id = idx[:, 0] * H + idx[:, 1]
    print("  item index mod 256 histogram (top):", collections.Counter((item_id % 256).tolist()).most_common(8))
    print("  item index // 256 histogram (top):", collections.Counter((item_id // 256).tolist()).most_common(8))
    rows = d.any(-1)                   # [Lo, Ls, H]
    srow = rows.permute(0, 2, 1)[items]   # [n items, Ls]
    print("  rows differing per item (mean):", float(srow.float().sum(1).mean()), " row histogram by 16-row tile:", srow.view(-1, Ls // 16, 16).any(-1).float().mean(0).tolist())
    dd = d.permute(0, 2, 1, 3)[items]     # [n, Ls, 64]
    print("  columns (d) differing fraction by 16-col tile:", dd.view(dd.shape[0], Ls, 4, 16).any(-1).any(1).float().mean(0).tolist())
    mag = (a.float() - b.float()).abs()
    print("  max |diff|", float(mag.max()), " rel to |a| max", float(mag.max() / a.float().abs().max()))
