#!/usr/bin/env python3
"""BASELINE.json configs[4]: structure-module-only microbench (CoordUpdateWithMsaAndPair = node/edge embeddings, kNN
graph, SE(3)-Transformer, coordinate update) at B=8, L=256, k=128.  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
B, N, L, k = 8, 128, 256, 128
torch.manual_seed(0)
m = R.CoordUpdateWithMsaAndPair(384, 288, 32, 32, 32, n_neighbors=k, p_dropout=0.0).cuda().eval()
g = torch.Generator().manual_seed(3)
steps = torch.randn(B, L, 3, generator=g)
ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
xyz = (ca[:, :, None, :] + 0.5 * torch.randn(B, L, 3, 3, generator=g)).cuda()
msa = torch.randn(B, N, L, 384, generator=g).cuda()
pair = torch.randn(B, L, L, 288, generator=g).cuda()
oh = torch.nn.functional.one_hot(torch.randint(0, 21, (B, L), generator=g), 21).float().cuda()
aa = torch.arange(L).unsqueeze(0).repeat(B, 1).cuda()
with torch.no_grad():
    for _ in range(2):
        m.run(xyz, msa, pair, aa, oh)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        m.run(xyz, msa, pair, aa, oh)
    torch.cuda.synchronize()
dt = (time.time() - t0) / 5
print(f"structure module, B={B} L={L} k={k}: {dt * 1e3:.1f} ms per call, {B * L / dt:.0f} residues/s")
