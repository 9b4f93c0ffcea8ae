#!/usr/bin/env python3
"""The fused outer-product kernel alone at the bench shape, a few launches (GPU box; for rocprofv3 --pmc passes on one kernel:
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d out -- python3 tools/outer_only.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
B, N, L, P, Dout = 4, 128, 256, 32, 288
torch.manual_seed(0)
m = R.OuterProductMean(P, Dout).cuda()
xt = torch.randn(B, L, P, N, device="cuda").bfloat16()
yt = (torch.randn(B, L, P, N, device="cuda") * 0.05).bfloat16()
ln2 = R.LayerNorm(Dout).cuda()
feat = torch.empty(B, L, L, 720, device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("ITERS", "3"))):
    m.run_into(xt, yt, ln2, feat, 720)
torch.cuda.synchronize()
print("done")
