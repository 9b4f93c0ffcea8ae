"""Phase stamps of csrc/outer_pairs.hip (library built with -DOP_STAMP, RFMI_LIB=...): share of every phase in waves 0 and 7."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
B, N, L, P, Dout = 4, 128, 256, 32, 288
torch.manual_seed(0)
m = R.OuterProductMean(P, Dout).cuda()
xt = torch.randn(B, L, P, N, device="cuda").bfloat16()
yt = (torch.randn(B, L, P, N, device="cuda") * 0.05).bfloat16()
o = m.run(xt, yt, N); torch.cuda.synchronize()
o = m.run(xt, yt, N); torch.cuda.synchronize()
t = o.view(-1)[:32].view(torch.int64).cpu().tolist()
names = ["vmcnt wait", "lgkm + barrier", "prep", "MFMA interval (26 MFMAs + 3 DMAs)", "statistics + pack + exchange write", "epilogue", "first fragments landed (issue 8 reads, wait for the partner fragment)"]
for w, base in ((0, 0), (7, 8)):
    tot = sum(t[base:base + 7])
    print(f"wave {w}: total {tot / 1e3:.0f}k cycles: " + ", ".join(f"{n} {100 * v / tot:.1f}%" for n, v in zip(names, t[base:base + 7])))
