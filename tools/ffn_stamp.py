import os, sys, torch
sys.path.insert(0, "/root/repo")
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
for D, M in ((384, 131072), (288, 262144)):
    hid = 4 * D
    w1 = torch.randn(hid, D, device="cuda") * 0.05; w2 = torch.randn(D, hid, device="cuda") * 0.05
    wp = ops.ffn_pack(w1, w2, torch.bfloat16)
    b1 = torch.zeros(hid, device="cuda"); b2 = torch.zeros(D, device="cuda")
    ln = torch.nn.LayerNorm(D).cuda()
    xn = torch.randn(M, D, device="cuda").bfloat16(); x = torch.randn(M, D, device="cuda")
    for withln in (False, True):
        ops.ffn_fused(xn, wp, b1, b2, x, ln if withln else None); torch.cuda.synchronize()
        t = x.view(-1)[:32].view(torch.int64).cpu().tolist()
        names = ["vmcnt wait", "lgkm+barrier", "issue", "step A", "pack+xchg wait (B head)", "step B body(+loop)", "epilogue"]
        for w, base in ((0, 0), (7, 8)):
            tot = sum(t[base:base + 7])
            print(f"D={D} ln={withln} wave {w}: total {tot/1e3:.0f}k cycles(100MHz ticks?) " + ", ".join(f"{n} {100*v/tot:.1f}%" for n, v in zip(names, t[base:base+7])))
