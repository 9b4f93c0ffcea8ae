import os, sys, torch
sys.path.insert(0, "/root/repo")
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for D, M in ((384, 131072), (384, 32768), (288, 262144)):
    for hid in (D, 2 * D, 4 * D, 8 * D):
        w1 = torch.randn(hid, D, device="cuda") * 0.05; w2 = torch.randn(D, hid, device="cuda") * 0.05
        wp = ops.ffn_pack(w1, w2, torch.bfloat16)
        b1 = torch.zeros(hid, device="cuda"); b2 = torch.zeros(D, device="cuda")
        xn = torch.randn(M, D, device="cuda").bfloat16(); x = torch.randn(M, D, device="cuda")
        t = timeit(lambda: ops.ffn_fused(xn, wp, b1, b2, x, None))
        print(f"D={D} M={M} hidden={hid}: {t:.0f} us  ({M // 128} tiles, {4 + hid // 16} steps per tile)", flush=True)
