import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
torch.manual_seed(0)
for name, gen, B, L1, L2, D, H, axis in [("pair row", True, 4, 256, 256, 288, 8, 1), ("pair col", True, 4, 256, 256, 288, 8, 2), ("msa col", False, 4, 128, 256, 384, 12, 1)]:
    m = R.PerformerSelfAttention(dim=D, heads=H, generalized_attention=gen).cuda()
    inner = 64 * H
    qkv = torch.randn(B * L1 * L2, 3 * inner, device="cuda").bfloat16()
    o = torch.empty(B * L1 * L2, inner, device="cuda", dtype=torch.bfloat16)
    Ls, Lo = (L1, L2) if axis == 1 else (L2, L1)
    ss, so = (L2, 1) if axis == 1 else (1, L2)
    RB, W3 = L1 * L2, 3 * inner
    pc = m.proj_scaled(log2e=not gen)
    f = lambda: ops.favor_attention(qkv, pc, o, (RB * W3, so * W3, ss * W3, 64), (RB * inner, so * inner, ss * inner), 0, inner, 2 * inner, B, Lo, H, Ls, 64, 266, not gen, 1e-3 if gen else 1e-4)
    t = timeit(f)
    fl = B * Lo * H * (4 * 2 * Ls * 64 * 266)
    print(f"{name}: {t*1e3:.0f} us, {fl/t/1e9:.0f} TF/s (algorithmic, 266 features)", flush=True)
