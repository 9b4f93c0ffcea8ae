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
    # head-major q|k|v tiles: [Lo, 3H, B, Ls, 64] (every item's K / V / Q tile is one contiguous Ls x 128-byte block)
    qh = torch.randn(Lo, 3 * H, B, Ls, 64, device="cuda").bfloat16()
    fh = lambda: ops.favor_attention(qh, pc, o, (Ls * 64, 3 * H * B * Ls * 64, 64, B * Ls * 64), (RB * inner, so * inner, ss * inner), 0, H * B * Ls * 64, 2 * H * B * Ls * 64, B, Lo, H, Ls, 64, 266, not gen, 1e-3 if gen else 1e-4)
    t = timeit(fh)
    print(f"   head-major q|k|v tiles: {t*1e3:.0f} us, {fl/t/1e9:.0f} TF/s", flush=True)
    if int(os.environ.get("RF_FAVOR_DBG", "0")) & 8:
        import ctypes as C
        buf = (C.c_ulonglong * 7)()
        ops.lib.rf_favor_phase_cycles.argtypes = [C.c_void_p, C.c_int]
        ops.lib.rf_favor_phase_cycles(None, 1)
        f(); torch.cuda.synchronize()
        ops.lib.rf_favor_phase_cycles(buf, 1)
        n = max(buf[6], 1)
        names = ["wait K/V", "phase A", "publish + barrier", "prefetch issue + combine", "phase B", "stores"]
        print("   cycles per item (wave 0): " + ", ".join(f"{nm} {buf[i]/n:.0f}" for i, nm in enumerate(names)) + f"  [{n} items]")
