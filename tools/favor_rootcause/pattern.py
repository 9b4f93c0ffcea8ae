"""What do the wrong rows of the non-reproducible FAVOR+ kernel (libfvx0.so) look like against the shipped kernel's rows?
Per wrong (item, wave) block: is it a per-row SCALING of the correct rows (wrong denominator), are all 16 rows affected, which
items / waves / launch positions are hit."""
import ctypes as C, os, sys
from collections import Counter
import torch
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE)
import run as RUN
H, Lo, Ls = 12, 1024, 128
inner, W3 = 64 * H, 3 * 64 * H
softmax = (sys.argv[1] if len(sys.argv) > 1 else "softmax") == "softmax"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = RUN.load("libfvx0.so"); lib.rf_favor_exp_set_dbgbuf(None)
torch.manual_seed(0)
qkv = torch.randn(Lo * Ls, W3, device="cuda").bfloat16()
pc = RUN.proj(not softmax, softmax)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from rosettafold_pytorch_amd import ops
eps = 1e-4 if softmax else 1e-3
xs = RUN.I64x4(Lo * Ls * W3, Ls * W3, W3, 64); os_ = RUN.I64x3(Lo * Ls * inner, Ls * inner, inner)
ref = torch.empty(Lo * Ls, inner, device="cuda", dtype=torch.bfloat16)
ops.favor_attention(qkv, pc, ref, (Lo * Ls * W3, Ls * W3, W3, 64), (Lo * Ls * inner, Ls * inner, inner), 0, inner, 2 * inner, 1, Lo, H, Ls, 64, 266, softmax, eps)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
R = ref.view(Lo, 8, 16, H, 64).float()
waves, firsts, nrows, kinds = Counter(), Counter(), Counter(), Counter()
shown = 0
for it in range(launches):
    o = torch.empty(Lo * Ls, inner, device="cuda", dtype=torch.bfloat16)
    assert lib.rf_favor_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(pc.data_ptr()), C.c_void_p(o.data_ptr()), C.byref(xs), C.byref(os_), 0, inner, 2 * inner, 1, Lo, H, Ls, 64, 266, 1 if softmax else 0, eps, stream) == 0
    torch.cuda.synchronize()
    O = o.view(Lo, 8, 16, H, 64).float()
    bad_rows = (O != R).any(-1)                      # [o, wave, row, h]
    blocks = bad_rows.any(2).nonzero().tolist()      # (o, wave, h)
    for oo, w_, hh in blocks:
        item = oo * H + hh
        waves[w_] += 1
        firsts["first item of its workgroup" if item < 256 else "later item"] += 1
        nr = int(bad_rows[oo, w_, :, hh].sum()); nrows[nr] += 1
        ob, gd = O[oo, w_, :, hh], R[oo, w_, :, hh]
        rows = bad_rows[oo, w_, :, hh]
        ratio = ob[rows] / gd[rows]
        big = gd[rows].abs() > 0.05 * gd.abs().max()
        spread = torch.stack([(r[b].max() - r[b].min()) if b.any() else torch.tensor(0., device=r.device) for r, b in zip(ratio, big)])
        rel = ((ob[rows] - gd[rows]).abs().amax(1) / gd[rows].abs().amax(1))
        kind = "row-wise scaling (ratio spread < 1/4 of the deviation)" if (spread < 0.25 * rel).all() else "not a scaling"
        kinds[kind] += 1
        if shown < 8:
            shown += 1
            print(f"launch {it} item {item} (pos {item // 256} in its workgroup) wave {w_}: {nr}/16 rows differ, max rel dev {rel.max():.3e}, "
                  f"ratio spread {spread.max():.3e} -> {kind}; mean ratio of row 0: {ratio[0][big[0]].mean():.4f}")
print("waves:", dict(waves)); print("position:", dict(firsts)); print("rows per block:", dict(nrows)); print("kind:", dict(kinds))
