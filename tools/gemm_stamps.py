#!/usr/bin/env python3
"""Per-workgroup phase timeline of one rf_gemm launch (timing experiment; run on the GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosettafold_pytorch_amd import ops, _lib as L  # noqa: E402


def main():
    M, N, K = [int(v) for v in os.environ.get("MNK", "262144,1536,288").split(",")]
    cfg = int(os.environ.get("CFG", "0"))
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    nblk = 65536
    stamps = torch.zeros(nblk, 8, device="cuda", dtype=torch.int64)
    if os.environ.get("FAST"):
        # LN=1: the residual + next-LayerNorm epilogue variant (needs a library built with -DFAST_STAMP_LN, loaded via RFMI_LIB)
        if os.environ.get("LN"):
            res = torch.randn(M, N, device="cuda")
            xn = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            g, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
            bias = torch.zeros(N, device="cuda")
            run = lambda: ops.linear(x, w, bias, out=res, residual=res, ln=(xn, g, b, 1e-5))
        else:
            run = lambda: ops.linear(x, w, None, out=out)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        L.lib.rf_debug_gemm_fast_stamps(stamps.data_ptr())
        run()
        torch.cuda.synchronize()
        L.lib.rf_debug_gemm_fast_stamps(None)
        s = stamps.cpu().numpy()
        s = s[s[:, 0] != 0].astype(np.float64)
        nt = s[:, 5].mean()
        print(f"persistent kernel: {len(s)} workgroups, {nt:.1f} tiles each, total {s[:, 0].mean():.0f} cycles = {s[:, 0].mean() / nt:.0f} per tile")
        for name, col in zip(["vmcnt wait", "barrier", "K-step body", "epilogue"], [1, 2, 3, 4]):
            print(f"  {name:12s} {s[:, col].mean() / nt:8.0f} cycles/tile  ({100 * s[:, col].mean() / s[:, 0].mean():.1f} %)")
        return
    for _ in range(3):
        ops.linear(x, w, None, out=out, tile_cfg=cfg)
    torch.cuda.synchronize()
    L.lib.rf_debug_gemm_stamps(stamps.data_ptr())
    ops.linear(x, w, None, out=out, tile_cfg=cfg)
    torch.cuda.synchronize()
    L.lib.rf_debug_gemm_stamps(None)
    s = stamps.cpu().numpy()
    s = s[s[:, 0] != 0]
    t00 = s[:, 0].min()
    t = (s[:, [0, 1, 2, 3, 6]] - t00) * 0.01  # us
    print(f"blocks {len(s)}  kernel span {t[:, 4].max():.1f} us")
    d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]], 1)
    for name, col in zip(["entry->first tile", "K loop", "epilogue issue", "store ack"], d.T):
        print(f"{name:18s} mean {col.mean():6.2f}  p10 {np.percentile(col, 10):6.2f}  p50 {np.percentile(col, 50):6.2f}  p90 {np.percentile(col, 90):6.2f} us")
    print(f"epilogue split: LDS staging (incl. barriers) mean {(s[:, 7] >> 32).mean() * 0.01:.2f} us, store loops mean {(s[:, 7] & 0xffffffff).mean() * 0.01:.2f} us")
    # concurrency: how many workgroups are inside their epilogue at a given instant (lockstep => bimodal)
    ts = np.linspace(t[:, 4].max() * 0.2, t[:, 4].max() * 0.8, 400)
    cnt = np.array([np.sum((t[:, 2] <= x) & (x < t[:, 3])) for x in ts])
    print(f"workgroups in epilogue at an instant: mean {cnt.mean():.0f} p10 {np.percentile(cnt, 10):.0f} p50 {np.percentile(cnt, 50):.0f} p90 {np.percentile(cnt, 90):.0f} max {cnt.max()}")
    # per-CU timeline: gap between one block's last stamp and the next block's entry on the same CU
    hw, xcc = s[:, 4], s[:, 5] & 0xF
    cu = (xcc << 16) | (hw & 0xFF00)  # cu_id, sh_id, se_id bits
    gaps, per_cu = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        per_cu.append(len(idx))
        gaps += list(t[idx[1:], 0] - t[idx[:-1], 4])
    gaps = np.array(gaps)
    print(f"CUs seen {len(per_cu)}, blocks per CU {min(per_cu)}..{max(per_cu)}")
    print(f"exit->next entry on the same CU: mean {gaps.mean():.2f} p10 {np.percentile(gaps, 10):.2f} p50 {np.percentile(gaps, 50):.2f} p90 {np.percentile(gaps, 90):.2f} us")
    print(f"per-block total (entry->ack) mean {(t[:, 4] - t[:, 0]).mean():.2f} us")


if __name__ == "__main__":
    main()
