#!/usr/bin/env python3
"""Bitwise reproducibility of DENSELY queued forwards (GPU box): K forwards of the config-2 model back to back with no host
synchronisation in between (the way bench.py times them), each compared with the first.  A race between consecutive kernels
or inside a persistent kernel shows up here and not in a forward that is synchronised block by block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rosettafold_pytorch_amd as R
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = bench.CONFIGS[2]
torch.manual_seed(1234)
model = R.RoseTTAFold(**dict(cfg["model"], p_dropout=0.0)).cuda().eval()
inp = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], 0, "cuda")
for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    R.set_compute_dtype(dt)
    model(*inp); torch.cuda.synchronize()
    outs = [model(*inp) for _ in range(K)]
    torch.cuda.synchronize()
    a = outs[0]
    for i, o in enumerate(outs[1:], 1):
        d = {k: (o[0][k] != a[0][k]).float().mean().item() for k in a[0]}
        d["xyz"] = (o[1] != a[1]).float().mean().item()
        print(f"{name}: forward {i} vs 0: fraction of differing elements {d}")
R.set_compute_dtype(torch.bfloat16)
