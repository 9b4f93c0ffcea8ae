#!/usr/bin/env python3
"""Run-to-run reproducibility of the whole benchmarked forward (config 2, default path: fused feed-forward, hipGraph replay):
N replays and N eager forwards, every output tensor compared bit for bit with the first (GPU box).
    python tools/determinism_forward.py [N=20] [dtype=bf16|fp16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import rosettafold_pytorch_amd as R  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
R.set_compute_dtype(dt)
cfg = bench.CONFIGS[2]
torch.manual_seed(1234)
model = R.RoseTTAFold(p_dropout=0.0, **cfg["model"]).cuda()
inputs = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], seed=0, device=torch.device("cuda", 0))


def flat(out):
    return [out[0][k].clone() for k in sorted(out[0])] + [out[1].clone(), out[2].clone()]


ref = flat(model(*inputs))
g = R.GraphedForward(model, *inputs)
bad = 0
for i in range(n):
    for name, out in (("replay", g(*inputs)), ("eager", model(*inputs))):
        cur = flat(out)
        if not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
            print(f"iteration {i} {name}: differs from the first forward")
torch.cuda.synchronize()
print(f"{dt}: {n} graph replays + {n} eager forwards of config 2 (B=4, N=128, L=256, 8+5 blocks): "
      f"{'all bitwise equal to the first forward' if bad == 0 else str(bad) + ' DIFFER'}")
sys.exit(1 if bad else 0)
