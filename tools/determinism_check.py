#!/usr/bin/env python3
"""Run the bench model twice on the same inputs: any race in a kernel shows up as run-to-run differences (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rosettafold_pytorch_amd as R
cfg = bench.CONFIGS[2]
torch.manual_seed(1234)
model = R.RoseTTAFold(**dict(cfg["model"], p_dropout=0.0)).cuda().eval()
msa, seq, aa = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], 0, "cuda")
outs = []
with torch.no_grad():
    for _ in range(3):
        lg, xyz, pl = model(msa, seq, aa)
        outs.append((lg["dist"].float().clone(), xyz.clone(), pl.clone()))
torch.cuda.synchronize()
for i in (1, 2):
    d = [(a - b).abs().max().item() for a, b in zip(outs[0], outs[i])]
    print(f"run 0 vs run {i}: max |diff| dist-logits {d[0]:.3e}, xyz {d[1]:.3e}, plddt {d[2]:.3e};  finite: {all(torch.isfinite(t).all().item() for t in outs[i])}")
