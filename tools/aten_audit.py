#!/usr/bin/env python3
"""Which PyTorch (ATen) operators still launch GPU work inside one timed forward of the bench model, and from which source
lines (run on the GPU box).  The product path is librfmi.so kernels; anything listed here is a leftover to replace."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from torch.profiler import profile, ProfilerActivity

sys.argv = [sys.argv[0]]
import bench
cfg = bench.CONFIGS[2]
R.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = R.RoseTTAFold(p_dropout=0.0, **cfg["model"]).cuda()
inputs = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], 0, "cuda")
with torch.no_grad():
    model(*inputs); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(*inputs); torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    where = next((s for s in ev.stack if "rosettafold-pytorch_amd" in s or "rosettafold_pytorch_amd" in s), "?")
    r = rows.setdefault((ev.name, where), [0, 0.0])
    r[0] += 1
    r[1] += ev.device_time_total
for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n:5d} x {us/1e3:8.3f} ms  {name:28s} {where}")
print("--- copies (aten::copy_ / clone / contiguous) by source line, with or without device time recorded")
cp = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        where = next((s for s in ev.stack if "rosettafold-pytorch_amd" in s or "rosettafold_pytorch_amd" in s), "?")
        r = cp.setdefault((ev.name, where), [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total
for (name, where), (n, us) in sorted(cp.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{n:5d} x {us/1e3:8.3f} ms  {name:20s} {where}")
