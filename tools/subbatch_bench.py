#!/usr/bin/env python3
"""Does running the batch in sub-batches (smaller working set -> more Infinity Cache hits) beat one B=4 forward?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rosettafold_pytorch_amd as R
cfg = bench.CONFIGS[2]
torch.manual_seed(1234)
model = R.RoseTTAFold(**dict(cfg["model"], p_dropout=0.0)).cuda().eval()
msa, seq, aa = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], 0, "cuda")
def run(sb):
    with torch.no_grad():
        for b0 in range(0, cfg["B"], sb):
            model(msa[b0:b0 + sb], seq[b0:b0 + sb], aa[b0:b0 + sb])
for sb in (4, 2, 1, 4):
    run(sb); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2): run(sb)
    torch.cuda.synchronize()
    print(f"sub-batch {sb}: {(time.time() - t0) / 2 * 1e3:.1f} ms per 4 samples", flush=True)
