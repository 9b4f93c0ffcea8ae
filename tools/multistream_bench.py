#!/usr/bin/env python3
"""Do independent sub-batches on separate HIP streams overlap usefully (tail effects, memory-bound next to MFMA-bound
kernels) against one B=4 forward on one stream?  (run on the GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rosettafold_pytorch_amd as R
cfg = bench.CONFIGS[2]
torch.manual_seed(1234)
model = R.RoseTTAFold(**dict(cfg["model"], p_dropout=0.0)).cuda().eval()
msa, seq, aa = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], 0, "cuda")
B = cfg["B"]


def run(nstreams, streams):
    sb = B // nstreams
    outs = []
    cur = torch.cuda.current_stream()
    for k in range(nstreams):
        s = streams[k]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(model(msa[k * sb:(k + 1) * sb], seq[k * sb:(k + 1) * sb], aa[k * sb:(k + 1) * sb]))
    for s in streams[:nstreams]:
        cur.wait_stream(s)
    return outs


streams = [torch.cuda.Stream() for _ in range(4)]
ref = model(msa, seq, aa)
torch.cuda.synchronize()
for ns in (1, 2, 4, 1):
    o = run(ns, streams); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        o = run(ns, streams)
    torch.cuda.synchronize()
    d = torch.cat([x[0]["dist"] for x in o])
    print(f"{ns} stream(s) x B={B // ns}: {(time.time() - t0) / 2 * 1e3:.1f} ms per {B} samples; max |dist - single| = {(d - ref[0]['dist']).abs().max().item():.3e}", flush=True)
