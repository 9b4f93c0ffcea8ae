#!/usr/bin/env python3
"""Is the forward identical on a non-default HIP stream?  (run on the GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rosettafold_pytorch_amd as R
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = bench.CONFIGS[which]
torch.manual_seed(1234)
model = R.RoseTTAFold(**dict(cfg["model"], p_dropout=0.0)).cuda().eval()
msa, seq, aa = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], 0, "cuda")
def fwd(stream=None, sync=True):
    if stream is None:
        o = model(msa, seq, aa)
    else:
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            o = model(msa, seq, aa)
        torch.cuda.current_stream().wait_stream(stream)
    if sync:
        torch.cuda.synchronize()
    return o
def diff(a, b):
    return {k: (a[0][k] - b[0][k]).abs().max().item() for k in a[0]} | {"xyz": (a[1] - b[1]).abs().max().item()}
s1 = torch.cuda.Stream()
a = fwd(); b = fwd()
print("default vs default:", diff(a, b))
c = fwd(s1); d = fwd(s1)
print("side vs side      :", diff(c, d))
print("default vs side   :", diff(a, c))
