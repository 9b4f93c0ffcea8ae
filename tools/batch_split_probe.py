"""Does the config-2 forward run faster as one batch of 4 or as smaller batches (whose tensors fit the 256 MB Infinity Cache)?
(GPU box)   python tools/batch_split_probe.py
One hipGraph per batch size at config-2 dimensions (N=128, L=256, 8+5 blocks), 10 replays each; prints ms per forward and per sample."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rosettafold_pytorch_amd as R  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = R.RoseTTAFold(**bench.CONFIGS[2]["model"]).to(dev)
N, L = 128, 256
for B in (1, 2, 4):
    g = torch.Generator().manual_seed(B)
    msa = torch.randint(0, 21, (B, N, L), generator=g)
    inputs = (msa.to(dev), msa[:, 0].clone().to(dev), torch.arange(L).repeat(B, 1).to(dev))
    with torch.no_grad():
        model(*inputs)
        graphed = R.GraphedForward(model, *inputs)
        for _ in range(2):
            graphed(*inputs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            graphed(*inputs)
        torch.cuda.synchronize()
    ms = 1e2 * (time.perf_counter() - t0)
    print(f"B={B}: {ms:8.2f} ms per forward, {ms / B:8.2f} ms per sample, {B * L / ms * 1e3:8.1f} residues/s", flush=True)
    del graphed
