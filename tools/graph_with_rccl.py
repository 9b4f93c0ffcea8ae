#!/usr/bin/env python3
"""hipGraph capture of the forward while an RCCL process group (and its watchdog thread) is alive -- the situation of every rank
of the multi-GPU bench (GPU box; a one-rank "nccl" group is what one GPU can host)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29733")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench  # noqa: E402
import rosettafold_pytorch_amd as R  # noqa: E402

t = torch.ones(4, device="cuda")
dist.all_reduce(t)          # the communicator exists, the watchdog has work to poll
dist.barrier()
cfg = bench.CONFIGS[1]
torch.manual_seed(1234)
model = R.RoseTTAFold(p_dropout=0.0, **cfg["model"]).cuda()
inputs = bench.make_inputs(cfg["B"], cfg["N"], cfg["L"], seed=0, device=torch.device("cuda", 0))
eager = model(*inputs)
g = R.GraphedForward(model, *inputs)
for _ in range(3):
    out = g(*inputs)
    dist.all_reduce(t)      # collectives between replays, as bench.py's step() issues them
torch.cuda.synchronize()
same = all(torch.equal(out[0][k], eager[0][k]) for k in eager[0]) and torch.equal(out[1], eager[1]) and torch.equal(out[2], eager[2])
print("capture + 3 replays with a live RCCL process group: replay == eager bit for bit:", same)
dist.barrier()
dist.destroy_process_group()
assert same
