"""Streaming kernels on a cache-resident tensor and on a working set larger than the 256 MB Infinity Cache (GPU box):
    python tools/bw_probe.py
One [1, 256, 256, 288] fp32 picture (75 MB) in a hot loop vs. eight of them round-robin (600 MB): torch's own copy / reduction as
the reference points, then the library's conditioning and LayerNorm kernels.  Prints microseconds per call and the effective GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R  # noqa: E402,F401
from rosettafold_pytorch_amd import ops  # noqa: E402

NSET = 8
xs = [torch.randn(1, 256, 256, 288, device="cuda") for _ in range(NSET)]
y16 = [torch.empty_like(x, dtype=torch.bfloat16) for x in xs]
y32 = [torch.empty_like(x) for x in xs]
mean = ops.channel_mean(xs[0])
g, b_ = torch.ones(288, device="cuda"), torch.zeros(288, device="cuda")
NB = xs[0].numel() * 4


def t(fn, cold, n=32):
    idx = (lambda i: i % NSET) if cold else (lambda i: 0)
    for i in range(8):
        fn(idx(i))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for i in range(n):
        fn(idx(i))
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


CASES = [
    ("torch copy fp32->bf16", lambda i: y16[i].copy_(xs[i]), 1.5),
    ("torch copy fp32->fp32", lambda i: y32[i].copy_(xs[i]), 2.0),
    ("torch sum over pixels", lambda i: xs[i].sum(dim=(1, 2)), 1.0),
    ("rf_center_apply -> 16 bit", lambda i: ops.center_apply(xs[i], mean, out=y16[i]), 1.5),
    ("rf_channel_mean", lambda i: ops.channel_mean(xs[i]), 1.0),
    ("rf_axpby cast -> 16 bit", lambda i: ops.axpby(xs[i], 1.0, None, 0.0, y16[i]), 1.5),
    ("rf_layernorm -> 16 bit", lambda i: ops.layernorm(xs[i], g, b_, out=y16[i]), 1.5),
]
for name, fn, passes in CASES:
    hot, cold = t(fn, False), t(fn, True)
    print(f"{name:28s} hot {hot:7.1f} us ({passes * NB / hot / 1e3:6.0f} GB/s)   cold {cold:7.1f} us ({passes * NB / cold / 1e3:6.0f} GB/s)")
