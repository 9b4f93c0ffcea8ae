#!/usr/bin/env python3
"""Microbenchmark of rf_gemm (bf16 MFMA path) on the forward path's shapes vs torch.matmul
(hipBLASLt) as a calibration point.  Run on the GPU box: python tools/gemm_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosettafold_pytorch_amd import ops, _lib as L  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    shapes = [  # (M, N, K, tag)
        (262144, 1152, 288, "pair FF1"), (262144, 288, 1152, "pair FF2"), (262144, 1024, 288, "pair qk proj"),
        (131072, 1536, 384, "msa FF1"), (131072, 384, 1536, "msa FF2"), (131072, 1152, 384, "msa qkp proj"),
        (262144, 288, 1024, "outer->pair"), (262144, 1536, 288, "pair qkv"), (262144, 288, 512, "pair attn out"),
        (131072, 2304, 384, "msa qkv"), (131072, 384, 768, "msa attn out"), (262144, 288, 288, "pair proj"),
        (8192, 8192, 4096, "square-ish"), (4096, 4096, 4096, "4k cube"),
    ]
    cfgs = [int(c) for c in os.environ.get("CFGS", "0,1,3,13,14,15,16").split(",")]
    only = os.environ.get("ONLY")
    if only:
        shapes = [s for s in shapes if s[3] in only.split(",")]
    for M, N, K, tag in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16()
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t = timeit(lambda: torch.nn.functional.linear(x, w, b.bfloat16()))
        line = f"{tag:14s} M={M:7d} N={N:5d} K={K:5d} torch {fl / t / 1e9:7.1f} TF/s {t * 1e3:6.0f}us |"
        for c in cfgs:
            try:
                t = timeit(lambda: ops.linear(x, w, b, out=out, tile_cfg=c))
                line += f" cfg{c}:{fl / t / 1e9:6.1f} ({t * 1e3:4.0f}us)"
            except Exception as ex:  # noqa: BLE001
                line += f" cfg{c}:ERR"
        print(line, flush=True)


if __name__ == "__main__":
    main()
