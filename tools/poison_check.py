#!/usr/bin/env python3
"""Uninitialised-read screen (run on the GPU box): poison the caching allocator's free blocks with NaN / huge values
between two forwards of the same model and inputs; any output change means some kernel reads memory it (or its producer)
never wrote.  Also compares two configurations (small unfused paths, fused bench paths)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R

def poison(val):
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    chunks = []
    torch.cuda.empty_cache()
    for _ in range(6):
        t = torch.empty(512 * 1024 * 1024 // 4, device="cuda", dtype=torch.float32)
        t.fill_(val)
        chunks.append(t)
    torch.cuda.synchronize()
    del chunks  # blocks return to the caching allocator, still holding `val`

def run(cfg, B, N, L, tag):
    torch.manual_seed(1234)
    m = R.RoseTTAFold(**cfg).cuda()
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (B, N, L), generator=g).cuda()
    seq, aa = msa[:, 0].clone(), torch.arange(L).unsqueeze(0).repeat(B, 1).cuda()
    outs = []
    for val in (0.0, float("nan"), 3.0e30, -7.5):
        poison(val)
        lg, xyz, pl = m(msa, seq, aa)
        torch.cuda.synchronize()
        outs.append({**{k: v.clone() for k, v in lg.items()}, "xyz": xyz.clone(), "plddt": pl.clone()})
    ok = True
    for i, o in enumerate(outs[1:], 1):
        for k in o:
            same = torch.equal(o[k], outs[0][k])
            if not same:
                ok = False
                d = (o[k].float() - outs[0][k].float())
                print(f"[{tag}] poison #{i}: {k} differs: nan={torch.isnan(o[k]).any().item()} max|d|={d[~torch.isnan(d)].abs().max().item() if (~torch.isnan(d)).any() else float('nan')}")
    print(f"[{tag}] {'CLEAN: outputs independent of stale memory' if ok else 'UNINITIALISED READ DETECTED'}")
    return ok

small = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
             n_encoder_layers=1, max_len=64, n_neighbors=[16, 16], p_dropout=0.0)
bench = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=1, n_three_track_blocks=2,
             n_encoder_layers=1, max_len=260, n_neighbors=[128, 64], p_dropout=0.0)
a = run(small, 2, 8, 32, "small dims (unfused paths)")
b = run(bench, 1, 128, 256, "bench dims (fused paths)")
sys.exit(0 if (a and b) else 1)
