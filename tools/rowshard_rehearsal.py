#!/usr/bin/env python3
"""Pair-track row-block sharding at configs[3] size on the one-GPU box: ONE sample (N = 64, L = 1024, d_msa 384, d_pair 288,
1 two-track + 1 three-track + final block) over WORLD ranks that share cuda:0 (gloo rendezvous: RCCL wants one GPU per rank, so
this rehearses the code path and the numerics, not the speed).  Prints per-rank wall time, peak memory, and the logit maps'
rel-L2 against the single-process forward of the same mode.
    python tools/rowshard_rehearsal.py [world=2] [L=1024]"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFG = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=1, n_three_track_blocks=1,
           n_encoder_layers=1, max_len=1100, n_neighbors=[32], p_dropout=0.0)
N = 64


def inputs(L):
    g = torch.Generator().manual_seed(31)
    msa = torch.randint(0, 21, (1, N, L), generator=g)
    return msa, msa[:, 0].clone(), torch.arange(L).unsqueeze(0)


def worker(rank, world, port, ckpt, opath, L):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    model = R.RoseTTAFold(**CFG)
    R.load_checkpoint(model, ckpt)
    model = model.to("cuda:0")
    shard.forward_row_sharded(model, *inputs(L))  # warm-up: weight copies
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    logits, xyz, plddt = shard.forward_row_sharded(model, *inputs(L))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rank {rank}: rows {shard.shard_range(L, world, rank)}, forward {dt * 1e3:.0f} ms wall (ranks share one GPU; exchanges through "
          f"the host), peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
    torch.save({k: v.cpu() for k, v in logits.items()}, f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    import rosettafold_pytorch_amd as R
    ckpt, opath = "/tmp/rowshard_model.pt", "/tmp/rowshard_out.pt"
    torch.manual_seed(84)
    model = R.RoseTTAFold(**CFG)
    R.save_checkpoint(model, ckpt)
    if world == 1:   # in this process (a one-rank group): the sharded code path without exchanges, profilable with rocprofv3
        worker(0, 1, 29611, ckpt, opath, L)
    else:
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=worker, args=(r, world, 29611, ckpt, opath, L)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
            assert p.exitcode == 0
    outs = [torch.load(f"{opath}.{r}") for r in range(world)]
    model = model.to("cuda:0")
    ins = [t.cuda() for t in inputs(L)]
    model(*ins)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = model(*ins)[0]
    torch.cuda.synchronize()
    print(f"single process: forward {1e3 * (time.perf_counter() - t0):.0f} ms")
    for k in ref:
        got = torch.cat([o[k] for o in outs], 1)
        a, b = got.double(), ref[k].cpu().double()
        print(f"  {k}: {tuple(got.shape)} sharded vs single process (bf16 mode both) rel-L2 {((a - b).norm() / b.norm()).item():.3e}, "
              f"argmax agreement {(a.argmax(-1) == b.argmax(-1)).double().mean().item():.4f}")


if __name__ == "__main__":
    main()
