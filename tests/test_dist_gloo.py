"""N>1 path on CPU: world_size 2, gloo.  Shard ranges partition the batch; the result gather returns every rank's
tensors bit-exactly on rank 0 and nothing elsewhere."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rosettafold_pytorch_amd import shard


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, L = 2, 6
    g = torch.Generator().manual_seed(100 + rank)
    logits = {k: torch.randn(B, L, L, n, generator=g) for k, n in shard.LOGIT_BINS.items()}
    xyz, plddt = torch.randn(B, L, 3, 3, generator=g), torch.randn(B, L, generator=g)
    res = shard.gather_results(logits, xyz, plddt, dst=0)
    ok = True
    if rank == 0:
        assert len(res) == world
        for r, (lg, x, p) in enumerate(res):
            gg = torch.Generator().manual_seed(100 + r)
            for k, n in shard.LOGIT_BINS.items():
                ok &= torch.equal(lg[k], torch.randn(B, L, L, n, generator=gg))
            ok &= torch.equal(x, torch.randn(B, L, 3, 3, generator=gg)) and torch.equal(p, torch.randn(B, L, generator=gg))
    else:
        ok = res is None
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _worker_uneven(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, Btot = 5, 3  # 3 samples over 2 ranks: shards of 2 and 1
    lo, hi = shard.shard_range(Btot, world, rank)
    sizes = [shard.shard_range(Btot, world, r)[1] - shard.shard_range(Btot, world, r)[0] for r in range(world)]
    g = torch.Generator().manual_seed(7)
    logits = {k: torch.randn(Btot, L, L, n, generator=g) for k, n in shard.LOGIT_BINS.items()}
    xyz, plddt = torch.randn(Btot, L, 3, 3, generator=g), torch.randn(Btot, L, generator=g)
    res = shard.gather_results({k: v[lo:hi] for k, v in logits.items()}, xyz[lo:hi], plddt[lo:hi], dst=0, batch_sizes=sizes)
    ok = True
    if rank == 0:
        cat = {k: torch.cat([r[0][k] for r in res]) for k in shard.LOGIT_KEYS}
        ok = all(torch.equal(cat[k], logits[k]) for k in cat)
        ok &= torch.equal(torch.cat([r[1] for r in res]), xyz) and torch.equal(torch.cat([r[2] for r in res]), plddt)
    else:
        ok = res is None
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_results_uneven_shards_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == {0: True, 1: True}


def test_shard_ranges_partition():
    for n in (1, 7, 32):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_gather_results_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == {0: True, 1: True}
