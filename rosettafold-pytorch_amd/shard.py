"""Batch sharding across the GPUs of one node (one process per GPU) and the single collective of the path.

The reference has no parallelism at all; every op of the forward is per-sample, so independent
MSAs shard across ranks with replicated weights and NO data-path collective.  The only exchange is the gather of
the results on rank 0: one flat fp32 buffer per rank (logits | xyz | plddt), `torch.distributed.gather`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist

LOGIT_KEYS = ("theta", "phi", "dist", "omega")
LOGIT_BINS = {"theta": 37, "phi": 19, "dist": 37, "omega": 37}


def shard_range(n_items, world, rank):
    """Contiguous slice [lo, hi) of n_items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(logits, xyz, plddt):
    return torch.cat([logits[k].reshape(-1).float() for k in LOGIT_KEYS] + [xyz.reshape(-1).float(),
                                                                           plddt.reshape(-1).float()])


def unpack_results(flat, B, L):
    out, o = {}, 0
    for k in LOGIT_KEYS:
        n = B * L * L * LOGIT_BINS[k]
        out[k] = flat[o:o + n].view(B, L, L, LOGIT_BINS[k])
        o += n
    xyz = flat[o:o + B * L * 9].view(B, L, 3, 3)
    o += B * L * 9
    return out, xyz, flat[o:o + B * L].view(B, L)


def gather_results(logits, xyz, plddt, dst=0):
    """Gather every rank's results on `dst`; returns a list of (logits, xyz, plddt) on dst, None elsewhere.
    All ranks must hold the same per-rank batch size (weak scaling)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [(logits, xyz, plddt)]
    B, L = plddt.shape
    flat = pack_results(logits, xyz, plddt)
    if dist.get_backend() == "gloo" and flat.is_cuda:
        flat = flat.cpu()  # gloo has no device gather; RCCL ("nccl") gathers in HBM
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(flat) for _ in range(world)] if rank == dst else None
    dist.gather(flat, bufs, dst=dst)
    if rank != dst:
        return None
    return [unpack_results(b, B, L) for b in bufs]
