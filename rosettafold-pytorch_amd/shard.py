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


def pack_results(logits, xyz, plddt, numel=None):
    """One flat fp32 buffer (logits | xyz | plddt), zero-padded to `numel` elements.  Device tensors are packed by the
    library's own copy kernel (rf_axpby: one launch per piece on the forward's stream, no ATen kernel in the timed region);
    host tensors (the gloo tests) by torch.cat."""
    pieces = [logits[k] for k in LOGIT_KEYS] + [xyz, plddt]
    n = sum(t.numel() for t in pieces)
    total = n if numel is None else max(n, numel)
    if n > 0 and all(t.is_cuda and t.dtype == torch.float32 for t in pieces):
        from . import ops
        flat = torch.empty(total, device=pieces[0].device, dtype=torch.float32)
        o = 0
        for t in pieces:
            ops.axpby(t.contiguous(), 1.0, None, 0.0, flat[o:o + t.numel()])
            o += t.numel()
        if total > n:
            ops.fill(flat[n:], 0.0)
        return flat
    flat = torch.cat([t.reshape(-1).float() for t in pieces])
    return flat if total == n else torch.cat([flat, flat.new_zeros(total - n)])


def unpack_results(flat, B, L):
    out, o = {}, 0
    for k in LOGIT_KEYS:
        n = B * L * L * LOGIT_BINS[k]
        out[k] = flat[o:o + n].view(B, L, L, LOGIT_BINS[k])
        o += n
    xyz = flat[o:o + B * L * 9].view(B, L, 3, 3)
    o += B * L * 9
    return out, xyz, flat[o:o + B * L].view(B, L)


def result_numel(B, L):
    return B * L * L * sum(LOGIT_BINS.values()) + B * L * 9 + B * L


def gather_results(logits, xyz, plddt, dst=0, batch_sizes=None):
    """Gather every rank's results on `dst`; returns a list of (logits, xyz, plddt) on dst, None elsewhere.
    batch_sizes: per-rank batch sizes when they differ (uneven shards: every rank pads its flat buffer to the largest
    shard, `torch.distributed.gather` needs equal sizes); None = all ranks hold the same batch size (weak scaling)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [(logits, xyz, plddt)]
    B, L = plddt.shape
    world, rank = dist.get_world_size(), dist.get_rank()
    if batch_sizes is None:
        batch_sizes = [B] * world
    if len(batch_sizes) != world or batch_sizes[rank] != B:
        raise ValueError(f"batch_sizes {batch_sizes} does not describe this rank (rank {rank} holds {B})")
    flat = pack_results(logits, xyz, plddt, result_numel(max(batch_sizes), L))
    if dist.get_backend() == "gloo" and flat.is_cuda:
        flat = flat.cpu()  # gloo has no device gather; RCCL ("nccl") gathers in HBM
    bufs = [torch.empty_like(flat) for _ in range(world)] if rank == dst else None
    dist.gather(flat, bufs, dst=dst)
    if rank != dst:
        return None
    return [unpack_results(b[:result_numel(n, L)], n, L) for b, n in zip(bufs, batch_sizes) if n > 0]


def forward_sharded(model, msa, seq, aa_idx, dst=0):
    """Data-parallel forward of a GLOBAL batch (BASELINE.json configs[2]: bsz=32 over 8 GPUs): every rank receives the
    same global (msa [B,N,L], seq [B,L], aa_idx [B,L]) tensors (host or device), runs the forward on its contiguous
    slice `shard_range(B, world, rank)` on its own GPU and the results meet on `dst` in one gather.  Returns
    (logits, xyz, plddt) of the whole batch on dst (concatenated in batch order), None on the other ranks.  Samples are
    independent, so the result equals the single-process forward of the same batch sample by sample."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = msa.shape[0]
    sizes = [shard_range(B, world, r)[1] - shard_range(B, world, r)[0] for r in range(world)]
    lo, hi = shard_range(B, world, rank)
    dev = next(model.parameters()).device
    L = msa.shape[-1]
    if hi > lo:
        logits, xyz, plddt = model(msa[lo:hi].to(dev), seq[lo:hi].to(dev), aa_idx[lo:hi].to(dev))
    else:  # more ranks than samples: this rank contributes nothing
        logits = {k: torch.zeros(0, L, L, n, device=dev) for k, n in LOGIT_BINS.items()}
        xyz, plddt = torch.zeros(0, L, 3, 3, device=dev), torch.zeros(0, L, device=dev)
    parts = gather_results(logits, xyz, plddt, dst=dst, batch_sizes=sizes)
    if parts is None:
        return None
    out = {k: torch.cat([p[0][k] for p in parts]) for k in LOGIT_KEYS}
    return out, torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts])
