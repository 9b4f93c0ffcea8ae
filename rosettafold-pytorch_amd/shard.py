"""Batch sharding across the GPUs of one node (one process per GPU) and the single collective of the path.

The reference has no parallelism at all; every op of the forward is per-sample, so independent
MSAs shard across ranks with replicated weights and NO data-path collective.  The only exchange is the gather of
the results on rank 0: one flat fp32 buffer per rank (logits | xyz | plddt), `torch.distributed.gather`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist

# RF_SHARD_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank group (tests/test_rowshard_gpu.py runs the device
# branches of all_reduce / all_gather through RCCL that way on the one-GPU box)
_FORCE = bool(int(__import__("os").environ.get("RF_SHARD_FORCE_COLLECTIVES", "0")))


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
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
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


# ----------------------------------------------------------------------------------------------------------------------
# Pair-track row-block sharding, first piece: one axial-attention layer on a block of rows
# ----------------------------------------------------------------------------------------------------------------------
def all_reduce_sum(t, group=None):
    """In-place sum over the ranks of `group` (RCCL on device tensors; the gloo rehearsal goes through the host)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE):
        return t
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def pair_axial_layer_row_sharded(layer, x_rows, group=None):
    """PairUpdateWithAxialAttentionLayer (rf.py:501-528) on the row block x_rows = pair[:, r0:r1] held by this rank; the
    ranks of `group` (default: all) hold the other row blocks, in any split.  The attention along the rows' index (RowWise,
    rf.py:31-41: sequences over i for fixed j) spans the ranks: its Performer contexts k'^T [v | 1] -- [B * L * heads, 80, 288]
    fp32 per layer, independent of the block height -- are summed with one all-reduce; everything else is local.  Returns
    the updated row block (fp32)."""
    from . import model as M
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    x = M.fresh_f32(x_rows)
    layer.run(x, row_group=g)   # (a PairUpdateWithAxialAttention stack, rf.py:531-547, takes the same call)
    return x


def _peer(group, r):
    return r if group is None or group is dist.group.WORLD else dist.get_global_rank(group, r)


def check_row_split(L_rows, world, halo):
    """Every rank evaluates the SAME condition from (L, world) alone -- the lowest block of shard_range(L, world, .) is
    L // world rows -- so an unusable split raises on every rank before anyone posts a send or a receive (a rank-local
    `h < d` check leaves the peers blocked in RCCL while one rank raises)."""
    min_h = L_rows // max(world, 1)
    if world > 1 and min_h < max(halo, 1):
        raise ValueError(f"{L_rows} picture rows over {world} ranks leave a block of {min_h} rows: lower than the halo of {halo} "
                         f"rows the dilated 3x3 convolutions exchange with ONE neighbour (use fewer ranks)")


def exchange_row_halos(x, d, group=None):
    """x: [B, h, W, C] contiguous block of picture rows on this rank (the ranks of `group` hold consecutive blocks in rank
    order).  Returns [B, h + 2 d, W, C]: the block with d rows of each neighbour attached (zeros beyond the picture: the 'same'
    padding of the dilated 3x3 convolutions, resnet.py:19-38).  One send + one receive per neighbour (RCCL point-to-point on
    device buffers; the gloo rehearsal stages through the host); slicing and placement by rf_copy4d."""
    from . import ops
    B, h, W, Cc = x.shape
    n = dist.get_world_size(group) if dist.is_initialized() else 1
    r = dist.get_rank(group) if dist.is_initialized() else 0
    check_row_split(W, n, d)  # (square pictures: W = the number of rows of the whole picture) -- the same verdict on every rank
    if h < d:
        raise ValueError(f"row block of {h} rows is lower than the halo of {d}: halos would span more than one neighbour")
    row = W * Cc
    xh = ops.zeros(B, h + 2 * d, W, Cc, device=x.device, dtype=x.dtype)
    ops.copy4d(x, (h * row, row, Cc, 1), xh, ((h + 2 * d) * row, row, Cc, 1), (B, h, W, Cc), y_off=d * row)
    if n == 1:
        return xh
    host = dist.get_backend(group) == "gloo"
    cut = lambda off: ops.copy4d(x, (h * row, row, Cc, 1), torch.empty(B, d, W, Cc, device=x.device, dtype=x.dtype),  # noqa: E731
                                 (d * row, row, Cc, 1), (B, d, W, Cc), x_off=off)
    ops_, recv = [], {}
    for peer, send_off, place_off in ((r - 1, 0, 0), (r + 1, (h - d) * row, (d + h) * row)):
        if 0 <= peer < n:
            out = cut(send_off)
            buf = torch.empty(B, d, W, Cc, device="cpu" if host else x.device, dtype=x.dtype)
            ops_.append(dist.P2POp(dist.isend, out.cpu() if host else out, _peer(group, peer), group))
            ops_.append(dist.P2POp(dist.irecv, buf, _peer(group, peer), group))
            recv[place_off] = buf
    if not host:
        torch.cuda.current_stream().synchronize()  # the send buffers are complete before RCCL's stream reads them
    for w in dist.batch_isend_irecv(ops_):
        w.wait()
    for place_off, buf in recv.items():
        ops.copy4d(buf.to(x.device), (d * row, row, Cc, 1), xh, ((h + 2 * d) * row, row, Cc, 1), (B, d, W, Cc), y_off=place_off)
    return xh


def drop_row_halos(y, d):
    """[B, h + 2 d, W, C] -> the contiguous inner [B, h, W, C]."""
    from . import ops
    B, hh, W, Cc = y.shape
    h, row = hh - 2 * d, W * Cc
    out = torch.empty(B, h, W, Cc, device=y.device, dtype=y.dtype)
    return ops.copy4d(y, (hh * row, row, Cc, 1), out, (h * row, row, Cc, 1), (B, h, W, Cc), x_off=d * row)


def resblock_row_sharded(block, x_rows, rows_global, group=None):
    """ResBlock2D (resnet.py:15-44) on the block of picture rows x_rows = x[:, r0:r1] (NHWC, fp32) held by this rank: two dilated
    3x3 convolutions (halo rows from the neighbouring ranks), two InstanceNorms (all-reduced sums), residual, ELU.  Returns the
    updated row block (fp32)."""
    from . import model as M, ops
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    xf = M.fresh_f32(x_rows)
    return block.run(ops.cast(xf, M.T()), xf, row_group=g, rows_global=rows_global)[1]


def transpose_row_sharded(x_rows, group=None):
    """x_rows: fp32 / 16-bit [B, h, L, C] = rows shard_range(L, world, rank) of a square [B, L, L, C] tensor.  Returns the same
    rows of its transpose: out[b, i, j] = x[b, j, i].  Rank r needs, from every rank s, the sub-block x[s rows, r columns]: one
    send + one receive per pair of ranks (the diagonal block stays local); the transposition happens in the placing rf_copy4d."""
    from . import ops
    B, h, Lr, Cc = x_rows.shape
    n = dist.get_world_size(group) if dist.is_initialized() else 1
    r = dist.get_rank(group) if dist.is_initialized() else 0
    r0, r1 = shard_range(Lr, n, r)
    if r1 - r0 != h:
        raise ValueError(f"rank {r} holds {h} rows; the contiguous split of {Lr} rows over {n} ranks gives it {r1 - r0}")
    out = torch.empty_like(x_rows)
    row = Lr * Cc
    host = n > 1 and dist.get_backend(group) == "gloo"
    ops_, recv = [], {}
    for s_ in range(n):
        s0, s1 = shard_range(Lr, n, s_)
        hs = s1 - s0
        if hs == 0 or h == 0:
            continue
        if s_ == r:  # the diagonal block never leaves: one transposing copy  out[b, i, r0 + j] = x[b, j, r0 + i]
            ops.copy4d(x_rows, (h * row, Cc, row, 1), out, (h * row, row, Cc, 1), (B, h, h, Cc), x_off=r0 * Cc, y_off=r0 * Cc)
            continue
        # my rows, columns of rank s: [B, h, hs, C] contiguous
        blk = ops.copy4d(x_rows, (h * row, row, Cc, 1), torch.empty(B, h, hs, Cc, device=x_rows.device, dtype=x_rows.dtype),
                         (h * hs * Cc, hs * Cc, Cc, 1), (B, h, hs, Cc), x_off=s0 * Cc)
        buf = torch.empty(B, hs, h, Cc, device="cpu" if host else x_rows.device, dtype=x_rows.dtype)  # rank s's rows, my columns
        ops_.append(dist.P2POp(dist.isend, blk.cpu() if host else blk, _peer(group, s_), group))
        ops_.append(dist.P2POp(dist.irecv, buf, _peer(group, s_), group))
        recv[s_] = buf
    if ops_:
        if not host:
            torch.cuda.current_stream().synchronize()
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
    for s_, buf in recv.items():
        s0, s1 = shard_range(Lr, n, s_)
        hs = s1 - s0
        # buf[b, j - s0, i - r0, c] -> out[b, i - r0, j, c]
        ops.copy4d(buf.to(x_rows.device), (hs * h * Cc, Cc, h * Cc, 1), out, (h * row, row, Cc, 1), (B, h, hs, Cc), y_off=s0 * Cc)
    return out


def prediction_head_row_sharded(head, pair_rows, group=None):
    """PredictionHead (rf.py:1130-1172) on this rank's rows shard_range(L, world, rank) of the pair tensor: returns the same
    rows of the four logit maps (fp32 NHWC dict)."""
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    return head.run(pair_rows.float().contiguous(), row_group=g)


def group_size(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def group_rank(group=None):
    return dist.get_rank(group) if dist.is_initialized() else 0


def all_gather_positions(rows, full, group=None):
    """rows: fp32 [B, N, h, D] = positions shard_range(L, world, rank) of full [B, N, L, D]; writes every rank's slice into
    `full` on every rank (one all_gather of slices padded to the largest share; placement by rf_copy4d)."""
    from . import ops
    B, N, h, D = rows.shape
    Lr = full.shape[2]
    n, r = group_size(group), group_rank(group)
    if n == 1 and not (_FORCE and dist.is_initialized()):
        ops.copy4d(rows, (N * h * D, h * D, D, 1), full, (N * Lr * D, Lr * D, D, 1), (B, N, h, D))
        return full
    hmax = max(shard_range(Lr, n, k)[1] - shard_range(Lr, n, k)[0] for k in range(n))
    host = dist.get_backend(group) == "gloo"
    send = ops.zeros(B, N, hmax, D, device=rows.device, dtype=rows.dtype)
    if h > 0:
        ops.copy4d(rows, (N * h * D, h * D, D, 1), send, (N * hmax * D, hmax * D, D, 1), (B, N, h, D))
    if host:
        send = send.cpu()
    else:
        torch.cuda.current_stream().synchronize()
    bufs = [torch.empty_like(send) for _ in range(n)]
    dist.all_gather(bufs, send, group=group)
    for k, buf in enumerate(bufs):
        k0, k1 = shard_range(Lr, n, k)
        if k1 > k0:
            ops.copy4d(buf.to(full.device), (N * hmax * D, hmax * D, D, 1), full, (N * Lr * D, Lr * D, D, 1),
                       (B, N, k1 - k0, D), y_off=k0 * D)
    return full


def msa_update_with_pair_row_sharded(module, msa, pair_rows, group=None):
    """MsaUpdateWithPair (rf.py:550-610) with the pair tensor held as row blocks: msa [B,N,L,D] is replicated, pair_rows are this
    rank's rows shard_range(L, world, rank).  Exchanges: the transposed sub-blocks of the symmetrisation (once per block of
    layers) and one all-gather of the updated msa positions per layer.  Returns the updated msa (whole, fp32) on every rank."""
    from . import model as M
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    out = M.fresh_f32(msa)
    module.run(out, pair_rows.float().contiguous(), row_group=g)
    return out


def pair_update_with_msa_row_sharded(module, msa, pair_rows, att, group=None):
    """PairUpdateWithMsa (rf.py:430-498) with the pair tensor held as row blocks: msa [B,N,L,D] and the tied-attention map att
    [B,L,L,H] are replicated, pair_rows are this rank's rows shard_range(L, world, rank).  Exchanges: one halo row per
    neighbour and convolution, 2*B*C doubles per InstanceNorm.  Returns the new rows (fp32)."""
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    return module.run_rows(msa.float().contiguous(), pair_rows.float().contiguous(), att.float().contiguous(), g)


def two_track_block_row_sharded(block, msa, pair_rows, group=None):
    """TwoTrackBlock (rf.py:923-968) with the pair tensor held as row blocks: msa [B,N,L,D] replicated in, replicated out;
    pair_rows = rows shard_range(L, world, rank) in, the same rows out.  Returns (msa, pair_rows), both fp32."""
    from . import model as M
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    m = M.fresh_f32(msa)
    return m, block.run(m, pair_rows.float().contiguous(), row_group=g)


def take_rows(full, group=None):
    """This rank's rows shard_range(L, world, rank) of a replicated [B, L, ...] tensor, as a contiguous copy."""
    from . import ops
    B, Lr = full.shape[:2]
    inner = full.numel() // max(B * Lr, 1)
    r0, r1 = shard_range(Lr, group_size(group), group_rank(group))
    h = r1 - r0
    out = torch.empty((B, h) + tuple(full.shape[2:]), device=full.device, dtype=full.dtype)
    if h > 0:
        ops.copy4d(full.contiguous(), (Lr * inner, inner, 0, 1), out, (h * inner, inner, 0, 1), (B, h, 1, inner), x_off=r0 * inner)
    return out


def all_gather_rows(rows, group=None):
    """rows [B, h, L, C] (shard_range(L, world, rank) of a square picture) -> the whole [B, L, L, C] on every rank."""
    B, h, Lr, Cc = rows.shape
    full = torch.empty(B, Lr, Lr, Cc, device=rows.device, dtype=rows.dtype)
    all_gather_positions(rows.view(B, 1, h, Lr * Cc), full.view(B, 1, Lr, Lr * Cc), group)
    return full


def forward_row_sharded(model, msa, seq, aa_idx, group=None):
    """RoseTTAFold.forward (rf.py:1273-1289) of ONE batch spread over the ranks of `group` by pair-track row blocks -- the
    configuration the reference cannot shard at all (BASELINE configs[3]: B = 1, L = 1024).  Every rank passes the same
    inputs.  Returns (logits, xyz, plddt) like forward(), the logit maps holding this rank's rows shard_range(L, world, rank)."""
    from . import model as M, structure as S
    g = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
    dev = next(model.parameters()).device
    msa, seq, aa_idx = msa.to(dev).contiguous(), seq.to(dev).contiguous(), aa_idx.to(dev).contiguous()
    mono = M.check_index_range(msa, seq, aa_idx, model.msa_emb.to_embedding.num_embeddings,
                               min(model.msa_emb.pos_enc.max_len, model.pair_emb.pos_enc.max_len))
    # the widest halo of the forward: dilation 8 in the prediction head's ResNets (resnet.py:47-83); decided from (L, world) alone,
    # identically on every rank, before the first exchange
    if model.training:
        raise NotImplementedError("the row-sharded forward is the inference path: call model.eval() (training-mode dropout is built "
                                  "for the single-process forward only)")
    check_row_split(msa.shape[2], group_size(g), 8)
    with torch.no_grad():
        out = model.forward_validated(msa, seq, aa_idx, mono, row_group=g)
    S.check_edge_capacity()
    return out


# ---- pre-haloed pictures (B == 1: a block of rows is one contiguous slab, so producers can write straight into the interior) ----
def haloed_buffer(h, W, Cc, d, device, dtype):
    """[1, h + 2 d, W, C] with the 2 d halo rows zeroed (the 'same' padding at the picture's edge; exchange_row_halos_inplace
    overwrites the inner edges); the interior is left for the producer."""
    from . import ops
    xh = torch.empty(1, h + 2 * d, W, Cc, device=device, dtype=dtype)
    if d > 0:
        ops.fill(xh[0, :d], 0.0)
        ops.fill(xh[0, d + h:], 0.0)
    return xh


def interior(xh, d):
    return xh[:, d:xh.shape[1] - d] if d > 0 else xh


def haloed_parent(x, d):
    """The pre-haloed buffer x is the interior of (halo d), or None."""
    b = x._base
    if b is None or x.dim() != 4 or b.dim() != 4 or x.shape[0] != 1:
        return None
    _, h, W, Cc = x.shape
    if tuple(b.shape) == (1, h + 2 * d, W, Cc) and x.data_ptr() == b.data_ptr() + d * W * Cc * x.element_size() and b.is_contiguous():
        return b
    return None


def exchange_row_halos_inplace(xh, d, group=None):
    """Fill the halo rows of a pre-haloed [1, h + 2 d, W, C] picture from the neighbouring ranks' edge rows (zeros stay at the
    picture's edge).  Only the 2 d halo rows move."""
    n, r = group_size(group), group_rank(group)
    if n == 1 or d == 0:
        return xh
    h = xh.shape[1] - 2 * d
    check_row_split(xh.shape[2], n, d)  # collective verdict first (see check_row_split)
    if h < d:
        raise ValueError(f"row block of {h} rows is lower than the halo of {d}")
    host = dist.get_backend(group) == "gloo"
    ops_, recv = [], []
    for peer, send_view, recv_view in ((r - 1, xh[0, d:2 * d], xh[0, :d]), (r + 1, xh[0, h:h + d], xh[0, d + h:])):
        if 0 <= peer < n:
            buf = torch.empty(recv_view.shape, device="cpu", dtype=xh.dtype) if host else recv_view
            ops_.append(dist.P2POp(dist.isend, send_view.cpu() if host else send_view, _peer(group, peer), group))
            ops_.append(dist.P2POp(dist.irecv, buf, _peer(group, peer), group))
            recv.append((recv_view, buf))
    if not host:
        torch.cuda.current_stream().synchronize()
    for w in dist.batch_isend_irecv(ops_):
        w.wait()
    if host:
        for view, buf in recv:
            view.copy_(buf)
    return xh

