"""Pair-track row-block sharding, first piece (reference semantics rf.py:501-528): one
PairUpdateWithAxialAttentionLayer on two row blocks of unequal height held by two ranks (both on cuda:0, gloo rendezvous --
RCCL needs one GPU per rank).  The RowWise attention spans the ranks and all-reduces its Performer contexts; the stacked row
blocks must equal the single-process layer on the whole tensor (fp32 mode: the same kernels, only the context partial sums are
added in another order; 16-bit modes: within the layer tolerance of tests/test_config2_gpu.py)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

D, H, L1, L2 = 96, 8, 48, 48
CUT = 32  # rank 0: rows [0, 32), rank 1: rows [32, 48)


def _x():
    return torch.randn(1, L1, L2, D, generator=torch.Generator().manual_seed(5))


def _worker(rank, world, port, wpath, opath, dtype, mode="contexts", cut=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RF_ROWSHARD_ATTENTION=mode)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    layer = R.PairUpdateWithAxialAttention(D, 2 * D, H, 0.0, 2)
    layer.load_state_dict(torch.load(wpath))
    layer = layer.to("cuda:0")
    cut = CUT if cut is None else cut
    lo, hi = (0, cut) if rank == 0 else (cut, L1)
    out = shard.pair_axial_layer_row_sharded(layer, _x()[:, lo:hi].to("cuda:0"))
    torch.cuda.synchronize()
    torch.save(out.cpu(), f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,cut", [("contexts", CUT), ("transpose", L1 // 2)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)],
                         ids=["fp32", "bf16", "fp16"])
def test_axial_layer_row_sharded_world2(tmp_path, dtype, tol, mode, cut):
    """mode "contexts": the crossing direction all-reduces its Performer contexts (any split of the rows: 32 + 16 here);
    mode "transpose" (the default of the sharded forward): it runs on transposed blocks (the contiguous even split)."""
    import rosettafold_pytorch_amd as R
    wpath, opath = str(tmp_path / "layer.pt"), str(tmp_path / "rows.pt")
    torch.manual_seed(77)
    layer = R.PairUpdateWithAxialAttention(D, 2 * D, H, 0.0, 2)   # a stack of two axial layers (rf.py:531-547)
    torch.save(layer.state_dict(), wpath)
    ctx = mp.get_context("spawn")
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, wpath, opath, dtype, mode, cut)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = torch.cat([torch.load(f"{opath}.{r}") for r in range(2)], 1)
    R.set_compute_dtype(dtype)
    try:
        ref = layer.to("cuda:0")(_x().to("cuda:0")).cpu()
    finally:
        R.set_compute_dtype(torch.bfloat16)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    # ... and against the CPU oracle's restatement of the reference layer on the whole tensor
    from oracle import rf_oracle as O
    st = {"m." + k: v.detach().float().cpu() for k, v in layer.state_dict().items()}
    ora = O.pair_update_with_axial_attention(st, "m", _x(), 2)
    err_o = ((got - ora).abs().max() / ora.abs().max()).item()
    print(f"\n[row shard {dtype} {mode}] two row blocks ({cut} + {L1 - cut} of {L1}): vs one process max-rel {err:.3e}, vs oracle {err_o:.3e}")
    assert got.shape == ref.shape and err < tol and err_o < tol, (err, err_o)


# ---- ResBlock2D on row blocks: halo rows from the neighbours, all-reduced InstanceNorm sums -----------------------------
CH, DIL, PH, PW = 64, 4, 40, 24   # picture rows split 24 + 16 (both >= the halo of 4)
PCUT = 24


def _pic():
    return torch.randn(1, PH, PW, CH, generator=torch.Generator().manual_seed(9))  # NHWC


def _res_worker(rank, world, port, wpath, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    blk = R.ResBlock2D(CH, 3, DIL, 0.0)
    blk.load_state_dict(torch.load(wpath))
    blk = blk.to("cuda:0")
    lo, hi = (0, PCUT) if rank == 0 else (PCUT, PH)
    out = shard.resblock_row_sharded(blk, _pic()[:, lo:hi].to("cuda:0"), PH)
    torch.cuda.synchronize()
    torch.save(out.cpu(), f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)],
                         ids=["fp32", "bf16", "fp16"])
def test_resblock_row_sharded_world2(tmp_path, dtype, tol):
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import ops
    wpath, opath = str(tmp_path / "blk.pt"), str(tmp_path / "rows.pt")
    torch.manual_seed(78)
    blk = R.ResBlock2D(CH, 3, DIL, 0.0)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.InstanceNorm2d):
                m.weight.normal_(1.0, 0.2)
                m.bias.normal_(0.0, 0.2)
    torch.save(blk.state_dict(), wpath)
    ctx = mp.get_context("spawn")
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_res_worker, args=(r, 2, port, wpath, opath, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = torch.cat([torch.load(f"{opath}.{r}") for r in range(2)], 1)
    R.set_compute_dtype(dtype)
    try:
        x = _pic().to("cuda:0")
        ref = blk.to("cuda:0").run(ops.cast(x, R.model.T()), x)[1].cpu()
    finally:
        R.set_compute_dtype(torch.bfloat16)
    from oracle import rf_oracle as O
    st = {"m." + k: v.detach().float().cpu() for k, v in blk.state_dict().items()}
    ora = O.resblock2d(st, "m", _pic().permute(0, 3, 1, 2), DIL).permute(0, 2, 3, 1)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    err_o = ((got - ora).abs().max() / ora.abs().max()).item()
    print(f"\n[row shard resblock {dtype}] rows 24 + 16 of 40, dilation {DIL}: vs one process max-rel {err:.3e}, vs oracle {err_o:.3e}")
    assert got.shape == ref.shape and err < tol and err_o < tol, (err, err_o)


# ---- PredictionHead on row blocks: transposed sub-blocks for the symmetrisation + four row-sharded ResNets -----------------
HC, HL, HNB = 64, 42, 3   # 42 rows over 2 ranks (21 + 21), three residual blocks (dilations 1, 2, 4)


def _pair():
    return torch.randn(1, HL, HL, HC, generator=torch.Generator().manual_seed(13))


def _head_worker(rank, world, port, wpath, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    head = R.PredictionHead(HC, HNB, 0.0)
    head.load_state_dict(torch.load(wpath))
    head = head.to("cuda:0")
    lo, hi = shard.shard_range(HL, world, rank)
    out = shard.prediction_head_row_sharded(head, _pair()[:, lo:hi].to("cuda:0"))
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in out.items()}, f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.float16, 1e-2)], ids=["fp32", "fp16"])
def test_prediction_head_row_sharded_world2(tmp_path, dtype, tol):
    import rosettafold_pytorch_amd as R
    wpath, opath = str(tmp_path / "head.pt"), str(tmp_path / "rows.pt")
    torch.manual_seed(79)
    head = R.PredictionHead(HC, HNB, 0.0)
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.InstanceNorm2d):
                m.weight.normal_(1.0, 0.2)
                m.bias.normal_(0.0, 0.2)
    torch.save(head.state_dict(), wpath)
    ctx = mp.get_context("spawn")
    port = 39500 + os.getpid() % 2000
    procs = [ctx.Process(target=_head_worker, args=(r, 2, port, wpath, opath, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    parts = [torch.load(f"{opath}.{r}") for r in range(2)]
    R.set_compute_dtype(dtype)
    try:
        ref = {k: v.cpu() for k, v in head.to("cuda:0").run(_pair().to("cuda:0")).items()}
    finally:
        R.set_compute_dtype(torch.bfloat16)
    from oracle import rf_oracle as O
    st = {"m." + k: v.detach().float().cpu() for k, v in head.state_dict().items()}
    ora = O.prediction_head(st, "m", _pair(), HNB)
    for k in ("theta", "phi", "dist", "omega"):
        got = torch.cat([p[k] for p in parts], 1)
        err = ((got - ref[k]).abs().max() / ref[k].abs().max()).item()
        err_o = ((got - ora[k]).abs().max() / ora[k].abs().max()).item()
        print(f"\n[row shard head {dtype}] {k}: vs one process max-rel {err:.3e}, vs oracle {err_o:.3e}")
        assert got.shape == ref[k].shape and err < tol and err_o < tol, (k, err, err_o)


# ---- MsaUpdateWithPair with the pair tensor on row blocks: transposed exchange + per-layer all-gather of msa positions -----
MD, MP, MH, MN, ML = 96, 64, 4, 6, 48   # (16-bit attention maps want L % 8 == 0, like the unsharded path)


def _msa_pair():
    g = torch.Generator().manual_seed(17)
    return torch.randn(1, MN, ML, MD, generator=g), torch.randn(1, ML, ML, MP, generator=g)


def _mup_worker(rank, world, port, wpath, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    mod = R.MsaUpdateWithPair(MD, MP, MH, n_encoder_layers=2, p_dropout=0.0)
    mod.load_state_dict(torch.load(wpath))
    mod = mod.to("cuda:0")
    msa, pair = _msa_pair()
    lo, hi = shard.shard_range(ML, world, rank)
    out = shard.msa_update_with_pair_row_sharded(mod, msa.to("cuda:0"), pair[:, lo:hi].to("cuda:0"))
    torch.cuda.synchronize()
    torch.save(out.cpu(), f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)],
                         ids=["fp32", "bf16", "fp16"])
def test_msa_update_with_pair_row_sharded_world2(tmp_path, dtype, tol):
    import rosettafold_pytorch_amd as R
    wpath, opath = str(tmp_path / "mup.pt"), str(tmp_path / "msa.pt")
    torch.manual_seed(80)
    mod = R.MsaUpdateWithPair(MD, MP, MH, n_encoder_layers=2, p_dropout=0.0)
    torch.save(mod.state_dict(), wpath)
    ctx = mp.get_context("spawn")
    port = 41500 + os.getpid() % 2000
    procs = [ctx.Process(target=_mup_worker, args=(r, 2, port, wpath, opath, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    outs = [torch.load(f"{opath}.{r}") for r in range(2)]
    assert torch.equal(outs[0], outs[1]), "the all-gather leaves the same msa on every rank"
    msa, pair = _msa_pair()
    R.set_compute_dtype(dtype)
    try:
        ref = mod.to("cuda:0")(msa.to("cuda:0"), pair.to("cuda:0")).cpu()
    finally:
        R.set_compute_dtype(torch.bfloat16)
    from oracle import rf_oracle as O
    st = {"m." + k: v.detach().float().cpu() for k, v in mod.state_dict().items()}
    ora = O.msa_update_with_pair(st, "m", msa, pair, 2, MH)
    err = ((outs[0] - ref).abs().max() / ref.abs().max()).item()
    err_o = ((outs[0] - ora).abs().max() / ora.abs().max()).item()
    print(f"\n[row shard msa-update-with-pair {dtype}] pair rows 24 + 24 of 48: vs one process max-rel {err:.3e}, vs oracle {err_o:.3e}")
    assert err < tol and err_o < tol, (err, err_o)


# ---- PairUpdateWithMsa on row blocks: rectangular outer product + feature assembly, halo rows, InstanceNorm sums ------------
UD, UPJ, UP, UH, UN, UL = 96, 32, 64, 4, 8, 48


def _pum_inputs():
    g = torch.Generator().manual_seed(19)
    msa, pair = torch.randn(1, UN, UL, UD, generator=g), torch.randn(1, UL, UL, UP, generator=g)
    return msa, pair, torch.rand(1, UL, UL, UH, generator=g).softmax(2)


def _pum_worker(rank, world, port, wpath, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    mod = R.PairUpdateWithMsa(d_msa=UD, d_proj=UPJ, d_pair=UP, n_heads=UH, p_dropout=0.0)
    mod.load_state_dict(torch.load(wpath))
    mod = mod.to("cuda:0")
    msa, pair, att = _pum_inputs()
    lo, hi = shard.shard_range(UL, world, rank)
    out = shard.pair_update_with_msa_row_sharded(mod, msa.to("cuda:0"), pair[:, lo:hi].to("cuda:0"), att.to("cuda:0"))
    torch.cuda.synchronize()
    torch.save(out.cpu(), f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)],
                         ids=["fp32", "bf16", "fp16"])
def test_pair_update_with_msa_row_sharded_world2(tmp_path, dtype, tol):
    import rosettafold_pytorch_amd as R
    wpath, opath = str(tmp_path / "pum.pt"), str(tmp_path / "rows.pt")
    torch.manual_seed(81)
    mod = R.PairUpdateWithMsa(d_msa=UD, d_proj=UPJ, d_pair=UP, n_heads=UH, p_dropout=0.0)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.InstanceNorm2d):
                m.weight.normal_(1.0, 0.2)
                m.bias.normal_(0.0, 0.2)
    torch.save(mod.state_dict(), wpath)
    ctx = mp.get_context("spawn")
    port = 43500 + os.getpid() % 2000
    procs = [ctx.Process(target=_pum_worker, args=(r, 2, port, wpath, opath, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = torch.cat([torch.load(f"{opath}.{r}") for r in range(2)], 1)
    msa, pair, att = _pum_inputs()
    R.set_compute_dtype(dtype)
    try:
        ref = mod.to("cuda:0")(msa.to("cuda:0"), pair.to("cuda:0"), att.to("cuda:0")).cpu()
    finally:
        R.set_compute_dtype(torch.bfloat16)
    from oracle import rf_oracle as O
    st = {"m." + k: v.detach().float().cpu() for k, v in mod.state_dict().items()}
    ora = O.pair_update_with_msa(st, "m", msa, pair, att)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    err_o = ((got - ora).abs().max() / ora.abs().max()).item()
    print(f"\n[row shard pair-update-with-msa {dtype}] rows 24 + 24 of 48: vs one process max-rel {err:.3e}, vs oracle {err_o:.3e}")
    assert got.shape == ref.shape and err < tol and err_o < tol, (err, err_o)


# ---- a whole TwoTrackBlock with the pair tensor on row blocks (rf.py:923-968) -------------------------------------------------
TM, TP, TN, TL = 96, 72, 8, 48


def _ttb_inputs():
    g = torch.Generator().manual_seed(23)
    return torch.randn(1, TN, TL, TM, generator=g), torch.randn(1, TL, TL, TP, generator=g)


def _ttb_worker(rank, world, port, wpath, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    blk = R.TwoTrackBlock(TM, TP, 1, 0.0)
    blk.load_state_dict(torch.load(wpath))
    blk = blk.to("cuda:0")
    msa, pair = _ttb_inputs()
    lo, hi = shard.shard_range(TL, world, rank)
    m, p = shard.two_track_block_row_sharded(blk, msa.to("cuda:0"), pair[:, lo:hi].to("cuda:0"))
    torch.cuda.synchronize()
    torch.save({"msa": m.cpu(), "pair": p.cpu()}, f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float16, 1e-2)], ids=["fp32", "fp16"])
def test_two_track_block_row_sharded_world2(tmp_path, dtype, tol):
    import rosettafold_pytorch_amd as R
    wpath, opath = str(tmp_path / "ttb.pt"), str(tmp_path / "out.pt")
    torch.manual_seed(82)
    blk = R.TwoTrackBlock(TM, TP, 1, 0.0)
    torch.save(blk.state_dict(), wpath)
    ctx = mp.get_context("spawn")
    port = 45500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ttb_worker, args=(r, 2, port, wpath, opath, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    outs = [torch.load(f"{opath}.{r}") for r in range(2)]
    assert torch.equal(outs[0]["msa"], outs[1]["msa"]), "msa is replicated again after the block"
    pair_got = torch.cat([o["pair"] for o in outs], 1)
    msa, pair = _ttb_inputs()
    R.set_compute_dtype(dtype)
    try:
        rm, rp = blk.to("cuda:0")(msa.to("cuda:0"), pair.to("cuda:0"))
    finally:
        R.set_compute_dtype(torch.bfloat16)
    from oracle import rf_oracle as O
    st = {"m." + k: v.detach().float().cpu() for k, v in blk.state_dict().items()}
    om, op = O.two_track_block(st, "m", msa, pair, 1)
    e = lambda a, b: ((a.cpu() - b.cpu()).abs().max() / b.abs().max()).item()  # noqa: E731
    errs = {"msa vs one process": e(outs[0]["msa"], rm), "pair vs one process": e(pair_got, rp),
            "msa vs oracle": e(outs[0]["msa"], om), "pair vs oracle": e(pair_got, op)}
    print(f"\n[row shard two-track block {dtype}] " + ", ".join(f"{k} {v:.3e}" for k, v in errs.items()))
    assert all(v < tol for v in errs.values()), errs


# ---- the whole forward, ONE sample over two ranks (the case the reference cannot shard: configs[3] is B = 1) ---------------------
FCFG = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
            n_encoder_layers=1, max_len=64, n_neighbors=[16, 16], p_dropout=0.0)
FN, FL = 8, 32


def _f_inputs():
    g = torch.Generator().manual_seed(29)
    msa = torch.randint(0, 21, (1, FN, FL), generator=g)
    return msa, msa[:, 0].clone(), torch.arange(FL).unsqueeze(0)


def _f_worker(rank, world, port, ckpt, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    model = R.RoseTTAFold(**FCFG)
    R.load_checkpoint(model, ckpt)
    model = model.to("cuda:0")
    logits, xyz, plddt = shard.forward_row_sharded(model, *_f_inputs())
    torch.cuda.synchronize()
    torch.save({"logits": {k: v.cpu() for k, v in logits.items()}, "xyz": xyz.cpu(), "plddt": plddt.cpu()}, f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_forward_row_sharded_world2(tmp_path):
    import rosettafold_pytorch_amd as R
    ckpt, opath = str(tmp_path / "model.pt"), str(tmp_path / "out.pt")
    torch.manual_seed(83)
    model = R.RoseTTAFold(**FCFG)
    R.save_checkpoint(model, ckpt)
    ctx = mp.get_context("spawn")
    port = 47500 + os.getpid() % 2000
    procs = [ctx.Process(target=_f_worker, args=(r, 2, port, ckpt, opath, torch.float32)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    outs = [torch.load(f"{opath}.{r}") for r in range(2)]
    assert torch.equal(outs[0]["xyz"], outs[1]["xyz"]) and torch.equal(outs[0]["plddt"], outs[1]["plddt"]), \
        "the replicated structure track sees the same gathered pair tensor on every rank"
    R.set_compute_dtype(torch.float32)
    try:
        rl, rx, rp = model.to("cuda:0")(*[t.cuda() for t in _f_inputs()])
    finally:
        R.set_compute_dtype(torch.bfloat16)
    e = lambda a, b: ((a.cpu() - b.cpu()).abs().max() / b.abs().max().clamp_min(1e-20)).item()  # noqa: E731
    errs = {k: e(torch.cat([o["logits"][k] for o in outs], 1), rl[k]) for k in rl}
    errs["xyz"], errs["plddt"] = e(outs[0]["xyz"], rx), e(outs[0]["plddt"], rp)
    print("\n[row shard forward fp32] one sample over two ranks vs one process: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert all(v < 1e-3 for v in errs.values()), errs
    for k in rl:   # ... and the distogram argmax bins are the single-process ones
        assert torch.equal(torch.cat([o["logits"][k] for o in outs], 1).argmax(-1), rl[k].cpu().argmax(-1)), k


# ---- the same at the benchmark's layer dimensions (d_msa 384, d_pair 288, N 64, L 256): the row blocks go through the production
# kernels (persistent / register-resident GEMMs, fused feed-forward, conv3x3 halo kernel on haloed pictures, fused tied attention)
BCFG = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=1, n_three_track_blocks=1,
            n_encoder_layers=1, max_len=300, n_neighbors=[32], p_dropout=0.0)
BN, BL = 64, 256


def _b_inputs():
    g = torch.Generator().manual_seed(31)
    msa = torch.randint(0, 21, (1, BN, BL), generator=g)
    return msa, msa[:, 0].clone(), torch.arange(BL).unsqueeze(0)


def _b_worker(rank, world, port, ckpt, opath, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(dtype)
    model = R.RoseTTAFold(**BCFG)
    R.load_checkpoint(model, ckpt)
    model = model.to("cuda:0")
    logits, xyz, plddt = shard.forward_row_sharded(model, *_b_inputs())
    torch.cuda.synchronize()
    torch.save({"logits": {k: v.cpu() for k, v in logits.items()}, "xyz": xyz.cpu(), "plddt": plddt.cpu()}, f"{opath}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-2), (torch.bfloat16, 1.5e-1)], ids=["fp16", "bf16"])
def test_forward_row_sharded_benchmark_dims(tmp_path, dtype, tol):
    """Logit maps of the sharded 16-bit forward against the SINGLE-PROCESS fp32-mode forward (rel-L2): the bound a 1+1-block model
    of these dimensions meets unsharded (tests/test_depth_gpu.py reports 3e-3 / 3e-2 at 2+2 blocks)."""
    import rosettafold_pytorch_amd as R
    ckpt, opath = str(tmp_path / "model.pt"), str(tmp_path / "out.pt")
    torch.manual_seed(84)
    model = R.RoseTTAFold(**BCFG)
    R.save_checkpoint(model, ckpt)
    ctx = mp.get_context("spawn")
    port = 49500 + os.getpid() % 2000
    procs = [ctx.Process(target=_b_worker, args=(r, 2, port, ckpt, opath, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    outs = [torch.load(f"{opath}.{r}") for r in range(2)]
    model = model.to("cuda:0")
    ins = [t.cuda() for t in _b_inputs()]
    res = {}
    try:
        for mode in (torch.float32, dtype):
            R.set_compute_dtype(mode)
            res[mode] = {k: v.cpu() for k, v in model(*ins)[0].items()}
    finally:
        R.set_compute_dtype(torch.bfloat16)
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()  # noqa: E731
    for k in res[torch.float32]:
        got = torch.cat([o["logits"][k] for o in outs], 1)
        e_sh, e_1p = l2(got, res[torch.float32][k]), l2(res[dtype][k], res[torch.float32][k])
        print(f"\n[row shard forward {dtype}, benchmark dims] {k}: sharded vs fp32 mode rel-L2 {e_sh:.3e} (single process, same mode: {e_1p:.3e})")
        assert e_sh < tol and e_sh < 3 * e_1p + 1e-3, (k, e_sh, e_1p)


# ---- the device branches of the collectives through RCCL (backend "nccl"), as far as one GPU allows: a one-rank group ------------
def _nccl_worker(port, ckpt, opath):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RF_SHARD_FORCE_COLLECTIVES="1",
                      RF_ROWSHARD_ATTENTION="contexts")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    R.set_compute_dtype(torch.float32)
    model = R.RoseTTAFold(**FCFG)
    R.load_checkpoint(model, ckpt)
    model = model.to("cuda:0")
    logits, xyz, plddt = shard.forward_row_sharded(model, *_f_inputs())
    torch.cuda.synchronize()
    # ... and the one collective of the batch-sharded bench path (bench.py step(): shard.gather_results -> dist.gather on the
    # flat device buffer packed by rf_axpby), also forced through RCCL
    plain = model(*[t.cuda() for t in _f_inputs()])
    (gl, gx, gp), = shard.gather_results(*plain, dst=0)
    assert all(torch.equal(gl[k], plain[0][k]) for k in gl) and torch.equal(gx, plain[1]) and torch.equal(gp, plain[2])
    torch.save({"logits": {k: v.cpu() for k, v in logits.items()}, "xyz": xyz.cpu(), "plddt": plddt.cpu()}, opath)
    dist.barrier()
    dist.destroy_process_group()


def test_forward_row_sharded_rccl_one_rank(tmp_path):
    """dist.all_reduce (Performer contexts, InstanceNorm sums) and dist.all_gather (msa positions, pair rows) on DEVICE tensors
    through RCCL -- a one-rank group is all a one-GPU box can host, the collectives are forced to run anyway."""
    import rosettafold_pytorch_amd as R
    ckpt, opath = str(tmp_path / "model.pt"), str(tmp_path / "out.pt")
    torch.manual_seed(83)
    model = R.RoseTTAFold(**FCFG)
    R.save_checkpoint(model, ckpt)
    p = mp.get_context("spawn").Process(target=_nccl_worker, args=(51500 + os.getpid() % 2000, ckpt, opath))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    got = torch.load(opath)
    R.set_compute_dtype(torch.float32)
    try:
        rl, rx, rp = model.to("cuda:0")(*[t.cuda() for t in _f_inputs()])
    finally:
        R.set_compute_dtype(torch.bfloat16)
    e = lambda a, b: ((a.cpu() - b.cpu()).abs().max() / b.abs().max().clamp_min(1e-20)).item()  # noqa: E731
    errs = {k: e(got["logits"][k], rl[k]) for k in rl}
    errs["xyz"], errs["plddt"] = e(got["xyz"], rx), e(got["plddt"], rp)
    print("\n[row shard forward, RCCL one-rank group] vs plain forward: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert all(v < 1e-3 for v in errs.values()), errs
