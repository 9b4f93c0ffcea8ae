"""GPU checks of the drop-in boundary pieces added in round 2: the call surface of the small public classes the
reference's tests import (tests/test_module.py:9-31, 35-100), the reference's error behaviour for bad indices, the
weight-cache invalidation rules, the fixed-capacity edge buffers, and the small helper kernels."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import rosettafold_pytorch_amd as R  # noqa: E402
from rosettafold_pytorch_amd import ops, structure as S, _lib  # noqa: E402
from oracle import rf_oracle as O  # noqa: E402

DEV = "cuda"


def rn(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed + len(s) + sum(s)))


def state(mod, prefix="m"):
    return {prefix + "." + k: v.detach().float().cpu() for k, v in mod.state_dict().items()}


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.fixture(autouse=True)
def fp32_mode():
    R.set_compute_dtype(torch.float32)
    yield
    R.set_compute_dtype(torch.bfloat16)


# ---------------------------------------------------------------- reference tests/test_module.py:35-66
def test_sinusoidal_positional_encoding_is_sinusoidal():
    bsz, n_seq, max_len, d = 4, 10, 128, 128
    pe = R.SinusoidalPositionalEncoding(dim=d, max_len=max_len, p_dropout=0.0).to(DEV)
    x = torch.randn(bsz, n_seq, max_len, d, device=DEV)
    aa = torch.arange(0, max_len).unsqueeze(0).repeat(bsz, 1).to(DEV)
    p = pe(x, aa) - x
    s = p[..., 0::2].square() + p[..., 1::2].square()
    assert torch.isclose(s, torch.tensor([1.0], device=DEV), atol=1e-5).all()
    ref = x.cpu() + O.sinusoid_table(d, max_len)[aa.cpu()][:, None]
    assert rel(pe(x, aa), ref) < 1e-6


def test_sinusoidal_positional_encoding_2d():
    bsz, max_len, d = 2, 64, 32
    pe = R.SinusoidalPositionalEncoding2D(dim=d, max_len=max_len, p_dropout=0.0).to(DEV)
    x = torch.randn(bsz, max_len, max_len, d, device=DEV)
    aa = torch.stack([torch.arange(max_len), torch.arange(max_len).flip(0)]).to(DEV)
    y = pe(x, aa)
    assert y.shape == (bsz, max_len, max_len, d)
    t = O.sinusoid_table(d // 2, max_len)[aa.cpu()]  # [b, l, d/2]
    ref = x.cpu() + torch.cat([t[:, :, None, :].expand(-1, -1, max_len, -1), t[:, None, :, :].expand(-1, max_len, -1, -1)], -1)
    assert rel(y, ref) < 1e-6


def test_residual_rowwise_colwise():
    torch.manual_seed(3)
    ff = R.FeedForward(32, 64, 0.0).to(DEV)
    x = torch.randn(2, 5, 7, 32, device=DEV)
    y = R.Residual(ff)(x)
    assert rel(y, O.feed_forward(state(ff), "m", x.cpu()) + x.cpu()) < 1e-5
    att = R.PerformerSelfAttention(dim=32, heads=2, generalized_attention=True).to(DEV)
    P = state(att)
    xc = x.cpu()
    col = O.performer_self_attention(P, "m", xc.reshape(10, 7, 32), 2, True).view(2, 5, 7, 32)
    row = O.performer_self_attention(P, "m", xc.permute(0, 2, 1, 3).reshape(14, 5, 32), 2, True).view(2, 7, 5, 32).permute(0, 2, 1, 3)
    assert rel(R.ColWise(att)(x), col) < 1e-4
    assert rel(R.RowWise(att)(x), row) < 1e-4
    # a user module inside RowWise / ColWise (generic path: explicit transposes)
    assert rel(R.RowWise(ff)(x), O.feed_forward(state(ff), "m", xc)) < 1e-5
    assert rel(R.ColWise(ff)(x), O.feed_forward(state(ff), "m", xc)) < 1e-5
    # the model's own composition called generically, like the reference's nn.Sequential (rf.py:519-528)
    lay = R.PairUpdateWithAxialAttentionLayer(32, 64, 2, 0.0, {}).to(DEV)
    xp = torch.randn(1, 6, 6, 32, device=DEV)
    assert rel(lay.layer(xp), O.pair_axial_layer(state(lay), "m", xp.cpu(), 2)) < 1e-4
    assert rel(lay(xp), lay.layer(xp)) < 1e-5


# ---------------------------------------------------------------- error behaviour (IndexError like nn.Embedding)
CFG = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
           n_encoder_layers=1, max_len=40, n_neighbors=[8, 8], p_dropout=0.0)


def test_index_errors_like_the_reference():
    torch.manual_seed(0)
    m = R.RoseTTAFold(**CFG).to(DEV)
    msa = torch.randint(0, 21, (1, 4, 16), device=DEV)
    aa = torch.arange(16, device=DEV)[None]
    with pytest.raises(IndexError):
        m(msa, msa[:, 0], aa + 30)          # residue index beyond max_len
    bad = msa.clone()
    bad[0, 2, 3] = 21
    with pytest.raises(IndexError):
        m(bad, msa[:, 0], aa)               # token outside [0, 21)
    with pytest.raises(IndexError):
        R.MsaEmbedding(21, 32, 40).to(DEV)(msa, aa - 1)
    with pytest.raises(IndexError):
        R.PairEmbedding(21, 32, 40).to(DEV)(msa[:, 0], aa + 30)
    out = m(msa, msa[:, 0], aa + 20)        # chain-break style offset inside max_len is fine
    assert torch.isfinite(out[1]).all()


def test_non_monotonic_aa_idx_uses_the_general_capacity():
    """repeated / non-monotonic residue indices put up to L edges in a row; the reference's topk/where
    handles any graph.  The wrapper then sizes the edge buffers as B*L*L and the result still matches the oracle."""
    Lr, k = 24, 4
    torch.manual_seed(2)
    m = R.CoordUpdateWithMsaAndPair(32, 24, 8, 8, 8, n_neighbors=k, p_dropout=0.0).to(DEV)
    g = torch.Generator().manual_seed(3)
    steps = torch.randn(1, Lr, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
    xyz = ca[:, :, None, :] + 0.5 * torch.randn(1, Lr, 3, 3, generator=g)
    xyz[:, :, 1] = ca
    msa, pair = rn(1, 4, Lr, 32), rn(1, Lr, Lr, 24)
    oh = torch.nn.functional.one_hot(torch.randint(0, 21, (1, Lr), generator=g), 21).float()
    aa = torch.zeros(1, Lr, dtype=torch.long)  # every |i-j| separation is 0 < 9: the complete graph minus self loops
    st, xo = m(xyz.to(DEV), msa.to(DEV), pair.to(DEV), aa.to(DEV), oh.to(DEV))
    rs, rx = O.coord_update(state(m), "m", xyz, msa, pair, aa, oh, k, 8)
    assert rel(st, rs) < 5e-4 and rel(xo, rx) < 5e-4


def test_graph_overflow_is_reported_not_silent():
    """if the caller's `monotonic` promise is wrong the static capacity k + 2*(kmin-1) per row overflows;
    rf_edges_from_mask drops the surplus (count[1] = true count) and check_edge_capacity() turns that into an error at
    the end of the public forward."""
    Lr, k = 48, 4
    g = torch.Generator().manual_seed(5)
    xyz = torch.randn(1, Lr, 3, 3, generator=g).to(DEV)
    edge = torch.randn(1, Lr, Lr, 8, generator=g).to(DEV)
    aa = torch.zeros(1, Lr, dtype=torch.long, device=DEV)          # complete graph: 47 edges per row > 4 + 16
    S._PENDING_EDGE_COUNTS.clear()
    gr = S.build_graph(xyz, edge, aa, k, monotonic=True)           # the lie
    with pytest.raises(_lib.RfmiError, match="kNN graph overflow"):
        S.check_edge_capacity()
    assert gr["count"].tolist() == [gr["cap"], Lr * (Lr - 1)]
    assert not S._PENDING_EDGE_COUNTS                               # drained
    S.build_graph(xyz, edge, aa, k, monotonic=False)
    S.check_edge_capacity()                                         # the general capacity holds every graph


def test_edges_from_mask_never_writes_past_capacity():
    B, Lr, cap = 1, 32, 100
    mask = torch.ones(B, Lr, Lr, device=DEV, dtype=torch.uint8)  # 1024 edges >> capacity
    pad = 4096
    src = torch.full((cap + pad,), -7, device=DEV, dtype=torch.int32)
    dst = torch.full((cap + pad,), -7, device=DEV, dtype=torch.int32)
    eid = torch.empty(B, Lr, Lr, device=DEV, dtype=torch.int32)
    count = torch.empty(2, device=DEV, dtype=torch.int32)
    ws = torch.empty(2 * B * Lr, device=DEV, dtype=torch.int32)
    rc = _lib.lib.rf_edges_from_mask(ops.ptr(mask), ops.ptr(src), ops.ptr(dst), ops.ptr(eid), ops.ptr(count), ops.ptr(ws), B, Lr,
                                     cap, ops.stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert count.tolist() == [cap, Lr * Lr]
    assert (src[cap:] == -7).all() and (dst[cap:] == -7).all()
    assert (src[:cap] >= 0).all() and int(eid.max()) == cap - 1 and int((eid >= 0).sum()) == cap


def test_knn_mask_ignores_nan_coordinates():
    Lr = 16
    xyz = torch.randn(1, Lr, 3, 3, device=DEV)
    xyz[0, 5] = float("nan")
    aa = (torch.arange(Lr, device=DEV) * 20)[None]  # no sequence neighbours
    mask = ops.knn_mask(xyz.contiguous(), aa, 4, 9)
    assert int(mask[0, :, 5].sum()) == 0           # a NaN residue is nobody's neighbour
    assert int(mask[0, 0].sum()) == 4 and int(mask[0, 0, 0]) == 0


# ---------------------------------------------------------------- weight caches
def test_load_state_dict_after_forward_is_seen():
    """GSE3Res packs radial weights once; loading a checkpoint after a forward must refresh them."""
    Lr = 16
    torch.manual_seed(5)
    a = R.CoordUpdateWithMsaAndPair(32, 24, 8, 8, 8, n_neighbors=4, p_dropout=0.0).to(DEV)
    torch.manual_seed(6)
    b = R.CoordUpdateWithMsaAndPair(32, 24, 8, 8, 8, n_neighbors=4, p_dropout=0.0).to(DEV)
    g = torch.Generator().manual_seed(3)
    xyz = torch.cumsum(torch.randn(1, Lr, 1, 3, generator=g) * 2.5, 1) + 0.5 * torch.randn(1, Lr, 3, 3, generator=g)
    msa, pair = rn(1, 4, Lr, 32), rn(1, Lr, Lr, 24)
    oh = torch.nn.functional.one_hot(torch.randint(0, 21, (1, Lr), generator=g), 21).float()
    aa = torch.arange(Lr)[None]
    args = [t.to(DEV) for t in (xyz, msa, pair, aa, oh)]
    ref = b(*args)
    a(*args)                                  # fills a's caches with its own weights
    a.load_state_dict(b.state_dict())
    got = a(*args)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    with torch.no_grad():                     # in-place edit without load_state_dict: caught by the fingerprint
        for p in a.se3_transformer.parameters():
            p.mul_(1.5)
        for p in b.se3_transformer.parameters():
            p.mul_(1.5)
    R.invalidate_weight_caches(b)
    got, ref = a(*args), b(*args)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    # edits through the `.data` alias bump neither data_ptr nor _version -- the documented contract is that the
    # caller invalidates (model.weights_fingerprint); without it the stale copies are used, with it the edit is seen
    for m in (a, b):
        for p in m.se3_transformer.parameters():
            p.data.mul_(0.5)
    R.invalidate_weight_caches(b)
    stale, ref = a(*args), b(*args)
    assert not torch.equal(stale[1], ref[1])
    R.invalidate_weight_caches(a)
    got = a(*args)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


# ---------------------------------------------------------------- helper kernels
def test_fill_onehot_seqsep():
    for dt in (torch.float32, torch.bfloat16):
        y = torch.empty(1003, device=DEV, dtype=dt)
        ops.fill(y, 1.5)
        assert (y == 1.5).all()
    z = ops.zeros(7, 5, device=DEV, dtype=torch.int32)
    assert (z == 0).all()
    idx = torch.randint(0, 21, (3, 17), device=DEV)
    assert torch.equal(ops.onehot(idx, 21), torch.nn.functional.one_hot(idx, 21).float())
    aa = torch.stack([torch.arange(10), torch.arange(10) * 40]).to(DEV)
    out = ops.zeros(2, 10, 10, 8, device=DEV, dtype=torch.float32)
    ops.seqsep_feature(aa, out, 8, 3)
    d = aa.unsqueeze(-1) - aa.unsqueeze(-2)
    ref = (torch.sign(d) * torch.log(torch.abs(d) + 1)).clamp(0.0, 5.5)
    assert torch.allclose(out[..., 3], ref.float(), atol=1e-6) and (out[..., :3] == 0).all() and (out[..., 4:] == 0).all()


def test_instnorm_requires_the_workspace():
    """The library has no atomic accumulation path (DESIGN.md 'Reproducibility'): no workspace, no launch."""
    x = torch.randn(1, 8, 8, 16, device=DEV)
    sums = torch.empty(32, device=DEV, dtype=torch.float64)
    rc = _lib.lib.rf_instnorm_stats(ops.ptr(x), _lib.RF_F32, ops.ptr(sums), 1, 64, 16, None, 0, ops.stream())
    assert rc == _lib.lib.rf_version() * 0 - 1  # RF_EINVAL


def test_ops_refuse_a_foreign_device_context():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    x = torch.zeros(8, 8, device="cuda:1")
    with torch.cuda.device(0), pytest.raises(_lib.RfmiError):
        ops.fill(x, 1.0)


def test_a3m_to_pdb_end_to_end():
    """alignment text -> model inputs -> HIP forward -> decoded maps and a PDB backbone."""
    from rosettafold_pytorch_amd import featurize as F
    a3m = ">q\nMKVLAAGIVGLSEERARELA\n>h1\nMKVLAtAGIVGLSEDRARELA\n>h2\n-KVLSAGIVGL-EERARDLA\n"
    msa, seq, aa = F.featurize(a3m, chain_lengths=[12, 8], device=DEV, max_len=260)
    assert msa.shape == (1, 3, 20) and int(aa.max()) == 219
    torch.manual_seed(3)
    m = R.RoseTTAFold(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1,
                      n_three_track_blocks=2, n_encoder_layers=1, max_len=260, n_neighbors=[128, 128], p_dropout=0.0).to(DEV)
    logits, xyz, plddt = m(msa, seq, aa)
    d = F.decode_logits(logits)
    assert d["p_contact"].shape == (1, 20, 20) and torch.isfinite(d["dist_expected"]).all()
    assert (d["p_contact"] >= 0).all() and (d["p_contact"] <= 1 + 1e-5).all()
    assert (d["phi"] >= 0).all() and (d["phi"] <= 3.1416).all() and (d["omega"].abs() <= 3.1416).all()
    txt = F.to_pdb(xyz[0], seq[0], torch.sigmoid(plddt[0]), aa_idx=aa[0], chain_lengths=[12, 8])
    x2, s2, _ = F.from_pdb_backbone(txt)
    assert torch.allclose(x2, xyz[0].cpu(), atol=1e-3) and torch.equal(s2, seq[0].cpu())
