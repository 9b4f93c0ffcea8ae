"""GPU parity at BASELINE.json configs[3] (long-sequence stress: B=1, N=64, L=1024, full model dims) against the CPU oracle:
one tied MSA-row encoder layer (north-star row a6 at L=1024) and one FULL two-track block (MSA update with the pair
track, outer-product pair update + ResNet, axial pair attention) in the 16-bit modes the bench times.  The oracle's
Performer attention is evaluated in chunks of sequences (sequences are independent; the feature tensors of 1024 x 8
sequences of 1024 rows do not fit a host), everything else is the oracle as it stands.

Stated tolerances (max |a-b| / max |ref|, relative L2): layer 2e-2 / 1.5e-2 (bf16), 4e-3 / 3e-3 (fp16), 2e-5 (fp32);
the same bounds for the block (one encoder layer per stack, ~12 layers deep; observed 5.3e-3 / 4.2e-3 bf16,
6.1e-4 / 5.5e-4 fp16: the outputs carry the O(1) residual streams).
"""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import rosettafold_pytorch_amd as R  # noqa: E402
from oracle import rf_oracle as O  # noqa: E402

DEV = "cuda"
N4, L4, DM, DP = 64, 1024, 384, 288


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def rel2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rn(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed + len(s) + sum(s)))


def state(mod, prefix="m"):
    return {prefix + "." + k: v.detach().float().cpu() for k, v in mod.state_dict().items()}


@pytest.fixture(scope="module", autouse=True)
def oracle_threads():
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    yield


@pytest.fixture(scope="module")
def chunked_performer():
    """O.performer_self_attention over 32 sequences at a time (exact: sequences do not interact)."""
    orig = O.performer_self_attention

    def chunked(P, pre, x, heads, generalized, dim_head=64):
        if x.shape[0] <= 32:
            return orig(P, pre, x, heads, generalized, dim_head)
        return torch.cat([orig(P, pre, x[i:i + 32], heads, generalized, dim_head) for i in range(0, x.shape[0], 32)])
    O.performer_self_attention = chunked
    yield
    O.performer_self_attention = orig


@pytest.fixture(scope="module")
def tied_case():
    torch.manual_seed(11)
    m = R.EncoderLayer(d_msa=DM, d_ff=4 * DM, n_heads=12, p_dropout=0.0, tied=True, return_att=True).to(DEV)
    x = rn(1, N4, L4, DM)
    t0 = time.time()
    with torch.no_grad():
        ro, ra = O.encoder_layer_tied(state(m), "m", x, 12)
    print(f"\n[config4 tied_row_layer] oracle {time.time() - t0:.1f}s")
    return m, x, ro, ra


@pytest.mark.parametrize("mode,tol", [(torch.float32, (2e-5, 2e-5)), (torch.bfloat16, (2e-2, 1.5e-2)), (torch.float16, (4e-3, 3e-3))],
                         ids=["fp32", "bf16", "fp16"])
def test_tied_row_layer_L1024(tied_case, mode, tol):
    from rosettafold_pytorch_amd import ops
    m, x, ro, ra = tied_case
    R.set_compute_dtype(mode)
    before = ops.COUNTERS["tied_logits_long"]
    try:
        out, att = m(x.to(DEV))
    finally:
        R.set_compute_dtype(torch.bfloat16)
    # the 16-bit modes take the long-row kernel (contraction-split logits over 128 x 256 tiles), not the generic GEMM path
    assert ops.COUNTERS["tied_logits_long"] - before == (0 if mode == torch.float32 else 1)
    for name, g, r in (("out", out, ro), ("att", att, ra)):
        e, e2 = rel(g, r), rel2(g, r)
        print(f"\n[config4 tied_row_layer.{name} {str(mode).split('.')[-1]}] max-rel {e:.3e}  rel-L2 {e2:.3e}")
        assert e < tol[0] and e2 < tol[1], (name, e, e2)
    assert torch.equal(att, att.transpose(1, 2))


@pytest.fixture(scope="module")
def block_case(chunked_performer):
    torch.manual_seed(12)
    m = R.TwoTrackBlock(DM, DP, n_encoder_layers=1, p_dropout=0.0).to(DEV)
    msa, pair = rn(1, N4, L4, DM), rn(1, L4, L4, DP)
    t0 = time.time()
    with torch.no_grad():
        rm, rp = O.two_track_block(state(m), "m", msa, pair, 1)
    print(f"\n[config4 two_track_block] oracle {time.time() - t0:.1f}s")
    return m, msa, pair, rm, rp


@pytest.mark.parametrize("mode,tol", [(torch.bfloat16, (2e-2, 1.5e-2)), (torch.float16, (4e-3, 3e-3))], ids=["bf16", "fp16"])
def test_two_track_block_config4(block_case, mode, tol):
    """MsaUpdateWithPair -> PairUpdateWithMsa (outer product over N=64, 720-channel ResNet at 1024 x 1024) ->
    PairUpdateWithAxialAttention, exactly the kernels `bench.py --config 4` times."""
    m, msa, pair, rm, rp = block_case
    R.set_compute_dtype(mode)
    try:
        gm, gp = m(msa.to(DEV), pair.to(DEV))
        gm, gp = gm.cpu(), gp.cpu()
    finally:
        R.set_compute_dtype(torch.bfloat16)
    for name, g, r in (("msa", gm, rm), ("pair", gp, rp)):
        e, e2 = rel(g, r), rel2(g, r)
        print(f"\n[config4 two_track_block.{name} {str(mode).split('.')[-1]}] max-rel {e:.3e}  rel-L2 {e2:.3e}")
        assert e < tol[0] and e2 < tol[1], (name, e, e2)


# ---- round 4: the rest of the forward at configs[3] (round-3 review: only an L = 512 finiteness smoke covered these) ----------------
def _trace(b, l, seed=3):
    g = torch.Generator().manual_seed(seed)
    steps = torch.randn(b, l, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)   # protein-like CA trace: non-degenerate kNN / masks
    xyz = ca[:, :, None, :] + 0.5 * torch.randn(b, l, 3, 3, generator=g)
    xyz[:, :, 1] = ca
    return xyz


def test_structure_track_L1024():
    """The three-track additions at N = 64, L = 1024 (k = 128: ~147k edges): CoordUpdateWithMsaAndPair (kNN graph, SE(3)
    transformer with the fused radial kernel, rf.py:786-862) and MsaUpdateWithPairAndCoord (rf.py:891-920) against the oracle.
    The structure track is fp32 in every mode; the MSA update is run in the exact-fp32 mode."""
    torch.manual_seed(13)
    cu = R.CoordUpdateWithMsaAndPair(DM, DP, 32, 32, 32, n_neighbors=128, p_dropout=0.0).to(DEV)
    mu = R.MsaUpdateWithPairAndCoord(DM, 32, 32, 4 * DM, p_dropout=0.0).to(DEV)
    msa, pair, xyz = rn(1, N4, L4, DM, seed=1), rn(1, L4, L4, DP, seed=2), _trace(1, L4)
    seq = torch.randint(0, 21, (1, L4), generator=torch.Generator().manual_seed(5))
    onehot, aa = torch.nn.functional.one_hot(seq, 21).float(), torch.arange(L4).unsqueeze(0)
    t0 = time.time()
    with torch.no_grad():
        rs, rx = O.coord_update(state(cu), "m", xyz, msa, pair, aa, onehot, 128, 32)
        rmsa = O.msa_update_with_pair_and_coord(state(mu), "m", rx, rs, msa)
    print(f"\n[config4 structure track] oracle {time.time() - t0:.1f}s")
    R.set_compute_dtype(torch.float32)
    try:
        gs, gx = cu(xyz.to(DEV), msa.to(DEV), pair.to(DEV), aa.to(DEV), onehot.to(DEV))
        gmsa = mu(gx, gs, msa.to(DEV))
    finally:
        R.set_compute_dtype(torch.bfloat16)
    for name, g, r in (("state", gs, rs), ("xyz", gx, rx), ("msa", gmsa, rmsa)):
        e, e2 = rel(g, r), rel2(g, r)
        print(f"\n[config4 structure.{name} fp32] max-rel {e:.3e}  rel-L2 {e2:.3e}")
        assert e < 1e-4 and e2 < 2e-5, (name, e, e2)


def test_prediction_head_L1024():
    """PredictionHead (rf.py:1130-1172: 4 ResNets of 4 dilated blocks) on a 1024 x 1024 pair map: exact-fp32 mode against the
    oracle (distogram bins bit-exact on >= 99.99 % of the 1M pairs: single pairs are ties to the last bit at random init), and the
    16-bit mode's map against the same oracle output within the per-module bound."""
    torch.manual_seed(14)
    head = R.PredictionHead(DP, 4, 0.0).to(DEV)
    pair = rn(1, L4, L4, DP, seed=7)
    t0 = time.time()
    with torch.no_grad():
        ref = O.prediction_head(state(head), "m", pair)
    print(f"\n[config4 prediction_head] oracle {time.time() - t0:.1f}s")
    for mode, tol in ((torch.float32, 5e-5), (torch.float16, 6e-3)):
        R.set_compute_dtype(mode)
        try:
            got = {k: v.cpu() for k, v in head(pair.to(DEV)).items()}
        finally:
            R.set_compute_dtype(torch.bfloat16)
        for k in ("theta", "phi", "dist", "omega"):
            e2 = rel2(got[k], ref[k])
            agree = (got[k].argmax(-1) == ref[k].argmax(-1)).float().mean().item()
            print(f"\n[config4 head.{k} {str(mode).split('.')[-1]}] rel-L2 {e2:.3e}  argmax agreement {agree:.6f}")
            assert e2 < tol, (k, mode, e2)
            if mode == torch.float32:
                assert agree >= 0.9999, (k, agree)
