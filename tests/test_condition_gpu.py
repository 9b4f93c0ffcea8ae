"""Operand conditioning of the 16-bit modes (csrc/condition.hip, include/rfmi.h): the five entry points against plain torch fp32,
the two identities they implement -- W (f - m) + (b + W m) == W f + b and conv3x3(x - c) + [taps outside the picture] ==
conv3x3(x) - const -- and their effect where the model uses them: PairUpdateWithMsa (rf.py:430-498) and PredictionHead
(rf.py:1130-1172) on inputs that carry a large per-sample constant must agree with the exact-fp32 mode several times better
WITH the conditioning than without, and no worse on inputs without one."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import rosettafold_pytorch_amd as R  # noqa: E402
from rosettafold_pytorch_amd import model as M, ops  # noqa: E402
from rosettafold_pytorch_amd._lib import RfmiError  # noqa: E402

DEV = "cuda"


def rn(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed + len(s) + sum(s))).to(DEV)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.fixture(autouse=True)
def _restore_mode():
    yield
    R.set_compute_dtype(torch.bfloat16)
    M.RT.condition = True
    M.RT.head_center = True


@pytest.mark.parametrize("B,R_,C", [(1, 256, 64), (3, 37, 64), (2, 50, 405), (1, 1, 8)])
def test_center_rows(B, R_, C):
    x = rn(B, R_, C) + 5.0
    ref_mean = x.mean(1)
    y = x.clone()
    mean = ops.center_rows(y)
    assert torch.allclose(mean, ref_mean, atol=1e-5)
    assert torch.allclose(y, x - ref_mean[:, None], atol=1e-5)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(1, 64, 64, 288), (2, 17, 23, 128), (1, 5, 7, 4), (2, 32, 32, 36), (1, 40, 33, 1024)])
def test_channel_mean_and_center_apply(B, H, W, C, out_dtype):
    x = rn(B, H, W, C) + 3.0 * rn(1, 1, 1, C, seed=1)
    mean = ops.channel_mean(x)
    ref = x.double().mean((1, 2))
    assert torch.allclose(mean.double(), ref, atol=1e-5)
    # the vectorised fp32 kernel (rf_channel_mean) and the InstanceNorm statistics path (rf_instnorm_stats + rf_instnorm_mean, the
    # one a row-sharded picture takes) give the same means
    assert torch.allclose(ops.center_channels(x.clone()), x - mean[:, None, None, :], atol=1e-5)
    want = (x - mean[:, None, None, :]).to(out_dtype)
    y = ops.center_apply(x, mean, out_dtype=out_dtype)
    assert y.dtype == out_dtype and torch.equal(y, want)
    if out_dtype == torch.float32:   # default: in place
        assert y.data_ptr() == x.data_ptr()


def test_fold_mean_both_forms():
    N, ld, K, B = 288, 736, 64, 3
    w, mean, bias = rn(N, ld), rn(B, K), rn(N)
    got = ops.fold_mean(w, mean, bias, k0=288, nseg=2, seg_stride=64)
    want = bias[None] + mean @ (w[:, 288:352] + w[:, 352:416]).T
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-4)
    w9, m9 = rn(96, 9 * 40), rn(B, 40)
    got = ops.fold_mean(w9, m9, None, nseg=9, seg_stride=40, sum_seg=False)
    want = torch.einsum("ntk,bk->btn", w9.view(96, 9, 40), m9)
    assert got.shape == (B, 9, 96) and torch.allclose(got, want, rtol=1e-5, atol=1e-4)
    with pytest.raises(RfmiError):   # the last segment would read past the row
        ops.fold_mean(w9, m9, None, k0=8, nseg=9, seg_stride=40, sum_seg=False)


@pytest.mark.parametrize("dil", [1, 2])
@pytest.mark.parametrize("B,H,W,C", [(1, 16, 16, 32), (2, 9, 12, 24)])
def test_conv3x3_border_identity(B, H, W, C, dil):
    """conv(x - c) fixed at the border == conv(x) - (sum of all taps . c): everywhere, to fp32 rounding."""
    wt = rn(C, C, 3, 3) / (3 * C ** 0.5)
    c = 10.0 * rn(B, C, seed=2)
    x = rn(B, H, W, C) + c[:, None, None, :]
    conv = lambda t: F.conv2d(t.permute(0, 3, 1, 2), wt, padding=dil, dilation=dil).permute(0, 2, 3, 1).contiguous()  # noqa: E731
    w32 = wt.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    taps = ops.fold_mean(w32, c.contiguous(), None, nseg=9, seg_stride=C, sum_seg=False)
    y = ops.conv3x3_border_fix(conv(x - c[:, None, None, :]), taps, dil)
    want = conv(x) - taps.sum(1)[:, None, None, :]
    assert torch.allclose(y, want, atol=2e-4), (y - want).abs().max()
    # a row block of the picture (neighbours above and below supply the halo rows): only left | right are edges
    if H >= 4 * dil + 1:
        full = conv(x - c[:, None, None, :])
        blk = full[:, 2 * dil:H - 2 * dil].clone()
        got = ops.conv3x3_border_fix(blk, taps, dil, edges=12)
        assert torch.allclose(got, want[:, 2 * dil:H - 2 * dil], atol=2e-4)
        top = ops.conv3x3_border_fix(full[:, :H - 2 * dil].clone(), taps, dil, edges=13)
        assert torch.allclose(top, want[:, :H - 2 * dil], atol=2e-4)
        bot = ops.conv3x3_border_fix(full[:, 2 * dil:].clone(), taps, dil, edges=14)
        assert torch.allclose(bot, want[:, 2 * dil:], atol=2e-4)


def test_border_fix_16bit_output_touches_only_the_border():
    B, H, W, C = 1, 12, 12, 16
    y = rn(B, H, W, C).to(torch.bfloat16)
    taps = rn(B, 9, C)
    z = ops.conv3x3_border_fix(y.clone(), taps, 1)
    assert torch.equal(z[:, 1:-1, 1:-1], y[:, 1:-1, 1:-1])
    corner = y[0, 0, 0].float() - (taps[0, 0] + taps[0, 1] + taps[0, 2] + taps[0, 3] + taps[0, 6])
    assert torch.equal(z[0, 0, 0], corner.to(torch.bfloat16))
    edge = y[0, 5, 11].float() - (taps[0, 2] + taps[0, 5] + taps[0, 8])
    assert torch.equal(z[0, 5, 11], edge.to(torch.bfloat16))


# ---- in the model -----------------------------------------------------------------------------------------------------------------
def _pum(d_msa=64, d_pair=288, seed=0):
    torch.manual_seed(seed)
    m = R.PairUpdateWithMsa(d_msa=d_msa, d_proj=32, d_pair=d_pair, n_heads=4, p_dropout=0.0).to(DEV)
    with torch.no_grad():   # biases / LayerNorm offsets of a trained-size order: what makes the common mode at random init
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.5 * torch.randn(p.shape, generator=torch.Generator().manual_seed(seed + p.numel())).to(DEV))
    return m


def _pum_inputs(B, N, L, d_msa, d_pair, offset, seed=0):
    msa = rn(B, N, L, d_msa, seed=seed) + offset * rn(B, 1, 1, d_msa, seed=seed + 1)
    pair = rn(B, L, L, d_pair, seed=seed + 2) + offset * rn(B, 1, 1, d_pair, seed=seed + 3)
    att = torch.softmax(rn(B, L, L, 4, seed=seed + 4), 2)
    return msa, pair, att


@pytest.mark.parametrize("mode", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B", [1, 2])
def test_pair_update_with_msa_conditioning(mode, B):
    m = _pum()
    msa, pair, att = _pum_inputs(B, 16, 32, 64, 288, offset=8.0)
    R.set_compute_dtype(torch.float32)
    ref = m(msa, pair, att)
    R.set_compute_dtype(mode)
    M.RT.condition = False
    plain = rel(m(msa, pair, att), ref)
    M.RT.condition = True
    cond = rel(m(msa, pair, att), ref)
    print(f"\n[PairUpdateWithMsa {mode} B={B}] plain {plain:.3e}  conditioned {cond:.3e}")
    assert cond < 0.5 * plain, (plain, cond)
    # and without a common mode in the inputs the two forms are equally good (same roundings up to the centring)
    msa, pair, att = _pum_inputs(B, 16, 32, 64, 288, offset=0.0, seed=7)
    R.set_compute_dtype(torch.float32)
    ref = m(msa, pair, att)
    R.set_compute_dtype(mode)
    M.RT.condition = False
    plain = rel(m(msa, pair, att), ref)
    M.RT.condition = True
    cond = rel(m(msa, pair, att), ref)
    assert cond < 1.3 * plain + 1e-4, (plain, cond)


def test_pair_update_with_msa_fp32_mode_is_untouched():
    m = _pum()
    msa, pair, att = _pum_inputs(1, 16, 32, 64, 288, offset=8.0)
    R.set_compute_dtype(torch.float32)
    M.RT.condition = False
    a = m(msa, pair, att)
    M.RT.condition = True
    assert torch.equal(m(msa, pair, att), a)


@pytest.mark.parametrize("mode", [torch.float16, torch.bfloat16])
def test_prediction_head_conditioning(mode):
    torch.manual_seed(1)
    head = R.PredictionHead(in_channels=64, n_res_blocks=2, p_dropout=0.0).to(DEV)
    pair = rn(1, 48, 48, 64) + 8.0 * rn(1, 1, 1, 64, seed=3)
    R.set_compute_dtype(torch.float32)
    ref = head(pair)
    R.set_compute_dtype(mode)
    M.RT.head_center = False
    plain = max(rel(v, ref[k]) for k, v in head(pair).items())
    M.RT.head_center = True
    cond = max(rel(v, ref[k]) for k, v in head(pair).items())
    print(f"\n[PredictionHead {mode}] plain {plain:.3e}  conditioned {cond:.3e}")
    assert cond < 0.5 * plain, (plain, cond)


# ---- the attention layers' value path (RFModule.value_conditioning) ------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sample_mean(dtype):
    B, R_, C = 2, 5000, 40
    x = (rn(B, R_, C) + 4.0 * rn(1, 1, C, seed=1)).to(dtype)
    got = ops.sample_mean(x, nsample=512)
    rows = (torch.arange(512, device=DEV) * R_) // 512
    want = x[:, rows].float().mean(1)
    assert torch.allclose(got, want, atol=1e-4)
    assert (got - x.float().mean(1)).abs().max() < 0.25          # and it is an estimate of the mean (sigma / sqrt(512) = 0.044)
    few = ops.sample_mean(x[:, :7].contiguous(), nsample=512)     # more samples asked than rows: every row once
    assert torch.allclose(few, x[:, :7].float().mean(1), atol=1e-4)


def _offset_biases(m, seed, scale=0.5):
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(scale * torch.randn(p.shape, generator=torch.Generator().manual_seed(seed + p.numel())).to(DEV))


@pytest.mark.parametrize("mode", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind,B,N,L", [("tied", 1, 16, 64), ("tied", 2, 16, 64), ("tied", 1, 128, 256), ("performer", 1, 64, 32),
                                         ("performer", 2, 64, 32), ("performer", 1, 128, 64)])
def test_encoder_layer_value_conditioning(kind, B, N, L, mode):
    """The attention update of an MSA encoder layer (rf.py:284-354: to_out(attention(LN(x)))) on a stream with a per-sample
    constant: the error of its position-dependent part against the exact-fp32 mode drops several-fold."""
    torch.manual_seed(5)
    D = 384 if (N, L) == (128, 256) else 96
    lay = R.EncoderLayer(d_msa=D, d_ff=2 * D, n_heads=D // 32, p_dropout=0.0, tied=(kind == "tied"), performer=(kind == "performer")).to(DEV)
    _offset_biases(lay, 11)
    # the constant is shared by the samples of a batch (biases, LayerNorm offsets, the mean embedding) up to a small per-sample
    # part: ONE conditioning constant serves the whole batch (RFModule.value_conditioning)
    x = rn(B, N, L, D) + 6.0 * rn(1, 1, 1, D, seed=1) + 0.5 * rn(B, 1, 1, D, seed=2)

    def update():
        xn = M.ln(lay.ln, x)
        res = ops.zeros(B, N, L, D, device=DEV, dtype=torch.float32)
        if kind == "tied":
            lay.attn.attend(xn, res, False)
        else:
            lay.attn.attend(xn, res, 2)
        return res - res.mean(dim=(1, 2), keepdim=True)   # the part of the update that varies over the positions
    R.set_compute_dtype(torch.float32)
    ref = update()
    R.set_compute_dtype(mode)
    M.RT.condition = False
    plain = rel(update(), ref)
    M.RT.condition = True
    cond = rel(update(), ref)
    print(f"\n[attention update {kind} B={B} N={N} L={L} {mode}] plain {plain:.3e}  conditioned {cond:.3e}")
    assert cond < 0.5 * plain, (plain, cond)


def test_value_conditioning_is_exact_algebra_in_fp32_terms():
    """v - c, then W_o (o - c) + (b_o + W_o c): with the attention weights summing to one the result is the unconditioned one --
    checked on the tied layer in fp16 with inputs WITHOUT a common mode (conditioning must then change nothing beyond rounding)."""
    torch.manual_seed(6)
    lay = R.EncoderLayer(d_msa=96, d_ff=192, n_heads=3, p_dropout=0.0, tied=True).to(DEV)
    x = rn(1, 16, 64, 96)
    R.set_compute_dtype(torch.float32)
    ref = lay(x) - x
    R.set_compute_dtype(torch.float16)
    M.RT.condition = False
    plain = rel(lay(x) - x, ref)
    M.RT.condition = True
    cond = rel(lay(x) - x, ref)
    assert cond < 1.5 * plain + 1e-4 and plain < 1e-2, (plain, cond)
