"""The algebra behind csrc/condition.hip, in float64 on the CPU (no library call): the identities the operand conditioning of the
16-bit modes relies on hold for ANY constant, so the conditioned forward computes the reference's function (rf.py:252-267,
313-318, 452-457, 476-498, 1130-1140) exactly; only where the 16-bit rounding lands changes.  tests/test_condition_gpu.py checks the
kernels and their effect on the GPU."""
import numpy as np
import torch
import torch.nn.functional as F

G = torch.Generator().manual_seed(0)


def rn(*s):
    return torch.randn(*s, generator=G, dtype=torch.float64)


def test_linear_with_a_constant_removed_from_its_operand():
    """W (f - m) + (b + W m) == W f + b   (PairUpdateWithMsa's tiled 1-D features, rf.py:476-498)"""
    f, W, b, m = rn(50, 24), rn(16, 24), rn(16), rn(24)
    assert torch.allclose(F.linear(f - m, W, b + W @ m), F.linear(f, W, b), atol=1e-12)


def test_instance_norm_drops_a_per_channel_constant_of_a_bias_free_conv1x1_input():
    """PredictionHead (rf.py:1130-1140; resnet.py:57-60): conv1x1 (no bias) -> InstanceNorm sees W (x - c) like W x"""
    x, W, c = rn(2, 12, 9, 9), rn(8, 12, 1, 1), rn(12)
    a = F.instance_norm(F.conv2d(x, W))
    b = F.instance_norm(F.conv2d(x - c[None, :, None, None], W))
    assert torch.allclose(a, b, atol=1e-10)


def test_conv3x3_of_a_centred_picture_plus_the_border_terms():
    """conv3x3(x - c) - [taps that fall outside the picture] . c == conv3x3(x) - (sum of ALL taps) . c: a per-channel constant,
    which the InstanceNorm behind the convolution drops (rf.py:452-453)"""
    for dil in (1, 2):
        B, C, H, W_ = 2, 6, 9, 11
        x, wt, c = rn(B, C, H, W_), rn(5, C, 3, 3), rn(B, C)
        full = F.conv2d(x, wt, padding=dil, dilation=dil)
        centred = F.conv2d(x - c[:, :, None, None], wt, padding=dil, dilation=dil)
        taps = torch.einsum("ockl,bc->bklo", wt, c)                      # [B, 3, 3, Co]: W_tap c
        fix = torch.zeros_like(centred)
        for i in range(H):
            for j in range(W_):
                for kh in range(3):
                    for kw in range(3):
                        ii, jj = i + (kh - 1) * dil, j + (kw - 1) * dil
                        if ii < 0 or ii >= H or jj < 0 or jj >= W_:
                            fix[:, :, i, j] += taps[:, kh, kw]
        want = full - taps.sum((1, 2))[:, :, None, None]
        assert torch.allclose(centred - fix, want, atol=1e-11)
        assert torch.allclose(F.instance_norm(centred - fix), F.instance_norm(full), atol=1e-9)


def test_attention_on_centred_values():
    """softmax rows sum to one: sum_j a_ij (v_j - c) = o_i - c, so W_o (o - c) + (b_o + W_o c) == W_o o + b_o with
    c = W_v mu + b_v for ANY mu (rf.py:252-267): the value block of the projection takes the bias -W_v mu, to_out the bias
    b_o + W_o b_v + (W_o W_v) mu"""
    L, D = 17, 12
    xn, Wv, bv, Wo, bo, mu = rn(L, D), rn(D, D), rn(D), rn(D, D), rn(D), rn(D)
    att = torch.softmax(rn(L, L), -1)
    ref = F.linear(att @ F.linear(xn, Wv, bv), Wo, bo)
    v_c = F.linear(xn, Wv, -Wv @ mu)
    got = F.linear(att @ v_c, Wo, bo + Wo @ bv + (Wo @ Wv) @ mu)
    assert torch.allclose(got, ref, atol=1e-11)


def test_linear_attention_on_centred_values():
    """FAVOR+ (rf.py:313-318; performer): o_i = (q'_i . sum_s k'_s v_s^T) / (q'_i . sum_s k'_s): weights that sum to one as well"""
    S, M_, D = 23, 9, 8
    qf, kf, v, c = rn(S, M_).abs(), rn(S, M_).abs(), rn(S, D), rn(D)
    def favor(vals):
        ctx, ksum = kf.t() @ vals, kf.sum(0)
        return (qf @ ctx) / (qf @ ksum)[:, None]
    assert torch.allclose(favor(v - c) + c, favor(v), atol=1e-11)
