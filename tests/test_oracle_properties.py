"""Property tests of the oracle for the parts no reference fixture can pin (dgl / performer / lie_learn are absent):
SE(3) equivariance of the restated structure module, FAVOR+ feature identities, sinusoid identity."""
import math
import os
import subprocess
import sys

import torch

from oracle import rf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_rot(g):
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def test_qj_constants_satisfy_intertwiner():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "derive_qj.py")], capture_output=True).returncode == 0


def test_se3_transformer_equivariance():
    """rotate + translate the coordinates: type-0 output invariant, type-1 output rotates (fp64)."""
    import rosettafold_pytorch_amd as R
    torch.manual_seed(3)
    B, N, L, DM, DP = 1, 4, 12, 24, 16
    m = R.CoordUpdateWithMsaAndPair(DM, DP, 8, 8, 4, n_neighbors=5, p_dropout=0.0)
    P = {"m." + k: v.detach().double() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    msa, pair = torch.randn(B, N, L, DM, generator=g).double(), torch.randn(B, L, L, DP, generator=g).double()
    xyz = (torch.cumsum(torch.randn(B, L, 1, 3, generator=g), 1) * 3 + torch.randn(B, L, 3, 3, generator=g)).double()
    aa = torch.arange(L)[None]
    oh = torch.nn.functional.one_hot(torch.randint(0, 21, (B, L), generator=g), 21).double()
    Rm, t = rand_rot(g).double(), torch.randn(3, generator=g).double()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        s0, x0 = O.coord_update(P, "m", xyz, msa, pair, aa, oh, 5, 4)
        s1, x1 = O.coord_update(P, "m", xyz @ Rm.t() + t, msa, pair, aa, oh, 5, 4)
    finally:
        torch.set_default_dtype(old)
    assert torch.allclose(s0, s1, atol=1e-9)
    assert torch.allclose(x0 @ Rm.t() + t, x1, atol=1e-8)


def test_favor_softmax_features_estimate_the_softmax_kernel():
    """E_P[phi(q).phi(k)] = exp(q.k/sqrt(d)) up to the common stabilisers: the feature map is the FAVOR+ one."""
    g = torch.Generator().manual_seed(0)
    d, m = 16, 20000
    P = torch.randn(m, d, generator=g)
    q, k = 0.5 * torch.randn(1, 1, d, generator=g), 0.5 * torch.randn(1, 3, d, generator=g)
    fq, fk = O.favor_softmax_features(q, P, True, eps=0.0), O.favor_softmax_features(k, P, False, eps=0.0)
    est = (fq @ fk.transpose(-1, -2))[0, 0]
    exact = torch.exp((q @ k.transpose(-1, -2))[0, 0] / math.sqrt(d))
    assert torch.allclose(est / est[0], exact / exact[0], rtol=0.08)


def test_linear_attention_matches_quadratic_form():
    g = torch.Generator().manual_seed(1)
    q, k, v = torch.rand(2, 3, 10, 7, generator=g), torch.rand(2, 3, 10, 7, generator=g), torch.randn(2, 3, 10, 5, generator=g)
    a = q @ k.transpose(-1, -2)
    ref = (a / a.sum(-1, keepdim=True)) @ v
    assert torch.allclose(O.linear_attention(q, k, v), ref, atol=1e-5)


def test_sinusoid_identity():
    pe = O.sinusoid_table(32, 50)  # reference tests/test_module.py:35-50
    assert torch.allclose(pe[:, 0::2] ** 2 + pe[:, 1::2] ** 2, torch.ones(50, 16), atol=1e-6)
