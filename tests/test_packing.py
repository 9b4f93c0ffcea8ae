"""Host-side weight packing (no GPU): the fragment orders documented in include/rfmi.h for rf_ffn_fused and
rf_outer_product_pairs, checked element by element against the index formulas of the header."""
import torch

import rosettafold_pytorch_amd as R  # noqa: F401  (loads both libraries: works without a GPU)
from rosettafold_pytorch_amd import ops


def test_ffn_pack_matches_the_header_formula():
    D, hid = 288, 64
    g = torch.Generator().manual_seed(0)
    w1 = torch.randn(hid, D, generator=g)
    w2 = torch.randn(D, hid, generator=g)
    p = ops.ffn_pack(w1, w2, torch.float32).view(hid // 32, -1)      # [chunk][2 KS pieces of W1 | NT pieces of W2][512]
    KS, NT = D // 32, D // 16
    for c in range(hid // 32):
        pieces = p[c].view(2 * KS + NT, 64, 8)
        for ks in range(KS):
            for ht in range(2):
                for lane in (0, 5, 17, 63):
                    fr, fq = lane & 15, lane >> 4
                    for j in range(8):
                        assert pieces[ks * 2 + ht, lane, j] == w1[32 * c + 16 * ht + fr, 32 * ks + 8 * fq + j]
        for nt in range(NT):
            for lane in (0, 9, 31, 63):
                fr, fq = lane & 15, lane >> 4
                for j in range(8):
                    assert pieces[2 * KS + nt, lane, j] == w2[16 * nt + fr, 32 * c + 16 * (j >> 2) + 4 * fq + (j & 3)]


def test_outer_fold_layouts_match_the_header_formulas():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(288, 1024, generator=g)
    gamma, beta, bias = torch.rand(1024, generator=g) + 0.5, torch.randn(1024, generator=g), torch.randn(288, generator=g)
    wq, s, c = ops.outer_fold(w, gamma, beta, bias, torch.float32, pairs=True)
    wp = w * gamma[None, :]
    assert tuple(wq.shape) == (32, 18, 64, 8)
    for v in (0, 7, 31):
        for ot in (0, 8, 17):
            for lane in (0, 21, 63):
                fr, fq = lane & 15, lane >> 4
                for e in range(8):
                    assert wq[v, ot, lane, e] == wp[16 * ot + fr, (16 * (e >> 2) + 4 * fq + (e & 3)) * 32 + v]
    assert torch.allclose(s, wp.sum(1), rtol=1e-5, atol=1e-4) and torch.allclose(c, w @ beta + bias, rtol=1e-5, atol=1e-4)
    wc, s2, c2 = ops.outer_fold(w, gamma, beta, bias, torch.float32, pairs=False)
    assert tuple(wc.shape) == (16, 288, 64) and torch.equal(s, s2) and torch.equal(c, c2)
    for ch in (0, 5, 15):
        for o in (0, 100, 287):
            for f in (0, 9, 63):
                ug, vg, uu, vv = ch // 4, ch % 4, f // 8, f % 8
                assert wc[ch, o, f] == wp[o, (8 * ug + uu) * 32 + 8 * vg + vv]
