"""A3M -> model inputs and model outputs -> maps / PDB (rosettafold-pytorch_amd/featurize.py).  CPU only."""
import math

import pytest
import torch

from rosettafold_pytorch_amd import featurize as F

A3M = """>query
MKV-LAAGX
>hit1 with insertions
MKVaaLLAAGA
>hit2
.KVGLAAG-
>dup
MKV-LAAGX
"""


def test_tokenize_alphabet():
    t = F.tokenize("ARNDCQEGHILKMFPSTWYV-XbZ.")
    assert t.tolist() == list(range(21)) + [20, 20, 20, 20]
    assert len(F.ALPHABET) == 21 and F.GAP == 20  # d_input=21 of the reference's embeddings (rf.py:106-120)


def test_parse_a3m_drops_insertions_and_duplicates():
    msa, names = F.parse_a3m(A3M)
    assert msa.shape == (3, 9) and msa.dtype == torch.long
    assert names == ["query", "hit1 with insertions", "hit2"]
    assert msa[0].tolist() == F.tokenize("MKV-LAAGX").tolist()
    assert msa[1].tolist() == F.tokenize("MKVLLAAGA").tolist()  # lower-case insertion columns dropped
    assert msa[2, 0].item() == 20 and msa[2, -1].item() == 20  # '.' and '-' -> gap
    assert F.parse_a3m(A3M, dedup=False)[0].shape[0] == 4
    assert F.parse_a3m(A3M, max_seqs=2)[0].shape[0] == 2


def test_parse_a3m_errors():
    with pytest.raises(ValueError):
        F.parse_a3m("")
    with pytest.raises(ValueError):
        F.parse_a3m(">a\nMKV\n>b\nMK\n")


def test_featurize_shapes_and_chain_break():
    msa, seq, aa = F.featurize(A3M, chain_lengths=[4, 5])
    assert msa.shape == (1, 3, 9) and seq.shape == (1, 9) and aa.shape == (1, 9)
    assert torch.equal(seq[0], msa[0, 0])
    assert aa[0].tolist() == [0, 1, 2, 3, 204, 205, 206, 207, 208]
    with pytest.raises(IndexError):
        F.featurize(A3M, chain_lengths=[4, 5], max_len=100)
    with pytest.raises(ValueError):
        F.residue_index(9, [4, 4])
    b = F.collate([F.featurize(A3M), F.featurize(A3M, n_seq=1)])
    assert b[0].shape == (2, 3, 9) and (b[0][1, 1:] == F.GAP).all() and b[1].shape == (2, 9)


def test_decode_logits_bins():
    B, L = 1, 4
    lg = {k: torch.full((B, L, L, n), -30.0) for k, n in (("dist", 37), ("omega", 37), ("theta", 37), ("phi", 19))}
    lg["dist"][0, 0, 1, 4] = 30.0     # bin 4: 4.0 .. 4.5 A
    lg["dist"][0, 0, 2, 36] = 30.0    # no contact
    lg["dist"][0, 0, 3, 20] = 30.0    # 12.0 .. 12.5 A
    lg["omega"][0, 0, 1, 0] = 30.0    # -180 .. -170 deg
    lg["theta"][0, 0, 1, 27] = 30.0   # 90 .. 100 deg
    lg["phi"][0, 0, 1, 9] = 30.0      # 90 .. 100 deg
    d = F.decode_logits(lg)
    assert d["dist_argmax"][0, 0, 1] == 4 and abs(d["dist_expected"][0, 0, 1].item() - 4.25) < 1e-3
    assert d["p_contact"][0, 0, 1] > 0.999 and d["p_contact"][0, 0, 3] < 1e-3 and d["p_no_contact"][0, 0, 2] > 0.999
    assert abs(d["dist_expected"][0, 0, 3].item() - 12.25) < 1e-3
    assert abs(math.degrees(d["omega"][0, 0, 1].item()) + 175.0) < 1e-2
    assert abs(math.degrees(d["theta"][0, 0, 1].item()) - 95.0) < 1e-2
    assert abs(math.degrees(d["phi"][0, 0, 1].item()) - 95.0) < 1e-2
    assert F.dist_bin_centers()[0].item() == pytest.approx(2.25) and F.dist_bin_centers()[-1].item() == pytest.approx(19.75)


def test_pdb_round_trip():
    g = torch.Generator().manual_seed(0)
    L = 7
    xyz = torch.randn(L, 3, 3, generator=g) * 10
    seq = torch.randint(0, 21, (L,), generator=g)
    pl = torch.rand(L, generator=g)
    txt = F.to_pdb(xyz, seq, pl, chain_lengths=[3, 4])
    lines = txt.splitlines()
    assert sum(ln.startswith("ATOM") for ln in lines) == 3 * L and lines[-1] == "END"
    assert all(len(ln) == 78 for ln in lines if ln.startswith("ATOM"))  # fixed PDB columns
    x2, s2, b2 = F.from_pdb_backbone(txt)
    assert torch.allclose(x2, xyz, atol=6e-4) and torch.equal(s2, seq.clamp(max=20))
    assert torch.allclose(b2 / 100, pl, atol=6e-3)
    assert " A" in lines[0] and any(ln[21] == "B" for ln in lines if ln.startswith("ATOM"))
    with pytest.raises(ValueError):
        F.to_pdb(xyz[:, :2], seq)
