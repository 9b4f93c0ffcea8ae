"""Weight I/O: a reference checkpoint = registered state_dict + the list-held layers
(rf.py:602-605, 699-702).  The strict loader refuses a checkpoint that would leave the list-held layers at random
initialisation.  CPU only (modules are constructed and loaded, never run)."""
import pytest
import torch
import torch.nn as nn

import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import weights as W

CFG = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=4, n_two_track_blocks=1, n_three_track_blocks=2,
           n_encoder_layers=2, max_len=64, n_neighbors=[16, 16], p_dropout=0.0)


class _RefLike(nn.Module):
    """A module tree shaped like the reference's: one registered Linear and a plain Python list of layers."""

    def __init__(self):
        super().__init__()
        self.front = nn.Linear(3, 3)
        self.encoder_layers = [nn.Linear(3, 3) for _ in range(2)]  # hidden from state_dict(), like rf.py:602


def test_export_hidden_lists_names():
    m = _RefLike()
    assert "encoder_layers.0.weight" not in m.state_dict()
    h = W.export_hidden_lists(m)
    assert sorted(h) == ["encoder_layers.0.bias", "encoder_layers.0.weight", "encoder_layers.1.bias", "encoder_layers.1.weight"]
    assert torch.equal(h["encoder_layers.1.weight"], m.encoder_layers[1].weight)


def test_strict_loader_needs_the_hidden_lists(tmp_path):
    torch.manual_seed(0)
    src = R.RoseTTAFold(**CFG)
    torch.manual_seed(1)
    dst = R.RoseTTAFold(**CFG)
    full = {k: v.clone() for k, v in src.state_dict().items()}
    hid = set(W.hidden_list_keys(src))
    assert len(hid) > 0 and all((".encoder_layers." in k or ".blocks." in k) for k in hid)
    registered = {k: v for k, v in full.items() if k not in hid}   # what a reference state_dict() can hold
    hidden = {k: v for k, v in full.items() if k in hid}           # what export_hidden_lists() supplies
    with pytest.raises(KeyError, match="list-held"):
        W.load_reference_weights(dst, registered)
    rep = W.load_reference_weights(dst, registered, allow_missing_hidden=True)
    assert sorted(rep["missing_hidden"]) == sorted(hid)
    k0 = sorted(k for k in hid if full[k].dim() == 2)[0]  # a Linear weight (LayerNorm weights start equal: all ones)
    assert not torch.equal(dst.state_dict()[k0], full[k0])          # kept its own initialisation
    rep = W.load_reference_weights(dst, registered, hidden)
    assert rep["missing_hidden"] == [] and rep["loaded"] == len(full)
    for k, v in dst.state_dict().items():
        assert torch.equal(v, full[k]), k
    # unknown / missing registered keys and shape mismatches are errors, not warnings
    with pytest.raises(KeyError, match="does not have"):
        W.load_reference_weights(dst, dict(full, bogus=torch.zeros(1)))
    some = next(k for k in registered if k.endswith("to_q.weight"))
    with pytest.raises(KeyError, match="registered"):
        W.load_reference_weights(dst, {k: v for k, v in full.items() if k != some})
    with pytest.raises(ValueError, match="shape"):
        W.load_reference_weights(dst, dict(full, **{some: torch.zeros(2, 2)}))
    # round trip through a file
    path = tmp_path / "ckpt.pt"
    W.save_checkpoint(src, str(path))
    torch.manual_seed(2)
    third = R.RoseTTAFold(**CFG)
    W.load_checkpoint(third, str(path))
    assert all(torch.equal(v, full[k]) for k, v in third.state_dict().items())


def test_weight_cache_fingerprint_tracks_in_place_edits():
    """kernel-ready weight copies must not outlive p.copy_ / nn.init / load_state_dict."""
    from rosettafold_pytorch_amd.model import weights_fingerprint
    m = R.FeedForward(8, 16)
    f0 = weights_fingerprint(m)
    with torch.no_grad():
        m.net[0].weight.mul_(2.0)
    f1 = weights_fingerprint(m)
    assert f0 != f1
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert weights_fingerprint(m) != f1
