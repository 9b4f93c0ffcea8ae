"""Pin the CPU oracle (oracle/rf_oracle.py) against golden vectors captured from the
reference's own modules by tools/make_goldens.py.  fp32, rtol 1e-4/atol 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import rf_oracle as O


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    I = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in:")}
    Y = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out:")}
    X = {k[2:]: z[k] for k in z.files if k.startswith("x:")}
    return P, I, Y, X


def pre(P, prefix):
    """Re-key a module-level state dict under `prefix.` so oracle functions can address it."""
    return {prefix + "." + k: v for k, v in P.items()}


def close(a, b, rtol=1e-4, atol=1e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_pair_embedding_template(golden_dir):
    """use_template=True branch (rf.py:141-169)."""
    P, I, Y, X = load(golden_dir, "pair_embedding_template")
    close(O.pair_embedding(pre(P, "m"), "m", I["seq"], I["aa_idx"], int(X["max_len"]), template=I["template"]), Y["y"])


def test_msa_embedding(golden_dir):
    P, I, Y, X = load(golden_dir, "msa_embedding")
    close(O.msa_embedding(pre(P, "m"), "m", I["msa"], I["aa_idx"], int(X["max_len"])), Y["y"])


def test_pair_embedding(golden_dir):
    P, I, Y, X = load(golden_dir, "pair_embedding")
    close(O.pair_embedding(pre(P, "m"), "m", I["seq"], I["aa_idx"], int(X["max_len"])), Y["y"])


def test_poswise_weight(golden_dir):
    P, I, Y, X = load(golden_dir, "poswise_weight")
    y = O.poswise_weight(pre(P, "m"), "m", I["x"], int(X["n_heads"]))
    close(y, Y["y"])
    close(y.sum(1), torch.ones_like(y.sum(1)))  # reference tests/test_module.py:180-200


def test_soft_tied_attention(golden_dir):
    P, I, Y, X = load(golden_dir, "soft_tied_attention")
    out, att = O.soft_tied_attention(pre(P, "m"), "m", I["x"], int(X["n_heads"]))
    close(out, Y["out"])
    close(att, Y["att"])


def test_encoder_layer_tied(golden_dir):
    P, I, Y, X = load(golden_dir, "encoder_layer_tied")
    out, att = O.encoder_layer_tied(pre(P, "m"), "m", I["x"], int(X["n_heads"]))
    close(out, Y["out"])
    close(att, Y["att"])


def test_outer_product_mean(golden_dir):
    P, I, Y, _ = load(golden_dir, "outer_product_mean")
    close(O.outer_product_mean(pre(P, "m"), "m", I["x"], I["y"]), Y["y"])


def test_pair_update_with_msa(golden_dir):
    P, I, Y, _ = load(golden_dir, "pair_update_with_msa")
    close(O.pair_update_with_msa(pre(P, "m"), "m", I["msa"], I["pair"], I["att"]), Y["y"], 1e-4, 2e-5)


def test_msa_update_with_pair_layer(golden_dir):
    P, I, Y, X = load(golden_dir, "msa_update_with_pair_layer")
    close(O.msa_update_with_pair_layer(pre(P, "m"), "m", I["msa"], I["pair"], int(X["n_heads"])), Y["y"])


def test_msa_update_with_pair_hidden_list(golden_dir):
    P, I, Y, X = load(golden_dir, "msa_update_with_pair")
    assert any(k.startswith("encoder_layers.1.") for k in P)  # hidden-list weights were exported
    y = O.msa_update_with_pair(pre(P, "m"), "m", I["msa"], I["pair"], int(X["n_layers"]), int(X["n_heads"]))
    close(y, Y["y"])


def test_graph_transformer_block(golden_dir):
    P, I, Y, X = load(golden_dir, "graph_transformer_block")
    close(O.graph_transformer_block(pre(P, "m"), "m", I["node"], I["edge"], int(X["n_heads"])), Y["y"])


def test_initial_coord_generation(golden_dir):
    P, I, Y, X = load(golden_dir, "initial_coord_generation")
    y = O.initial_coord_generation(pre(P, "m"), "m", I["msa"], I["pair"], I["seq_onehot"], I["aa_idx"],
                                   int(X["n_layers"]), int(X["n_heads"]))
    close(y, Y["y"])


@pytest.mark.parametrize("k", [4, 32])
def test_knn_graph(golden_dir, k):
    _, I, Y, X = load(golden_dir, f"knn_graph_k{k}")
    xyz, idx, edge = I["xyz"], I["idx"], I["edge"]
    L = xyz.shape[1]
    b, i, j = O.knn_graph(xyz, idx, int(X["n_neighbors"]))
    assert torch.equal(b * L + i, Y["src"]) and torch.equal(b * L + j, Y["dst"])  # bit-exact indices
    close(xyz[b, j, 1] - xyz[b, i, 1], Y["d"])
    close(edge[b, i, j], Y["w"])
    if k >= L:  # complete graph including self loops
        assert b.numel() == xyz.shape[0] * L * L


def test_msa_update_with_pair_and_coord(golden_dir):
    P, I, Y, _ = load(golden_dir, "msa_update_with_pair_and_coord")
    close(O.msa_update_with_pair_and_coord(pre(P, "m"), "m", I["xyz"], I["state"], I["msa"]), Y["y"])


def test_resnet(golden_dir):
    P, I, Y, X = load(golden_dir, "resnet")
    close(O.resnet(pre(P, "m"), "m", I["x"], int(X["n_blocks"])), Y["y"], 1e-4, 2e-5)


def test_prediction_head(golden_dir):
    P, I, Y, X = load(golden_dir, "prediction_head")
    out = O.prediction_head(pre(P, "m"), "m", I["pair"], int(X["n_blocks"]))
    for k in ("theta", "phi", "dist", "omega"):
        close(out[k], Y[k], 1e-4, 5e-5)
    assert out["phi"].shape[-1] == 19 and out["dist"].shape[-1] == 37


def test_spherical_harmonics(golden_dir):
    _, I, Y, _ = load(golden_dir, "spherical_harmonics")
    Ys = O.real_sh(I["d"])
    close(Ys[0], Y["Y0"], 1e-10, 1e-12)
    close(Ys[1], Y["Y1"], 1e-10, 1e-12)
    close(Ys[2], Y["Y2"], 1e-10, 1e-12)


def test_radial_func(golden_dir):
    P, I, Y, _ = load(golden_dir, "radial_func")
    close(O.radial_func(pre(P, "m"), "m", I["feat"]).view(Y["y"].shape), Y["y"])


def test_pairwise_conv(golden_dir):
    P, I, Y, _ = load(golden_dir, "pairwise_conv")
    # the oracle's gconv_partial with one (d_in=1, d_out=1) pair and an identity source feature
    # reproduces PairwiseConv's kernel: feed unit vectors to read the kernel columns back
    E, mo, mi = I["feat"].shape[0], 5, 4
    R = O.radial_func(pre(P, "m.kernel_unary.(1,1)"), "m.kernel_unary.(1,1).rp", I["feat"]).view(E, mo, 1, mi, 1, 3)
    kern = (R * I["basis11"]).sum(-1).reshape(E, mo * 3, mi * 3)
    close(kern, Y["y"])


def test_g1x1_gnorm_selfint(golden_dir):
    P, I, Y, _ = load(golden_dir, "g1x1")
    o = O.g1x1(pre(P, "m"), "m", {0: I["h0"], 1: I["h1"]}, [0, 1])
    close(o[0], Y["o0"]); close(o[1], Y["o1"])
    P, I, Y, _ = load(golden_dir, "gnorm_bias")
    o = O.gnorm_bias(pre(P, "m"), "m", {0: I["h0"], 1: I["h1"]})
    close(o[0], Y["o0"]); close(o[1], Y["o1"])
    P, I, Y, _ = load(golden_dir, "gattentive_selfint")
    o = O.gattentive_selfint(pre(P, "m"), "m", {0: I["h0"], 1: I["h1"]}, {0: 4, 1: 3})
    close(o[0], Y["o0"]); close(o[1], Y["o1"])


# ---- config-1-sized cases (B=1, N=8, L=64, d_msa=96, d_pair=64): the model's real head counts ----
def load16(golden_dir, name):
    P, I, Y, X = load(golden_dir, name)
    return {k: v.float() for k, v in P.items()}, {k: v.float() for k, v in I.items()}, Y, X


def test_c1_soft_tied_attention(golden_dir):
    P, I, Y, X = load16(golden_dir, "c1_soft_tied_attention")
    out, att = O.soft_tied_attention(pre(P, "m"), "m", I["x"], int(X["n_heads"]))
    close(out, Y["out"])
    close(att, Y["att"])


def test_c1_pair_update_with_msa(golden_dir):
    P, I, Y, X = load16(golden_dir, "c1_pair_update_with_msa")
    s = int(X["stride"])
    y = O.pair_update_with_msa(pre(P, "m"), "m", I["msa"], I["pair"], I["att"])
    close(y[:, ::s, ::s], Y["y_sub2"], rtol=2e-4, atol=2e-5)
