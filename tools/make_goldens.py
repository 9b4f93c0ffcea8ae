#!/usr/bin/env python3
"""Capture golden vectors from the reference's own modules (build container only).

Runs the importable parts of /root/reference at tiny
shapes with seeded inputs and writes inputs + all weights (including the layers the
reference hides in plain Python lists) + outputs to tests/golden/*.npz.  The four packages
the reference imports but this image lacks (dgl, pytorch_lightning, performer_pytorch,
lie_learn) get inert stand-ins holding NO arithmetic, so every captured number comes from
the reference's own code + PyTorch.  Nothing under /root/reference is copied.

    python tools/make_goldens.py        # from anywhere; needs /root/reference
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def install_standins():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    sys.modules["pytorch_lightning"] = pl

    pp = types.ModuleType("performer_pytorch")

    class SelfAttention(nn.Module):  # placeholder: raises if ever called
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            raise RuntimeError("performer_pytorch is not available (stand-in)")

    pp.SelfAttention = SelfAttention
    sys.modules["performer_pytorch"] = pp

    dgl = types.ModuleType("dgl")
    dgl.__version__ = "0.9.0"

    class GraphRecorder:  # records (src, dst, num_nodes) and edge data; no arithmetic
        def __init__(self, edges, num_nodes=None):
            self.src, self.dst = edges
            self.num_nodes = num_nodes
            self.edata = {}
            self.ndata = {}

        def to(self, device):
            return self

    dgl.graph = GraphRecorder
    dgl.DGLGraph = GraphRecorder
    sys.modules["dgl"] = dgl
    fn = types.ModuleType("dgl.function")
    sys.modules["dgl.function"] = fn
    dgl.function = fn
    for name in ("dgl.nn", "dgl.nn.pytorch"):
        sys.modules[name] = types.ModuleType(name)
    sm = types.ModuleType("dgl.nn.pytorch.softmax")
    sm.edge_softmax = None
    sys.modules["dgl.nn.pytorch.softmax"] = sm
    gl = types.ModuleType("dgl.nn.pytorch.glob")
    gl.AvgPooling = nn.Identity
    gl.MaxPooling = nn.Identity
    sys.modules["dgl.nn.pytorch.glob"] = gl


def np_(t):
    return t.detach().cpu().numpy()


def full_state(module):
    """state_dict plus the weights of layers hidden in plain python lists."""
    sd = {k: v for k, v in module.state_dict().items()}
    for name, sub in module.named_modules():
        for attr in ("encoder_layers", "blocks"):
            lst = getattr(sub, attr, None)
            if isinstance(lst, list):
                for i, layer in enumerate(lst):
                    layer.eval()
                    for k, v in layer.state_dict().items():
                        key = f"{name + '.' if name else ''}{attr}.{i}.{k}"
                        sd[key] = v
    return sd


def save(name, module, inputs, outputs, extra=None):
    data = {}
    if module is not None:
        for k, v in full_state(module).items():
            data["w:" + k] = np_(v)
    for k, v in inputs.items():
        data["in:" + k] = np_(v) if torch.is_tensor(v) else np.asarray(v)
    for k, v in outputs.items():
        data["out:" + k] = np_(v) if torch.is_tensor(v) else np.asarray(v)
    for k, v in (extra or {}).items():
        data["x:" + k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{name:40s} {os.path.getsize(path) / 1024:8.1f} KB")


def main():
    os.makedirs(OUT, exist_ok=True)
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    scratch = tempfile.mkdtemp(prefix="rf_golden_")
    os.chdir(scratch)  # the reference's Q_J cache decorator creates ./cache
    install_standins()
    sys.path.insert(0, REF)
    import rosettafold_pytorch.rosettafold_pytorch as rf
    from rosettafold_pytorch.resnet import ResNet
    from rosettafold_pytorch.equivariant_attention import modules as ea
    from rosettafold_pytorch.equivariant_attention.fibers import Fiber
    from rosettafold_pytorch.equivariant_attention.from_se3cnn import utils_steerable as us

    torch.set_grad_enabled(False)
    B, N, L, dm, dp, H = 2, 5, 12, 16, 16, 2
    g = torch.Generator().manual_seed(7)

    def rn(*s):
        return torch.randn(*s, generator=g)

    aa_idx = torch.stack([torch.arange(L), torch.arange(L) + torch.tensor([0] * 6 + [15] * 6)])
    msa_tok = torch.randint(0, 21, (B, N, L), generator=g)
    seq = msa_tok[:, 0].clone()

    # ---- embeddings
    torch.manual_seed(1)
    m = rf.MsaEmbedding(d_input=21, d_msa=dm, max_len=40, p_pe_drop=0.0).eval()
    save("msa_embedding", m, {"msa": msa_tok, "aa_idx": aa_idx}, {"y": m(msa_tok, aa_idx)}, {"max_len": 40})
    m = rf.PairEmbedding(d_input=21, d_pair=dp, max_len=40, p_pe_drop=0.0).eval()
    save("pair_embedding", m, {"seq": seq, "aa_idx": aa_idx}, {"y": m(seq, aa_idx)}, {"max_len": 40})

    # ---- MSA row attention
    x = rn(B, N, L, dm)
    m = rf.PositionWiseWeightFactor(dm, H, 0.0).eval()
    save("poswise_weight", m, {"x": x}, {"y": m(x)}, {"n_heads": H})
    m = rf.SoftTiedAttentionOverResidues(dm, H, 0.0, return_att=True).eval()
    o, a = m(x)
    save("soft_tied_attention", m, {"x": x}, {"out": o, "att": a}, {"n_heads": H})
    m = rf.EncoderLayer(d_msa=dm, d_ff=dm * 4, n_heads=H, p_dropout=0.0, tied=True, return_att=True).eval()
    o, a = m(x)
    save("encoder_layer_tied", m, {"x": x}, {"out": o, "att": a}, {"n_heads": H})

    # ---- pair update with msa
    dproj = 4
    m = rf.OuterProductMean(dproj, dp).eval()
    xa, xb = rn(B, N, L, dproj), rn(B, N, L, dproj)
    save("outer_product_mean", m, {"x": xa, "y": xb}, {"y": m(xa, xb)})
    m = rf.PairUpdateWithMsa(d_msa=dm, d_proj=dproj, d_pair=dp, n_heads=H, p_dropout=0.0).eval()
    pair = rn(B, L, L, dp)
    att = torch.rand(B, L, L, H, generator=g)
    save("pair_update_with_msa", m, {"msa": x, "pair": pair, "att": att}, {"y": m(x, pair, att)})

    # ---- msa update with pair (hidden list)
    m = rf.MsaUpdateWithPairLayer(dm, dp, 4, 0.0).eval()
    save("msa_update_with_pair_layer", m, {"msa": x, "pair": pair}, {"y": m(x, pair)}, {"n_heads": 4})
    m = rf.MsaUpdateWithPair(dm, dp, 4, n_encoder_layers=2, p_dropout=0.0).eval()
    for lyr in m.encoder_layers:
        lyr.eval()
    save("msa_update_with_pair", m, {"msa": x, "pair": pair}, {"y": m(x, pair)}, {"n_heads": 4, "n_layers": 2})

    # ---- initial coordinates
    dn, de = 8, 8
    m = rf.GraphTransformerBlock(dn, dn, de, 4, 0.0).eval()
    node, edge = rn(B, L, dn), rn(B, L, L, de)
    save("graph_transformer_block", m, {"node": node, "edge": edge}, {"y": m(node, edge, None)}, {"n_heads": 4})
    m = rf.InitialCoordGenerationWithMsaAndPair(dm, dp, d_node=dn, d_edge=de, n_heads=4, n_layers=2, p_dropout=0.0).eval()
    for blk in m.blocks:
        blk.eval()
    onehot = torch.nn.functional.one_hot(seq, 21).float()
    save("initial_coord_generation", m, {"msa": x, "pair": pair, "seq_onehot": onehot, "aa_idx": aa_idx},
         {"y": m(x, pair, onehot, aa_idx)}, {"n_layers": 2, "n_heads": 4})

    # ---- kNN graph (edge list via the recorder) -- protein-like CA trace
    m = rf.CoordUpdateWithMsaAndPair(dm, dp, dn, de, 4, n_neighbors=4, p_dropout=0.0).eval()
    L2 = 16
    steps = rn(B, L2, 3)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
    xyz = ca[:, :, None, :] + 0.5 * rn(B, L2, 3, 3)
    xyz[:, :, 1] = ca
    idx2 = torch.stack([torch.arange(L2), torch.arange(L2) + torch.tensor([0] * 8 + [30] * 8)])
    edge_feat = rn(B, L2, L2, de)
    for k in (4, 32):
        G = m._knn_graph(xyz, edge_feat, idx2, n_neighbors=k)
        save(f"knn_graph_k{k}", None, {"xyz": xyz, "edge": edge_feat, "idx": idx2},
             {"src": G.src, "dst": G.dst, "d": G.edata["d"], "w": G.edata["w"]}, {"n_neighbors": k})
    save("se3_transformer_manifest", None, {}, {},
         {"keys": np.array([f"{k}|{tuple(v.shape)}" for k, v in m.se3_transformer.state_dict().items()])})

    # ---- msa update with pair and coord
    ds = 8
    m = rf.MsaUpdateWithPairAndCoord(dm, ds, 32, dm * 4, p_dropout=0.0).eval()
    state = rn(B, L2, ds)
    x2 = rn(B, N, L2, dm)
    save("msa_update_with_pair_and_coord", m, {"xyz": xyz, "state": state, "msa": x2}, {"y": m(xyz, state, x2)})

    # ---- prediction head
    m = ResNet(2, 8, 8, 5, p_dropout=0.0).eval()
    img = rn(B, 8, L, L)
    save("resnet", m, {"x": img}, {"y": m(img)}, {"n_blocks": 2})
    m = rf.PredictionHead(8, 4, 0.0).eval()
    p8 = rn(B, L, L, 8)
    o = m(p8)
    save("prediction_head", m, {"pair": p8}, {k: v for k, v in o.items()}, {"n_blocks": 4})

    # ---- SE(3) pieces that do not need dgl / lie_learn
    vec = torch.cat([rn(64, 3), torch.eye(3), -torch.eye(3), torch.zeros(1, 3)]).double()
    sph = us.get_spherical_from_cartesian_torch(vec)
    Y = us.precompute_sh(sph, 2)
    save("spherical_harmonics", None, {"d": vec}, {"sph": sph, "Y0": Y[0], "Y1": Y[1], "Y2": Y[2]})
    E = 10
    feat = rn(E, de + 1)
    m = ea.RadialFunc(3, 4, 5, edge_dim=de).eval()
    save("radial_func", m, {"feat": feat}, {"y": m(feat)})
    m = ea.PairwiseConv(1, 4, 1, 5, edge_dim=de).eval()
    basis = {"1,1": rn(E, 1, 3, 1, 3, 3)}
    save("pairwise_conv", m, {"feat": feat, "basis11": basis["1,1"]}, {"y": m(feat, basis)})
    fin, fout = Fiber(dictionary={0: 6, 1: 5}), Fiber(dictionary={0: 4, 1: 3})
    h = {"0": rn(7, 6, 1), "1": rn(7, 5, 3)}
    m = ea.G1x1SE3(fin, fout).eval()
    o = m(h)
    save("g1x1", m, {"h0": h["0"], "h1": h["1"]}, {"o0": o["0"], "o1": o["1"]})
    m = ea.GNormBias(fin).eval()
    o = m({k: v.clone() for k, v in h.items()})
    save("gnorm_bias", m, {"h0": h["0"], "h1": h["1"]}, {"o0": o["0"], "o1": o["1"]})
    m = ea.GAttentiveSelfInt(fin, fout).eval()
    o = m(h)
    save("gattentive_selfint", m, {"h0": h["0"], "h1": h["1"]}, {"o0": o["0"], "o1": o["1"]})
    print("scratch dir:", scratch)


if __name__ == "__main__":
    main()
