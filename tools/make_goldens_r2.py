#!/usr/bin/env python3
"""Round-2 additions to the golden fixtures (build container only; needs /root/reference):

  * rosettafold_state_manifest.npz -- `state_dict()` key|shape manifest of the reference's RoseTTAFold at the README /
    bench configuration and at the configuration of the reference's own shape test (tests/test_module.py:792-824),
    plus the keys of the layers the reference hides in plain Python lists (rf.py:602-605, 699-702).  Captured from the
    reference imported with the inert stand-ins of tools/make_goldens.py (the Performer stand-in holds no parameters,
    so its keys are absent from the manifest).
  * c1_*.npz -- config-1-sized cases (B=1, N=8, L=64, d_msa=96, d_pair=64) for the two north-star rows:
    SoftTiedAttentionOverResidues with 12 heads and PairUpdateWithMsa with d_proj=32.  Inputs and weights are rounded
    to fp16-representable values BEFORE the reference runs and stored as fp16 (exact); the pair output is stored on a
    stride-2 sub-grid to bound the fixture size.

  * pair_embedding_template.npz -- PairEmbedding(use_template=True) (rf.py:141-169), the branch of the public module
    that RoseTTAFold.forward itself never takes.

    python tools/make_goldens_r2.py [--only template]
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402


def h16(t):
    return t.half().float()


def round_module_fp16(m):
    for p in m.parameters():
        p.copy_(h16(p))
    for lst_name in ("encoder_layers", "blocks"):
        for sub in m.modules():
            lst = getattr(sub, lst_name, None)
            if isinstance(lst, list):
                for layer in lst:
                    for p in layer.parameters():
                        p.copy_(h16(p))


def save16(name, module, inputs, outputs, extra=None):
    data = {}
    for k, v in MG.full_state(module).items():
        data["w:" + k] = MG.np_(v).astype(np.float16)
    for k, v in inputs.items():
        data["in:" + k] = MG.np_(v).astype(np.float16)
    for k, v in outputs.items():
        data["out:" + k] = MG.np_(v)
    for k, v in (extra or {}).items():
        data["x:" + k] = np.asarray(v)
    path = os.path.join(MG.OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{name:40s} {os.path.getsize(path) / 1024:8.1f} KB")


def manifest(module):
    return np.array([f"{k}|{tuple(v.shape)}" for k, v in module.state_dict().items()])


def hidden_manifest(module):
    sd = module.state_dict()
    return np.array([f"{k}|{tuple(v.shape)}" for k, v in MG.full_state(module).items() if k not in sd])


def main():
    os.makedirs(MG.OUT, exist_ok=True)
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    os.chdir(tempfile.mkdtemp(prefix="rf_golden_r2_"))
    MG.install_standins()
    sys.path.insert(0, MG.REF)
    import rosettafold_pytorch.rosettafold_pytorch as rf

    torch.set_grad_enabled(False)
    # ---- template branch of PairEmbedding
    g = torch.Generator().manual_seed(77)
    B, L, dp, dt = 2, 12, 32, 16
    seq = torch.randint(0, 21, (B, L), generator=g)
    aa_idx = torch.stack([torch.arange(L), torch.arange(L) * 2 + 3])
    templ = torch.randn(B, L, L, dt, generator=g)
    torch.manual_seed(5)
    m = rf.PairEmbedding(d_input=21, d_pair=dp, max_len=40, p_pe_drop=0.0, use_template=True, d_template=dt).eval()
    MG.save("pair_embedding_template", m, {"seq": seq, "aa_idx": aa_idx, "template": templ}, {"y": m(seq, aa_idx, templ)},
            {"max_len": 40})
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "template":
        return
    # ---- state_dict manifests
    readme = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=8,
                  n_three_track_blocks=5, n_encoder_layers=4, max_len=260, n_neighbors=[128, 128, 64, 64, 64],
                  p_dropout=0.1, use_template=False)
    reftest = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=4, n_two_track_blocks=4,
                   n_three_track_blocks=4, n_encoder_layers=4, max_len=64, n_neighbors=[128, 128, 64, 64],
                   p_dropout=0.1, use_template=False)
    data = {}
    for tag, cfg in (("readme", readme), ("reftest", reftest)):
        m = rf.RoseTTAFold(**cfg)
        data[f"x:{tag}_keys"] = manifest(m)
        data[f"x:{tag}_hidden_keys"] = hidden_manifest(m)
        data[f"x:{tag}_cfg"] = np.array([f"{k}={v}" for k, v in cfg.items()])
        print(tag, len(data[f"x:{tag}_keys"]), "registered keys,", len(data[f"x:{tag}_hidden_keys"]), "hidden-list keys")
        del m
    path = os.path.join(MG.OUT, "rosettafold_state_manifest.npz")
    np.savez_compressed(path, **data)
    print(f"{'rosettafold_state_manifest':40s} {os.path.getsize(path) / 1024:8.1f} KB")

    # ---- config-1-sized cases
    g = torch.Generator().manual_seed(2024)
    B, N, L, dm, dp = 1, 8, 64, 96, 64
    x = h16(torch.randn(B, N, L, dm, generator=g))
    torch.manual_seed(3)
    m = rf.SoftTiedAttentionOverResidues(dm, 12, 0.0, return_att=True).eval()
    round_module_fp16(m)
    o, a = m(x)
    save16("c1_soft_tied_attention", m, {"x": x}, {"out": o, "att": a}, {"n_heads": 12})
    m = rf.PairUpdateWithMsa(d_msa=dm, d_proj=32, d_pair=dp, n_heads=12, p_dropout=0.0).eval()
    round_module_fp16(m)
    pair = h16(torch.randn(B, L, L, dp, generator=g))
    att = h16(torch.rand(B, L, L, 12, generator=g))
    y = m(x, pair, att)
    save16("c1_pair_update_with_msa", m, {"msa": x, "pair": pair, "att": att}, {"y_sub2": y[:, ::2, ::2].contiguous()},
           {"stride": 2})


if __name__ == "__main__":
    main()
