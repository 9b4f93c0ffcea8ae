#!/usr/bin/env python3
"""Run-to-run bitwise comparison of every layer kind of the forward at the BENCH shapes (B=4, N=128, L=256, bf16): a race
or an uninitialised read shows up as a difference.  (run on the GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
DEV = "cuda"
B, N2, L2, DM, DP, DN, DE, DS = 4, 128, 256, 384, 288, 32, 32, 32
R.set_compute_dtype(torch.bfloat16)
def rn(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed + len(s) + sum(s))).to(DEV)
def build(ctor, seed=11):
    torch.manual_seed(seed)
    return ctor().to(DEV)
def flat(o):
    if isinstance(o, dict):
        return [o[k] for k in sorted(o)]
    return [t for t in (o if isinstance(o, (tuple, list)) else [o]) if torch.is_tensor(t)]
bad = 0
def same(name, fn, n=4):
    global bad
    outs = [flat(fn()) for _ in range(n)]
    torch.cuda.synchronize()
    ok = all(all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:])
    md = max((a.float() - b.float()).abs().max().item() for o in outs[1:] for a, b in zip(outs[0], o))
    bad += 0 if ok else 1
    print(f"{name:40s} bitwise identical: {ok}   max |diff| {md:.3e}", flush=True)
def xyz_trace(b, l, seed=3):
    g = torch.Generator().manual_seed(seed)
    steps = torch.randn(b, l, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=-1, keepdim=True), 1)
    xyz = ca[:, :, None, :] + 0.5 * torch.randn(b, l, 3, 3, generator=g)
    xyz[:, :, 1] = ca
    return xyz.to(DEV)
msa, pair = rn(B, N2, L2, DM), rn(B, L2, L2, DP, seed=1)
m = build(lambda: R.EncoderLayer(d_msa=DM, d_ff=4 * DM, n_heads=12, p_dropout=0.0, tied=True, return_att=True))
same("tied row layer", lambda: m(msa))
m2 = build(lambda: R.EncoderLayer(d_msa=DM, d_ff=4 * DM, n_heads=12, p_dropout=0.0, tied=False, performer=True))
def col():
    x = msa.clone(); m2.run(x, seq_axis=1); return x
same("performer column layer", col)
att = torch.rand(B, L2, L2, 12, generator=torch.Generator().manual_seed(2)).softmax(2).to(DEV)
m3 = build(lambda: R.PairUpdateWithMsa(d_msa=DM, d_proj=32, d_pair=DP, n_heads=12, p_dropout=0.0))
same("pair update with msa", lambda: m3(msa, pair, att))
m4 = build(lambda: R.OuterProductMean(32, DP))
xa, xb = rn(B, N2, L2, 32), rn(B, N2, L2, 32, seed=1) * 0.1
same("outer product mean", lambda: m4(xa, xb))
m5 = build(lambda: R.PairUpdateWithAxialAttentionLayer(DP, 4 * DP, 8, 0.0, {}))
same("pair axial layer", lambda: m5(pair))
m6 = build(lambda: R.MsaUpdateWithPair(DM, DP, 4, n_encoder_layers=1, p_dropout=0.0))
same("msa update with pair", lambda: m6(msa, pair))
m7 = build(lambda: R.CoordUpdateWithMsaAndPair(DM, DP, DN, DE, DS, n_neighbors=128, p_dropout=0.0))
xyz = xyz_trace(B, L2)
seq = torch.randint(0, 21, (B, L2), generator=torch.Generator().manual_seed(1))
oh = torch.nn.functional.one_hot(seq, 21).float().to(DEV)
aa = torch.arange(L2).unsqueeze(0).repeat(B, 1).to(DEV)
same("coord update (SE3, k=128)", lambda: m7(xyz, msa, pair, aa, oh))
m8 = build(lambda: R.MsaUpdateWithPairAndCoord(DM, DS, 32, 4 * DM, p_dropout=0.0))
st = rn(B, L2, DS)
same("msa update with pair and coord", lambda: m8(xyz, st, msa))
m9 = build(lambda: R.PredictionHead(DP, 4, 0.0))
same("prediction head", lambda: m9(pair))
m10 = build(lambda: R.InitialCoordGenerationWithMsaAndPair(DM, DP, DN, DE, p_dropout=0.0))
if m10 is not None:
    same("initial coord generation", lambda: m10(msa, pair, oh, aa))
print("NON-DETERMINISTIC MODULES:", bad)
