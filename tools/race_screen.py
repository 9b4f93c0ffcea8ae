#!/usr/bin/env python3
"""Race screen (run on the GPU box): the same forward, stage by stage, with and without a second stream hammering the GPU
(uneven load changes wave timing: a kernel with a latent LDS / global-memory race then gives run-to-run differences)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rosettafold_pytorch_amd as R
from rosettafold_pytorch_amd import ops

CFG = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
           n_encoder_layers=1, max_len=64, n_neighbors=[16, 16], p_dropout=0.0)
B, N, L = int(os.environ.get("RS_B", 2)), 8, 32
torch.manual_seed(1234)
m = R.RoseTTAFold(**CFG).cuda()
g = torch.Generator().manual_seed(0)
msa = torch.randint(0, 21, (B, N, L), generator=g).cuda()
seq, aa = msa[:, 0].clone(), torch.arange(L).unsqueeze(0).repeat(B, 1).cuda()
side = torch.cuda.Stream()
big = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)

def stages():
    st = {}
    with torch.no_grad():
        x = m.msa_emb(msa, aa); p = m.pair_emb(seq, aa)
        oh = ops.onehot(seq, 21)
        blk = m.two_track_blocks[0]
        att = blk.msa_update_using_self_att.run(x); st["msa_selfatt"] = x.clone(); st["att"] = att.clone()
        p = blk.pair_update_with_msa.run(x, p, att); st["pair_with_msa"] = p.clone()
        blk.pair_update_with_axial_attention.run(p); st["pair_axial"] = p.clone()
        blk.msa_update_with_pair.run(x, p); st["msa_with_pair"] = x.clone()
        xyz = m.initial_coord_generation_with_msa_and_pair.run(x, p, oh, aa); st["init_xyz"] = xyz.clone()
        t3 = m.three_track_blocks[0]
        p2 = t3.run(x, p); st["t3_pair"] = p2.clone(); st["t3_msa"] = x.clone()
        state, xyz2 = t3.coord_update_with_msa_and_pair.run(xyz, x, p2, aa, oh); st["state"] = state.clone(); st["xyz2"] = xyz2.clone()
        x2 = t3.msa_update_with_pair_and_coord.run(xyz2, state, x); st["msa_coord"] = x2.clone()
        lg = m.prediction_head.run(p2)
        st.update({"head_" + k: v.clone() for k, v in lg.items()})
    torch.cuda.synchronize()
    return st

ref = stages()
bad_total = {}
for it in range(int(os.environ.get("RS_ITERS", 30))):
    load = it % 3
    if load:
        with torch.cuda.stream(side):
            for _ in range(3 * load):
                big @ big
    st = stages()
    for k in ref:
        if not torch.equal(ref[k], st[k]):
            bad_total[k] = bad_total.get(k, 0) + 1
torch.cuda.synchronize()
print("stages that changed run to run:", bad_total if bad_total else "none (bitwise stable under load)")
