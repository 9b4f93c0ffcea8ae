#!/usr/bin/env python3
"""Cross-process determinism (run on the GPU box): the same forward in fresh processes, one after the other and two at
a time on the same GPU, must give bit-identical outputs.  Prints which outputs differ, per stage of the model."""
import os, sys, subprocess, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
           n_encoder_layers=1, max_len=64, n_neighbors=[16, 16], p_dropout=0.0)

def child(path):
    import rosettafold_pytorch_amd as R
    torch.manual_seed(1234)
    m = R.RoseTTAFold(**CFG).cuda()
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (2, 8, 32), generator=g).cuda()
    seq, aa = msa[:, 0].clone(), torch.arange(32).unsqueeze(0).repeat(2, 1).cuda()
    stages = {}
    with torch.no_grad():
        x = m.msa_emb(msa, aa); p = m.pair_emb(seq, aa)
        stages["msa_emb"], stages["pair_emb"] = x.clone(), p.clone()
        blk = m.two_track_blocks[0]
        att = blk.msa_update_using_self_att.run(x); stages["msa_selfatt"] = x.clone(); stages["att"] = att.clone()
        p = blk.pair_update_with_msa.run(x, p, att); stages["pair_with_msa"] = p.clone()
        blk.pair_update_with_axial_attention.run(p); stages["pair_axial"] = p.clone()
        blk.msa_update_with_pair.run(x, p); stages["msa_with_pair"] = x.clone()
        lg, xyz, pl = m(msa, seq, aa)
    stages.update({"logits_" + k: v for k, v in lg.items()}); stages["xyz"] = xyz; stages["plddt"] = pl
    torch.save({k: v.cpu() for k, v in stages.items()}, path)

if len(sys.argv) > 2 and sys.argv[1] == "child":
    child(sys.argv[2]); sys.exit(0)
d = tempfile.mkdtemp()
def launch(i):
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", os.path.join(d, f"o{i}.pt")], stderr=subprocess.DEVNULL)
for i in range(2):
    launch(i).wait()
ps = [launch(i) for i in (2, 3)]
[p.wait() for p in ps]
outs = [torch.load(os.path.join(d, f"o{i}.pt")) for i in range(4)]
for i, tag in ((1, "sequential #2"), (2, "concurrent A"), (3, "concurrent B")):
    bad = [k for k in outs[0] if not torch.equal(outs[0][k], outs[i][k])]
    print(f"{tag} vs sequential #1: {'identical' if not bad else 'DIFFERS in ' + str(bad)}")
