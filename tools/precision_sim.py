"""CPU experiment (build container, no GPU): how much of the whole-model gap of a 16-bit-operand forward is operand
rounding, and what would fp16 operands (11-bit significand, same MFMA rate as bf16 on gfx950) buy over bf16 (8-bit)?

The oracle's contractions (F.linear, einsum, conv2d, the FAVOR+ feature products) get their operands rounded to the
given 16-bit type (fp32 accumulation, fp32 norms / softmax / residual streams), i.e. the precision policy of the HIP path;
the SE(3) stack stays fp32 as it does on the GPU.  Prints relative L2 / distogram-argmax agreement against the plain
fp32 oracle.

    python tools/precision_sim.py [L] [N] [n_two] [n_three] [n_enc]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RF_ALLOW_NO_GPU", "1")
from oracle import rf_oracle as O  # noqa: E402

RND = {"dtype": None, "off": 0}


def r(x):
    if RND["dtype"] is None or RND["off"] or not torch.is_floating_point(x):
        return x
    return x.to(RND["dtype"]).float()


class _F:
    def __getattr__(self, k):
        return getattr(F, k)

    @staticmethod
    def linear(x, w, b=None):
        return F.linear(r(x), r(w), b)

    @staticmethod
    def conv2d(x, w, *a, **k):
        return F.conv2d(r(x), r(w), *a, **k)


class _T:
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def einsum(eq, *ops):
        return torch.einsum(eq, *[r(o) for o in ops])


def patch():
    O.F = _F()
    O.torch = _T()
    sm, rl = O.favor_softmax_features, O.favor_relu_features

    def fsm(data, proj, is_query, eps=1e-4):
        return sm(r(data), r(proj * data.shape[-1] ** -0.25) * data.shape[-1] ** 0.25, is_query, eps)

    def frl(data, proj, eps=1e-3):
        return rl(r(data), r(proj * data.shape[-1] ** -0.25) * data.shape[-1] ** 0.25, eps)

    O.favor_softmax_features, O.favor_relu_features = fsm, frl

    def linatt(q, k, v):  # as the fused kernel: 16-bit q', k', v and context; fp32 sums and normaliser
        ksum = r(k).sum(-2)
        dinv = 1.0 / torch.einsum("...nd,...d->...n", r(q), ksum)
        ctx = torch.einsum("...nd,...ne->...de", r(k), r(v))
        return torch.einsum("...de,...nd->...ne", r(ctx), r(q)) * dinv.unsqueeze(-1)

    O.linear_attention = linatt
    se3 = O.se3_transformer

    def se3_fp32(*a, **k):
        RND["off"] += 1
        try:
            return se3(*a, **k)
        finally:
            RND["off"] -= 1

    O.se3_transformer = se3_fp32


def main():
    Lr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    n2 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    n3 = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    ne = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    import rosettafold_pytorch_amd as R
    cfg = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=n2,
               n_three_track_blocks=n3, n_encoder_layers=ne, max_len=Lr + 4, n_neighbors=[128, 128, 64, 64, 64],
               p_dropout=0.0)
    torch.manual_seed(1234)
    model = R.RoseTTAFold(**cfg)
    P = {k: v.detach().float() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (1, N, Lr), generator=g)
    seq = msa[:, 0].clone()
    aa = torch.arange(Lr).unsqueeze(0)
    patch()
    outs = {}
    with torch.no_grad():
        for name, dt in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
            RND["dtype"] = dt
            t0 = time.time()
            outs[name] = O.rosettafold_forward(P, msa, seq, aa, cfg)
            print(f"{name}: {time.time() - t0:.1f}s", flush=True)
    lg, xyz, pl = outs["fp32"]
    d = lg["dist"]
    top2 = d.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.02 * (d.max() - d.min())
    for name in ("bf16", "fp16"):
        lb, xb, pb = outs[name]
        rel = {k: ((lb[k] - lg[k]).double().norm() / lg[k].double().norm()).item() for k in lg}
        agree = (lb["dist"].argmax(-1) == d.argmax(-1)).float().mean().item()
        ac = (lb["dist"].argmax(-1) == d.argmax(-1))[clear].float().mean().item()
        print(name, "rel-L2", {k: round(v, 4) for k, v in rel.items()}, "xyz",
              round(((xb - xyz).double().norm() / xyz.double().norm()).item(), 4), "dist argmax agree", round(agree, 4),
              "clear-margin", round(ac, 4), "clear frac", round(clear.float().mean().item(), 3))


if __name__ == "__main__":
    main()
