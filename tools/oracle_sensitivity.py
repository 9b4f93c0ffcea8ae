"""How far does the CPU ORACLE itself move when its weights are perturbed at the size of a 16-bit operand rounding?

    python tools/oracle_sensitivity.py [--n-two 8 --n-three 5] [--levels 12,9] [--round fp16,bf16] [--threads 6]

CPU only (no GPU, no HIP library call): oracle/rf_oracle.py at config-2 dimensions (B=1, N=128, L=256, d_msa=384, d_pair=288,
n_enc=4) with the seeded weights / inputs of tools/depth_parity.py, run once as is and once per perturbation:
  * `--levels k`: every weight tensor multiplied element-wise by (1 + 2^-k u), u ~ U(-1, 1)  (relative noise 2^-k);
  * `--round t`:  every weight tensor rounded to fp16 / bf16 and back (what a 16-bit operand copy of the WEIGHTS alone does;
                  the 16-bit modes of the library additionally round every activation operand).
Reports the same block-by-block curve and final metrics as tools/depth_parity.py (perturbed oracle against the unperturbed
oracle).  This is the yardstick for the whole-model gap of the 16-bit modes: if the oracle moves by as much under a
perturbation of the same size, the gap is the network's own sensitivity at random init (kNN top-k, GNormBias, distance
bins), not a kernel defect.  Output: one JSON object (profiles/r04_oracle_sensitivity.json).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rosettafold_pytorch_amd as R  # noqa: E402  (parameter containers only: nothing here launches a kernel)
from depth_parity import oracle_trace, compare  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-two", type=int, default=8)
    ap.add_argument("--n-three", type=int, default=5)
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--L", type=int, default=256)
    ap.add_argument("--levels", default="12,9")
    ap.add_argument("--round", default="fp16,bf16")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = dict(d_input=21, d_msa=384, d_pair=288, d_node=32, d_edge=32, d_state=32, n_two_track_blocks=args.n_two,
               n_three_track_blocks=args.n_three, n_encoder_layers=4, max_len=args.L + 4, n_neighbors=[128, 128, 64, 64, 64],
               p_dropout=0.0)
    torch.manual_seed(1234)
    model = R.RoseTTAFold(**cfg)
    P = {k: v.detach().float() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (1, args.N, args.L), generator=g)
    seq = msa[:, 0].clone()
    aa = torch.arange(args.L).unsqueeze(0)
    res = {"config": {"B": 1, "N": args.N, "L": args.L, "blocks": f"{args.n_two}+{args.n_three}", "n_enc": 4, "threads": args.threads,
                      "reference": "unperturbed CPU oracle (oracle/rf_oracle.py)", "weights": "torch.manual_seed(1234) default init"}}
    t0 = time.perf_counter()
    base = oracle_trace(P, msa, seq, aa, cfg)
    res["oracle_seconds"] = time.perf_counter() - t0
    print(f"[sens] base oracle: {res['oracle_seconds']:.0f}s", file=sys.stderr, flush=True)

    def dump():
        if args.out:
            with open(args.out, "w") as fh:
                json.dump(res, fh, indent=1)

    for lv in [int(x) for x in args.levels.split(",") if x]:
        gp = torch.Generator().manual_seed(1000 + lv)
        Pp = {k: (v * (1.0 + 2.0 ** -lv * (2.0 * torch.rand(v.shape, generator=gp) - 1.0))) if v.is_floating_point() else v
              for k, v in P.items()}
        t0 = time.perf_counter()
        res[f"weights_noise_2^-{lv}"] = compare(oracle_trace(Pp, msa, seq, aa, cfg), base)
        print(f"[sens] noise 2^-{lv}: {time.perf_counter() - t0:.0f}s", file=sys.stderr, flush=True)
        dump()
    for name in [x for x in args.round.split(",") if x]:
        dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[name]
        Pp = {k: v.to(dt).float() if v.is_floating_point() else v for k, v in P.items()}
        t0 = time.perf_counter()
        res[f"weights_rounded_{name}"] = compare(oracle_trace(Pp, msa, seq, aa, cfg), base)
        print(f"[sens] rounded {name}: {time.perf_counter() - t0:.0f}s", file=sys.stderr, flush=True)
        dump()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
