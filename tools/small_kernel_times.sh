#!/bin/bash
# (GPU box) rocprofv3 kernel statistics of one eager bench run, filtered to the small elementwise / normalisation kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_small
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-parity > gpurun_out/prof_small.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_small/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / 3e6:.1f} ms")
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("instnorm", "axpby", "copy4d", "tile_1d", "weighted_msa", "softmax_batched", "att_sym", "layernorm", "poswise", "onehot", "fill", "cast", "linattn", "favor_softmax", "se3_", "graph_att", "dist_att", "edge_", "knn", "coord", "center")):
        print(f"{n[:70]:70s} {int(r['Calls']) // 3:5d}/step {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / 3e6:6.2f} ms/step")
PY
rm -rf gpurun_out/prof_small
