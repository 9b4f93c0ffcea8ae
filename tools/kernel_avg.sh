#!/bin/bash
# (GPU box) average duration of the kernels whose name matches $1 in one eager bench run (rocprofv3 kernel trace statistics)
#   [RFMI_LIB=...] bash tools/kernel_avg.sh favor_attention
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_avg_$$
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-parity > $D.log 2>&1
python3 - "$D" "$1" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / 3e6:.1f} ms")
for r in rows:
    if sys.argv[2] in r["Name"]:
        print(f"{r['Name'][:80]:80s} {int(r['Calls']) // 3:5d}/step {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / 3e6:6.2f} ms/step")
PY
rm -rf $D $D.log
