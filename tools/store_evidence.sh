#!/bin/bash
# (build container) copy the summaries tools/collect_profiles.sh left in gpurun_out/ into profiles/ under the names bench.py and
# tests/test_evidence.py read:  bash tools/store_evidence.sh [pmc|bench]
set -e
cd "$(dirname "$0")/.."
if [ "${1:-pmc}" = pmc ]; then
  cp gpurun_out/r04_traffic_pmc.json gpurun_out/r04_traffic_pmc.log gpurun_out/r04_mfma_pmc.json gpurun_out/r04_mfma_pmc.log profiles/
  cp gpurun_out/r04_kernel_stats_bench_config2.csv profiles/r04_kernel_stats_bench_config2.csv
  for c in 4 5; do
    cp gpurun_out/r04_config${c}_traffic_pmc.json gpurun_out/r04_config${c}_traffic_pmc.log gpurun_out/r04_config${c}_mfma_pmc.json gpurun_out/r04_config${c}_mfma_pmc.log profiles/
    cp gpurun_out/r04_config${c}_kernel_stats_bench_config${c}.csv profiles/r04_config${c}_kernel_stats.csv
  done
else
  cp gpurun_out/r04_bench_default.json gpurun_out/r04_bench_config4.json gpurun_out/r04_bench_config5.json profiles/
fi
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print("csrc tree", bench.tree_hash())
for f in ("r04_traffic_pmc", "r04_mfma_pmc", "r04_config4_traffic_pmc", "r04_config5_traffic_pmc"):
    print(f, json.load(open("profiles/%s.json" % f))["tree"])
PY
