#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace statistics + the two PMC passes of one bench forward,
# summaries copied next to the other gpurun_out files (copy the ones to keep into profiles/ afterwards).
#   bash tools/collect_profiles.sh <tag> [config]
set -u
TAG=${1:-r04}
CFG=${2:-2}     # bench.py --config (2 = the headline workload; 4 / 5: tag the files r04_config4 / r04_config5)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-parity > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "gpurun_out/${TAG}_kernel_stats_bench_config${CFG}.csv"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline --no-parity > "$OUT/fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline --no-parity > "$OUT/write.log" 2>&1
python3 tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" "gpurun_out/${TAG}_traffic_pmc.json" > "gpurun_out/${TAG}_traffic_pmc.log" 2>&1
# matrix-pipe utilisation (its own pass: SQ + GRBM counters, no tracing domains besides the kernel trace)
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline --no-parity > "$OUT/mfma.log" 2>&1
python3 tools/pmc_mfma.py "$OUT/mfma" "gpurun_out/${TAG}_mfma_pmc.json" > "gpurun_out/${TAG}_mfma_pmc.log" 2>&1
# drop the bulky raw traces (only the summaries travel back)
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/mfma"
tail -3 "$OUT/stats.log"; cat "gpurun_out/${TAG}_traffic_pmc.log" "gpurun_out/${TAG}_mfma_pmc.log"
