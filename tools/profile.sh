#!/bin/bash
# tools/profile.sh <tag> -- run on the GPU box (via gpurun).  Produces under gpurun_out/:
#   <tag>_bench.json          bench.py default run (no profiler attached)
#   <tag>_stats/              rocprofv3 --kernel-trace --stats of the same command
#   <tag>_pmc_{fetch,write}/  separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE)
# Summaries are copied into profiles/ by tools/summarize_profile.py afterwards.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5"
$BENCH > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cat "$OUT/${TAG}_bench.json"
# kernel durations of the same command (the `configs` entries included: every dominant kernel shows up)
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_stats" -o stats -- $BENCH --no-cpu-baseline > "$OUT/${TAG}_stats.log" 2>&1
# counter passes: no clock pre-warm (hundreds of launches under the counter collector), 5 steps, headline only
PMC="$BENCH --no-cpu-baseline --no-configs --prewarm-ms 0 --steps 5 --placement-candidates 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/${TAG}_pmc_fetch" -o pmc -- $PMC > "$OUT/${TAG}_pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/${TAG}_pmc_write" -o pmc -- $PMC > "$OUT/${TAG}_pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -30
# keep the merge-back small: drop everything but the csv/json/log files
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.log" ! -name "*.err" ! -name "*.txt" -size +1M -delete
du -sh "$OUT"
