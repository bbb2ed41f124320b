#!/bin/bash
# round 5 profile set (run via gpurun; VERDICT r04 next #5: evidence in which every row is ONE workload).
#   A  bench.py, the driver's command, no profiler                          -> r05_bench.json (+ r05_bench_details.json)
#   B  rocprofv3 --kernel-trace --stats over the HEADLINE ALONE              -> r05_headline_stats/, r05_headline_*.json
#      (`--no-configs --no-cpu-baseline`): the record names the timed dispatches, tools/summarize_r05.py cuts the trace to them
#   C  the same with --placement-candidates 1 (no probe: every launch of the row runs on ONE pair of buffers)
#   D  FETCH_SIZE / WRITE_SIZE passes over the headline alone (5 steps, no pre-warm)
#   E  one bench.py per other decode workload (--format / --ways): its decoder's row alone
#   F  tools/time_slots.py per encoder configuration under --stats and the two PMC passes: encoders in all three layouts
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05p
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 20 --warmup 5"
$BENCH --details "$OUT/bench_details.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"; echo
HEAD="$BENCH --no-configs --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/headline_stats" -o h -- $HEAD --details "$OUT/headline_details.json" > "$OUT/headline_line.json" 2> "$OUT/headline.err"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/noprobe_stats" -o h -- $HEAD --placement-candidates 1 --details "$OUT/noprobe_details.json" > "$OUT/noprobe_line.json" 2> "$OUT/noprobe.err"
PMC="$HEAD --prewarm-ms 0 --steps 5 --placement-candidates 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o pmc -- $PMC --details "$OUT/pmc_fetch_details.json" > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o pmc -- $PMC --details "$OUT/pmc_write_details.json" > "$OUT/pmc_write.log" 2>&1
for w in "byte 64 14" "byte 64 12" "word 128 12" "word 256 12"; do
  set -- $w
  tag="dec_$1$2_sb$3"
  extra=""
  [ "$1" = "byte" ] && [ "$3" = "12" ] && extra="--scale-bits 12"
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/$tag" -o h -- $HEAD --format $1 --ways $2 --chunk 16384 $extra --placement-candidates 1 --details "$OUT/${tag}_details.json" > "$OUT/${tag}_line.json" 2> "$OUT/${tag}.err"
done
for c in word byte c4 c2; do
  SLOTS="python $REPO/tools/time_slots.py --configs $c --rounds 1 --launches 10"
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/enc_${c}_stats" -o h -- $SLOTS > "$OUT/enc_${c}.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/enc_${c}_fetch" -o pmc -- $SLOTS --launches 3 > "$OUT/enc_${c}_fetch.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/enc_${c}_write" -o pmc -- $SLOTS --launches 3 > "$OUT/enc_${c}_write.log" 2>&1
done
cd "$REPO"
# keep the merge-back small: csv / json / log only, and no per-dispatch traces but the headline's
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.log" ! -name "*.err" -delete
find "$OUT" -name "*_kernel_trace.csv" ! -path "*headline_stats*" ! -path "*noprobe_stats*" -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
python tools/summarize_r05.py "$OUT" || true
