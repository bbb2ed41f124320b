#!/bin/bash
# round 6 profile set (run via gpurun): every row ONE workload, as in round 5 (tools/r05_profile.sh).
#   A  bench.py, the driver's command, no profiler                                  -> r06_bench.json (+ r06_bench_details.json)
#   B  rocprofv3 --kernel-trace --stats over the HEADLINE ALONE, cut to the timed dispatches the record names
#   D  FETCH_SIZE / WRITE_SIZE passes over the headline alone                          -> r06_traffic.json (bench.py quotes it)
#   G  per-chunk models: tools/time_adaptive.py under --stats and the two PMC passes (k_encode_adaptive, the decoders)
#   H  the reference's own layouts (decoders: decode_groups.hip; encoders: the lane kernels): tools/time_lanes.py word 8-way / byte 2-way, 1024-symbol chunks,
#      --stats, the two PMC passes, and the SQ / LDS / TA counter sets of tools/pmc_kernel.sh
#   I  the same counter sets for the per-chunk-model decoder and encoder (what bounds them)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r06p
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 20 --warmup 5"
$BENCH --details "$OUT/bench_details.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 400 "$OUT/bench.json"; echo
HEAD="$BENCH --no-configs --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/headline_stats" -o h -- $HEAD --details "$OUT/headline_details.json" > "$OUT/headline_line.json" 2> "$OUT/headline.err"
PMC="$HEAD --prewarm-ms 0 --steps 5 --placement-candidates 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o pmc -- $PMC --details "$OUT/pmc_fetch_details.json" > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o pmc -- $PMC --details "$OUT/pmc_write_details.json" > "$OUT/pmc_write.log" 2>&1
AD="python $REPO/tools/time_adaptive.py 30"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/adaptive_stats" -o h -- $AD > "$OUT/adaptive.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/adaptive_fetch" -o pmc -- $AD > "$OUT/adaptive_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/adaptive_write" -o pmc -- $AD > "$OUT/adaptive_write.log" 2>&1
for w in "word 8 12" "byte 2 14"; do
  set -- $w
  L="python $REPO/tools/time_lanes.py --fmt $1 --ways $2 --chunk 1024 --log2n 30 --sb $3 --steps 10 --encode"
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/lanes_$1$2_stats" -o h -- $L > "$OUT/lanes_$1$2.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/lanes_$1$2_fetch" -o pmc -- $L > "$OUT/lanes_$1$2_fetch.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/lanes_$1$2_write" -o pmc -- $L > "$OUT/lanes_$1$2_write.log" 2>&1
done
cd "$REPO"
PMC_CMD="python $REPO/tools/time_lanes.py --fmt word --ways 8 --chunk 1024 --log2n 30 --sb 12 --steps 5" bash tools/pmc_kernel.sh r06p/cnt_word8 k_decode_word_groups > "$OUT/cnt_word8.log" 2>&1
PMC_CMD="python $REPO/tools/time_lanes.py --fmt byte --ways 2 --chunk 1024 --log2n 30 --sb 14 --steps 5" bash tools/pmc_kernel.sh r06p/cnt_byte2 k_decode_byte_pairs > "$OUT/cnt_byte2.log" 2>&1
PMC_CMD="python $REPO/tools/time_lanes.py --fmt word --ways 8 --chunk 1024 --log2n 30 --sb 12 --steps 2 --encode" bash tools/pmc_kernel.sh r06p/cnt_word8enc k_encode_word_groups > "$OUT/cnt_word8enc.log" 2>&1
PMC_CMD="python $REPO/tools/time_adaptive.py 30" bash tools/pmc_kernel.sh r06p/cnt_adec "k_decode<12, 1" > "$OUT/cnt_adec.log" 2>&1
PMC_CMD="python $REPO/tools/time_adaptive.py 30" bash tools/pmc_kernel.sh r06p/cnt_aenc "k_encode_adaptive<1, 1, 16" > "$OUT/cnt_aenc.log" 2>&1
# keep the merge-back small: csv / json / log / txt only, and no per-dispatch traces but the headline's
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.log" ! -name "*.err" ! -name "*.txt" -delete
find "$OUT" -name "*_kernel_trace.csv" ! -path "*headline_stats*" -size +2M -delete
find "$OUT" -name "*.db" -delete
find "$OUT" -path "*_sq*" -name "*counter_collection.csv" -size +4M -delete
du -sh "$OUT"
python tools/summarize_r06.py "$OUT" || true
