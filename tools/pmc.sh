#!/bin/bash
# tools/pmc.sh <tag> -- SQ/LDS counter passes over the bench decode kernel (run via gpurun).
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs --prewarm-ms 0 --placement-candidates 1 ${BENCH_ARGS:-}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -f csv -d "$OUT/${TAG}_sq$i" -o pmc -- $BENCH > "$OUT/${TAG}_sq$i.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, collections, glob, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/{tag}_sq*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_decode_word64" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_sq_summary.txt", "w") as fh:
    for k in sorted(agg):
        line = "%-24s n=%d avg=%.4g" % (k, len(agg[k]), sum(agg[k]) / len(agg[k]))
        print(line); fh.write(line + "\n")
PY
