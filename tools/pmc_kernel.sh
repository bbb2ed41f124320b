#!/bin/bash
# tools/pmc_kernel.sh <tag> <kernel-substring> [bench args] -- SQ/LDS/TA counter passes, summarised for
# the kernels whose name contains the substring (run via gpurun).  Same counter sets as pmc.sh plus
# the vector-memory ones.
set -u
TAG=${1:-k}
KERN=${2:-k_encode}
shift 2 || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH=${PMC_CMD:-"python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"} # PMC_CMD: any other command
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE TA_BUSY_sum TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -f csv -d "$OUT/${TAG}_sq$i" -o pmc -- $BENCH > "$OUT/${TAG}_sq$i.log" 2>&1
done
python3 - "$OUT" "$TAG" "$KERN" <<'PY'
import csv, collections, glob, sys
out, tag, kern = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(f"{out}/{tag}_sq*/pmc_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    # (a kernel that is launched twice per call -- the sized-slot encoder's coders and its normally empty redo pass -- :
    #  only the launches that last at least a tenth of the longest one count)
    longest = max([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows] or [0])
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if d * 10 >= longest:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(d)
with open(f"{out}/{tag}_sq_summary.txt", "w") as fh:
    lines = ["kernel filter: %s; avg duration under counters %.1f us" % (kern, sum(dur) / max(len(dur), 1) / 1e3)]
    lines += ["%-30s n=%d avg=%.4g" % (k, len(agg[k]), sum(agg[k]) / len(agg[k])) for k in sorted(agg)]
    for line in lines:
        print(line); fh.write(line + "\n")
PY
