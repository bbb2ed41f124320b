#!/bin/bash
# tools/pmc_all.sh <tag> -- SQ/LDS/TA counter passes over one bench.py run (headline + configs), summarised per kernel
set -u
TAG=${1:-all}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH=${PMC_CMD:-"python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prewarm-ms 0 --config-steps 2"}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE TA_BUSY_sum TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -f csv -d "$OUT/${TAG}_sq$i" -o pmc -- $BENCH > "$OUT/${TAG}_sq$i.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, collections, glob, re, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"rans_amd::", "", n)
    n = re.sub(r"\(rans_amd::(Dec|Enc)Params\)", "", n)
    return n.replace("void ", "")[:60]
for f in glob.glob(f"{out}/{tag}_sq*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"{out}/{tag}_sq_summary.txt", "w") as fh:
    for k in sorted(agg):
        a = {c: sum(v) / len(v) for c, v in agg[k].items()}
        us = sum(dur[k]) / len(dur[k]) / 1e3
        cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0  # per XCD
        def pct(x, per):  # cycles per unit / kernel cycles
            return 100.0 * x / per / cyc if cyc else 0.0
        line = ("%-60s %8.1f us n=%d | VALU %.0f%% SALU %.0f%% LDSpipe %.0f%% (conflict %.0f%% of it) TA %.0f%% | wait_any %.0f%% of wave cycles | "
                "insts/wave-k: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f+%.0f | TCC rd %.3g wr %.3g" % (
            k, us, len(dur[k]), pct(4 * a.get("SQ_ACTIVE_INST_VALU", 0), 1024), pct(4 * a.get("SQ_INST_CYCLES_SALU", 0), 1024),
            pct(a.get("SQ_LDS_IDX_ACTIVE", 0), 256),
            100.0 * a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1), pct(a.get("TA_TA_BUSY_sum", 0), 256),
            100.0 * a.get("SQ_WAIT_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1),
            a.get("SQ_INSTS_VALU", 0) / 1e3, a.get("SQ_INSTS_SALU", 0) / 1e3, a.get("SQ_INSTS_LDS", 0) / 1e3,
            a.get("SQ_INSTS_VMEM_RD", 0) / 1e3, a.get("SQ_INSTS_VMEM_WR", 0) / 1e3,
            a.get("TCP_TCC_READ_REQ_sum", 0), a.get("TCP_TCC_WRITE_REQ_sum", 0)))
        print(line); fh.write(line + "\n")
PY
