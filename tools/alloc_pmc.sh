#!/bin/bash
# tools/alloc_pmc.sh <tag> -- hardware counters of the SAME decode kernel over the same bytes in different allocations
# (tools/alloc_probe.py pmc): one rocprofv3 --pmc pass per counter group (each pass is its own process and draws its own
# allocations; inside a pass every placement gets its counters AND its duration, so counters are correlated with speed
# placement by placement).  Run via gpurun; summary in gpurun_out/<tag>_alloc_pmc.md.
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/alloc_probe.py pmc --launches 6 --torch-pairs 8 --hip-pairs 2 --shifts 0"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" \
           "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY GRBM_EA_BUSY TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -f csv -d "$OUT/${TAG}_alloc_pmc$i" -o pmc -- $CMD > "$OUT/${TAG}_alloc_pmc$i.log" 2>&1
  echo "pass $i rc=$? : $set"
done
python3 "$REPO/tools/alloc_pmc_summary.py" "$OUT" "$TAG" | tee "$OUT/${TAG}_alloc_pmc.md"
