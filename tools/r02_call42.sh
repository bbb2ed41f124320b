#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/c42 -o st -- python $R/tools/time_lanes.py --fmt r64 --ways 2 --encode --no-check > $R/gpurun_out/c42.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/c42/**/st_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
