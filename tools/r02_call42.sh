#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "X=0"; do
  rm -rf $R/gpurun_out/c42
  env $v timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/c42 -o st -- python $R/tools/time_lanes.py --fmt r64 --ways 2 --encode > $R/gpurun_out/c42.log 2>&1
  echo "== $v"; tail -1 $R/gpurun_out/c42.log
  python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/c42/**/st_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rans_amd' in r["Name"] and ('compact' in r["Name"] or 'encode' in r["Name"]): print(r["Name"][32:75], r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
