#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
export TMPDIR=/tmp
( timeout 240 python -m pytest tests -m gpu -q -x --timeout 90 -k "lane or layout or stress or scale or chunked or two_way or tiny or unaligned" 2>&1 | tail -6 ) > gpurun_out/c44_pytest.log
tail -3 gpurun_out/c44_pytest.log
cd /tmp
rm -rf $R/gpurun_out/c44
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/c44 -o st -- python $R/tools/time_lanes.py --fmt r64 --ways 2 --encode > $R/gpurun_out/c44.log 2>&1
grep -v rocprofv3 $R/gpurun_out/c44.log | tail -1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/c44/**/st_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rans_amd' in r["Name"] and ('compact' in r["Name"] or 'encode' in r["Name"]): print(r["Name"][32:75], r["Calls"], r["AverageNs"], r["MinNs"])
PY
