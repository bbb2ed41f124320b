cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r5ah -o a -- python $GRAFT_REPO_ROOT/tools/time_adaptive.py > /dev/null 2>&1
python3 - <<'PY'
import csv,os,glob
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5ah/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "rans_amd" in r["Name"]: print("%-110s calls %4s avg %9.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3))
PY
find $GRAFT_REPO_ROOT/gpurun_out/r5ah -name "*.db" -delete
