#!/bin/bash
# tools/traffic_kernel.sh <tag> [bench args] -- HBM bytes per launch of every kernel of one bench.py
# run: separate FETCH_SIZE / WRITE_SIZE passes (KiB; read side doubled per MI355X_MICROARCH "HBM").
set -u
TAG=${1:-t}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d "$OUT/${TAG}_$c" -o pmc -- $BENCH > "$OUT/${TAG}_$c.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, collections, sys, re
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{tag}_{c}/pmc_counter_collection.csv")):
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][c].append(float(r["Counter_Value"]))
            agg[m.group(1)]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"{out}/{tag}_traffic.txt", "w") as fh:
    for k, v in agg.items():
        rd = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) * 1024 * 2
        wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) * 1024
        us = sum(v["us"]) / len(v["us"])
        line = "%-28s read %.4g B  write %.4g B  total %.4g B  avg %.1f us  -> %.0f GB/s" % (k, rd, wr, rd + wr, us, (rd + wr) / us / 1e3)
        print(line); fh.write(line + "\n")
PY
