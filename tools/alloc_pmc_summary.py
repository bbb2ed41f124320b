"""Summarise tools/alloc_pmc.sh: per pass, per placement: mean duration of its timed k_decode_word64 dispatches and the mean
of every counter of the pass; then the correlation of each counter with the duration over the placements of its pass."""
import collections
import csv
import glob
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
print("# %s: counters of k_decode_word64 per placement (tools/alloc_pmc.sh)\n" % tag)
dirs = [p for p in glob.glob(os.path.join(out, tag + "_alloc_pmc[0-9]*")) if os.path.isdir(p)]
for d in sorted(dirs, key=lambda p: int(re.findall(r"(\d+)$", p)[0])):
    log = d + ".log"
    seq = [ln.split() for ln in open(log) if ln.startswith("PMC-PLACEMENT")] if os.path.exists(log) else []
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files or not seq:
        print("## %s: no data (%d csv, %d placements)\n" % (os.path.basename(d), len(files), len(seq)))
        continue
    rows = collections.defaultdict(dict)  # dispatch -> counter -> value
    times = {}
    for r in csv.DictReader(open(files[0])):
        if "k_decode_word64" not in r["Kernel_Name"]:
            continue
        did = int(r["Dispatch_Id"])
        rows[did][r["Counter_Name"]] = rows[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if "End_Timestamp" in r and r["End_Timestamp"]:
            times[did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(rows)
    per = sum(int(s[3]) for s in seq)
    ids = ids[-per:]  # the placement sequence is the tail of the process
    counters = sorted({c for v in rows.values() for c in v})
    print("## pass %s: %s\n" % (os.path.basename(d), " ".join(counters)))
    print("| placement | ms (events, in-process) | us (rocprof) | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    at, table = 0, []
    for s in seq:
        name, launches, ms = s[1], int(s[3]), float(s[5])
        mine = ids[at + 3:at + launches]  # the first three are untimed warm-ups
        at += launches
        if not mine:
            continue
        dur = sum(times.get(i, 0.0) for i in mine) / len(mine)
        vals = [sum(rows[i].get(c, 0.0) for i in mine) / len(mine) for c in counters]
        table.append((name, ms, dur, vals))
        print("| %s | %.4f | %.1f | %s |" % (name, ms, dur, " | ".join("%.4g" % v for v in vals)))
    if len(table) >= 3:
        xs = [t[2] for t in table]
        mx = sum(xs) / len(xs)
        print("\ncorrelation with the duration over these placements: " + ", ".join(
            "%s %+.2f" % (c, (lambda ys: (sum((x - mx) * (y - sum(ys) / len(ys)) for x, y in zip(xs, ys)) /
                                          ((sum((x - mx) ** 2 for x in xs) * sum((y - sum(ys) / len(ys)) ** 2 for y in ys)) ** 0.5 or 1.0)))(
                [t[3][k] for t in table])) for k, c in enumerate(counters)))
    print()
