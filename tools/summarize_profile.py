#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/<tag>_* into profiles/<tag>_*.{md,json,csv}.

  python tools/summarize_profile.py r01

Inputs (written by tools/profile.sh / tools/pmc.sh on the GPU box):
  gpurun_out/<tag>_bench.json                     the un-profiled bench line
  gpurun_out/<tag>_stats/stats_kernel_stats.csv   rocprofv3 --kernel-trace --stats
  gpurun_out/<tag>_pmc_fetch|write/pmc_counter_collection.csv   FETCH_SIZE / WRITE_SIZE passes
  gpurun_out/<tag>_sq_summary.txt                 SQ counter averages (tools/pmc.sh)
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled (the decoder's stream fetch is exactly that access pattern).
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_name(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    out = {"tag": tag}
    lines = ["# rocprofv3 summary `%s`" % tag, ""]

    bj = os.path.join(src, tag + "_bench.json")
    bench = None
    if os.path.exists(bj):
        bench = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, tag + "_bench.json"))
        lines += ["bench.py (no profiler): value %.1f %s, %.4f ms/step, roofline.frac %.4f, kernel %.4f ms avg"
                  % (bench["value"], bench["unit"], bench["ms_per_step"], bench["roofline"]["frac"],
                     bench["roofline"]["kernel_ms_avg"]), ""]

    st = os.path.join(src, tag + "_stats", "stats_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, tag + "_kernel_stats.csv"))
        lines += ["## rocprofv3 --kernel-trace --stats (same command)", "",
                  "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(st)):
            name = r["Name"]
            if "rans_amd" not in name:
                continue
            short = short_name(name)
            lines.append("| %s | %s | %.1f | %.1f | %.1f | %s |" % (short, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                                  r["Percentage"]))
            if "k_decode_word64" in name:  # the headline kernel
                out["decode_kernel_avg_us_rocprof"] = float(r["AverageNs"]) / 1e3
        lines.append("")

    traffic = {}
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(src, "%s_pmc_%s" % (tag, which), "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "rans_amd" in r["Kernel_Name"]:
                vals[short_name(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            traffic.setdefault(k, {})[counter] = sum(v) / len(v)
    if traffic:
        lines += ["## HBM traffic per launch (separate --pmc passes)", "",
                  "| kernel | FETCH_SIZE KiB (raw) | read bytes (x1024 x2) | WRITE_SIZE KiB | write bytes | total |",
                  "|---|---|---|---|---|---|"]
        for k, v in traffic.items():
            rd = v.get("FETCH_SIZE", 0) * 1024 * 2
            wr = v.get("WRITE_SIZE", 0) * 1024
            lines.append("| %s | %.0f | %.4g | %.0f | %.4g | %.4g |" % (k, v.get("FETCH_SIZE", 0), rd,
                                                                      v.get("WRITE_SIZE", 0), wr, rd + wr))
            if "k_decode_word64" in k:
                out["hbm_bytes_per_launch"] = rd + wr
                out["hbm_read_bytes"] = rd
                out["hbm_write_bytes"] = wr
        if bench and "hbm_bytes_per_launch" in out:
            alg = bench["roofline"]["algorithmic_bytes_per_launch"]
            lines += ["", "decode: algorithmic bytes per launch %d, measured HBM traffic %.4g (x%.3f)"
                      % (alg, out["hbm_bytes_per_launch"], out["hbm_bytes_per_launch"] / alg)]
        lines.append("")

    sq = os.path.join(src, tag + "_sq_summary.txt")
    if os.path.exists(sq):
        lines += ["## SQ counters, decode kernel (avg per launch)", "", "```"] + open(sq).read().splitlines() + ["```", ""]

    # which kernel sources the measurement belongs to: bench.py quotes it as roofline.traffic only on a match
    sys.path.insert(0, ROOT)
    try:
        import bench
        out["kernel_source_tag"] = bench.kernel_source_tag()
    except Exception as e:  # noqa: BLE001
        out["kernel_source_tag"] = None
        print("kernel_source_tag unavailable: %r" % (e,))
    json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
    open(os.path.join(dst, tag + "_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
