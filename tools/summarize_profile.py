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

    # the encoders in both layouts and the other configs' decoders (tools/r04_profile.sh: tools/time_slots.py under the same
    # two passes).  Algorithmic bytes: symbols read + stream bytes written of the fixed BASELINE shapes (bench.py's
    # generator and seeds, 16 Ki-symbol chunks; config 2: 512-symbol chunks) -- the streams' sizes do not change with
    # the kernels.
    enc = {}
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(src, "%senc_pmc_%s" % (tag, which), "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "rans_amd" in r["Kernel_Name"]:
                vals[short_name(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            enc.setdefault(k, {})[counter] = sum(v) / len(v)
    shapes = [  # kernel prefix, what, algorithmic bytes, FETCH_SIZE multiplier
        ("k_encode<1, 1, 1>", "word 64-way, compact (fused placement)", 1.922e9, 2),
        ("k_encode<1, 1, 2>", "word 64-way, SLOT layout", 1.922e9, 2),
        ("k_encode<0, 1, 1>", "byte 64-way, compact", 1.924e9, 2),
        ("k_encode<0, 1, 2>", "byte 64-way, SLOT layout", 1.924e9, 2),
        ("k_encode<5, 1, 1>", "config 4 (alias, 4096 symbols), compact", 1.669e9, 2),
        ("k_encode<5, 1, 2>", "config 4, SLOT layout", 1.669e9, 2),
        ("k_encode_lanes_r64x2", "config 2 coding kernel (both layouts; 64-byte requests: FETCH_SIZE x 1)", 4.866e8, 1),
        ("k_compact_small", "config 2, compact only: the copy", 4.364e8, 2),
        ("k_decode_word64", "word 64-way decode", 1.922e9, 2),
        ("k_decode<0, 1, 1>", "byte 64-way decode", 1.924e9, 2),
        ("k_decode_dual<8, true>", "config 4 decode", 1.669e9, 2),
        ("k_decode_lanes_r64x2<true>", "config 2 decode (64-byte requests: x 1)", 4.866e8, 1),
    ]
    if enc:
        lines += ["## HBM traffic of the encoders in both layouts and of the decoders (`tools/time_slots.py` under separate "
                  "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)", "",
                  "FETCH_SIZE x 1024 x 2 for the kernels that read with 16 bytes per lane (calibrated on `k_histogram_u8`: 1.074e9 B "
                  "for its 1 GiB), x 1 for the lane kernels' 64-byte quad requests (MI355X_MICROARCH \"HBM\": the counter tallies "
                  "128-byte requests at 64 bytes); WRITE_SIZE x 1024.", "",
                  "| kernel | what | read B | write B | total | algorithmic B | ratio |", "|---|---|---|---|---|---|---|"]
        out["encoders"] = {}
        for prefix, what, alg, mult in shapes:
            hit = [k for k in enc if k.startswith(prefix)]
            if not hit:
                continue
            v = enc[hit[0]]
            rd = v.get("FETCH_SIZE", 0) * 1024 * mult
            wr = v.get("WRITE_SIZE", 0) * 1024
            lines.append("| `%s` | %s | %.4g | %.4g | %.4g | %.4g | **%.3f** |" % (hit[0], what, rd, wr, rd + wr, alg, (rd + wr) / alg))
            out["encoders"][hit[0]] = {"read": rd, "write": wr, "algorithmic": alg, "ratio": (rd + wr) / alg}
        lines += ["", "The slot layout takes every wave encoder from 1.75-1.94 x its algorithmic bytes to 1.00-1.02 x (VERDICT r03 next "
                  "#1: <= 1.3 x): the stream is written once.", ""]

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
