#!/usr/bin/env python3
"""Condense tools/r05_profile.sh's output directory into profiles/r05_*.

  python tools/summarize_r05.py gpurun_out/r05p

Every row of the tables it writes is ONE workload (round 4's kernel_stats.csv pooled five).
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide (16 B / lane) streaming read -- read side x 2 for the wave kernels, x 1 for the lane kernels' 64-byte
quad requests (calibrated in round 4 on k_histogram_u8: 1.074e9 B for its 1 GiB).
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def last_json_line(path):
    if not os.path.exists(path):
        return None
    for ln in reversed(open(path).read().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


def stats_rows(d):
    f = [os.path.join(d, x) for x in os.listdir(d)] if os.path.isdir(d) else []
    f = [x for x in f if x.endswith("kernel_stats.csv")]
    if not f:
        return []
    return [r for r in csv.DictReader(open(f[0])) if "rans_amd" in r["Name"]]


def split_redo(rows, name_key, grid_key):
    """The sized-slot encoder launches k_encode<.., 3> twice per call: the coders, then the (normally empty) redo pass,
    which ends after a few microseconds.  Give the redo launches a name of their own (told apart by their duration: under a
    tenth of the kernel's longest launch)."""
    longest = collections.defaultdict(float)
    for r in rows:
        longest[short(r[name_key])] = max(longest[short(r[name_key])], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    for r in rows:
        n = short(r[name_key])
        if n.endswith(", 3>") and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 0.1 * longest[n] or longest[n] < 20000):
            n += " redo"
        out.append((n, r))
    return out


def pmc_avg(d, counter):
    f = [os.path.join(d, x) for x in os.listdir(d)] if os.path.isdir(d) else []
    f = [x for x in f if x.endswith("counter_collection.csv")]
    vals = collections.defaultdict(list)
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and "rans_amd" in r["Kernel_Name"]]
        for n, r in split_redo(rows, "Kernel_Name", "Grid_Size"):
            vals[n].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


def trace_rows(d):
    """Per-kernel rows from the per-dispatch trace (calls, avg / min / max us), the redo launches apart."""
    f = [os.path.join(d, x) for x in os.listdir(d)] if os.path.isdir(d) else []
    f = [x for x in f if x.endswith("kernel_trace.csv")]
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f[0])) if "rans_amd" in r["Kernel_Name"]]
    agg = collections.OrderedDict()
    for n, r in split_redo(rows, "Kernel_Name", "Grid_Size_X"):
        agg.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return [(n, len(v), sum(v) / len(v), min(v), max(v)) for n, v in agg.items()]


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05p")
    dst = os.path.join(ROOT, "profiles")
    L = ["# r05 rocprofv3 evidence -- one workload per row (`tools/r05_profile.sh`)", ""]
    out = {"tag": "r05"}

    line = last_json_line(os.path.join(src, "bench.json"))
    if line:
        json.dump(line, open(os.path.join(dst, "r05_bench.json"), "w"))
        if os.path.exists(os.path.join(src, "bench_details.json")):
            shutil.copy(os.path.join(src, "bench_details.json"), os.path.join(dst, "r05_bench_details.json"))
        rl = line["roofline"]
        L += ["## A. `python bench.py --gpus 1 --steps 20 --warmup 5` (no profiler; the judged line, %d bytes)" % len(json.dumps(line, separators=(",", ":"))), "",
              "value %.1f GB/s, %.4f ms/step, kernel %.4f ms avg (HIP events), roofline.frac **%.4f**; first (un-probed) pair %.4f ms = %.4f"
              % (line["value"], line["ms_per_step"], rl["kernel_ms_avg"], rl["frac"], line.get("placement", {}).get("first_pair_ms", 0),
                 line.get("frac_first_pair", 0)), ""]

    # B / C: the headline alone under --kernel-trace --stats
    for tag, what in (("headline", "B. headline alone, placement probe on (the driver's command minus configs and CPU leg)"),
                      ("noprobe", "C. headline alone, `--placement-candidates 1` (every launch on the first pair of buffers)")):
        ln = last_json_line(os.path.join(src, tag + "_line.json"))
        det = json.load(open(os.path.join(src, tag + "_details.json"))) if os.path.exists(os.path.join(src, tag + "_details.json")) else None
        rows = stats_rows(os.path.join(src, tag + "_stats"))
        if not ln or not rows:
            continue
        sd = os.path.join(src, tag + "_stats")
        for x in os.listdir(sd):
            if x.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(sd, x), os.path.join(dst, "r05_kernel_stats.csv" if tag == "headline" else "r05_kernel_stats_noprobe.csv"))
        L += ["## " + what, "", "line under the profiler: kernel_ms_avg %.4f ms (HIP events over the %d timed launches), frac %.4f" %
              (ln["roofline"]["kernel_ms_avg"], ln["steps"], ln["roofline"]["frac"]), "",
              "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
        for r in rows:
            L.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                           float(r["MaxNs"]) / 1e3))
        L.append("")
        # cut the per-dispatch trace to the timed launches
        tr = [os.path.join(sd, x) for x in os.listdir(sd) if x.endswith("kernel_trace.csv")]
        if tr and det:
            disp = [r for r in csv.DictReader(open(tr[0])) if "k_decode_word64" in r["Kernel_Name"]]
            disp.sort(key=lambda r: int(r["Start_Timestamp"]))
            a, b = det["roofline"]["timed_dispatches"]
            win = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in disp[a:b]]
            allv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in disp]
            if win:
                avg = sum(win) / len(win)
                L += ["`k_decode_word64`: %d dispatches in the trace; the record names dispatches [%d, %d) as the timed ones: "
                      "**avg %.1f us** (min %.1f, max %.1f) against the line's kernel_ms_avg %.1f us (%+.1f %%); the whole row: avg %.1f us"
                      % (len(disp), a, b, avg, min(win), max(win), ln["roofline"]["kernel_ms_avg"] * 1e3,
                         (avg / (ln["roofline"]["kernel_ms_avg"] * 1e3) - 1) * 100, sum(allv) / len(allv)), ""]
                out[tag] = {"timed_dispatch_avg_us": avg, "timed_dispatch_min_us": min(win), "timed_dispatch_max_us": max(win),
                            "line_kernel_ms_avg": ln["roofline"]["kernel_ms_avg"], "row_avg_us": sum(allv) / len(allv),
                            "dispatches": len(disp), "timed": [a, b],
                            "frac_from_trace": ln["roofline"]["algorithmic_bytes_per_launch"] / (avg * 1e-6) / 8e12}
                L += ["roofline from the trace: %d B / %.1f us / 8 TB/s = **%.4f**" % (ln["roofline"]["algorithmic_bytes_per_launch"], avg,
                                                                                       out[tag]["frac_from_trace"]), ""]
                json.dump({"dispatch_us": [round(v, 2) for v in allv], "timed": [a, b]},
                          open(os.path.join(dst, "r05_%s_dispatches.json" % tag), "w"))

    # D: traffic of the headline kernel
    fe, wr = pmc_avg(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), pmc_avg(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    k = [x for x in fe if x.startswith("k_decode_word64")]
    if k and line:
        rd, w = fe[k[0]] * 1024 * 2, wr.get(k[0], 0) * 1024
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        out.update({"hbm_bytes_per_launch": rd + w, "hbm_read_bytes": rd, "hbm_write_bytes": w})
        L += ["## D. HBM traffic of `k_decode_word64` (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, headline alone)", "",
              "read %.4g B (FETCH_SIZE x 1024 x 2), write %.4g B, total %.4g B = **%.3f x** the %d algorithmic bytes" % (rd, w, rd + w, (rd + w) / alg, alg), ""]

    # E: the other decoders, one bench.py each
    rows_e = []
    for x in sorted(os.listdir(src)):
        if x.startswith("dec_") and x.endswith("_line.json"):
            tag = x[:-len("_line.json")]
            ln = last_json_line(os.path.join(src, x))
            rows = stats_rows(os.path.join(src, tag))
            if not ln or not rows:
                continue
            kern = ln["roofline"]["kernel"]
            r = max((r for r in rows if "k_decode" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
            rows_e.append("| %s | `%s` | %s | %.1f | %.1f | %.1f | %.1f | %.4f |" % (
                ln["config"]["workload"][:60], short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                float(r["MaxNs"]) / 1e3, ln["roofline"]["kernel_ms_avg"] * 1e3, ln["roofline"]["frac"]))
    if rows_e:
        L += ["## E. the other decoders, one `bench.py --format/--ways --no-configs --no-cpu-baseline --placement-candidates 1` each", "",
              "| workload | kernel | calls | row avg us | min | max | line kernel_ms_avg us (timed launches) | line frac |", "|---|---|---|---|---|---|---|---|"] + rows_e + [""]

    # F: encoders in the three layouts (tools/time_slots.py one configuration at a time)
    alg = {"word": 1.922e9, "byte": 1.924e9, "c4": 1.669e9, "c2": 4.866e8}
    rows_f = []
    out["encoders"] = {}
    for c in ("word", "byte", "c4", "c2"):
        rows = trace_rows(os.path.join(src, "enc_%s_stats" % c))
        fe, wr = pmc_avg(os.path.join(src, "enc_%s_fetch" % c), "FETCH_SIZE"), pmc_avg(os.path.join(src, "enc_%s_write" % c), "WRITE_SIZE")
        for n, calls, avg, mn, mx in rows:
            if not (n.startswith("k_encode") or n.startswith("k_compact") or n.startswith("k_decode")):
                continue
            mult = 1 if "lanes" in n else 2
            rd, w = fe.get(n, 0) * 1024 * mult, wr.get(n, 0) * 1024
            ratio = (rd + w) / alg[c] if (rd + w) else 0
            layout = ("compact (fused placement)" if n.endswith(", 1>") else "slots" if n.endswith(", 2>") else "SIZED slots: the coders"
                      if n.endswith(", 3>") else "SIZED slots: the redo launch (nothing overflowed)" if n.endswith("redo") else
                      "decode of the compact / slot / sized container (pooled)" if n.startswith("k_decode") else "")
            rows_f.append("| %s | `%s` | %s | %d | %.1f | %.1f | %.4g | %.4g | %s |" % (c, n, layout, calls, avg, mn, rd, w,
                                                                                    "%.3f" % ratio if not n.endswith("redo") else "-"))
            out["encoders"]["%s %s" % (c, n)] = {"avg_us": avg, "min_us": mn, "read": rd, "write": w, "algorithmic": alg[c], "ratio": ratio}
    if rows_f:
        L += ["## F. encoders in the three layouts and the decoders of their containers (`tools/time_slots.py --configs X`, one X per run)", "",
              "(`k_encode<FMT, K, MODE>`: MODE 1 = compact with fused placement, 2 = slots, 3 = sized slots, whose redo launches -- a "
              "quarter of the grid, told apart by it -- have a row of their own; the lane kernel `k_encode_lanes_r64x2` serves all three "
              "layouts of config 2 and its row pools them, `k_encode<2, 1, 3>` there is the redo launch alone; decode rows pool the three "
              "containers.)  Rows from the per-dispatch trace; traffic from separate PMC passes of the same command (FETCH_SIZE x 1024 x 2 "
              "for the wave kernels, x 1 for the lane kernels' 64-byte requests; WRITE_SIZE x 1024).", "",
              "| config | kernel | layout | calls | avg us | min us | read B | write B | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|---|"] + rows_f + [""]

    sys.path.insert(0, ROOT)
    try:
        import bench
        out["kernel_source_tag"] = bench.kernel_source_tag()
    except Exception as e:  # noqa: BLE001
        out["kernel_source_tag"] = None
        print("kernel_source_tag unavailable: %r" % (e,))
    json.dump(out, open(os.path.join(dst, "r05_traffic.json"), "w"), indent=1)
    open(os.path.join(dst, "r05_rocprof_summary.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
