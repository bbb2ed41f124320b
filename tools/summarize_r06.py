#!/usr/bin/env python3
"""Condense tools/r06_profile.sh's output directory into profiles/r06_*.

  python tools/summarize_r06.py gpurun_out/r06p

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
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06p")
    dst = os.path.join(ROOT, "profiles")
    L = ["# r06 rocprofv3 evidence -- one workload per row (`tools/r06_profile.sh`)", ""]
    out = {"tag": "r06"}

    line = last_json_line(os.path.join(src, "bench.json"))
    if line:
        json.dump(line, open(os.path.join(dst, "r06_bench.json"), "w"))
        if os.path.exists(os.path.join(src, "bench_details.json")):
            shutil.copy(os.path.join(src, "bench_details.json"), os.path.join(dst, "r06_bench_details.json"))
        rl = line["roofline"]
        L += ["## A. `python bench.py --gpus 1 --steps 20 --warmup 5` (no profiler; the judged line, %d bytes)" % len(json.dumps(line, separators=(",", ":"))), "",
              "value %.1f GB/s, %.4f ms/step, roofline.frac **%.4f** (one clock: algorithmic bytes / ms_per_step); kernel %.4f ms avg by HIP events = "
              "%.4f; first (un-probed) pair, K timed steps: %.4f ms/step = %.4f" % (line["value"], line["ms_per_step"], rl["frac"], rl["kernel_ms_avg"],
                 rl.get("frac_kernel_events", 0), line.get("placement", {}).get("first_pair_ms_per_step", 0), line.get("frac_first_pair", 0)), "",
              "| config | decode ms | frac | compact encode ms | sized / one-kernel encode ms | container / input | oracle |", "|---|---|---|---|---|---|---|"]
        for r in line.get("configs", []):
            L.append("| %s | %s | %s | %s | %s | %s | %s |" % (r.get("name"), r.get("decode_ms"), r.get("decode_frac"), r.get("encode_ms", "-"),
                                                            r.get("enc_tight_ms", "-"), r.get("tight_size", "-"), r.get("oracle_ok")))
        L.append("")

    # B / C: the headline alone under --kernel-trace --stats
    for tag, what in (("headline", "B. headline alone, placement probe on (the driver's command minus configs and CPU leg)"),):
        ln = last_json_line(os.path.join(src, tag + "_line.json"))
        det = json.load(open(os.path.join(src, tag + "_details.json"))) if os.path.exists(os.path.join(src, tag + "_details.json")) else None
        rows = stats_rows(os.path.join(src, tag + "_stats"))
        if not ln or not rows:
            continue
        sd = os.path.join(src, tag + "_stats")
        for x in os.listdir(sd):
            if x.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(sd, x), os.path.join(dst, "r06_kernel_stats.csv" if tag == "headline" else "r06_kernel_stats_noprobe.csv"))
        L += ["## " + what, "", "line under the profiler: ms_per_step %.4f, kernel_ms_avg %.4f ms (HIP events over the %d timed launches), frac %.4f" %
              (ln["ms_per_step"], ln["roofline"]["kernel_ms_avg"], ln["steps"], ln["roofline"]["frac"]), "",
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
                          open(os.path.join(dst, "r06_%s_dispatches.json" % tag), "w"))

    # D: traffic of the headline kernel
    fe, wr = pmc_avg(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), pmc_avg(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    k = [x for x in fe if x.startswith("k_decode_word64")]
    if k and line:
        rd, w = fe[k[0]] * 1024 * 2, wr.get(k[0], 0) * 1024
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        out.update({"hbm_bytes_per_launch": rd + w, "hbm_read_bytes": rd, "hbm_write_bytes": w})
        L += ["## D. HBM traffic of `k_decode_word64` (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, headline alone)", "",
              "read %.4g B (FETCH_SIZE x 1024 x 2), write %.4g B, total %.4g B = **%.3f x** the %d algorithmic bytes" % (rd, w, rd + w, (rd + w) / alg, alg), ""]

    # G: per-chunk models (tools/time_adaptive.py: 1 GiB Zipf(256), 16 Ki-symbol chunks, 64-way, 12 bits; rows = 65536 x 512 B)
    n, rows_b = 1 << 30, 65536 * 512
    streams = {"word": 0.7886 * n, "byte": 0.7906 * n}  # (container of the compact path: profiles/r06_adaptive_baseline.log)
    fe, wr = pmc_avg(os.path.join(src, "adaptive_fetch"), "FETCH_SIZE"), pmc_avg(os.path.join(src, "adaptive_write"), "WRITE_SIZE")
    rows_g = []
    out["adaptive"] = {}
    names = {"k_encode_adaptive<1, 1, 16>": ("word", "ONE-kernel encoder (count + normalise + records + code), chunk resident in registers"),
             "k_encode_adaptive<0, 1, 16>": ("byte", "ONE-kernel encoder, chunk resident in registers"),
             "k_decode<12, 1, 1>": ("word", "decoder (four-wave workgroups, packed records)"),
             "k_decode<7, 1, 1>": ("byte", "decoder"),
             "k_chunk_models": ("both", "three-launch path: models"), "k_encode<12, 1, 0>": ("word", "three-launch path: coding into scratch"),
             "k_encode<0, 1, 0>": ("byte", "three-launch path: coding into scratch")}
    for nme, calls, avg, mn, mx in trace_rows(os.path.join(src, "adaptive_stats")):
        if nme not in names:
            continue
        fmt, what = names[nme]
        alg = n + streams.get(fmt, 0.79 * n) + rows_b
        rd, w = fe.get(nme, 0) * 1024 * 2, wr.get(nme, 0) * 1024
        rows_g.append("| `%s` | %s, %s | %d | %.1f | %.1f | %.4f | %.4g | %.4g | %s |" % (
            nme, fmt, what, calls, avg, mn, alg / (avg * 1e-6) / 8e12, rd, w, "%.3f" % ((rd + w) / alg) if rd + w else "-"))
        out["adaptive"][nme] = {"avg_us": avg, "min_us": mn, "read": rd, "write": w, "algorithmic": alg}
    if rows_g:
        L += ["## G. per-chunk models, 1 GiB Zipf(256), 16 Ki-symbol chunks, 64-way, 12 bits (`tools/time_adaptive.py 30`)", "",
              "(algorithmic bytes = symbols + streams + frequency rows, each once; FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 from separate PMC "
              "passes.  The one-kernel encoder keeps a 4 / 8 / 16 Ki-symbol chunk in registers between the count and the coding pass: every "
              "input byte crosses the fabric once.  Its two-pass form -- other chunk sizes, ragged chunks -- reads the chunk twice: 1.62 x, "
              "1.54 x with non-temporal coding-pass loads and stream stores: profiles/r06_adaptive_encoder_variants.log.)", "",
              "| kernel | what | calls | avg us | min us | frac of 8 TB/s | read B | write B | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|---|"] + rows_g + [""]

    # H: the reference's own layouts on the lane kernels
    rows_h = []
    out["lanes"] = {}
    for tag, what, alg in (("word8", "word 8-way, 1024-symbol chunks", n + 0.8077 * n), ("byte2", "byte 2-way (14 bits), 1024-symbol chunks", n + 0.7896 * n)):
        fe = pmc_avg(os.path.join(src, "lanes_%s_fetch" % tag), "FETCH_SIZE")
        wr = pmc_avg(os.path.join(src, "lanes_%s_write" % tag), "WRITE_SIZE")
        for nme, calls, avg, mn, mx in trace_rows(os.path.join(src, "lanes_%s_stats" % tag)):
            if not (nme.startswith("k_decode_lanes") or nme.startswith("k_encode_lanes") or nme.startswith("k_compact") or
                    nme.startswith("k_decode_word_groups") or nme.startswith("k_decode_byte_pairs") or nme.startswith("k_encode_word_groups")):
                continue
            # FETCH_SIZE counts fabric read requests at 64 bytes each (MI355X_MICROARCH.md): x 1 for the lane kernels' 64-byte
            # quad requests (calibrated in round 4) and for the pair decoder's 32-byte blocks (taken as 64-byte sector fetches),
            # x 2 for the word group decoder and encoder, whose groups read whole 128-byte lines like the headline's waves
            rd, w = fe.get(nme, 0) * 1024 * (2 if nme.startswith(("k_decode_word_groups", "k_encode_word_groups")) else 1), wr.get(nme, 0) * 1024
            rows_h.append("| %s | `%s` | %d | %.1f | %.1f | %.4f | %.4g | %.4g | %s |" % (
                what, nme, calls, avg, mn, alg / (avg * 1e-6) / 8e12, rd, w, "%.3f" % ((rd + w) / alg) if rd + w else "-"))
            out["lanes"]["%s %s" % (tag, nme)] = {"avg_us": avg, "min_us": mn, "read": rd, "write": w, "algorithmic": alg}
    if rows_h:
        L += ["## H. the reference's own layouts: 1 GiB Zipf(256); decoders 8 / 32 chunks per wave (`decode_groups.hip`), the word encoder 8 chunks per wave (`encode_groups.hip`), the byte encoder one LANE per chunk (`tools/time_lanes.py --chunk 1024 --encode`)", "",
              "| layout | kernel | calls | avg us | min us | frac of 8 TB/s | read B | write B | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|---|"] + rows_h + [""]

    # I: counters of the kernels whose bound DESIGN states -- derived as in rounds 2-5 (tools/summarize_counters.py)
    def read_counters(path):
        vals = {}
        for line in open(path):
            m = re.match(r"(\w+)\s+n=\d+\s+avg=([0-9.e+\-]+)", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
        return vals
    derived = ["## I. counters (separate `--pmc` passes, `tools/pmc_kernel.sh`): derived", "",
               "cycles = GRBM_GUI_ACTIVE / 8 (per XCD); VALU busy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / cycles; LDS pipe = SQ_LDS_IDX_ACTIVE / 256 CUs / "
               "cycles; per-round counts = SQ_INSTS_* / 2^24 (64-symbol rounds of a 1 GiB launch); waves per CU = SQ_WAVES / 256 (persistent grids).", "",
               "| kernel | ms under counters | waves per CU | VALU busy | LDS pipe | conflict share | waiting / issue stall / issuing (share of wave cycles) | VALU / SALU / LDS instr per round |",
               "|---|---|---|---|---|---|---|---|"]
    for tag, what in (("cnt_word8", "`k_decode_word_groups`, word 8-way, 1024-symbol chunks"), ("cnt_byte2", "`k_decode_byte_pairs`, byte 2-way"),
                      ("cnt_word8enc", "`k_encode_word_groups`, word 8-way (scratch slots)"),
                      ("cnt_adec", "`k_decode<word, per-chunk models>`"), ("cnt_aenc", "`k_encode_adaptive<word>`")):
        f = os.path.join(src, tag + "_sq_summary.txt")
        if not os.path.exists(f):
            continue
        v = read_counters(f)
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        ms = re.search(r"avg duration under counters ([0-9.]+) us", open(f).read())
        derived.append("| %s | %.3f | %.0f | %.0f %% | %.0f %% | %.0f %% | %.0f / %.0f / %.0f %% | %.2f / %.2f / %.2f |" % (
            what, float(ms.group(1)) / 1e3 if ms else 0, v["SQ_WAVES"] / 256, 100 * 4 * v["SQ_ACTIVE_INST_VALU"] / 1024 / cyc,
            100 * v["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
            100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
            100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"] / 2 ** 24, v["SQ_INSTS_SALU"] / 2 ** 24, v["SQ_INSTS_LDS"] / 2 ** 24))
    if len(derived) > 6:
        L += derived + [""]
    for tag, what in (("cnt_word8", "k_decode_word_groups, word 8-way"), ("cnt_byte2", "k_decode_byte_pairs, byte 2-way"), ("cnt_word8enc", "k_encode_word_groups, word 8-way"),
                      ("cnt_adec", "k_decode<word, per-chunk models>"), ("cnt_aenc", "k_encode_adaptive<word>")):
        f = os.path.join(src, tag + "_sq_summary.txt")
        if os.path.exists(f):
            L += ["## I. counters: " + what, "", "```"] + open(f).read().rstrip().split("\n") + ["```", ""]

    sys.path.insert(0, ROOT)
    try:
        import bench
        out["kernel_source_tag"] = bench.kernel_source_tag()
    except Exception as e:  # noqa: BLE001
        out["kernel_source_tag"] = None
        print("kernel_source_tag unavailable: %r" % (e,))
    json.dump(out, open(os.path.join(dst, "r06_traffic.json"), "w"), indent=1)
    open(os.path.join(dst, "r06_rocprof_summary.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
