#!/usr/bin/env python3
"""profiles/<tag>_counters.md from the SQ / LDS / TA summaries tools/pmc.sh and tools/pmc_kernel.sh leave under gpurun_out/
(tools/r04_profile.sh runs them for the headline decoder, the byte decoders and the slot-layout word and byte encoders).

    python tools/summarize_counters.py r04
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = float(1 << 24)  # 64-symbol rounds of a 1 GiB launch (config 4: 512 Mi symbols = 2^23 rounds: its per-round columns read double)

SETS = [  # (summary file suffix, title)
    ("", "k_decode_word64 (headline)"),
    ("bytef", "k_decode<byte, slot records> (scale_bits 12)"),
    ("byte", "k_decode<byte> (scale_bits 14)"),
    ("encw", "k_encode<word>, slot layout (16-byte records)"),
    ("encb", "k_encode<byte>, slot layout (mirrored sub-step)"),
    ("encw3", "k_encode<word>, SIZED slots (MODE 3: exact room check where the staged stream is flushed)"),
    ("encc4", "k_encode<alias, LDS remap>, config 4, slot layout"),
    ("encc43", "k_encode<alias, LDS remap>, config 4, SIZED slots (MODE 3: room check before every pair of rounds)"),
]


def read(path):
    vals = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+n=\d+\s+avg=([0-9.e+\-]+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    return vals


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out")
    head = ["# Round %s: SQ / LDS / TA counters of the dominant kernels (rocprofv3 --pmc, separate passes, `tools/%s_profile.sh`)"
            % (tag[1:].lstrip("0"), tag), "",
            "Derived as in rounds 2-3: cycles = GRBM_GUI_ACTIVE / 8 (per XCD); VALU busy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / "
            "cycles; LDS pipe = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles; TA = TA_TA_BUSY_sum / 256 / cycles; per-round counts = "
            "SQ_INSTS_* / (64-symbol rounds of the launch: 2^24).", "",
            "| kernel | VALU busy | LDS pipe | conflict share | TA | s_waitcnt / issue stall / issuing (share of wave cycles) | "
            "VALU / SALU / LDS instr per round |", "|---|---|---|---|---|---|---|"]
    raw = ["", "Raw averages per launch:", ""]
    for suffix, title in SETS:
        f = os.path.join(src, "%s%s_sq_summary.txt" % (tag, suffix))
        if not os.path.exists(f):
            continue
        v = read(f)
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        ta = "%.0f %%" % (100 * v["TA_TA_BUSY_sum"] / 256 / cyc) if "TA_TA_BUSY_sum" in v else "-"
        head.append("| %s | %.0f %% | %.0f %% | %.0f %% | %s | %.0f / %.0f / %.0f %% | %.2f / %.2f / %.2f |" % (
            title, 100 * 4 * v["SQ_ACTIVE_INST_VALU"] / 1024 / cyc, 100 * v["SQ_LDS_IDX_ACTIVE"] / 256 / cyc,
            100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], ta, 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
            100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"],
            v["SQ_INSTS_VALU"] / ROUNDS, v["SQ_INSTS_SALU"] / ROUNDS, v["SQ_INSTS_LDS"] / ROUNDS))
        raw += ["## " + title, "", "```"] + ["%-30s %.4g" % (k, v[k]) for k in sorted(v)] + ["```", ""]
    out = os.path.join(ROOT, "profiles", tag + "_counters.md")
    open(out, "w").write("\n".join(head + raw) + "\n")
    print("\n".join(head))


if __name__ == "__main__":
    main()
