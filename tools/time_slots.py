"""Compact layout (rans_amd_encode) against slot layout (rans_amd_encode_slots) at the BASELINE shapes: encode time, and
the decode time of the container each leaves behind, interleaved on one box; output verified both ways.

    python tools/time_slots.py [--configs word,byte,c4,c2] [--chunk 16384] [--rounds 3] [--launches 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from tools.config_sweep import zipf  # noqa: E402

# name: (format, scale_bits, alphabet, n_ways, log2 of the symbol count, chunk or None = --chunk)
CONFIGS = {"word": (R.FMT_WORD, 12, 256, 64, 30, None), "byte": (R.FMT_BYTE, 14, 256, 64, 30, None),
           "c4": (R.FMT_ALIAS, 16, 4096, 64, 29, None), "c2": (R.FMT_R64, 14, 256, 2, 28, 512),
           "word128": (R.FMT_WORD, 12, 256, 128, 30, None), "word256": (R.FMT_WORD, 12, 256, 256, 30, None),
           "r64": (R.FMT_R64, 14, 256, 64, 30, None)}


def timed(fn, launches):
    for _ in range(max(8, launches // 2)):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    return sum(ms) / len(ms), min(ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="word,byte,c4,c2")
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--tight-slot-add", type=int, default=0, help="bytes added to rans_amd_tight_slot_bytes() (experiments with the slot stride)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    for name in a.configs.split(","):
        fmt, sb, nsyms, ways, log2n, chunk = CONFIGS[name]
        chunk = chunk or a.chunk
        n = 1 << log2n
        d = zipf(n, nsyms, 1, dev)
        f, _ = R.normalize_freqs(ctx.count_freqs_device(d, nsyms), 1 << sb)
        m = ctx.model(fmt, f, sb)
        c_cont, c_offs, c_lens, c_total = ctx.encode(m, d, ways, chunk)
        s_cont, s_offs, s_lens, s_total = ctx.encode_slots(m, d, ways, chunk)
        t_cont, t_offs, t_lens, t_total, t_slot = ctx.encode_sized(m, d, ways, chunk, slot=(ctx.tight_slot_bytes(m, ways, chunk) + a.tight_slot_add) if a.tight_slot_add else None)  # sized slots (rans_amd_encode_slots_sized)
        ok = bool(torch.equal(c_lens, s_lens)) and bool(torch.equal(c_lens, t_lens))
        out = torch.empty_like(d)
        try:  # (a library variant that is wrong by construction -- a dropped store, timing only -- still gets its encode times)
            ctx.decode(m, c_cont, c_total, c_offs, c_lens, n, ways, chunk, d_out=out)
            ok = ok and bool(torch.equal(out, d))
            out.zero_()
            ctx.decode(m, s_cont, s_total, s_offs, s_lens, n, ways, chunk, d_out=out)
            ok = ok and bool(torch.equal(out, d))
            out.zero_()
            ctx.decode(m, t_cont, t_total, t_offs, t_lens, n, ways, chunk, d_out=out)
            ok = ok and bool(torch.equal(out, d))
        except R.RansAmdError as e:
            print("%-8s decode of the encoder's output failed: %s" % (name, e), flush=True)
            ok = False
        alg = n * d.element_size() + c_total
        k_dec = ctx.last_decode_kernel()
        for r in range(a.rounds):
            res = {}
            res["enc compact"] = timed(lambda: ctx.encode(m, d, ways, chunk, d_out=c_cont, sync=False, d_offsets=c_offs, d_lengths=c_lens), a.launches)
            k_c = ctx.last_encode_kernel()[0]
            res["enc slots  "] = timed(lambda: ctx.encode_slots(m, d, ways, chunk, d_out=s_cont, sync=False, d_offsets=s_offs, d_lengths=s_lens), a.launches)
            k_s = ctx.last_encode_kernel()[0]
            res["enc tight  "] = timed(lambda: ctx.encode_sized(m, d, ways, chunk, slot=t_slot, d_out=t_cont, sync=False, d_offsets=t_offs, d_lengths=t_lens), a.launches)
            res["dec compact"] = timed(lambda: ctx.decode(m, c_cont, c_total, c_offs, c_lens, n, ways, chunk, d_out=out, sync=False), a.launches)
            res["dec slots  "] = timed(lambda: ctx.decode(m, s_cont, s_total, s_offs, s_lens, n, ways, chunk, d_out=out, sync=False), a.launches)
            res["dec tight  "] = timed(lambda: ctx.decode(m, t_cont, t_total, t_offs, t_lens, n, ways, chunk, d_out=out, sync=False), a.launches)
            bad = ctx.decode_errors()
            for k, (mean, mn) in res.items():
                print("%-8s chunk %-6d round %d  %s  mean %.4f ms  min %.4f ms  frac %.4f   %s" % (
                    name, chunk, r, k, mean, mn, alg / mean / 1e6 / 8000.0,
                    (k_c if k == "enc compact" else k_s if k.startswith("enc") else k_dec)), flush=True)
        print("%-8s %s  stream %.4f B/sym  slot container %.2f x input  sized-slot container %.3f x input (slot %d)  bad chunks %d" % (
            name, "ok" if ok and bad == 0 else "MISMATCH", c_total / n, s_total / (n * d.element_size()),
            t_total / (n * d.element_size()), t_slot, bad), flush=True)
        del d, c_cont, s_cont, t_cont, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
