"""Interleaved A/B timing of the decoders on the BASELINE shapes (device-resident, HIP events from the library).

    python tools/time_decode.py [--configs c4,alias256,byte,word] [--rounds 3] [--launches 20] [--mib 1024]

Every variant is a setting of the context options (all write the same bytes); per round each variant runs
`launches` back-to-back decodes after a warm-up, the table prints mean / min kernel ms and the fraction of the
8 TB/s HBM roofline on the algorithmic bytes (stream read + symbols written).  Output is verified once per variant.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from tools.config_sweep import zipf  # noqa: E402

CONFIGS = {
    # name: (format, scale_bits, nsyms, n_ways, chunk_syms, symbols per MiB of output)
    "c4": (R.FMT_ALIAS, 16, 4096, 64, 32768, 1 << 19),
    "alias256": (R.FMT_ALIAS, 16, 256, 64, 32768, 1 << 20),
    "byte": (R.FMT_BYTE, 14, 256, 64, 32768, 1 << 20),
    "word": (R.FMT_WORD, 12, 256, 64, 32768, 1 << 20),
    "c2": (R.FMT_R64, 14, 256, 2, 512, 1 << 20),
}
VARIANTS = {"dual": 2, "single": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c4,alias256")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    ctx.set_timing(True)
    for name in a.configs.split(","):
        fmt, sb, nsyms, ways, chunk, per_mib = CONFIGS[name]
        chunk = a.chunk or chunk
        n = a.mib * per_mib
        d = zipf(n, nsyms, 1, dev)
        f, _ = R.normalize_freqs(ctx.count_freqs_device(d, nsyms), 1 << sb)
        m = ctx.model(fmt, f, sb)
        cont, offs, lens, total = ctx.encode(m, d, ways, chunk)
        out = torch.empty_like(d)
        alg = total + n * d.element_size()
        print("== %s: %d symbols, %d stream bytes, chunk %d" % (name, n, total, chunk), flush=True)
        for v, val in VARIANTS.items():
            ctx.set_option(R.OPT_DUAL_DECODE, val)
            out.zero_()
            ctx.decode(m, cont, total, offs, lens, n, ways, chunk, d_out=out)
            print("   %-7s kernel %-24s output %s" % (v, ctx.last_decode_kernel(), "ok" if torch.equal(out, d) else "MISMATCH"),
                  flush=True)
        for r in range(a.rounds):
            for v, val in VARIANTS.items():
                ctx.set_option(R.OPT_DUAL_DECODE, val)
                for _ in range(min(40, 2 * a.launches)):  # settle the clocks on this kernel
                    ctx.decode(m, cont, total, offs, lens, n, ways, chunk, d_out=out, sync=False)
                torch.cuda.synchronize()
                ms = []
                for _ in range(a.launches):
                    ctx.decode(m, cont, total, offs, lens, n, ways, chunk, d_out=out, sync=False)
                    torch.cuda.synchronize()
                    ms.append(ctx.last_kernel_ms()[0])
                mean = sum(ms) / len(ms)
                print("   round %d %-7s mean %.4f ms  min %.4f ms  frac %.4f" % (r, v, mean, min(ms), alg / mean / 1e6 / 8000.0),
                      flush=True)
        ctx.set_option(R.OPT_DUAL_DECODE, 1)
        del d, out, cont
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
