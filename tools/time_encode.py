"""Sustained encode timing of the wave-per-chunk encoders on the BASELINE shapes (device-resident, HIP events from the
library), output verified by decoding.  One process per library build / knob setting (they are read at load time):

    [RANS_AMD_LIB=... RANS_AMD_ENC_NO_RING=1] python tools/time_encode.py [--configs word,byte] [--chunk 32768] [--tag name]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from tools.config_sweep import zipf  # noqa: E402

CONFIGS = {"word": (R.FMT_WORD, 12, 256, 64), "byte": (R.FMT_BYTE, 14, 256, 64), "r64": (R.FMT_R64, 14, 256, 64),
           "c4": (R.FMT_ALIAS, 16, 4096, 64), "alias256": (R.FMT_ALIAS, 16, 256, 64)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="word,byte")
    ap.add_argument("--chunk", type=int, default=32768)
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--tag", default="")
    ap.add_argument("--placements", type=int, default=1, help="repeat with the container and the scratch re-allocated elsewhere")
    ap.add_argument("--ring", type=int, default=0, help="RANS_AMD_OPT_ENC_SCRATCH_RING")
    ap.add_argument("--fused", type=int, default=1, help="RANS_AMD_OPT_FUSED_PLACEMENT")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    ctx.set_timing(True)
    ctx.set_option(R.OPT_ENC_SCRATCH_RING, a.ring)
    ctx.set_option(R.OPT_FUSED_PLACEMENT, a.fused)
    for name in a.configs.split(","):
        fmt, sb, nsyms, ways = CONFIGS[name]
        n = (1 << a.log2n) // (2 if nsyms > 256 else 1)  # (u16 symbols: the same number of bytes)
        d = zipf(n, nsyms, 1, dev)
        f, _ = R.normalize_freqs(ctx.count_freqs_device(d, nsyms), 1 << sb)
        m = ctx.model(fmt, f, sb)
        means, pads = [], []
        for pl in range(a.placements):
            # timings move by several per cent with where the container and the scratch happen to lie: every placement
            # frees both, shifts the allocator by an odd amount and allocates them again
            if pl:
                del cont, offs, lens
                ctx.trim()
                torch.cuda.empty_cache()
                pads.append(torch.empty((pl * 7 + 3) * (1 << 20) + pl * 4096, dtype=torch.uint8, device=dev))
            cont, offs, lens, total = ctx.encode(m, d, ways, a.chunk)
            try:  # (measurement builds with output-changing knobs: the container may be garbage)
                out = ctx.decode(m, cont, total, offs, lens, n, ways, a.chunk)
                ok = bool(torch.equal(out, d))
                del out
            except R.RansAmdError:
                ok = False
            kern = ctx.last_encode_kernel()
            for r in range(a.rounds):
                for _ in range(min(30, 2 * a.launches)):
                    ctx.encode(m, d, ways, a.chunk, d_out=cont, sync=False, d_offsets=offs, d_lengths=lens)
                torch.cuda.synchronize()
                ms = []
                for _ in range(a.launches):
                    ctx.encode(m, d, ways, a.chunk, d_out=cont, sync=False, d_offsets=offs, d_lengths=lens)
                    torch.cuda.synchronize()
                    ms.append(ctx.last_kernel_ms()[1])
                mean = sum(ms) / len(ms)
                means.append(mean)
                print("%-10s %-5s chunk %-6d placement %d round %d  mean %.4f ms  min %.4f ms  frac %.4f  %s %s" % (
                    a.tag, name, a.chunk, pl, r, mean, min(ms), (total + n * d.element_size()) / mean / 1e6 / 8000.0, kern,
                    "ok" if ok else "MISMATCH"), flush=True)
        if len(means) > 1:
            srt = sorted(means)
            print("%-10s %-5s chunk %-6d SUMMARY over %d: median %.4f  min %.4f  max %.4f" % (
                a.tag, name, a.chunk, len(srt), srt[len(srt) // 2], srt[0], srt[-1]), flush=True)
        del pads
        del d, cont
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
