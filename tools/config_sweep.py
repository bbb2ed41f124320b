"""Times encode + decode of every BASELINE.json config on the GPU (device-resident, HIP events
from the library) and checks the round trip.  Diagnostics for DESIGN.md / profiles; the
headline number comes from bench.py.

    python tools/config_sweep.py [--quick] > gpurun_out/configs.md
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402


def zipf(n, K, seed, device):
    w = 1.0 / torch.arange(1, K + 1, dtype=torch.float64, device=device)
    cdf = torch.cumsum(w / w.sum(), 0).float()
    out = torch.empty(n, dtype=torch.uint8 if K <= 256 else torch.int16, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    step = min(n, 1 << 26)
    for i in range(0, n, step):
        m = min(step, n - i)
        v = torch.searchsorted(cdf, torch.rand(m, device=device, generator=g)).clamp_(max=K - 1)
        out[i:i + m] = v.to(out.dtype)
    return out


def run(ctx, name, fmt, sb, nsyms, n, ways, chunk, reps=5):
    dev = torch.device("cuda", 0)
    d = zipf(n, nsyms, 1, dev)
    counts = ctx.count_freqs_device(d, nsyms)
    f, _ = R.normalize_freqs(counts, 1 << sb)
    m = ctx.model(fmt, f, sb)
    ctx.set_timing(True)
    enc_best = 1e9
    for _ in range(3):
        cont, offs, lens, total = ctx.encode(m, d, ways, chunk)
        enc_best = min(enc_best, ctx.last_kernel_ms()[1])
    out = torch.empty_like(d)
    dec_best = 1e9
    for _ in range(reps):
        ctx.decode(m, cont, total, offs, lens, n, ways, chunk, d_out=out, sync=False)
        torch.cuda.synchronize()
        dec_best = min(dec_best, ctx.last_kernel_ms()[0])
    bad = ctx.decode_errors()
    ok = bool(torch.equal(out, d)) and bad == 0
    w = d.element_size()
    print("| %s | %d | %d | %d | %.4f | %.3f | %.1f | %.3f | %.1f | %.3f | %s |" % (
        name, n >> 20, ways, chunk, total / n, enc_best, n * w / enc_best / 1e6, dec_best, n * w / dec_best / 1e6,
        (n * w + total) / dec_best / 1e6 / 8000.0, "ok" if ok else "MISMATCH"), flush=True)
    del d, out, cont
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="substring filter on the config name")
    a = ap.parse_args()
    global run
    run_all = run

    def run(ctx, name, *rest, **kw):  # noqa: F811
        if a.only in name:
            run_all(ctx, name, *rest, **kw)
    ctx = R.Context(0)
    big = 28 if a.quick else 30
    print("| config | Mi symbols | ways | chunk | stream B/sym | enc ms | enc GB/s | dec ms | dec GB/s (out) | dec frac of 8 TB/s (in+out) | round trip |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    run(ctx, "C3 word 64-way 1 GiB", R.FMT_WORD, 12, 256, 1 << big, 64, 32768)
    run(ctx, "word 128-way", R.FMT_WORD, 12, 256, 1 << big, 128, 32768)
    run(ctx, "word 256-way", R.FMT_WORD, 12, 256, 1 << big, 256, 65536)
    run(ctx, "C1-on-GPU byte sb14 64-way", R.FMT_BYTE, 14, 256, 1 << big, 64, 32768)
    run(ctx, "byte sb16 64-way", R.FMT_BYTE, 16, 256, 1 << big, 64, 32768)
    run(ctx, "r64 sb14 64-way", R.FMT_R64, 14, 256, 1 << big, 64, 32768)
    run(ctx, "C2 r64 sb14 2-way 256 MiB", R.FMT_R64, 14, 256, 1 << 28, 2, 4096, reps=3)
    run(ctx, "C2 r64 2-way, 256-symbol chunks", R.FMT_R64, 14, 256, 1 << 28, 2, 256, reps=3)
    run(ctx, "C2 r64 2-way, 512-symbol chunks", R.FMT_R64, 14, 256, 1 << 28, 2, 512, reps=3)
    run(ctx, "C2 r64 2-way, 1024-symbol chunks", R.FMT_R64, 14, 256, 1 << 28, 2, 1024, reps=3)
    run(ctx, "word 2-way, 1024-symbol chunks", R.FMT_WORD, 12, 256, 1 << 28, 2, 1024, reps=3)
    run(ctx, "byte 2-way, 1024-symbol chunks", R.FMT_BYTE, 14, 256, 1 << 28, 2, 1024, reps=3)
    run(ctx, "alias 256 sym sb16 64-way", R.FMT_ALIAS, 16, 256, 1 << big, 64, 32768)
    run(ctx, "C4 alias 4096 sym sb16 64-way (u16)", R.FMT_ALIAS, 16, 4096, 1 << (big - 1), 64, 32768)


if __name__ == "__main__":
    main()
