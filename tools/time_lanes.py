"""Sustained decode (and encode) timing of one lane-per-stream configuration; one process per variant so that the
library's environment knobs (RANS_AMD_DEBUG, RANS_AMD_LANES, RANS_AMD_LIB) can differ between runs.

    python tools/time_lanes.py [--fmt r64] [--ways 2] [--chunk 512] [--log2n 28] [--extra 0] [--sb 14] [--steps 20] [--no-check] [--encode]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from bench import gen_zipf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fmt", default="r64")
    ap.add_argument("--ways", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--sb", type=int, default=14)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--encode", action="store_true")
    ap.add_argument("--extra", type=int, default=0, help="symbols on top of 2^log2n (a ragged last chunk)")
    a = ap.parse_args()
    fmt = {"r64": R.FMT_R64, "word": R.FMT_WORD, "byte": R.FMT_BYTE, "alias": R.FMT_ALIAS}[a.fmt]
    ctx = R.Context(0)
    dev = torch.device("cuda", 0)
    n = (1 << a.log2n) + a.extra
    d = gen_zipf(torch, n, 256, 1.0, 1, dev)
    f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 1 << a.sb)
    m = ctx.model(fmt, f, a.sb)
    ctx.set_timing(True)
    cont, offs, lens, total = ctx.encode(m, d, a.ways, a.chunk)
    out = torch.empty_like(d)
    for _ in range(30):
        ctx.decode(m, cont, total, offs, lens, n, a.ways, a.chunk, d_out=out, sync=False)
    torch.cuda.synchronize()
    ms = []
    for _ in range(a.steps):
        ctx.decode(m, cont, total, offs, lens, n, a.ways, a.chunk, d_out=out, sync=False)
        torch.cuda.synchronize()
        ms.append(ctx.last_kernel_ms()[0])
    bad = ctx.decode_errors()
    ok = a.no_check or (bool(torch.equal(out, d)) and bad == 0)
    line = "decode %s %d-way chunk %d n 2^%d%s sb %d: mean %.4f ms min %.4f ms  frac %.3f  %s" % (
        a.fmt, a.ways, a.chunk, a.log2n, " + %d" % a.extra if a.extra else "", a.sb, float(np.mean(ms)), float(np.min(ms)),
        (n + total) / (np.mean(ms) * 1e-3) / 8e12, "ok" if ok else "MISMATCH (bad=%d)" % bad)
    if a.encode:
        es = []
        for _ in range(a.steps):
            ctx.encode(m, d, a.ways, a.chunk)
            es.append(ctx.last_kernel_ms()[1])
        line += " | encode mean %.4f ms min %.4f" % (float(np.mean(es)), float(np.min(es)))
    print(line, flush=True)


if __name__ == "__main__":
    main()
