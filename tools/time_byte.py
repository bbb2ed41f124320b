"""Byte-format 64-way decode at one scale_bits, 1 GiB of Zipf(256) bytes, output verified.  One process per variant (the
measure build reads its knobs at first use):

    RANS_AMD_LIB=ryg_rans_amd/lib/libryg_rans_amd_measure.so [RANS_AMD_NO_BYTE_FUSED=1 | RANS_AMD_BYTE_FUSED13=1 |
        RANS_AMD_BYTE_FMT_OUT=1] python tools/time_byte.py --sb 12 --tag fused
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from bench import gen_zipf, settle  # noqa: E402
from ryg_rans_amd.placement import choose_one, time_launches  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sb", type=int, default=14)
ap.add_argument("--chunk", type=int, default=16384)
ap.add_argument("--log2n", type=int, default=30)
ap.add_argument("--tag", default="")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = R.Context(0)
n = 1 << a.log2n
d = gen_zipf(torch, n, 256, 1.0, 1, dev)
f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 1 << a.sb)
m = ctx.model(R.FMT_BYTE, f, a.sb)
cont, offs, lens, total = ctx.encode(m, d, 64, a.chunk)
outs = [torch.empty_like(d) for _ in range(6)]
run = lambda o: ctx.decode(m, cont, total, offs, lens, n, 64, a.chunk, d_out=o, sync=False)  # noqa: E731
settle(torch, lambda: run(outs[0]), 200.0)
pick, ms = choose_one(torch, run, outs)
out = outs[pick]
out.zero_()
run(out)
ok = ctx.decode_errors() == 0 and bool(torch.equal(out, d))
res = [time_launches(torch, lambda: run(out), 20) for _ in range(a.rounds)]
alg = n + total
print("%-12s sb %2d  %s  best-of-6 placement: %s  ms %s  frac %.4f  %s" % (
    a.tag, a.sb, ctx.last_decode_kernel(), " ".join("%.4f" % v for v in ms), " ".join("%.4f" % v for v in res),
    alg / min(res) / 1e6 / 8000.0, "ok" if ok else "MISMATCH"), flush=True)
