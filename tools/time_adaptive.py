"""Per-chunk models (rans_amd_encode_adaptive[_fmt] / decode_adaptive[_fmt]) at 1 GiB: model building + encode, decode; byte
and word format, 16 Ki-symbol chunks, 64-way, verified.

    python tools/time_adaptive.py [log2n]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from tools.config_sweep import zipf  # noqa: E402
from tools.time_slots import timed  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, chunk, ways = 1 << log2n, 16384, 64
ctx = R.Context(0)
d = zipf(n, 256, 1, torch.device("cuda", 0))
for fmt, name, sb in ((R.FMT_BYTE, "byte", 12), (R.FMT_WORD, "word", 12)):
    cont, offs, lens, freqs, total = ctx.encode_adaptive(d, ways, chunk, sb, fmt=fmt)
    out = ctx.decode_adaptive(cont, total, offs, lens, freqs, n, ways, chunk, sb, fmt=fmt)
    ok = bool(torch.equal(out, d))
    enc = timed(lambda: ctx.encode_adaptive(d, ways, chunk, sb, sync=False, fmt=fmt), 10)
    dec = timed(lambda: ctx.decode_adaptive(cont, total, offs, lens, freqs, n, ways, chunk, sb, d_out=out, sync=False, fmt=fmt), 10)
    alg = n + total
    print("%-5s per-chunk models: encode (models + coding + layout + compaction) %.3f ms, decode %.3f ms (%.3f of the roofline), "
          "container %.4f B/sym + %.4f B/sym of frequency rows, %s / %s, %s" % (
              name, enc[0], dec[0], alg / dec[0] / 1e6 / 8000.0, total / n, 512.0 / chunk, ctx.last_encode_kernel()[0],
              ctx.last_decode_kernel(), "ok" if ok else "MISMATCH"), flush=True)
