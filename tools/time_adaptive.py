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
    c1, o1, l1, f1, t1 = ctx.encode_adaptive_sized(d, ways, chunk, sb, fmt=fmt)
    out1 = ctx.decode_adaptive(c1, t1, o1, l1, f1, n, ways, chunk, sb, fmt=fmt)
    ok = ok and bool(torch.equal(out1, d)) and bool(torch.equal(f1, freqs)) and bool(torch.equal(l1, lens))
    enc1 = timed(lambda: ctx.encode_adaptive_sized(d, ways, chunk, sb, fmt=fmt, d_out=c1, d_offsets=o1, d_lengths=l1, d_freqs=f1,
                                                   sync=False), 10)
    k1 = ctx.last_encode_kernel()[0]
    dec1 = timed(lambda: ctx.decode_adaptive(c1, t1, o1, l1, f1, n, ways, chunk, sb, d_out=out1, sync=False, fmt=fmt), 10)
    rows = n // chunk * 512
    print("%-5s ONE kernel (%s): encode %.3f ms (%.3f of the roofline on n + stream + rows), its container %.4f B/sym, decodes in "
          "%.3f ms" % (name, k1, enc1[0], (alg + rows) / enc1[0] / 1e6 / 8000.0, t1 / n, dec1[0]), flush=True)
    print("%-5s per-chunk models: encode (models + coding + layout + compaction) %.3f ms, decode %.3f ms (%.3f of the roofline), "
          "container %.4f B/sym + %.4f B/sym of frequency rows, %s / %s, %s" % (
              name, enc[0], dec[0], alg / dec[0] / 1e6 / 8000.0, total / n, 512.0 / chunk, ctx.last_encode_kernel()[0],
              ctx.last_decode_kernel(), "ok" if ok else "MISMATCH"), flush=True)
