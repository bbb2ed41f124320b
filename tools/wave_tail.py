"""How even is the end of a decode launch?  Per-wave start / end ticks of ONE headline decode (measure build,
RANS_AMD_TRACE=<file>: wave, start, end [100 MHz ticks], XCD, shader cycles, rounds), after a warm-up: the distribution of the
waves' end times relative to the launch's span, and what fraction of the wave-slots' time lies behind a wave's end.

    RANS_AMD_LIB=ryg_rans_amd/lib/libryg_rans_amd_measure.so RANS_AMD_TRACE=/tmp/trace.txt python tools/wave_tail.py [chunk]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ryg_rans_amd as R  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
path = os.environ["RANS_AMD_TRACE"]
ctx = R.Context(0)
dev = torch.device("cuda", 0)
n = 1 << 30
d = bench.gen_zipf(torch, n, 256, 1.0, 1, dev)
f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 4096)
m = ctx.model(R.FMT_WORD, f, 12)
cont, offs, lens, total = ctx.encode(m, d, 64, chunk)
out = torch.empty_like(d)
for _ in range(200):
    ctx.decode(m, cont, total, offs, lens, n, 64, chunk, d_out=out, sync=False)
torch.cuda.synchronize()
ctx.decode(m, cont, total, offs, lens, n, 64, chunk, d_out=out)  # (the trace file holds the last launch)
rows = np.loadtxt(path, dtype=np.float64)
start, end, xcd, cyc, rounds = rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4], rows[:, 5]
t0, t1 = start.min(), end.max()
span = t1 - t0
rel_end = (end - t0) / span
rel_start = (start - t0) / span
print("chunk %d: %d waves, span %.1f us; wave start: max %.4f of the span; wave end: min %.3f, p10 %.3f, median %.3f, p90 %.3f"
      % (chunk, len(rows), span / 100.0, rel_start.max(), rel_end.min(), np.percentile(rel_end, 10), np.median(rel_end), np.percentile(rel_end, 90)))
print("idle share of the wave slots behind their wave's end: %.2f %%; before its start: %.2f %%" %
      (100 * (1 - rel_end).mean(), 100 * rel_start.mean()))
print("rounds per wave: min %d median %d max %d (= %.1f .. %.1f chunks)" % (rounds.min(), np.median(rounds), rounds.max(),
                                                                         rounds.min() * 64 / chunk, rounds.max() * 64 / chunk))
# per SIMD-ish view: a CU's 32 waves are two blocks of 16; when does the LAST wave of each CU end, when the first?
per_block = rel_end.reshape(-1, 16)
print("per block of 16 waves: first wave out at %.3f (mean), last at %.3f (mean) of the span" % (per_block.min(axis=1).mean(), per_block.max(axis=1).mean()))
# who is fast?  rounds decoded per wave by position: wave slot inside the block (slot & 3 = SIMD), block parity, XCD
w = rows[:, 0].astype(np.int64)
slot, blk = w % 16, w // 16
for name, key in (("wave slot in its block", slot), ("SIMD (slot & 3)", slot & 3), ("block parity", blk & 1), ("XCD", xcd.astype(np.int64)),
                  ("block index mod 8", blk % 8)):
    print("rounds per wave by %s:" % name, " ".join("%d:%.0f" % (k, rounds[key == k].mean()) for k in np.unique(key)))
cu = blk // 2
per_cu = np.array([rounds[cu == c].sum() for c in np.unique(cu)])
print("rounds per CU (pair of blocks): min %.0f median %.0f max %.0f" % (per_cu.min(), np.median(per_cu), per_cu.max()))
