"""rans_amd_container_compact over the sized-slot container of the BASELINE shapes: what the compact layout costs a caller who
encodes with rans_amd_encode_slots_sized and needs the oracle's offsets afterwards (files: rans_amd_container_pack).

    python tools/time_compact.py [--configs word,byte,c4,c2]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from tools.config_sweep import zipf  # noqa: E402
from tools.time_slots import CONFIGS, timed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="word,byte,c4,c2")
ap.add_argument("--chunk", type=int, default=16384)
a = ap.parse_args()
ctx = R.Context(0)
dev = torch.device("cuda", 0)
for name in a.configs.split(","):
    fmt, sb, nsyms, ways, log2n, chunk = CONFIGS[name]
    chunk = chunk or a.chunk
    n = 1 << log2n
    d = zipf(n, nsyms, 1, dev)
    f, _ = R.normalize_freqs(ctx.count_freqs_device(d, nsyms), 1 << sb)
    m = ctx.model(fmt, f, sb)
    c_cont, c_offs, c_lens, c_total = ctx.encode(m, d, ways, chunk)
    t_cont, t_offs, t_lens, t_total, t_slot = ctx.encode_sized(m, d, ways, chunk)
    nchunks = c_lens.numel()
    dst, doffs, total = ctx.compact(t_cont, t_total, t_offs, t_lens, nchunks)
    ok = total == c_total and bool(torch.equal(doffs, c_offs))
    enc = timed(lambda: ctx.encode_sized(m, d, ways, chunk, slot=t_slot, d_out=t_cont, sync=False, d_offsets=t_offs, d_lengths=t_lens), 20)
    cmp_ = timed(lambda: ctx.compact(t_cont, t_total, t_offs, t_lens, nchunks, d_dst=dst, sync=False, d_dst_offsets=doffs), 20)
    fused = timed(lambda: ctx.encode(m, d, ways, chunk, d_out=c_cont, sync=False, d_offsets=c_offs, d_lengths=c_lens), 20)
    print("%-6s sized encode %.4f ms + compaction %.4f ms = %.4f ms;  rans_amd_encode (compact, fused placement) %.4f ms;  %s" % (
        name, enc[0], cmp_[0], enc[0] + cmp_[0], fused[0], "same container" if ok else "MISMATCH"), flush=True)
    del d, c_cont, t_cont, dst
    torch.cuda.empty_cache()
