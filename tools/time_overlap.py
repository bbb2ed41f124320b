"""Independent shards pipelined over two contexts on two streams against one stream: whole-job decode rate of the
headline configuration (the tail of one launch -- waves finish between 0.79 and 1.0 of its duration -- overlaps the head
of the next).  Not what bench.py reports: its per-launch roofline needs launches that do not share the GPU."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R
from tools.config_sweep import zipf

dev = torch.device("cuda", 0)
n, chunk, K = 1 << 30, 16384, 40
ctxs = [R.Context(0), R.Context(0)]
d = zipf(n, 256, 1, dev)
f, _ = R.normalize_freqs(ctxs[0].count_freqs_device(d, 256), 4096)
ms = [c.model(R.FMT_WORD, f, 12) for c in ctxs]
cont, offs, lens, total = ctxs[0].encode(ms[0], d, 64, chunk)
outs = [torch.empty_like(d), torch.empty_like(d)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two):
    for i in range(K):
        j = i & 1 if two else 0
        with torch.cuda.stream(streams[j]):
            ctxs[j].decode(ms[j], cont, total, offs, lens, n, 64, chunk, d_out=outs[j], sync=False)


for two in (False, True, False, True):
    run(two)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    streams[0].wait_event(e0); streams[1].wait_event(e0)
    run(two)
    run(two)
    for s in streams:
        e = torch.cuda.Event(); e.record(s); torch.cuda.current_stream().wait_event(e)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (2 * K)
    print("%s: %.4f ms per decode, %.0f GB/s decoded" % ("two streams" if two else "one stream ", t, n / t / 1e6), flush=True)
assert all(c.decode_errors() == 0 for c in ctxs) and torch.equal(outs[0], d) and torch.equal(outs[1], d)
