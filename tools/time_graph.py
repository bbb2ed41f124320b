"""Eager launches against one hipGraph of the same launches: ms per decode of the headline configuration."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R
from tools.config_sweep import zipf

dev = torch.device("cuda", 0)
ctx = R.Context(0)
n, chunk, K = 1 << 30, 16384, 50
d = zipf(n, 256, 1, dev)
f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 4096)
m = ctx.model(R.FMT_WORD, f, 12)
cont, offs, lens, total = ctx.encode(m, d, 64, chunk)
out = torch.empty_like(d)
s = torch.cuda.Stream()


def run_eager():
    for _ in range(K):
        ctx.decode(m, cont, total, offs, lens, n, 64, chunk, d_out=out, sync=False)


with torch.cuda.stream(s):
    run_eager()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run_eager()
    for name, fn in (("eager", run_eager), ("graph", g.replay), ("eager", run_eager), ("graph", g.replay)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%s: %.4f ms per decode" % (name, e0.elapsed_time(e1) / (4 * K)), flush=True)
    assert ctx.decode_errors() == 0 and torch.equal(out, d)
