"""Placement lottery: the same decode with the output (or the container) in different allocations."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R
from bench import gen_zipf, timed_launches

dev = torch.device("cuda", 0)
ctx = R.Context(0)
n = 1 << 30
d = gen_zipf(torch, n, 256, 1.0, 1, dev)
f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 4096)
m = ctx.model(R.FMT_WORD, f, 12)
cont, offs, lens, total = ctx.encode(m, d, 64, 16384)
nb = (total + 4095) & ~4095

def t(cont, out):
    ms, mn = timed_launches(torch, lambda: ctx.decode(m, cont, total, offs, lens, n, 64, 16384, d_out=out, sync=False), 12, 2)
    return ms

outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(10)]
conts = []
for _ in range(6):
    c = torch.empty(nb, dtype=torch.uint8, device=dev); c.copy_(cont[:nb]); conts.append(c)
print("cont ptrs", ["%x" % c.data_ptr() for c in conts])
print("out  ptrs", ["%x" % o.data_ptr() for o in outs])
print("rows = containers, columns = outputs")
for c in conts:
    print(" ".join("%.3f" % t(c, o) for o in outs), flush=True)

def t_fill(o):
    for _ in range(3):
        o.fill_(3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        o.fill_(7)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10

def t_read(o):
    for _ in range(3):
        s = o.view(torch.int64).sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        s = o.view(torch.int64).sum()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10

print("fill_ ms per out buffer:", " ".join("%.3f" % t_fill(o) for o in outs))
print("sum   ms per out buffer:", " ".join("%.3f" % t_read(o) for o in outs))
print("decode again           :", " ".join("%.3f" % t(conts[0], o) for o in outs))
