"""Why does the same kernel on the same data time differently in the headline loop and in the `configs` entry?"""
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
cont, offs, lens, total = ctx.encode(m, d, 64, 32768)
out = torch.empty(n, dtype=torch.uint8, device=dev)

def t(label, cont, offs, lens, out):
    ms, mn = timed_launches(torch, lambda: ctx.decode(m, cont, total, offs, lens, n, 64, 32768, d_out=out, sync=False), 20, 2)
    print("%-50s mean %.4f min %.4f" % (label, ms, mn), flush=True)

t("A first buffers", cont, offs, lens, out)
t("A again", cont, offs, lens, out)
out2 = torch.empty_like(d)
t("B new out (empty_like)", cont, offs, lens, out2)
cont2, offs2, lens2, total2 = ctx.encode(m, d, 64, 32768)
t("C second encode's container, out2", cont2, offs2, lens2, out2)
t("D second container, first out", cont2, offs2, lens2, out)
t("E first container again, first out", cont, offs, lens, out)
big = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
out3 = torch.empty_like(d)
t("F out allocated after 3 GiB more", cont, offs, lens, out3)
print("ptrs cont %x cont2 %x out %x out2 %x out3 %x" % (cont.data_ptr(), cont2.data_ptr(), out.data_ptr(), out2.data_ptr(), out3.data_ptr()))

def in_arena(arena, off, src_t):
    nb = (total + 4095) & ~4095
    v = arena[off:off + nb]
    v.copy_(src_t[:nb])
    return v

for gib in (1, 2, 4, 8, 16):
    arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    c = in_arena(arena, 0, cont)
    t("G%d cont at start of a %d GiB arena, out first" % (gib, gib), c, offs, lens, out)
    if gib >= 2:
        o = arena[(1 << 30):(2 << 30)]
        t("H%d cont + out both inside the %d GiB arena" % (gib, gib), c, offs, lens, o)
    del arena, c
    torch.cuda.empty_cache()
# many small-ish allocations first, then a fresh container copy
c = torch.empty(total + 4096, dtype=torch.uint8, device=dev); c[:total].copy_(cont[:total])
t("I fresh copy of the container (own allocation)", c, offs, lens, out)
