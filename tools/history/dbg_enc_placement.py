"""Which buffer's placement makes the word encoder bimodal (0.69-0.72 against 0.76-0.82 ms for 1 GiB)?  Needs the measure
build (RANS_AMD_LIB=lib/libryg_rans_amd_measure.so): sweeps the offset of the scratch slots and of the status words inside
their allocations, then the container's and the symbols' allocations."""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R
from tools.config_sweep import zipf

dev = torch.device("cuda", 0)
ctx = R.Context(0)
ctx.set_timing(True)
L = R.lib()
L.rans_amd_measure_ptr.restype = C.c_uint64
L.rans_amd_measure_ptr.argtypes = [C.c_void_p, C.c_int]
L.rans_amd_measure_shift.restype = None
L.rans_amd_measure_shift.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 1 << 30
d = zipf(n, 256, 1, dev)
f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 4096)
m = ctx.model(R.FMT_WORD, f, 12)
MAXSH = 256 << 20
L.rans_amd_measure_shift(ctx._h, 0, MAXSH)
L.rans_amd_measure_shift(ctx._h, 1, MAXSH)
cont, offs, lens, total = ctx.encode(m, d, 64, chunk)
L.rans_amd_measure_shift(ctx._h, 0, 0)
L.rans_amd_measure_shift(ctx._h, 1, 0)


def t(d_, cont_):
    for _ in range(24):
        ctx.encode(m, d_, 64, chunk, d_out=cont_, sync=False, d_offsets=offs, d_lengths=lens)
    torch.cuda.synchronize()
    ms = []
    for _ in range(12):
        ctx.encode(m, d_, 64, chunk, d_out=cont_, sync=False, d_offsets=offs, d_lengths=lens)
        torch.cuda.synchronize()
        ms.append(ctx.last_kernel_ms()[1])
    return sum(ms) / len(ms)


t(d, cont)
print("chunk %d: syms %x cont %x scratch %x status %x" % (chunk, d.data_ptr(), cont.data_ptr(), L.rans_amd_measure_ptr(ctx._h, 0),
                                                        L.rans_amd_measure_ptr(ctx._h, 1)))
KB, MB = 1 << 10, 1 << 20
shifts = [0, 4 * KB, MB]  # (the long sweep showed nothing: profiles/r03_encoder_placement.md)
for which, name in ((0, "scratch"), (1, "status")):
    print("%s offset:" % name)
    for sh in shifts:
        L.rans_amd_measure_shift(ctx._h, which, sh)
        print("  +%-10d %.3f" % (sh, t(d, cont)), flush=True)
    L.rans_amd_measure_shift(ctx._h, which, 0)
print("container re-allocated:", end=" ")
conts = [torch.empty_like(cont) for _ in range(6)]
for c in conts:
    print("%.3f(%x)" % (t(d, c), c.data_ptr()), end=" ", flush=True)
print()

# is it the address translation?  random 4-byte reads / writes over the first GiB of each container allocation
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.randint(0, (1 << 30) // 4, (1 << 24,), device=dev, generator=g)
src = torch.ones(1 << 24, dtype=torch.int32, device=dev)


def t_rand(c, write):
    v = c.view(torch.int32)
    for _ in range(2):
        v.index_put_((idx,), src) if write else v.index_select(0, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        v.index_put_((idx,), src) if write else v.index_select(0, idx)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


for c in [cont] + conts:
    enc = t(d, c)
    print("cont %x: encode %.3f ms, 16 Mi random reads %.3f ms, random writes %.3f ms" % (c.data_ptr(), enc, t_rand(c, False), t_rand(c, True)),
          flush=True)
