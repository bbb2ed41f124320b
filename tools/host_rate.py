"""PCIe-inclusive rate of the host-buffer wrappers (rans_amd_encode_host / rans_amd_decode_host):
pageable host memory in, pageable host memory out, staging buffers kept in the context between calls."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R

n = 1 << 28
rng = np.random.default_rng(1)
w = 1.0 / np.arange(1, 257); cdf = np.cumsum(w / w.sum())
data = np.searchsorted(cdf, rng.random(n)).clip(max=255).astype(np.uint8)
ctx = R.Context(0)
m = ctx.model_for(R.FMT_WORD, data, 256, 12)
for _ in range(2):
    t0 = time.perf_counter(); s = ctx.encode_host(m, data, 64); t1 = time.perf_counter()
    out = ctx.decode_host(m, s, n, 64); t2 = time.perf_counter()
    assert np.array_equal(out, data)
    print("host wrappers, one 64-way stream of %d MiB: encode_host %.1f ms (%.2f GB/s), decode_host %.1f ms (%.2f GB/s)"
          % (n >> 20, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9))

# a book1-sized input (768 771 symbols, the reference's own test file size) through the reference's 8-way layout: what a
# call costs when the staging buffers are already there (first call: they are allocated)
small = data[:768771]
for i in range(4):
    t0 = time.perf_counter(); s8 = ctx.encode_host(m, small, 8); t1 = time.perf_counter()
    o8 = ctx.decode_host(m, s8, small.size, 8); t2 = time.perf_counter()
    assert np.array_equal(o8, small)
    print("host wrappers, 768771 symbols, 8-way, call %d: encode_host %.2f ms, decode_host %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3))

# chunked path with the caller moving the buffers over PCIe (what a host application pays)
import torch
n2 = 1 << 30
d = torch.from_numpy(np.tile(data, n2 // n))
d_dev = d.cuda()
cont, offs, lens, total = ctx.encode(m, d_dev, 64, 32768)
h_cont = cont[:total].cpu()
for pinned in (False, True):
    h_in = h_cont.pin_memory() if pinned else h_cont
    h_out = torch.empty(n2, dtype=torch.uint8).pin_memory() if pinned else torch.empty(n2, dtype=torch.uint8)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c_dev = h_in.cuda(non_blocking=True)
        o_dev = ctx.decode(m, c_dev, total, offs, lens, n2, 64, 32768)
        h_out.copy_(o_dev, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.equal(h_out, d)
    print("chunked decode of 1 GiB incl. H2D of the container and D2H of the symbols (%s host memory): %.1f ms = %.1f GB/s"
          % ("pinned" if pinned else "pageable", dt * 1e3, n2 / dt / 1e9))
