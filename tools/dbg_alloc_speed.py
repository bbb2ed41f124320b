"""Random 4-byte reads over consecutive 2 GiB allocations of a fresh process: which allocations are the slow ones
(profiles/r03_encoder_placement.md)?  Then the same after freeing everything, and with 64 MiB allocations."""
import os, sys
import torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)


def probe(v, idx):
    for _ in range(2):
        v.index_select(0, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        v.index_select(0, idx)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


def sweep(count, gib, label):
    words = (gib << 30) // 4
    idx = torch.randint(0, words, (1 << 24,), device=dev, generator=g)
    bufs = []
    print(label)
    for i in range(count):
        b = torch.empty(words, dtype=torch.int32, device=dev)
        b.zero_()
        bufs.append(b)
        print("  #%-2d %x  %.3f ms" % (i, b.data_ptr(), probe(b, idx)), flush=True)
    return bufs


free, total = torch.cuda.mem_get_info()
print("free %.1f GiB of %.1f" % (free / 2**30, total / 2**30))
a = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 24, 2, "2 GiB allocations, fresh process")
del a
torch.cuda.empty_cache()
a = sweep(8, 2, "after freeing them all")
del a
torch.cuda.empty_cache()
a = sweep(6, 8, "8 GiB allocations")
