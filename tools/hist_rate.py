"""Times the device histogram (rans_amd_count_freqs) on 1 GiB of Zipf bytes, uniform bytes and a
constant buffer (the worst case for same-counter conflicts), and on u16 symbols.
    python tools/hist_rate.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ryg_rans_amd as R  # noqa: E402
from config_sweep import zipf  # noqa: E402

ctx = R.Context(0)
dev = torch.device("cuda", 0)
n = 1 << 30
cases = [("zipf(256,1) u8", zipf(n, 256, 1, dev), 256),
         ("uniform u8", torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev), 256),
         ("constant u8", torch.full((n,), 7, dtype=torch.uint8, device=dev), 256),
         ("zipf(256,1) u8, +1 byte offset", zipf(n + 1, 256, 1, dev)[1:], 256),
         ("zipf(4096,1) u16", zipf(n // 2, 4096, 1, dev), 4096)]
for name, d, nsyms in cases:
    ref = torch.bincount(d.view(-1).to(torch.int64) if d.numel() <= (1 << 24) else d[:1 << 24].to(torch.int64),
                         minlength=nsyms).cpu().numpy()
    got_small = ctx.count_freqs_device(d[:1 << 24], nsyms)
    assert np.array_equal(got_small, ref[:nsyms]), name
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = ctx.count_freqs_device(d, nsyms)
        best = min(best, time.perf_counter() - t0)
    assert int(c.sum()) == d.numel()
    print("%-34s %8.3f ms  %7.1f GB/s (call incl. readback)" % (name, best * 1e3, d.numel() * d.element_size() / best / 1e9))
    del d
    torch.cuda.empty_cache()
