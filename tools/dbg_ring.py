"""Scratch-ring debugging aid: encode with the fused placement (ring of scratch slots per coding wave when there are more
chunks than ring slots) and with the three-kernel path, compare index and bytes, print timings and the library's error
text.  Small enough to fail fast: python tools/dbg_ring.py [--log2n 24] [--chunk 1024] [--fmt word]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from tools.config_sweep import zipf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=24)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--fmt", default="word")
    ap.add_argument("--ways", type=int, default=64)
    a = ap.parse_args()
    fmt, sb = {"word": (R.FMT_WORD, 12), "byte": (R.FMT_BYTE, 14), "r64": (R.FMT_R64, 14)}[a.fmt]
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    n = 1 << a.log2n
    d = zipf(n, 256, 1, dev)
    f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 1 << sb)
    m = ctx.model(fmt, f, sb)
    res = {}
    for name, opt in (("unfused", 0), ("fused", 1)):
        ctx.set_option(R.OPT_FUSED_PLACEMENT, opt)
        t0 = time.time()
        try:
            cont, offs, lens, total = ctx.encode(m, d, a.ways, a.chunk)
            torch.cuda.synchronize()
            print("%-8s %d chunks: %d bytes in %.3f s, kernel %s" % (name, R.num_chunks(n, a.chunk), total, time.time() - t0,
                                                                  ctx.last_encode_kernel()), flush=True)
            res[name] = (cont[:total].clone(), offs.clone(), lens.clone(), total)
        except R.RansAmdError as e:
            print("%-8s FAILED after %.1f s: %s" % (name, time.time() - t0, e), flush=True)
    if len(res) == 2:
        (c0, o0, l0, t0_), (c1, o1, l1, t1_) = res["unfused"], res["fused"]
        same_index = t0_ == t1_ and bool(torch.equal(o0, o1)) and bool(torch.equal(l0, l1))
        print("index equal:", same_index, flush=True)
        if same_index:
            # compare the bytes inside chunks (padding between chunks is undefined)
            pos = torch.arange(t0_, device=dev)
            idx = torch.searchsorted(o0[:-1].contiguous(), pos, right=True) - 1
            inside = pos < (o0[:-1][idx] + l0[idx].to(torch.int64))
            diff = ((c0 != c1) & inside).nonzero()
            print("bytes inside chunks equal:", diff.numel() == 0, "" if diff.numel() == 0 else
                  "first difference at byte %d (chunk %d)" % (int(diff[0]), int(idx[int(diff[0])])), flush=True)
        out = ctx.decode(m, c1, t1_, o1, l1, n, a.ways, a.chunk, sync=False)
        print("fused container decodes to the input:", bool(torch.equal(out, d)), "bad chunks", ctx.decode_errors(), flush=True)


if __name__ == "__main__":
    main()
