"""First-contact GPU probe: runs every (format, n_ways) through encode/decode against
the oracle, reports pass/fail per case without stopping, then times the word
decoder at a few sizes.  Diagnostics tool (uses the oracle => not product code).

    python tools/gpu_probe.py [--quick] > gpurun_out/probe.log
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ryg_rans_amd as R  # noqa: E402
from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD, FMT_NAMES, Oracle  # noqa: E402


def case(ctx, orc, fmt, sb, n_ways, data, nsyms=256):
    f, _ = orc.normalize(orc.count_freqs(data, nsyms), 1 << sb)
    om = orc.model(f, sb, with_alias=(fmt == FMT_ALIAS))
    gm = ctx.model(fmt, f, sb)
    want = orc.encode(fmt, om, data, n_ways)
    res = []
    try:
        out, rc = ctx.decode_host(gm, want, data.size, n_ways, check=False)
        ok = rc == 0 and np.array_equal(out, data)
        if not ok:
            bad = np.nonzero(out != data)[0]
            res.append("DEC FAIL rc=%d first_bad=%s nbad=%d" % (rc, bad[:4], bad.size))
        else:
            res.append("dec ok")
    except Exception as e:  # noqa: BLE001
        res.append("DEC EXC %r" % e)
    try:
        got = ctx.encode_host(gm, data, n_ways)
        if got.size == want.size and np.array_equal(got, want):
            res.append("enc ok")
        else:
            nb = -1
            if got.size == want.size:
                nb = int(np.nonzero(got != want)[0][0])
            res.append("ENC FAIL size %d vs %d first_diff=%d" % (got.size, want.size, nb))
    except Exception as e:  # noqa: BLE001
        res.append("ENC EXC %r" % e)
    return " | ".join(res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    print("torch", torch.__version__, "devices", torch.cuda.device_count(), torch.cuda.get_device_name(0))
    ctx = R.Context(0)
    orc = Oracle()
    data = orc.gen_zipf(70003, K=256, s=1.0, seed=1)
    for fmt, sb in ((FMT_WORD, 12), (FMT_BYTE, 14), (FMT_R64, 14), (FMT_ALIAS, 16), (FMT_BYTE, 16)):
        for n_ways in (64, 1, 2, 33, 128, 256, 512):
            try:
                print("%-6s sb=%2d N=%3d : %s" % (FMT_NAMES[fmt], sb, n_ways, case(ctx, orc, fmt, sb, n_ways, data)),
                      flush=True)
            except Exception:  # noqa: BLE001
                traceback.print_exc()
    d16 = orc.gen_zipf(50001, K=4096, s=1.0, seed=1)
    for n_ways in (64, 256):
        print("alias4096 N=%3d : %s" % (n_ways, case(ctx, orc, FMT_ALIAS, 16, n_ways, d16, nsyms=4096)), flush=True)

    # ---- timing: word format, 64-way and wider, device resident --------------
    ctx.set_timing(True)
    for log2n in ((24, 28) if args.quick else (24, 28, 30)):
        n = 1 << log2n
        w = 1.0 / torch.arange(1, 257, dtype=torch.float64, device="cuda")
        cdf = torch.cumsum(w / w.sum(), 0).float()
        d_syms = torch.empty(n, dtype=torch.uint8, device="cuda")
        step = 1 << 24
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        for i in range(0, n, step):
            d_syms[i:i + step] = torch.searchsorted(cdf, torch.rand(step, device="cuda", generator=g)).clamp_(max=255).to(torch.uint8)
        counts = ctx.count_freqs_device(d_syms, 256)
        f, _ = R.normalize_freqs(counts, 4096)
        gm = ctx.model(FMT_WORD, f, 12)
        for n_ways, chunk in ((64, 32768), (64, 16384), (64, 65536), (128, 32768), (256, 65536)):
            try:
                t0 = time.time()
                cont, offs, lens, total = ctx.encode(gm, d_syms, n_ways, chunk)
                torch.cuda.synchronize()
                t_enc = time.time() - t0
                out = torch.empty_like(d_syms)
                best = 1e9
                for _ in range(5):
                    ctx.decode(gm, cont, total, offs, lens, n, n_ways, chunk, d_out=out, sync=False)
                    torch.cuda.synchronize()
                    ms, ems = ctx.last_kernel_ms()
                    best = min(best, ms)
                bad = ctx.decode_errors()
                okk = bool(torch.equal(out, d_syms))
                print("n=2^%d N=%3d chunk=%6d: stream %.4f B/sym, enc %.1f ms (kernels %.2f ms), dec best %.3f ms = "
                      "%.1f GB/s out, %.1f GB/s in+out, roundtrip=%s bad=%d" %
                      (log2n, n_ways, chunk, total / n, t_enc * 1e3, ems, best, n / best / 1e6,
                       (n + total) / best / 1e6, okk, bad), flush=True)
            except Exception:  # noqa: BLE001
                traceback.print_exc()
        del d_syms
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
