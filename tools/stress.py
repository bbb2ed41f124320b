"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random format, scale_bits, alphabet
skew, n, N, chunk size, buffer misalignment and kernel-family options of the context; every case checks
  * GPU encode == oracle encode (lengths, offsets, every chunk's bytes),
  * GPU decode of the ORACLE container == input, GPU decode of its own container == input,
  * the slot layout (rans_amd_encode_slots): every chunk == the oracle's stream at the end of its slot, decode from it,
    rans_amd_container_compact of it == the compact container, a random chunk range of it decoded on its own,
  * sized slots (rans_amd_encode_slots_sized) with the model's slot or a random one: every chunk == the oracle's stream in
    its slot or in the overflow region, the index rule, decode from it.
    python tools/stress.py [--cases 300] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ryg_rans_amd as R  # noqa: E402
from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD, Oracle  # noqa: E402


def run(cases, seed, ctx=None, oracle=None, big=False):
    """Returns the number of failing cases (details are printed)."""
    rng = np.random.default_rng(seed)
    oracle = oracle or Oracle()
    ctx = ctx or R.Context(0)
    fails = 0
    for case in range(cases):
        fmt = int(rng.choice([FMT_WORD, FMT_BYTE, FMT_R64, FMT_ALIAS]))
        sb = {FMT_WORD: 12, FMT_BYTE: int(rng.integers(8, 17)), FMT_R64: int(rng.integers(8, 17)),
              FMT_ALIAS: int(rng.integers(8, 17))}[fmt]
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 70000), rng.integers(70000, 400000)]))
        if big and rng.integers(0, 8) == 0:  # now and then: enough chunks for several rounds of every persistent grid
            n = int(rng.integers(3_000_000, 7_000_000))
        n_ways = int(rng.choice([1, 2, 4, 8, 64, 128, 256, 512, int(rng.integers(1, 513))]))
        chunk = int(rng.choice([n + 5, 16 * int(rng.integers(1, 300)), int(rng.integers(1, 5000)), 64 * int(rng.integers(1, 64)),
                                256 * int(rng.integers(1, 40))]))
        kind = int(rng.integers(0, 4))
        # alphabet: mostly 256 byte symbols; sometimes smaller, for alias a power of two, and
        # occasionally the 4096-symbol u16 alphabet of BASELINE config 4
        K = 256
        pick = int(rng.integers(0, 10))
        if fmt == FMT_ALIAS:
            if pick == 0 and sb >= 12:
                K = 4096
            elif pick <= 3:
                K = 1 << int(rng.integers(1, min(sb, 8) + 1))
        elif pick <= 2:
            K = int(rng.integers(2, 257))
        dt = np.uint8 if K <= 256 else np.uint16
        if kind == 0:
            data = oracle.gen_zipf(n, K=K, s=float(rng.uniform(0.3, 2.5)), seed=int(rng.integers(1, 1 << 30))).astype(dt)
        elif kind == 1:
            data = rng.integers(0, K, n).astype(dt)
        elif kind == 2:
            data = np.minimum(rng.geometric(float(rng.uniform(0.02, 0.7)), n) - 1, K - 1).astype(dt)
        else:
            data = (rng.integers(0, 2, n) * int(rng.integers(1, K))).astype(dt)
        if len(np.unique(data)) < 2:
            continue
        if big and rng.integers(0, 3) == 0:  # config 2's shape: the dedicated 2-way rans64 lane decoder / encoder (whole batches
            fmt, n_ways, K = FMT_R64, 2, 256  #  of 64 full chunks + a tail), any scale_bits with a cum2sym table
            sb = int(rng.integers(8, 17))
            chunk = 64 * int(rng.integers(1, 9))
            n = int(rng.integers(256 * 64 * chunk, 300 * 64 * chunk)) + int(rng.integers(0, 3)) * int(rng.integers(0, chunk))
            data = (oracle.gen_zipf(n, K=256, s=float(rng.uniform(0.3, 2.5)), seed=int(rng.integers(1, 1 << 30))) if kind != 1
                    else rng.integers(0, 256, n).astype(np.uint8))
            dt = np.uint8
        # kernel-family options of the context (every setting writes the same bytes): lane-kernel generation, the lane
        # encoders placing their chunks themselves (scanner wave per block), wave encoders with / without fused placement,
        # byte-stream decoders with two chunks per wave or one
        lanes = "auto"  # (the option that pinned a lane-kernel generation was retired in round 6: rng draw kept for the seed's sake)
        rng.choice(["staged", "regwin", "auto"])
        lanes_fused, wave_fused, dual = int(rng.integers(0, 2)), int(rng.integers(0, 4) != 0), int(rng.integers(0, 3))
        ctx.set_option(R.OPT_LANE_FUSED_PLACEMENT, lanes_fused)
        ctx.set_option(R.OPT_FUSED_PLACEMENT, wave_fused)
        ctx.set_option(R.OPT_DUAL_DECODE, dual)
        ctx.set_option(R.OPT_ENC_SCRATCH_RING, int(rng.integers(0, 2)))
        desc = dict(case=case, fmt=fmt, sb=sb, K=K, n=n, n_ways=n_ways, chunk=chunk, kind=kind, lanes=lanes,
                    lanes_fused=lanes_fused, wave_fused=wave_fused, dual=dual)
        try:
            counts = oracle.count_freqs(data, K)
            f, _ = oracle.normalize(counts, 1 << sb)
            if (1 << sb) < K:
                continue
            om = oracle.model(f, sb, with_alias=(fmt == FMT_ALIAS))
            gm = ctx.model(fmt, f, sb)
            want, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
            shift = int(rng.choice([0, 0, 1, 4, 16]))
            tdt = torch.uint8 if K <= 256 else torch.int16
            backing = torch.zeros(n + 64, dtype=tdt, device="cuda")
            d_syms = backing[shift:shift + n]
            d_syms.copy_(torch.from_numpy(data if K <= 256 else data.view(np.int16)))
            cont, d_offs, d_lens, total = ctx.encode(gm, d_syms, n_ways, chunk)
            ok = total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens) and \
                np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs)
            if ok:
                got = cont[:total].cpu().numpy()
                for c in range(len(lens)):
                    o, ln = int(offs[c]), int(lens[c])
                    if not np.array_equal(got[o:o + ln], want[o:o + ln]):
                        ok = False
                        desc["bad_chunk"] = c
                        break
            d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
            ob = torch.zeros(n + 64, dtype=tdt, device="cuda")
            oshift = int(rng.choice([0, 0, 1, 4, 16]))
            out = ctx.decode(gm, d_cont, want.size, torch.from_numpy(offs.astype(np.int64)).cuda(),
                             torch.from_numpy(lens.astype(np.int32)).cuda(), n, n_ways, chunk, d_out=ob[oshift:oshift + n])
            ok_dec = np.array_equal(ob[oshift:oshift + n].cpu().numpy().view(dt), data) and ctx.decode_errors() == 0
            ok_dec = ok_dec and int(ob[:oshift].to(torch.int64).sum()) == 0 and int(ob[oshift + n:].to(torch.int64).sum()) == 0
            if ok:
                out2 = ctx.decode(gm, cont, total, d_offs, d_lens, n, n_ways, chunk)
                ok_dec = ok_dec and np.array_equal(out2.cpu().numpy().view(dt), data)
            # round 4: the slot layout (every chunk == the oracle's stream, at the end of its slot), the decoders on it,
            # its compaction == the compact container, and a chunk RANGE of it decoded on its own
            ok_slots = True
            if ok:
                nchunks = len(lens)
                s_cont, s_offs, s_lens, s_total = ctx.encode_slots(gm, d_syms, n_ways, chunk)
                slot = R.slot_bytes(fmt, n, n_ways, chunk)
                so = s_offs.cpu().numpy().astype(np.int64)
                ok_slots = s_total == nchunks * slot and np.array_equal(s_lens.cpu().numpy().astype(np.uint32), lens) and \
                    np.array_equal(so[:nchunks], (np.arange(nchunks, dtype=np.int64) + 1) * slot - lens.astype(np.int64))
                if ok_slots:
                    sg = s_cont.cpu().numpy()
                    for c in range(nchunks):
                        a, o, ln = int(so[c]), int(offs[c]), int(lens[c])
                        if not np.array_equal(sg[a:a + ln], want[o:o + ln]):
                            ok_slots = False
                            desc["bad_slot_chunk"] = c
                            break
                if ok_slots:
                    out3 = ctx.decode(gm, s_cont, s_total, s_offs, s_lens, n, n_ways, chunk)
                    ok_slots = np.array_equal(out3.cpu().numpy().view(dt), data)
                if ok_slots:
                    c_cont, c_offs, c_total = ctx.compact(s_cont, s_total, s_offs, s_lens, nchunks)
                    ok_slots = c_total == want.size and np.array_equal(c_offs.cpu().numpy().astype(np.uint64), offs)
                    cg = c_cont[:c_total].cpu().numpy()
                    for c in range(nchunks):
                        o, ln = int(offs[c]), int(lens[c])
                        if ok_slots and not np.array_equal(cg[o:o + ln], want[o:o + ln]):
                            ok_slots = False
                            desc["bad_compacted_chunk"] = c
                if ok_slots and nchunks >= 3:  # chunks [lo, hi) of the slot container through rans_amd_decode's pointer arithmetic
                    lo = int(rng.integers(0, nchunks - 1))
                    hi = int(rng.integers(lo + 1, nchunks + 1))
                    n_range = min(n, hi * chunk) - lo * chunk
                    part = ctx.decode(gm, s_cont, s_total, s_offs[lo:], s_lens[lo:], n_range, n_ways, chunk)
                    ok_slots = np.array_equal(part.cpu().numpy().view(dt), data[lo * chunk:lo * chunk + n_range])
                    desc["range"] = (lo, hi)
            # round 5: SIZED slots -- the model's tight slot, or a random multiple of 64 bytes between one line and the worst
            # case (most of those overflow some or all chunks into the region behind the slots): every chunk == the
            # oracle's stream wherever it lies, the index follows the layout's rule, the container decodes as it is
            ok_sized = True
            if ok and ok_slots:
                nchunks = len(lens)
                worst = R.slot_bytes(fmt, n, n_ways, chunk)
                pick_slot = int(rng.integers(0, 3))
                slot_t = None if pick_slot == 0 else 64 * int(rng.integers(1, max(2, worst // 64 + 1)))
                t_cont, t_offs, t_lens, t_total, slot_t = ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot_t, overflow_chunks=nchunks)
                desc["sized_slot"] = (slot_t, worst)
                to = t_offs.cpu().numpy().astype(np.int64)
                eff = min(slot_t, worst)
                ok_sized = np.array_equal(t_lens.cpu().numpy().astype(np.uint32), lens)
                if ok_sized:
                    ends = to[:nchunks] + lens.astype(np.int64)
                    inside = to[:nchunks] < nchunks * eff
                    over = int((~inside).sum())
                    k = ends[~inside] - nchunks * eff
                    ok_sized = np.array_equal(ends[inside], (np.nonzero(inside)[0] + 1) * eff) and np.all(k % worst == 0) and \
                        sorted((k // worst).tolist()) == list(range(1, over + 1)) and int(to[nchunks]) == t_total == nchunks * eff + over * worst and \
                        bool(np.all(lens[inside] <= eff))
                    desc["overflowed"] = over
                if ok_sized:
                    tg = t_cont.cpu().numpy()
                    for c in range(nchunks):
                        a, o, ln = int(to[c]), int(offs[c]), int(lens[c])
                        if not np.array_equal(tg[a:a + ln], want[o:o + ln]):
                            ok_sized = False
                            desc["bad_sized_chunk"] = c
                            break
                if ok_sized:
                    out4 = ctx.decode(gm, t_cont, t_total, t_offs, t_lens, n, n_ways, chunk)
                    ok_sized = np.array_equal(out4.cpu().numpy().view(dt), data) and ctx.decode_errors() == 0
            if not (ok and ok_dec and ok_slots and ok_sized):
                fails += 1
                print("FAIL", desc, "encode_ok", ok, "decode_ok", ok_dec, "slots_ok", ok_slots, "sized_ok", ok_sized, flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("EXC", desc, repr(e), flush=True)
    for opt, val in ((R.OPT_LANE_KERNELS, 0), (R.OPT_LANE_FUSED_PLACEMENT, 0), (R.OPT_FUSED_PLACEMENT, 1), (R.OPT_DUAL_DECODE, 1),
                     (R.OPT_ENC_SCRATCH_RING, 0)):
        ctx.set_option(opt, val)
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="mix in inputs of 3-7 M symbols")
    a = ap.parse_args()
    fails = run(a.cases, a.seed, big=a.big)
    print("cases %d, failures %d" % (a.cases, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
