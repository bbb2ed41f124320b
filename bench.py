#!/usr/bin/env python3
"""bench.py -- headline benchmark: 64-way interleaved rANS decode (word format) of
synthetic order-0 byte streams, one 1 GiB shard per GPU (BASELINE.json configs[2] / [4]), 32 Ki-symbol chunks
(the `configs` entries: 16 Ki, --config-chunk).

  python bench.py                                  # 1 GPU
  python bench.py --gpus N --steps K --warmup W    # N GPUs of this node: spawns its own N ranks (torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W     # ... or is launched as N ranks by somebody else

A "step" = one pass of the decode hot path over the whole shard (device-resident
container -> device-resident symbols).  Per rank: generate Zipf(256, s=1) bytes on
the GPU (SURVEY 8(d) generator: splitmix64 + inverse CDF, seed = rank+1 -- the same
bytes tests/_oracle.py's gen_zipf makes on the CPU), build the order-0 model (GPU
histogram + exact normalize_freqs), encode with the GPU encoder (setup, untimed), then
W warm-up and K timed decodes.  Shards are independent: no collective on the data
path; RCCL only carries the barriers and the 48-byte per-rank result record (weak scaling).

Rank 0 prints ONE SHORT JSON line (<= 3.8 KB, the last line of stdout: judged_line()) and writes the full record --
probe matrices, per-thread CPU sweeps, per-config timings -- to bench_details.json (--details).  `value` = decoded
(uncompressed) GB/s of the whole job.  `roofline` = algorithmic bytes (compressed stream read + symbols written) of one
decode launch / its average duration measured with HIP events on the launch stream, vs the 8 TB/s HBM peak.  `clocks` =
shader clock and per-wave clocks per 64-symbol round measured by the kernel itself in one extra instrumented launch.
`configs` (N=1 only) = one row per other BASELINE configuration: decode, encode (compact layout, rans_amd_encode), the
write-once encoders, the reference's own CPU loop of that format on this box, and whether EVERY chunk equalled the oracle's.
`cpu_baseline` (rank 0, every N) = the reference's own fastest decoder (SSE4.1 8-way, oracle/_ref) on this box's host cores
over a bounded sample of the same data (`port_value`: this repo's AVX-512 host decoder, extra).  At N=1 the CPU leg
re-encodes EVERY chunk of every container with the oracle and compares the bytes; at N>1 every rank does that for a
256-chunk sample of its own shard and the counts are summed over the gathered records.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0  # measured float4 streaming copy (same guide)
MAX_CLOCK_HZ = 2.4e9
MAX_LINE_BYTES = 3800   # the judged stdout line (tests/test_bench_cpu.py pins it; the driver keeps ~8 KB of stdout)
PLACEMENT_STRIDE = [24 << 30]  # bytes between placement candidates in allocation order (--placement-stride-gib)
WAVES_PER_SIMD = 8      # resident waves per SIMD of the wave-per-chunk decoder (2 blocks x 16 waves per CU)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=30, help="symbols per GPU = 2^log2n (default 1 GiB)")
    ap.add_argument("--ways", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=32768,
                    help="symbols per independent chunk stream of the HEADLINE workload (32 Ki as in rounds 1-2 and BASELINE; "
                         "round 3 quoted 16 Ki, which the placement lottery of its boxes favoured: with the placement probe in "
                         "front, five fresh processes per size on two boxes put 32 Ki 1.9 %% ahead -- 0.379-0.383 against "
                         "0.386-0.390 ms -- for 0.8 %% fewer stream bytes; profiles/r04_chunk_sweep.log, DESIGN 5)")
    ap.add_argument("--config-chunk", type=int, default=16384,
                    help="symbols per chunk of the `configs` entries (16 Ki: config 4's and the byte format's optimum)")
    ap.add_argument("--format", default="word", choices=["word", "byte", "r64", "alias"])
    ap.add_argument("--scale-bits", type=int, default=0, help="probability bits of the model (default: the format's -- word 12, "
                                                               "byte / r64 14, alias 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (reference timing + oracle checks)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="setup: run the decoder this long before the warm-up steps (clocks settle); 0 = off")
    ap.add_argument("--dump-launch-ms", action="store_true", help="add the per-launch HIP-event times to the JSON line")
    ap.add_argument("--measure", action="store_true",
                    help="measurement run: RANS_AMD_* environment variables (another library build, the experiment knobs "
                         "of the -DRANS_AMD_MEASURE build) and the --debug-* aids are allowed; the line says so "
                         "(\"headline\": false) and is not a benchmark result")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (and run its barriers and the record all-gather) at world size 1 as "
                         "well: the RCCL path of the N-GPU run, exercised on one GPU")
    ap.add_argument("--oracle-sample", type=int, default=0, metavar="CHUNKS",
                    help="CPU leg: compare only this many sampled chunks per container with the oracle (default 0: EVERY "
                         "chunk of every container, threaded over the host cores)")
    ap.add_argument("--debug-out-offset", type=int, default=0, help="measurement aid: shift the output buffer (bytes, multiple of 16)")
    ap.add_argument("--debug-cont-offset", type=int, default=0, help="measurement aid: shift the container (bytes, multiple of 16)")
    ap.add_argument("--debug-same-chunk", type=int, default=0, metavar="K",
                    help="measurement aid: index entry c points at chunk c mod K (stream reads come from cache); the "
                         "round trip check is skipped")
    ap.add_argument("--config-steps", type=int, default=20, help="back-to-back launches per `configs` entry")
    ap.add_argument("--placement-candidates", type=int, default=8, metavar="K",
                    help="setup (untimed): allocate K candidate output buffers (and 3 copies of the container), time the decode "
                         "on every pair for a few launches and keep the fastest pair -- MI355X memory comes in two classes and "
                         "a kernel that streams one buffer in and another out is 4-6 %% faster when the two lie in different "
                         "ones (profiles/r04_allocation.md); 1 = take what the allocator returns first (rounds 1-3)")
    ap.add_argument("--placement-stride-gib", type=int, default=24, metavar="G",
                    help="allocate the placement candidates G GiB apart in allocation order (spacer allocations held during the "
                         "probe, never touched; 0 = one after the other, what round 4 did): a class of device memory is a window of "
                         "30-60 GiB of allocation order (profiles/r05_class_map.md)")
    ap.add_argument("--placement-spread", type=float, default=1.02, metavar="R",
                    help="setup (untimed): when the slowest probed pair is within this factor of the fastest, every candidate "
                         "lies in one class of memory -- keep them and allocate another round of candidates (twice at most)")
    ap.add_argument("--details", default=None, metavar="PATH",
                    help="where rank 0 writes the full record (every probe matrix, per-thread CPU sweep, per-config timing); "
                         "default bench_details.json beside this script.  stdout carries only the short judged line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for dry runs)")
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="dry-run aid: every rank uses this GPU (needs --backend gloo)")
    return ap.parse_args()


# ---- synthetic input: SURVEY.md 8(d) ---------------------------------------------------------

def zipf_cdf(K, s):
    """Running sums of 1/(k+1)^s in double precision, summed in index order (oracle/rans_oracle.c:580-587)."""
    run, cdf = 0.0, []
    for k in range(K):
        run += 1.0 / math.pow(float(k + 1), s)
        cdf.append(run)
    return cdf, run


def gen_zipf(torch, n, K, s, seed, device):
    """Zipf(K, s) symbols, bit-identical to tests/_oracle.py Oracle.gen_zipf(n, K, s, seed): splitmix64 is
    counter based (state_i = seed + (i + 1) * golden), u = (z >> 11) * 2^-53 * Z, symbol = first k with
    cdf[k] > u.  uint8 for K <= 256, else int16 holding the u16 symbol."""
    cdf_list, run = zipf_cdf(K, s)
    cdf = torch.tensor(cdf_list, dtype=torch.float64, device=device)
    out = torch.empty(n, dtype=torch.uint8 if K <= 256 else torch.int16, device=device)

    def as_i64(v):  # 64-bit constant as the signed value torch.int64 holds
        return v - (1 << 64) if v >= (1 << 63) else v

    golden, c1, c2 = as_i64(0x9E3779B97F4A7C15), as_i64(0xBF58476D1CE4E5B9), as_i64(0x94D049BB133111EB)

    def lsr(z, k):  # logical shift right of an int64 tensor
        return (z >> k) & ((1 << (64 - k)) - 1)

    step = min(n, 1 << 25)
    for i in range(0, n, step):
        m = min(step, n - i)
        idx = torch.arange(i + 1, i + 1 + m, dtype=torch.int64, device=device)
        z = idx * golden + as_i64(seed & ((1 << 64) - 1))  # wraps modulo 2^64 like the C code
        z = (z ^ lsr(z, 30)) * c1
        z = (z ^ lsr(z, 27)) * c2
        z = z ^ lsr(z, 31)
        u = lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0) * run
        sym = torch.searchsorted(cdf, u, right=True).clamp_(max=K - 1)
        out[i:i + m] = sym.to(out.dtype)
    return out


def gen_moving(torch, n, seed, device, group=1 << 22):
    """Bytes whose statistics MOVE (the per-chunk-model rows of `configs`, tests/test_gpu_scale.py): groups of `group`
    symbols, group g drawn from Zipf(K_g, s_g) with K_g in {256, 64, 16, 4} and s_g in {0.5, 1.0, 1.5, 2.0} (gen_zipf: the
    oracle's generator, seed + g) and rotated by 37 g modulo 256 -- a global model fits none of them."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    for g, lo in enumerate(range(0, n, group)):
        m = min(group, n - lo)
        part = gen_zipf(torch, m, (256, 64, 16, 4)[g % 4], (0.5, 1.0, 1.5, 2.0)[(g // 4) % 4], seed + g, device)
        out[lo:lo + m] = ((part.to(torch.int32) + 37 * g) % 256).to(torch.uint8)
    return out


# ---- timing helpers ----------------------------------------------------------------------------

def timed_launches(torch, fn, steps, warmup):
    """Mean / min milliseconds of `steps` back-to-back launches of fn (HIP events on the launch stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t_pre = time.perf_counter()  # the same steady-state rule as the headline: ~60 ms of this very work first
    while (time.perf_counter() - t_pre) * 1e3 < 60.0:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for k in range(steps):
        ev0[k].record()
        fn()
        ev1[k].record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    return sum(ms) / len(ms), min(ms)


def settle(torch, fn, ms=60.0):
    """Run fn back to back for `ms` milliseconds: the clocks of an idle GPU need that long to settle, and a probe that compares
    allocations must not compare a cold launch with a warm one."""
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < ms:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def trace(*a):
    """BENCH_TRACE=1: progress on stderr (which stage a failing run had reached)."""
    if os.environ.get("BENCH_TRACE"):
        print("[bench]", *a, file=sys.stderr, flush=True)


def measure_config(torch, R, ctx, short, name, fmt, sb, K, ways, chunk, log2n, seed, steps, device, d_syms=None, probe=1):
    """One `configs` entry: decode and encode of a BASELINE configuration, `steps` back-to-back launches each,
    round trip verified.  Returns (entry, artefacts for the CPU-side oracle check).  probe > 1: every timed call first
    chooses the buffer it WRITES among `probe` candidate allocations -- the decode the (container, output) pair among two
    copies of the container and `probe` outputs (ryg_rans_amd/placement.py; untimed)."""
    from ryg_rans_amd.placement import choose_one, spaced
    n = 1 << log2n
    sym_bytes = 1 if K <= 256 else 2
    trace("config", short, "n", n)
    if d_syms is None:
        d_syms = gen_zipf(torch, n, K, 1.0, seed, device)
    counts = ctx.count_freqs_device(d_syms, K)
    freqs, _ = R.normalize_freqs(counts, 1 << sb)
    model = ctx.model(fmt, freqs, sb)
    cont, offs, lens, total = ctx.encode(model, d_syms, ways, chunk)
    out = torch.empty_like(d_syms)
    placement = {"candidates": probe}
    cont_dec = cont
    if probe > 1:
        # the decode reads one buffer and writes another: the PAIR decides (two memory classes, profiles/r04_allocation.md),
        # so a second copy of the container is a candidate as well
        from ryg_rans_amd.placement import choose_pair
        outs, sp = spaced(torch, lambda: torch.empty_like(d_syms), probe, device, first=out, stride_bytes=PLACEMENT_STRIDE[0])
        conts = [cont, cont.clone()]  # (the copy: at the far end of the spaced outputs)
        settle(torch, lambda: ctx.decode(model, cont, total, offs, lens, n, ways, chunk, d_out=out, sync=False))
        ci, oi, matrix = choose_pair(torch, lambda c, o: ctx.decode(model, c, total, offs, lens, n, ways, chunk, d_out=o, sync=False),
                                     conts, outs)
        cont_dec, out = conts[ci], outs[oi]
        placement["decode_probe_ms"] = [[round(v, 4) for v in row] for row in matrix]
        placement["decode_chosen"] = [ci, oi]
        del conts, outs, sp
        torch.cuda.empty_cache()
    trace(" decode")
    dec_ms, dec_min = timed_launches(
        torch, lambda: ctx.decode(model, cont_dec, total, offs, lens, n, ways, chunk, d_out=out, sync=False), steps, 2)
    bad = ctx.decode_errors()
    exact = bool(torch.equal(out, d_syms)) and bad == 0
    kernel = ctx.last_decode_kernel()
    del cont_dec
    trace(" encode (compact)")
    cont2, offs2, lens2 = torch.empty_like(cont), torch.empty_like(offs), torch.empty_like(lens)
    if probe > 1:
        # (the compact encoders are the most placement-sensitive calls: 10-12 %)
        c2s, sp = spaced(torch, lambda: torch.empty_like(cont), 2 * probe, device, first=cont2, stride_bytes=PLACEMENT_STRIDE[0])
        settle(torch, lambda: ctx.encode(model, d_syms, ways, chunk, d_out=cont2, sync=False, d_offsets=offs2, d_lengths=lens2))
        pick, ms = choose_one(torch, lambda c: ctx.encode(model, d_syms, ways, chunk, d_out=c, sync=False, d_offsets=offs2,
                                                          d_lengths=lens2), c2s)
        cont2 = c2s[pick]
        placement["encode_compact_probe_ms"] = [round(v, 4) for v in ms]
        del c2s, sp
        torch.cuda.empty_cache()
    enc_ms, enc_min = timed_launches(
        torch, lambda: ctx.encode(model, d_syms, ways, chunk, d_out=cont2, sync=False, d_offsets=offs2, d_lengths=lens2),
        steps, 2)
    enc_kernel, enc_fused = ctx.last_encode_kernel()
    enc_kernels = enc_kernel + (" (places its chunks itself)" if enc_fused else " + k_layout + k_compact*")
    # (bytes between chunks are alignment padding nobody writes: compare index and sizes, not the raw buffers)
    same_container = bool(torch.equal(offs2, offs)) and bool(torch.equal(lens2, lens))
    del cont2
    # slot layout (rans_amd_encode_slots): every chunk written once, where it was coded -- what the reference does with
    # each of its buffers (main.cpp:176-188).  Encoder timed, the slot container decoded as it is (timed as well: its
    # chunk starts are not 16-byte aligned and it is 2.6 x as large), and every chunk of it goes to the oracle check.
    trace(" encode_slots")
    s_cont, s_offs, s_lens, s_total = ctx.encode_slots(model, d_syms, ways, chunk)
    slot = R.slot_bytes(fmt, n, ways, chunk)
    if probe > 1:
        scs, sp = spaced(torch, lambda: torch.empty_like(s_cont), min(probe, 4), device, first=s_cont, stride_bytes=PLACEMENT_STRIDE[0])
        settle(torch, lambda: ctx.encode_slots(model, d_syms, ways, chunk, d_out=s_cont, sync=False, d_offsets=s_offs, d_lengths=s_lens))
        pick, ms = choose_one(torch, lambda c: ctx.encode_slots(model, d_syms, ways, chunk, d_out=c, sync=False, d_offsets=s_offs,
                                                                d_lengths=s_lens), scs)
        s_cont = scs[pick]
        placement["encode_slots_probe_ms"] = [round(v, 4) for v in ms]
        del scs, sp
        torch.cuda.empty_cache()
    s_enc_ms, s_enc_min = timed_launches(
        torch, lambda: ctx.encode_slots(model, d_syms, ways, chunk, d_out=s_cont, sync=False, d_offsets=s_offs, d_lengths=s_lens),
        steps, 2)
    ctx.encode_status()
    s_kernel = ctx.last_encode_kernel()[0]
    slots_ok = ctx.last_encode_placement() == 2 and bool(torch.equal(s_lens, lens))
    trace(" decode_slots")
    out.zero_()
    s_dec_ms, s_dec_min = timed_launches(
        torch, lambda: ctx.decode(model, s_cont, s_total, s_offs, s_lens, n, ways, chunk, d_out=out, sync=False), steps, 2)
    slots_ok = slots_ok and ctx.decode_errors() == 0 and bool(torch.equal(out, d_syms))
    # sized slots (rans_amd_encode_slots_sized): one write per stream AND a container about the compact one's size -- slots
    # of rans_amd_tight_slot_bytes() (the model's expected chunk stream + 2 % + states + 4 sigma), chunks that do not fit
    # coded again behind them.  Encoder timed, the container decoded as it is (timed), every chunk goes to the oracle check.
    trace(" encode_sized")
    t_cont, t_offs, t_lens, t_total, t_slot = ctx.encode_sized(model, d_syms, ways, chunk)
    if probe > 1:
        tcs, sp = spaced(torch, lambda: torch.empty_like(t_cont), min(probe, 4), device, first=t_cont, stride_bytes=PLACEMENT_STRIDE[0])
        settle(torch, lambda: ctx.encode_sized(model, d_syms, ways, chunk, slot=t_slot, d_out=t_cont, sync=False, d_offsets=t_offs,
                                               d_lengths=t_lens))
        pick, ms = choose_one(torch, lambda c: ctx.encode_sized(model, d_syms, ways, chunk, slot=t_slot, d_out=c, sync=False,
                                                                d_offsets=t_offs, d_lengths=t_lens), tcs)
        t_cont = tcs[pick]
        placement["encode_tight_probe_ms"] = [round(v, 4) for v in ms]
        del tcs, sp
        torch.cuda.empty_cache()
    t_enc_ms, t_enc_min = timed_launches(
        torch, lambda: ctx.encode_sized(model, d_syms, ways, chunk, slot=t_slot, d_out=t_cont, sync=False, d_offsets=t_offs,
                                        d_lengths=t_lens), steps, 2)
    ctx.encode_status()
    t_kernel = ctx.last_encode_kernel()[0]
    t_total = int(t_offs[-1].item())
    nchunks = (n + chunk - 1) // chunk
    t_over = (t_total - nchunks * t_slot) // slot if t_slot < slot else 0
    tight_ok = bool(torch.equal(t_lens, lens))
    trace(" decode_tight")
    out.zero_()
    t_dec_ms, t_dec_min = timed_launches(
        torch, lambda: ctx.decode(model, t_cont, t_total, t_offs, t_lens, n, ways, chunk, d_out=out, sync=False), steps, 2)
    tight_ok = tight_ok and ctx.decode_errors() == 0 and bool(torch.equal(out, d_syms))
    alg = n * sym_bytes + total
    entry = {
        "short": short, "name": name, "format": R.FORMAT_NAMES[fmt], "scale_bits": sb, "alphabet": K, "n_ways": ways, "chunk_syms": chunk,
        "symbols": n, "decoded_bytes": n * sym_bytes, "stream_bytes": total,
        "algorithmic_bytes_per_launch": alg,
        "decode": {"kernel": kernel, "ms_mean": round(dec_ms, 4), "ms_min": round(dec_min, 4), "launches": steps,
                   "decoded_GBps": round(n * sym_bytes / dec_ms / 1e6, 1),
                   "achieved_GBps": round(alg / dec_ms / 1e6, 1), "frac": round(alg / dec_ms / 1e6 / HBM_PEAK_GBPS, 4)},
        # `encode`: the compact container of rans_amd_encode (chunk c at the sum of the aligned lengths before it) -- the key
        # rounds 1-3 reported under this name; `encode_slots`: the slot layout of rans_amd_encode_slots (one write per
        # stream, as the reference writes its buffers, main.cpp:176-188; a larger container)
        "encode": {"layout": "compact (rans_amd_encode)", "kernels": enc_kernels, "ms_mean": round(enc_ms, 4),
                   "ms_min": round(enc_min, 4), "launches": steps, "input_GBps": round(n * sym_bytes / enc_ms / 1e6, 1),
                   "achieved_GBps": round(alg / enc_ms / 1e6, 1), "frac": round(alg / enc_ms / 1e6 / HBM_PEAK_GBPS, 4)},
        "encode_slots": {"layout": "slots (rans_amd_encode_slots: chunk c = the last lengths[c] bytes of slot c, written once)",
                         "kernels": s_kernel, "ms_mean": round(s_enc_ms, 4), "ms_min": round(s_enc_min, 4), "launches": steps,
                         "input_GBps": round(n * sym_bytes / s_enc_ms / 1e6, 1), "achieved_GBps": round(alg / s_enc_ms / 1e6, 1),
                         "frac": round(alg / s_enc_ms / 1e6 / HBM_PEAK_GBPS, 4), "slot_bytes": slot, "container_bytes": s_total},
        "decode_slots": {"kernel": ctx.last_decode_kernel(), "ms_mean": round(s_dec_ms, 4), "ms_min": round(s_dec_min, 4),
                         "launches": steps, "frac": round(alg / s_dec_ms / 1e6 / HBM_PEAK_GBPS, 4)},
        "encode_tight": {"layout": "sized slots (rans_amd_encode_slots_sized, slot = rans_amd_tight_slot_bytes(); a chunk that "
                                   "does not fit is coded again into a worst-case slot behind the sized ones)",
                         "kernels": t_kernel + " + the redo launch", "ms_mean": round(t_enc_ms, 4), "ms_min": round(t_enc_min, 4),
                         "launches": steps, "input_GBps": round(n * sym_bytes / t_enc_ms / 1e6, 1),
                         "achieved_GBps": round(alg / t_enc_ms / 1e6, 1), "frac": round(alg / t_enc_ms / 1e6 / HBM_PEAK_GBPS, 4),
                         "slot_bytes": t_slot, "container_bytes": t_total, "overflowed_chunks": int(t_over),
                         "container_over_input": round(t_total / (n * sym_bytes), 4),
                         "container_over_compact": round(t_total / total, 4)},
        "decode_tight": {"kernel": ctx.last_decode_kernel(), "ms_mean": round(t_dec_ms, 4), "ms_min": round(t_dec_min, 4),
                         "launches": steps, "frac": round(alg / t_dec_ms / 1e6 / HBM_PEAK_GBPS, 4)},
        "bit_exact_roundtrip": exact and same_container and slots_ok and tight_ok,
        # the written buffer of every timed call was chosen among this many allocations (setup, untimed; see --placement-candidates)
        "placement": placement,
    }
    art = {"fmt": fmt, "sb": sb, "K": K, "ways": ways, "chunk": chunk, "n": n, "freqs": freqs, "d_syms": d_syms,
           "cont": cont, "offs": offs, "lens": lens, "total": total, "entry": entry,
           "slots": {"cont": s_cont, "offs": s_offs, "lens": s_lens, "total": s_total, "slot": slot},
           "tight": {"cont": t_cont, "offs": t_offs, "lens": t_lens, "total": t_total, "slot": t_slot, "worst": slot}}
    return entry, art


def measure_adaptive(torch, R, ctx, short, name, fmt, sb, ways, chunk, d_syms, steps):
    """One `configs` entry for per-chunk models: rans_amd_encode_adaptive_sized -- count + normalise + records +
    code in ONE kernel, the piece of a chunk sized from its own histogram -- and rans_amd_decode_adaptive_fmt on the
    container it leaves; `steps` back-to-back launches each, round trip verified; the artefacts go to the CPU leg, which
    compares EVERY chunk's row and stream with the oracle's."""
    n = d_syms.numel()
    trace("config", short, "n", n)
    cont, offs, lens, rows, total = ctx.encode_adaptive_sized(d_syms, ways, chunk, sb, fmt=fmt)
    out = torch.empty_like(d_syms)
    enc_ms, enc_min = timed_launches(
        torch, lambda: ctx.encode_adaptive_sized(d_syms, ways, chunk, sb, fmt=fmt, d_out=cont, d_offsets=offs, d_lengths=lens,
                                                 d_freqs=rows, sync=False), steps, 2)
    ctx.encode_status()
    enc_kernel = ctx.last_encode_kernel()[0]
    total = int(offs[-1].item())
    dec_ms, dec_min = timed_launches(
        torch, lambda: ctx.decode_adaptive(cont, total, offs, lens, rows, n, ways, chunk, sb, d_out=out, sync=False, fmt=fmt), steps, 2)
    exact = ctx.decode_errors() == 0 and bool(torch.equal(out, d_syms))
    stream = int(lens.to(torch.int64).sum().item())
    nchunks = (n + chunk - 1) // chunk
    alg = n + stream + nchunks * 512  # symbols + streams + frequency rows, each crossing HBM once
    entry = {
        "short": short, "name": name, "format": R.FORMAT_NAMES[fmt], "scale_bits": sb, "alphabet": 256, "n_ways": ways, "chunk_syms": chunk,
        "symbols": n, "decoded_bytes": n, "stream_bytes": stream, "row_bytes": nchunks * 512, "algorithmic_bytes_per_launch": alg,
        "decode": {"kernel": ctx.last_decode_kernel(), "ms_mean": round(dec_ms, 4), "ms_min": round(dec_min, 4), "launches": steps,
                   "decoded_GBps": round(n / dec_ms / 1e6, 1), "achieved_GBps": round(alg / dec_ms / 1e6, 1),
                   "frac": round(alg / dec_ms / 1e6 / HBM_PEAK_GBPS, 4)},
        "encode": {"layout": "pieces sized from each chunk's own histogram, in index order (rans_amd_encode_adaptive_sized: one kernel)",
                   "kernels": enc_kernel, "ms_mean": round(enc_ms, 4), "ms_min": round(enc_min, 4), "launches": steps,
                   "input_GBps": round(n / enc_ms / 1e6, 1), "achieved_GBps": round(alg / enc_ms / 1e6, 1),
                   "frac": round(alg / enc_ms / 1e6 / HBM_PEAK_GBPS, 4), "container_bytes": total,
                   "container_over_input": round(total / n, 4), "container_over_streams": round(total / max(1, stream), 4)},
        "bit_exact_roundtrip": exact, "per_chunk_models": True,
    }
    art = {"adaptive": True, "fmt": fmt, "sb": sb, "K": 256, "ways": ways, "chunk": chunk, "n": n, "d_syms": d_syms, "cont": cont,
           "offs": offs, "lens": lens, "rows": rows, "total": total, "entry": entry}
    return entry, art


def oracle_check_adaptive(art):
    """Per-chunk models at the bench size: EVERY chunk's frequency row == the oracle's normalize(count(chunk)) and EVERY chunk's
    stream == the oracle's stream of the chunk under that model (main.cpp:139-162 with the chunk as the input; threaded,
    oracle/rans_oracle.c orc_compare_chunks_adaptive); pieces in index order, whole 64-byte lines.  Returns chunks compared."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from _oracle import Oracle
    orc = Oracle()
    n, chunk = art["n"], art["chunk"]
    nchunks = (n + chunk - 1) // chunk
    offs = art["offs"].cpu().numpy().astype(np.uint64)
    lens = art["lens"].cpu().numpy().astype(np.uint32)
    ends = offs[:nchunks] + lens[:nchunks]
    assert np.all(ends % np.uint64(64) == 0) and np.all(ends[:-1] <= offs[1:nchunks]) and int(ends[-1]) == int(offs[nchunks]) == art["total"], \
        "per-chunk-model container: pieces are not whole lines in index order"
    count, bad = orc.compare_container_adaptive(art["fmt"], art["d_syms"].cpu().numpy(), art["ways"], chunk, art["sb"],
                                                art["cont"][:art["total"]].cpu().numpy(), offs, lens, art["rows"].cpu().numpy())
    assert bad == -1, "per-chunk models: chunk %d: %s differs from the oracle's" % bad
    return count


# ---- CPU leg (rank 0, N = 1): the only place bench.py touches oracle/ ---------------------------------

def oracle_check_chunks(art, sample=0):
    """Parity pin at the BASELINE sizes: EVERY chunk of a GPU-made container is re-encoded by the CPU oracle (threads
    over the host cores, oracle/rans_oracle.c orc_compare_chunks) and compared byte for byte, and the whole index is
    checked against the prefix sums of its 16-byte aligned lengths -- the reference checks all bytes, not a sample
    (main_simd.cpp:340-343).  sample > 0: only that many chunks (first, last, the offset-scan block edges, random ones).
    Returns the number of chunks compared."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from _oracle import FMT_ALIAS, Oracle
    orc = Oracle()
    fmt, n, chunk, ways = art["fmt"], art["n"], art["chunk"], art["ways"]
    lens32 = art["lens"].cpu().numpy().astype(np.uint32)
    lens = lens32.astype(np.uint64)
    offs = art["offs"].cpu().numpy().astype(np.uint64)
    nchunks = (n + chunk - 1) // chunk
    if art.get("worst"):  # sized slots: a chunk ends at its slot's end, or at the end of a worst-case slot of the overflow region
        slot, worst = np.uint64(art["slot"]), np.uint64(art["worst"])
        ends = offs[:nchunks] + lens[:nchunks]
        region = np.uint64(nchunks) * slot
        inside = offs[:nchunks] < region
        assert np.array_equal(ends[inside], (np.nonzero(inside)[0].astype(np.uint64) + np.uint64(1)) * slot), \
            "a chunk of a sized slot does not end at its slot's end"
        k = ends[~inside] - region
        assert np.all(k % worst == 0) and sorted((k // worst).tolist()) == list(range(1, int((~inside).sum()) + 1)), \
            "overflowed chunks do not fill the overflow slots one each"
        assert int(offs[nchunks]) == int(region) + int((~inside).sum()) * int(worst)
        assert np.all(lens[:nchunks][inside] <= slot)
    elif art.get("slot"):  # slot layout (rans_amd_encode_slots): chunk c is the last lens[c] bytes of slot c
        slot = np.uint64(art["slot"])
        want_offs = np.zeros(nchunks + 1, dtype=np.uint64)
        want_offs[:nchunks] = (np.arange(nchunks, dtype=np.uint64) + np.uint64(1)) * slot - lens[:nchunks]
        want_offs[nchunks] = np.uint64(nchunks) * slot
        assert np.array_equal(offs, want_offs), "slot index differs from (c + 1) * slot - length"
    else:
        aligned = (lens + np.uint64(15)) & ~np.uint64(15)
        want_offs = np.zeros(nchunks + 1, dtype=np.uint64)
        want_offs[1:] = np.cumsum(aligned[:nchunks])
        want_offs[nchunks] = want_offs[nchunks - 1] + lens[nchunks - 1]
        assert np.array_equal(offs, want_offs), "chunk index differs from the prefix sums of its lengths"
    assert int(offs[nchunks]) == art["total"]
    om = orc.model(art["freqs"], art["sb"], with_alias=(fmt == FMT_ALIAS))
    syms, cont = art["d_syms"], art["cont"]
    if sample <= 0:
        h_syms = syms.cpu().numpy()
        if h_syms.dtype == np.int16:
            h_syms = h_syms.view(np.uint16)
        h_cont = cont[:art["total"]].cpu().numpy()
        count, first_bad = orc.compare_container(fmt, om, h_syms, ways, chunk, h_cont, offs, lens32)
        assert first_bad < 0, "chunk %d differs from the oracle's stream" % first_bad
        return count
    rng = np.random.default_rng(2024)
    picks = {0, 1, nchunks - 1, nchunks - 2, nchunks // 2}
    picks |= {c for c in (8191, 8192, 8193, 16383, 16384) if c < nchunks}
    picks |= set(int(c) for c in rng.choice(nchunks, size=min(sample, nchunks), replace=False))  # (distinct: the count is a promise)
    picks = sorted(c for c in picks if 0 <= c < nchunks)
    for c in picks:
        lo, hi = c * chunk, min(n, (c + 1) * chunk)
        h_syms = syms[lo:hi].cpu().numpy()
        if h_syms.dtype == np.int16:
            h_syms = h_syms.view(np.uint16)
        ref = orc.encode(fmt, om, h_syms, ways)
        a, ln = int(offs[c]), int(lens[c])
        got = cont[a:a + ln].cpu().numpy()
        assert ref.size == ln and np.array_equal(got, ref), "chunk %d differs from the oracle's stream" % c
    return len(picks)


def decode_oracle_container(torch, R, ctx, model, art, device):
    """The decoder fed a container the ORACLE made (threaded encode of the whole shard on the host): independent of
    the GPU encoder.  Returns True when the GPU decodes it to the input without a failed chunk."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from _oracle import FMT_ALIAS, Oracle
    orc = Oracle()
    fmt, n, chunk, ways = art["fmt"], art["n"], art["chunk"], art["ways"]
    h_syms = art["d_syms"].cpu().numpy()
    if h_syms.dtype == np.int16:
        h_syms = h_syms.view(np.uint16)
    om = orc.model(art["freqs"], art["sb"], with_alias=(fmt == FMT_ALIAS))
    cont, offs, lens = orc.encode_chunked_mt(fmt, om, h_syms, ways, chunk, align=16)
    d_cont = torch.empty(cont.size + 64, dtype=torch.uint8, device=device)
    d_cont[:cont.size] = torch.from_numpy(cont).to(device)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(device)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(device)
    out = torch.zeros_like(art["d_syms"])
    ctx.decode(model, d_cont, cont.size, d_offs, d_lens, n, ways, chunk, d_out=out, sync=False)
    bad = ctx.decode_errors()
    return bool(torch.equal(out, art["d_syms"])) and bad == 0 and cont.size == art["total"]


def cpu_quota_cores():
    """CPU time this process tree may use per wall second, in cores (cgroup v2 cpu.max / v1 cfs quota), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(d_syms, freqs, n):
    """CPU decode of the same data on this box's host cores, threads PINNED (one per physical core first, SMT siblings
    after): the reference's own fastest decoder -- SSE4.1, two 4-lane vectors on 8-way streams, main_simd.cpp:313-332
    through oracle/_ref -- and, where the host has AVX-512, the 16-lane decoder of
    include/ryg_rans_amd/compat/rans_word_avx512.h on 32-way streams (SURVEY 8(f)4's stronger CPU baseline).
    `value` is the fastest of all (decoder, thread count) pairs; the reference's own best is always listed beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from _oracle import FMT_WORD, HostSimd, Oracle, Ref

    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    # A container's CFS quota (cgroup cpu.max) can be far below the CPUs its affinity mask shows: threads beyond it are
    # throttled, not run -- round 2's sweep "fell" from 32 to 256 threads on a 256-CPU box for exactly that reason, pinned
    # or not.  The sweep stops at the quota; both numbers are reported.
    quota = cpu_quota_cores()
    max_threads = max(1, usable if quota is None else min(usable, int(math.ceil(quota))))
    shard = min(1 << 24, n // max_threads)       # 16 Mi symbols per shard: the first 256 MiB on 16 threads, as the `configs` rows
    shard -= shard % 32
    orc = Oracle()
    if not Ref.available() and not HostSimd.available():
        # port: the scalar C restatement, one thread, 64-way stream
        m = min(n, 1 << 26)
        host = d_syms[:m].cpu().numpy()
        om = orc.model(freqs, 12)
        stream = orc.encode(FMT_WORD, om, host, 64)
        t0 = time.perf_counter()
        out = orc.decode(FMT_WORD, om, stream, m, 64)
        dt = time.perf_counter() - t0
        assert np.array_equal(out, host)
        return {"value": m / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": "first %d MiB of rank 0's shard, oracle scalar C, 64-way word stream" % (m >> 20)}

    hs = HostSimd()
    ref = Ref() if Ref.available() else None
    physical = min(max_threads, hs.physical_cores())
    host = d_syms[:max_threads * shard].cpu().numpy()
    om = orc.model(freqs, 12)
    f32 = np.ascontiguousarray(freqs, dtype=np.uint32)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)

    def pack(ways):
        """every shard as its own `ways`-way word stream (the oracle's encoder: equal to the reference's, tested),
        back to back with 64 bytes of padding each"""
        with ThreadPoolExecutor(min(max_threads, 64)) as ex:
            streams = list(ex.map(lambda i: orc.encode(FMT_WORD, om, host[i * shard:(i + 1) * shard], ways),
                                  range(max_threads)))
        offsets = np.zeros(max_threads, dtype=np.uint64)
        pos = 0
        for i, st in enumerate(streams):
            offsets[i] = pos
            pos += (st.size + 64 + 15) & ~15
        blob = np.zeros(pos + 64, dtype=np.uint8)
        for i, st in enumerate(streams):
            blob[int(offsets[i]):int(offsets[i]) + st.size] = st
        return blob, offsets

    decoders = []
    if ref is not None:
        decoders.append(("reference SSE4.1, two 4-lane vectors, 8-way streams (main_simd.cpp:313-332 via oracle/_ref)",
                         "reference", 8, C.cast(ref.lib.ref_decode_word_simd8, C.c_void_p)))
    if hs.has_avx512():
        decoders.append(("AVX-512, two 16-lane vectors, 32-way streams (include/ryg_rans_amd/compat/rans_word_avx512.h)",
                         "port", 32, C.cast(hs.lib.host_decode_word_avx512x2, C.c_void_p)))
    cands = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512) if t < physical} | {physical, max_threads})
    out = np.zeros(max_threads * shard, dtype=np.uint8)
    report, best = [], None
    for name, kind, ways, fn in decoders:
        blob, offsets = pack(ways)

        def run(nthreads, nreps):
            """nthreads pinned pthreads, one shard each, nreps passes; seconds per pass"""
            return hs.lib.host_time_threads(fn, f32.ctypes.data_as(u32p), blob.ctypes.data_as(u8p),
                                            offsets.ctypes.data_as(u64p), nthreads, shard,
                                            out.ctypes.data_as(u8p), nthreads, nreps, 1) / nreps

        t1 = run(1, 2)                                       # calibrate: ~0.15 s of work per timed run
        reps = max(2, min(64, int(0.15 / max(t1, 1e-4))))
        sweep = {}
        for t in cands:
            sweep[t] = t * shard / min(run(t, reps) for _ in range(2)) / 1e9
        out[:] = 0
        run(max_threads, 1)
        assert np.array_equal(out, host), "CPU decode mismatch (%s)" % name
        bt = max(sweep, key=sweep.get)
        entry = {"decoder": name, "kind": kind, "n_ways": ways, "best_GBps": round(sweep[bt], 2), "best_threads": bt,
                 "single_thread_GBps": round(sweep[1], 3),
                 "thread_sweep_GBps": {str(k): round(v, 2) for k, v in sweep.items()}}
        report.append(entry)
        if best is None or sweep[bt] > best[0]:
            best = (sweep[bt], bt, entry)
    # `value` is the REFERENCE's number (its fastest thread count); a faster decoder of this repo's own (the AVX-512 port)
    # is listed beside it as port_value -- a stronger baseline is extra, not the baseline
    head = report[0] if ref is not None else best[2]
    res = {"value": head["best_GBps"], "unit": "GB/s", "cores": head["best_threads"], "kind": head["kind"],
           "decoder": head["decoder"],
           "sample": "%d x %d MiB shards from the start of rank 0's data, each its own N-way word stream, one PINNED "
                     "pthread per shard (physical cores first, then SMT siblings), thread counts %s swept for every "
                     "decoder" % (max_threads, shard >> 20, cands),
           "decoders": report, "host_cpus": cores, "usable_cpus": usable, "physical_cores": physical,
           "cpu_quota_cores": quota}
    for r in report[1:] if ref is not None else []:
        res["port_value"], res["port_cores"], res["port_decoder"] = r["best_GBps"], r["best_threads"], r["decoder"]
    if ref is not None:
        r0 = report[0]
        res["reference_value"] = r0["best_GBps"]
        res["reference_cores"] = r0["best_threads"]
        res["single_thread_value"] = r0["single_thread_GBps"]
        # clocks per symbol of the reference loop on one core, as main.cpp:171,184-186 measures them
        blob, offsets = pack(8)
        fn = C.cast(ref.lib.ref_decode_word_simd8, C.c_void_p)
        c0 = ref.lib.ref_rdtsc()
        hs.lib.host_time_threads(fn, f32.ctypes.data_as(u32p), blob.ctypes.data_as(u8p), offsets.ctypes.data_as(u64p), 1, shard,
                                 out.ctypes.data_as(u8p), 1, 4, 1)
        res["single_thread_clocks_per_symbol"] = (ref.lib.ref_rdtsc() - c0) / 4 / shard
    return res


_CPU_BASELINE_CACHE = {}


def config_cpu_baseline(art):
    """The reference's OWN loop of this configuration's format beside its GPU numbers (SURVEY 8(d) "CPU baseline"):
    rans_byte 2-way (main.cpp:226-246 / 259-280), rans64 2-way (main64.cpp:228-248 / 261-282), alias 2-way
    (main_alias.cpp:353-373 / 386-405; 4096 symbols: the same loop over the sed-widened model of oracle/_ref), word 8-way
    with its SSE4.1 decoder (main_simd.cpp:287-300 / 313-332) -- through oracle/_ref (the unmodified reference compiled
    here), encode AND decode, timed with timer() and __rdtsc() exactly where the mains put them.  Sample: the first
    256 MiB of the configuration's own symbols as 16 MiB shards; one shard on one pinned thread (the reference as shipped:
    clocks/symbol), then one shard per thread on as many pinned threads as the CPU quota allows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from _oracle import FMT_ALIAS, HostSimd, Ref
    if not Ref.available():
        return {"value": None, "kind": "reference", "sample": "oracle/_ref not built on this box"}
    ref = Ref()
    if not ref.has_loop2():
        return {"value": None, "kind": "reference", "sample": "oracle/_ref predates ref_time_loop2_mt"}
    fmt, K, sb, n = art["fmt"], art["K"], art["sb"], art["n"]
    which = 12 if (fmt == FMT_ALIAS and K == 4096) else fmt
    key = (which, K, sb, art["d_syms"].data_ptr())
    if key in _CPU_BASELINE_CACHE:  # (the wider interleaves of the word format share the headline's data and reference loop)
        return _CPU_BASELINE_CACHE[key]
    if K not in (256, 4096) or (K == 4096 and fmt != FMT_ALIAS):
        return {"value": None, "kind": "reference", "sample": "the reference has no loop for this alphabet"}
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota_cores()
    max_threads = max(1, usable if quota is None else min(usable, int(math.ceil(quota))))
    cpus = HostSimd().cpu_order() if HostSimd.available() else []
    n_per = 1 << 24
    max_threads = max(1, min(max_threads, n // n_per, 16))
    host = art["d_syms"][:max_threads * n_per].cpu().numpy()
    if host.dtype == np.int16:
        host = host.view(np.uint16)
    sym_bytes = host.dtype.itemsize
    one = ref.time_loop2(which, art["freqs"], sb, host, n_per, 1, cpus)[0]
    many = ref.time_loop2(which, art["freqs"], sb, host, n_per, max_threads, cpus)
    ok = one["ok"] and all(t["ok"] for t in many)
    enc_wall = max(t["enc_s"] for t in many)  # the passes start together (barrier): wall time = the slowest thread
    dec_wall = max(t["dec_s"] for t in many)
    loops = {0: "rans_byte.h 2-way, main.cpp:226-246 / 259-280", 1: "rans_word_sse41.h 8-way, scalar encode main_simd.cpp:287-300 / "
             "SSE4.1 decode main_simd.cpp:313-332", 2: "rans64.h 2-way, main64.cpp:228-248 / 261-282",
             3: "alias lookup over rans_byte.h 2-way, main_alias.cpp:353-373 / 386-405",
             12: "alias lookup 2-way, main_alias.cpp:353-373 / 386-405 with LOG2NSYMS 12 and u16 symbols (the sed edits of SURVEY 8(c))"}
    gb = n_per * sym_bytes / 1e9
    _CPU_BASELINE_CACHE[key] = res = {
        "kind": "reference", "unit": "GB/s (uncompressed)", "loop": loops[which], "decode_ok": ok,
        "value": round(max_threads * gb / dec_wall, 3), "cores": max_threads,
        "decode": {"single_thread_GBps": round(gb / one["dec_s"], 4), "single_thread_clocks_per_symbol": round(one["dec_clocks"] / n_per, 2),
                   "threads": max_threads, "all_threads_GBps": round(max_threads * gb / dec_wall, 3)},
        "encode": {"single_thread_GBps": round(gb / one["enc_s"], 4), "single_thread_clocks_per_symbol": round(one["enc_clocks"] / n_per, 2),
                   "threads": max_threads, "all_threads_GBps": round(max_threads * gb / enc_wall, 3)},
        "stream_bytes_per_symbol": round(one["stream_bytes"] / n_per, 5),
        "sample": "first %d MiB of this configuration's symbols as %d shards of 16 Mi symbols; one shard on one pinned thread, then one "
                  "shard per pinned thread on %d threads (CPU quota %s, %d usable CPUs)"
                  % ((max_threads * n_per * sym_bytes) >> 20, max_threads, max_threads, quota, usable),
    }
    return res


def judged_line(full, details_path=None):
    """The ONE line rank 0 prints (last line of stdout, <= MAX_LINE_BYTES): the contract's fields, `roofline`,
    `cpu_baseline` (value = the REFERENCE's decoder on this box), `clocks`, the placement probe's summary (no matrix), and
    one row per `configs` entry.  Everything else lives in the details file.  Pure function of the full record, so that
    tests/test_bench_cpu.py can pin its size on a canned record."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    c = full.get("config", {})
    line["config"] = {k: c.get(k) for k in ("workload", "format", "n_ways", "chunk_syms", "scale_bits", "symbols_per_gpu",
                                            "compressed_bytes_per_symbol")}
    line["bit_exact_roundtrip"] = full.get("bit_exact_roundtrip")
    line["headline"] = full.get("headline")
    if full.get("knobs"):
        line["knobs"] = sorted(full["knobs"])
    rl = full.get("roofline", {})
    # (`frac` = algorithmic bytes / the wall time of the K timed steps / peak: the clock `value` and `ms_per_step` use;
    #  `frac_kernel_events` = the same bytes over the HIP-event mean of those launches; SURVEY 8(d): against the 8 TB/s spec
    #  AND the 6.29 TB/s measured copy)
    line["roofline"] = {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel",
                                               "kernel_ms_avg", "frac_kernel_events", "algorithmic_bytes_per_launch", "frac_job",
                                               "wave_span_ms_avg", "frac_of_measured_copy")}
    if (full.get("n_gpus") or 1) == 1:
        line["roofline"].pop("frac_job", None)  # (one GPU: the job IS the launch)
    if line["roofline"].get("traffic_source"):
        line["roofline"]["traffic_source"] = line["roofline"]["traffic_source"].split(" ")[0]  # (the file; the method is in the details)
    pl = full.get("placement", {})
    if "probe_ms_chosen" in pl:  # the un-probed figure beside the chosen one: what two plain hipMallocs would have got
        n_sym = c.get("symbols_per_gpu") or 0
        line["placement"] = {"chosen_ms": pl["probe_ms_chosen"], "first_pair_ms": pl["probe_ms_first_pair"],
                             "min_ms": pl["probe_ms_min"], "max_ms": pl["probe_ms_max"],
                             "pairs": pl["candidates"]["containers"] * pl["candidates"]["outputs"],
                             "stride_gib": pl.get("stride_gib", 0)}
        if (full.get("n_gpus") or 1) == 1:
            # K timed steps on the first pair, the same loop and clock as `value` (the probe's own figures above are HIP-event
            # means of its few launches: another clock).  The probe chose the first pair itself: it IS the headline.
            fp = pl.get("first_pair_ms_per_step") or (full.get("ms_per_step") if pl.get("chosen") == [0, 0] else None)
            if fp:
                line["placement"]["first_pair_ms_per_step"] = fp
                line["value_first_pair"] = round(n_sym / fp / 1e6, 2)
                line["frac_first_pair"] = round(rl.get("algorithmic_bytes_per_launch", 0) / fp / 1e6 / HBM_PEAK_GBPS, 4)
    ck = full.get("clocks", {})
    if "error" not in ck:
        line["clocks"] = {k: ck.get(k) for k in ("sclk_hz_measured", "per_wave_clocks_per_round_of_64", "clocks_per_symbol_per_simd")}
    cb = full.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {"value": None if cb.get("value") is None else round(cb["value"], 3), "unit": cb.get("unit"),
                                "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": (cb.get("sample") or "")[:72]}
        for k in ("single_thread_value", "single_thread_clocks_per_symbol", "port_value", "port_cores"):
            if cb.get(k) is not None:
                line["cpu_baseline"][k] = round(cb[k], 3)
    for k in ("oracle_chunks_checked", "oracle_chunks_total", "decodes_oracle_container", "error", "oracle_check_error"):
        if k in full:
            line[k] = full[k]
    rows = []
    ref_cores = None
    for e in full.get("configs", []):
        if "error" in e:
            rows.append({"error": e["error"][:120]})
            continue
        cpu = e.get("cpu_baseline") or {}
        total = e.get("oracle_chunks_total")
        row = {"name": e["short"], "decode_ms": e["decode"]["ms_mean"], "decode_frac": e["decode"]["frac"]}
        if e.get("per_chunk_models"):  # one kernel, pieces sized from each chunk's own histogram: a sized layout by construction
            row["enc_tight_ms"] = e["encode"]["ms_mean"]
            row["enc_tight_frac"] = e["encode"]["frac"]
            row["tight_size"] = e["encode"]["container_over_input"]
        else:
            row["encode_ms"] = e["encode"]["ms_mean"]  # (the compact layout, rans_amd_encode)
            row["encode_frac"] = e["encode"]["frac"]
            if "encode_tight" in e:  # (sized slots, rans_amd_encode_slots_sized, and the decode of the container they leave)
                row["enc_tight_ms"] = e["encode_tight"]["ms_mean"]
                row["dec_tight_ms"] = e["decode_tight"]["ms_mean"]
                row["tight_size"] = e["encode_tight"]["container_over_input"]  # (sized container / input bytes)
        if cpu.get("value") is not None:
            row["cpu_GBps"] = cpu["value"]  # (the reference's own loop of this format on cpu_ref_cores host threads)
            ref_cores = cpu.get("cores")
        row["oracle_ok"] = bool(e.get("bit_exact_roundtrip")) and total is not None and \
            e.get("oracle_chunks_checked") == total and e.get("oracle_chunks_checked_slots", total) == total and \
            e.get("oracle_chunks_checked_tight", total) == total
        rows.append(row)
    if ref_cores is not None:
        line["cpu_ref_cores"] = ref_cores  # (threads of every row's cpu_ref_GBps: the reference's own loop of that format)
    if rows:
        line["configs"] = rows
    if full.get("per_rank") and (full.get("n_gpus") or 1) > 1:
        line["per_rank_kernel_ms"] = full["per_rank"]["kernel_ms"]
    line["details"] = details_path
    # the size is a guarantee, not a hope: shed the optional per-config keys, then whole rows, until the line fits
    for drop in ("tight_size", "dec_tight_ms", "cpu_GBps", "enc_tight_frac", "encode_frac", None):
        if len(json.dumps(line, separators=(",", ":"))) <= MAX_LINE_BYTES:
            break
        if drop is None:
            while line.get("configs") and len(json.dumps(line, separators=(",", ":"))) > MAX_LINE_BYTES - 40:
                line["configs"].pop()
                line["configs_truncated"] = True
        else:
            for row in line.get("configs", []):
                row.pop(drop, None)
    return line


def kernel_source_tag():
    """sha256 (first 16 hex digits) of the sources the decode kernel is built from: a committed PMC traffic
    measurement is quoted only when it was taken on this very kernel."""
    h = hashlib.sha256()
    for f in ("decode_wave.hip", "decode_common.hpp", "device_common.hpp", "kernels.h"):
        with open(os.path.join(ROOT, "ryg_rans_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def spawn_ranks(args, visible):
    """Re-run this script as N ranks of one node (torch.distributed.run, rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    n = args.gpus if args.all_on_device is not None else max(1, min(args.gpus, visible))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv, skip = [], False
    for a in sys.argv[1:]:  # the same arguments, --gpus replaced by the clamped count
        if skip:
            skip = False
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus="):
            continue
        argv.append(a)
    env = dict(os.environ, BENCH_GPUS_REQUESTED=str(args.gpus), BENCH_GPUS_VISIBLE=str(visible), BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n == 1:  # one visible GPU: the plain single-process run, the line carries the request
        return subprocess.call([sys.executable, os.path.abspath(__file__), "--gpus", "1"] + argv, env=env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n)] + argv
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import ryg_rans_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # A judged line must not be foolable: the shipped library reads no environment, but RANS_AMD_LIB can put another
    # build (the -DRANS_AMD_MEASURE one, whose knobs drop stores or skip copies) under this script.  Every RANS_AMD_*
    # variable seen is reported in the line, and without --measure any of them -- or a measure build -- stops the run.
    knobs = {k: v for k, v in sorted(os.environ.items()) if k.startswith("RANS_AMD_")}
    measure_build = bool(R.lib().rans_amd_build_flags() & 1) if hasattr(R.lib(), "rans_amd_build_flags") else True
    debug_args = bool(args.debug_out_offset or args.debug_cont_offset or args.debug_same_chunk)
    if (knobs or measure_build or debug_args) and not args.measure:
        sys.exit("bench.py: RANS_AMD_* variables %s / measure build %s / --debug-* %s: not a headline run (pass --measure to "
                 "run it as a measurement)" % (sorted(knobs), measure_build, debug_args))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` started plainly: spawn the N ranks ourselves -- one process per GPU under
        # torch.distributed.run on the loopback interface, exactly the command the docstring shows -- and hand its
        # output through.  N is clamped to the GPUs this node shows (the line says so: gpus_requested / gpus_visible);
        # --all-on-device (dry runs: every rank on one GPU, gloo for the records) is not clamped.
        sys.exit(spawn_ranks(args, torch.cuda.device_count()))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d inside a %d-rank launch: the launcher's --nproc-per-node and --gpus must agree"
                 % (args.gpus, world))
    gpu_index = local_rank if args.all_on_device is None else args.all_on_device
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:  # --force-dist: a one-rank group, rendezvous on the loopback interface
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.all_on_device is not None:  # (ranks that share a device share its memory: no spacers)
        args.placement_stride_gib = 0
    PLACEMENT_STRIDE[0] = args.placement_stride_gib << 30
    fmt = {"word": R.FMT_WORD, "byte": R.FMT_BYTE, "r64": R.FMT_R64, "alias": R.FMT_ALIAS}[args.format]
    sb = args.scale_bits or {"word": 12, "byte": 14, "r64": 14, "alias": 16}[args.format]
    n = 1 << args.log2n

    # ---- setup (untimed): data, model, GPU encode ------------------------------
    trace("setup")
    ctx = R.Context(gpu_index)
    # every decode call of this process is counted (one launch of the decode kernel each; the ABI's placement probe reports
    # its own): the record then says WHICH dispatches of the headline kernel were the timed ones, so that a kernel trace
    # of this command (tools/r05_profile.sh) can be cut to exactly them
    launches = {"decode": 0}
    _decode = ctx.decode

    def counted_decode(*a, **k):
        launches["decode"] += 1
        return _decode(*a, **k)
    ctx.decode = counted_decode
    d_syms = gen_zipf(torch, n, 256, 1.0, rank + 1, device)
    counts = ctx.count_freqs_device(d_syms, 256)
    freqs, _ = R.normalize_freqs(counts, 1 << sb)
    model = ctx.model(fmt, freqs, sb)
    cont, offs, lens, total = ctx.encode(model, d_syms, args.ways, args.chunk)
    out = torch.empty(n + args.debug_out_offset, dtype=torch.uint8, device=device)[args.debug_out_offset:]
    placement = {"candidates": 1}
    first_pair = None
    if args.placement_candidates > 1 and not (args.debug_out_offset or args.debug_cont_offset or args.debug_same_chunk):
        # Setup, untimed: where the buffers lie is worth 4-6 % on this part (two classes of device memory; container and
        # output in DIFFERENT classes is the fast case, profiles/r04_allocation.md), and which class an allocation gets is
        # the driver's choice.  Allocate candidates, let the clocks settle on the first pair, time every pair, keep the best.
        # (round 5, profiles/r05_class_map.md: a class is a WINDOW of 30-60 GiB of allocation order, so candidates allocated
        #  one after the other mostly share one -- they are allocated --placement-stride-gib apart, with copies of the
        #  container a third and two thirds of the way; the spacers in between are never touched and freed with the rest)
        from ryg_rans_amd.placement import spaced
        conts, made = [cont], [1]

        def next_out():
            if made[0] in (args.placement_candidates // 3, 2 * args.placement_candidates // 3):
                conts.append(cont.clone())
            made[0] += 1
            return torch.empty(n, dtype=torch.uint8, device=device)
        outs, spacers = spaced(torch, next_out, args.placement_candidates, device, first=out,
                               stride_bytes=args.placement_stride_gib << 30)
        while len(conts) < 3:
            conts.append(cont.clone())
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for _ in range(16):
                ctx.decode(model, cont, total, offs, lens, n, args.ways, args.chunk, d_out=out, sync=False)
            torch.cuda.synchronize()
        # (a process whose first candidates ALL lie in one class -- every pair within 2 % -- keeps them alive and allocates
        #  more, twice at most: the other class is a window of a few GiB somewhere in allocation order)
        extended = 0
        while True:
            # (the C ABI's own probe -- rans_amd_probe_placement, what a C++ caller uses: examples/multi_gpu.cpp)
            ci, oi, matrix = ctx.probe_placement(model, conts, total, offs, lens, n, args.ways, args.chunk, outs, launches=6, sweeps=2)
            launches["decode"] += len(conts) * len(outs) * 2 * (6 + 2)  # (pairs x sweeps x (launches + 2 warm-up launches))
            flat = [v for row in matrix for v in row]
            if max(flat) >= args.placement_spread * min(flat) or extended == 2:
                break
            extended += 1
            conts += [cont.clone() for _ in range(2)]
            more, sp = spaced(torch, lambda: torch.empty(n, dtype=torch.uint8, device=device), args.placement_candidates, device,
                              stride_bytes=args.placement_stride_gib << 30, reserve_bytes=24 << 30)
            outs += more
            spacers += sp
        first_pair = None if (ci, oi) == (0, 0) else (conts[0], outs[0])  # (what two plain allocations would have got: timed below)
        cont, out = conts[ci], outs[oi]
        placement = {"candidates": {"containers": len(conts), "outputs": len(outs), "extended": extended}, "chosen": [ci, oi],
                     "probe_ms_chosen": round(matrix[ci][oi], 4), "probe_ms_min": round(min(flat), 4),
                     "probe_ms_max": round(max(flat), 4), "probe_ms_first_pair": round(matrix[0][0], 4),
                     "probe_ms": [[round(v, 4) for v in row] for row in matrix],
                     "note": "setup, untimed: 6 launches x 2 sweeps per pair; the other candidates are freed before the timed region"}
        placement["stride_gib"] = args.placement_stride_gib if spacers else 0
        del conts, outs, spacers
        torch.cuda.empty_cache()
    if args.debug_cont_offset:
        moved = torch.empty(cont.numel() + args.debug_cont_offset, dtype=torch.uint8, device=device)[args.debug_cont_offset:]
        moved[:total] = cont[:total]
        cont = moved
    if args.debug_same_chunk:
        k = args.debug_same_chunk
        idx = torch.arange(lens.numel(), device=device) % k
        offs = torch.cat([offs[:-1][idx], offs[-1:]])
        lens = lens[idx].contiguous()
    torch.cuda.synchronize()

    trace("placement done", placement.get("chosen"))

    def step():
        ctx.decode(model, cont, total, offs, lens, n, args.ways, args.chunk, d_out=out, sync=False)

    def barrier():
        if use_dist:
            dist.barrier()

    # steady state first: after an idle period the GPU needs tens of milliseconds of load before its clocks settle
    # (the first 20 launches after setup run ~6 % slower than the 80 that follow), and a long-running decode job
    # lives in the settled state -- so part of setup is to run the decoder for a moment before the W warm-up steps
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    timed_first = launches["decode"]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()  # HIP events on the launch stream (torch's current stream)
        step()
        ev1[k].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()

    # first wave start .. last wave end of each timed launch (recorded by the kernel itself, always on)
    try:
        spans = ctx.launch_spans(min(args.steps, 32))
    except Exception:  # noqa: BLE001  (an older library build in an A/B run)
        spans = []
    # the same K steps on the FIRST pair of allocations, same loop, same clock (setup's probe is what chose another pair):
    # `value_first_pair` is what a caller without the probe gets
    first_pair_elapsed = None
    if first_pair is not None and world == 1:
        fc, fo = first_pair

        def first_step():
            ctx.decode(model, fc, total, offs, lens, n, args.ways, args.chunk, d_out=fo, sync=False)
        for _ in range(max(args.warmup, 3)):
            first_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            first_step()
        torch.cuda.synchronize()
        first_pair_elapsed = time.perf_counter() - t1
    del first_pair

    # ---- verification (after the timed region) -------------------------------------
    trace("timed region done")
    bad = ctx.decode_errors()
    exact = bool(torch.equal(out, d_syms))  # (always the real comparison: a --measure run with an output-dropping knob says false)
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps

    # N > 1: EVERY rank pins its own shard against the oracle before the records are gathered -- a 256-chunk sample (first,
    # last, the offset-scan block edges, random ones) re-encoded on the host and compared byte for byte; the count travels
    # in the record and rank 0 sums it.  (N = 1 checks every chunk of every container further down.)
    sampled, sample_ok = 0, True
    if world > 1 and not args.no_cpu_baseline:
        try:
            sampled = oracle_check_chunks({"fmt": fmt, "sb": sb, "K": 256, "ways": args.ways, "chunk": args.chunk, "n": n,
                                           "freqs": freqs, "d_syms": d_syms, "cont": cont, "offs": offs, "lens": lens,
                                           "total": total}, args.oracle_sample or 256)
        except AssertionError as e:
            print("rank %d: %s" % (rank, e), file=sys.stderr)
            sample_ok = False

    from ryg_rans_amd.sharding import ShardRecord, aggregate, gather_records
    rec = ShardRecord(elapsed, float(n), float(total), kernel_ms, 1.0 if (exact and bad == 0 and sample_ok) else 0.0, float(sampled))
    # the only payload RCCL carries: 48 bytes per rank
    records = gather_records(rec, device=device if args.backend == "nccl" else "cpu", force=args.force_dist)

    exit_code = 0
    text = ""
    if rank == 0:
        agg = aggregate(records, args.steps)
        all_ok = agg["all_ok"]
        ms_per_step = agg["ms_per_step"]
        value = agg["symbols_per_s"] / 1e9  # 1 byte per symbol
        k_s = kernel_ms * 1e-3
        # ONE clock on the line: `value`, `ms_per_step` and `roofline.frac` all come from the wall time of the K timed steps
        # (rank 0's own; at N = 1 that is ms_per_step); the HIP-event mean of the same launches is listed beside it
        step_s = records[0].elapsed_s / args.steps
        achieved = (n + total) / step_s / 1e9
        achieved_events = (n + total) / k_s / 1e9
        result = {
            "metric": "decode GB/s (uncompressed), %d-way interleaved rANS (%s format)" % (args.ways, args.format),
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": agg["n_ranks"],  # counted from the records the all-gather delivered, not from the environment
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if fmt != R.FMT_R64 else "u64",
            "data": "synthetic",
            "config": {
                "workload": "%s format, %d-way interleaved decode of %d MiB Zipf(256,s=1) bytes per GPU, "
                            "%d-symbol chunks, tables in LDS (BASELINE configs[%d])"
                            % (args.format, args.ways, n >> 20, args.chunk, 2 if world == 1 else 4),
                "format": args.format, "n_ways": args.ways, "chunk_syms": args.chunk, "scale_bits": sb,
                "symbols_per_gpu": n, "compressed_bytes_per_symbol": round(total / n, 5),
                "generator": "splitmix64 + inverse CDF (SURVEY 8(d)), seed = rank + 1",
                "sharding": "one independent shard per GPU, no data-path collective",
            },
            "bit_exact_roundtrip": all_ok,
            "headline": not args.measure,
            "knobs": knobs,
            "library": {"path": os.path.relpath(R.LIB_PATH, ROOT), "measure_build": measure_build},
            "distributed": {"initialised": use_dist, "backend": args.backend if use_dist else None,
                            "records_gathered_on": ("device (RCCL)" if args.backend == "nccl" else "host") if use_dist else None},
            "prewarm_ms": args.prewarm_ms,
            "placement": placement,
            "per_rank": {"kernel_ms": [round(r.kernel_ms, 4) for r in records],
                         "elapsed_ms_per_step": [round(r.elapsed_s / args.steps * 1e3, 4) for r in records],
                         "stream_bytes": [int(r.stream_bytes) for r in records],
                         # every rank's own kernel against its own GPU's peak
                         "roofline_frac": [round((r.symbols + r.stream_bytes) / (r.kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                           for r in records]},
            "launch": {"self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1",
                       "gpus_requested": int(os.environ.get("BENCH_GPUS_REQUESTED", args.gpus)),
                       "gpus_visible": int(os.environ.get("BENCH_GPUS_VISIBLE", torch.cuda.device_count())),
                       "all_on_device": args.all_on_device},
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "clock": "wall time of the K timed steps (ms_per_step)",
                "achieved_kernel_events": round(achieved_events, 1), "frac_kernel_events": round(achieved_events / HBM_PEAK_GBPS, 4),
                "kernel": ctx.last_decode_kernel(), "kernel_ms_avg": round(kernel_ms, 4),
                # the timed launches are dispatches [first, first + steps) of this kernel in this process, counted from 0
                "timed_dispatches": [timed_first, timed_first + args.steps],
                "algorithmic_bytes_per_launch": n + total,
                # the job: algorithmic bytes of ALL ranks over the slowest rank's step time, against N GPUs' peak (at N = 1
                # this is `frac`); achieved / peak / frac above are rank 0's launches on rank 0's GPU
                "frac_job": round(sum(r.symbols + r.stream_bytes for r in records) / (ms_per_step * 1e-3)
                                  / 1e9 / (len(records) * HBM_PEAK_GBPS), 4),
                "achieved_job": round(sum(r.symbols + r.stream_bytes for r in records) / (ms_per_step * 1e-3) / 1e9, 1),
                "peak_job": len(records) * HBM_PEAK_GBPS,
                "frac_of_measured_copy": round(achieved / HBM_COPY_GBPS, 4),
                # what a launch spends inside its wavefronts (first wave start .. last wave end, mean over the timed
                # launches of rank 0); the rest of kernel_ms_avg is dispatch, the hand-over between back-to-back
                # kernels and the write-back at the kernel's end
                "wave_span_ms_avg": round(sum(v for v in spans if v > 0) / max(1, sum(1 for v in spans if v > 0)), 4)
                if spans else None,
            },
        }
        if first_pair_elapsed is not None:
            result["placement"]["first_pair_ms_per_step"] = round(first_pair_elapsed / args.steps * 1e3, 4)
        if args.dump_launch_ms:
            result["launch_ms"] = [round(a.elapsed_time(b), 4) for a, b in zip(ev0, ev1)]
            result["launch_span_ms"] = [round(v, 4) for v in spans]
        # per-wave clocks, measured by the kernel in one extra (untimed, instrumented) launch
        try:
            ctx.set_timing(2)
            step()
            wc = ctx.last_wave_clocks()
            ctx.set_timing(0)
            per_wave_round = wc["shader_cycles"] / max(1, wc["rounds"])
            result["clocks"] = {
                "sclk_hz_measured": round(wc["sclk_hz"]),
                "per_wave_clocks_per_round_of_64": round(per_wave_round, 1),
                "per_simd_clocks_per_round_of_64": round(per_wave_round / WAVES_PER_SIMD, 2),
                "clocks_per_symbol_per_simd": round(per_wave_round / WAVES_PER_SIMD / 64.0, 4),
                "gpu_aggregate_clocks_per_symbol": round(k_s * wc["sclk_hz"] / n, 6),
                "waves": wc["waves"], "rounds": wc["rounds"],
                "instrumented_launch_ms": round(wc["kernel_ticks_ms"], 4),
                "method": "s_memtime / wall_clock64 bracket around each wave's work (rans_amd_set_timing(ctx, 2)), "
                          "one extra launch after the timed region; the reference's analogue is the __rdtsc "
                          "bracket of main.cpp:171,184-186",
            }
        except Exception as e:  # noqa: BLE001
            result["clocks"] = {"error": repr(e)}
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this very command
        # (tools/profile.sh -> tools/summarize_profile.py -> profiles/<tag>_traffic.json); PMC passes cannot
        # share a process with the timed run, so a committed measurement is quoted only when it was taken on
        # this workload AND on the kernel sources of this checkout (its `kernel_source_tag`), else null.
        tj = os.environ.get("RANS_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "r06_traffic.json"))
        default_workload = (args.format == "word" and args.ways == 64 and args.chunk == 32768 and args.log2n == 30)
        if os.path.exists(tj) and default_workload:
            try:
                t = json.load(open(tj))
                if t.get("kernel_source_tag") == kernel_source_tag():
                    result["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                    result["roofline"]["traffic_source"] = os.path.relpath(tj, ROOT) + \
                        " (FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate --pmc passes, same kernel sources)"
            except (OSError, ValueError):
                pass
        arts = [{"fmt": fmt, "sb": sb, "K": 256, "ways": args.ways, "chunk": args.chunk, "n": n, "freqs": freqs,
                 "d_syms": d_syms, "cont": cont, "offs": offs, "lens": lens, "total": total, "entry": None}]
        if world == 1 and not args.no_configs:
            cfgs = []
            try:
                ks = args.config_steps
                cp = min(args.placement_candidates, 4)  # (the `configs` entries: up to four candidates per written buffer)
                cc = args.config_chunk
                # the headline configuration's encoder (and its decoder once more) at the `configs` chunk size
                e, a = measure_config(torch, R, ctx, "C3-word64", "C3 word 64-way 1 GiB, %d-symbol chunks (the headline's format: encoder, "
                                      "and decoder at this chunk size)" % cc, R.FMT_WORD, 12,
                                      256, args.ways, cc, args.log2n, 1, ks, device, d_syms=d_syms, probe=cp)
                cfgs.append(e)
                if cc == args.chunk:  # (the headline's own artefacts: container checked below, CPU loop timed beside it)
                    arts[0]["entry"] = e
                    arts[0]["slots"] = a["slots"]
                    arts[0]["tight"] = a["tight"]
                else:
                    arts.append(a)
                e, a = measure_config(torch, R, ctx, "C2-r64x2", "C2 rans64 2-way 256 MiB Zipf(256)", R.FMT_R64, 14, 256, 2, 512, 28, 1,
                                      ks, device, probe=cp)
                cfgs.append(e)
                arts.append(a)
                e, a = measure_config(torch, R, ctx, "C4-alias4096", "C4 alias 4096 symbols 64-way 512 Mi u16 symbols", R.FMT_ALIAS, 16,
                                      4096, 64, cc, 29, 1, ks, device, probe=cp)
                cfgs.append(e)
                arts.append(a)
                e, a = measure_config(torch, R, ctx, "byte14", "byte format 64-way 1 GiB Zipf(256)", R.FMT_BYTE, 14, 256, 64, cc,
                                      args.log2n, 1, ks, device, d_syms=d_syms, probe=cp)
                cfgs.append(e)
                arts.append(a)
                # the byte format at 12-bit probabilities: the decoder's fused slot records (one LDS gather per symbol, round 4)
                e, a = measure_config(torch, R, ctx, "byte12", "byte format 64-way 1 GiB Zipf(256), scale_bits 12 (slot-record decoder)",
                                      R.FMT_BYTE, 12, 256, 64, cc, args.log2n, 1, ks, device, d_syms=d_syms, probe=cp)
                cfgs.append(e)
                arts.append(a)
                # the layouts the reference itself ships: its 8-way word streams -- main_simd.cpp:287-332, the
                # SSE4.1 decoder's own input -- and its 2-way byte streams (main.cpp:226-280), one LANE per chunk (lanes.hip)
                # (1024-symbol chunks: a wave of these kernels walks 64 streams at once, and streams that lie more than a DRAM page
                #  apart cost it the row locality -- 0.91 ms at 1 Ki symbols, 1.27 at 4 Ki, 2.2 at 16 Ki: profiles/r06_lanes_chunk_sweep.log)
                e, a = measure_config(torch, R, ctx, "word8", "word 8-way 1 GiB Zipf(256), 1024-symbol chunks (the reference's SIMD layout)",
                                      R.FMT_WORD, 12, 256, 8, 1024, args.log2n, 1, ks, device, d_syms=d_syms, probe=1)
                cfgs.append(e)
                arts.append(a)
                e, a = measure_config(torch, R, ctx, "byte2", "byte 2-way 1 GiB Zipf(256), scale_bits 14, 1024-symbol chunks (main.cpp's layout)",
                                      R.FMT_BYTE, 14, 256, 2, 1024, args.log2n, 1, ks, device, d_syms=d_syms, probe=1)
                cfgs.append(e)
                arts.append(a)
                # per-chunk models: count + normalise + code in one kernel, every chunk its own model
                for afmt, aname in ((R.FMT_WORD, "word"), (R.FMT_BYTE, "byte")):
                    e, a = measure_adaptive(torch, R, ctx, aname + "-adaptive", "%s format, one model per %d-symbol chunk, 64-way, 1 GiB "
                                            "Zipf(256), 12 bits" % (aname, cc), afmt, 12, 64, cc, d_syms, ks)
                    cfgs.append(e)
                    arts.append(a)
                # "64-way and wider" (north_star): two and four states per lane over the headline's data
                for wide in (128, 256):
                    e, a = measure_config(torch, R, ctx, "word%d" % wide, "word %d-way 1 GiB Zipf(256) (%d states per lane)" % (wide, wide // 64),
                                          R.FMT_WORD, 12, 256, wide, cc, args.log2n, 1, ks, device, d_syms=d_syms, probe=cp)
                    cfgs.append(e)
                    arts.append(a)
            except Exception as e:  # noqa: BLE001
                cfgs.append({"error": repr(e)})
            result["configs"] = cfgs
            if any(not c.get("bit_exact_roundtrip", False) for c in cfgs):
                all_ok = False
        if world > 1 and not args.no_cpu_baseline:
            result["oracle_chunks_checked"] = int(sum(r.oracle_chunks for r in records))
            result["oracle_chunks_total"] = len(records) * ((n + args.chunk - 1) // args.chunk)
            result["oracle_chunks_checked_per_rank"] = [int(r.oracle_chunks) for r in records]
            try:  # (the other ranks wait in the barrier below: their host threads sleep, the reference gets the cores)
                result["cpu_baseline"] = cpu_baseline(d_syms, freqs, n)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference",
                                          "sample": "failed: %r" % (e,)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                trace("oracle checks")
                checked = {}
                t_or = time.perf_counter()
                for a in arts:
                    if a.get("adaptive"):
                        key = "%s-%d/%d-way/%d per-chunk models" % (R.FORMAT_NAMES[a["fmt"]], a["sb"], a["ways"], a["chunk"])
                        checked[key] = oracle_check_adaptive(a)
                        a["entry"]["oracle_chunks_checked"] = checked[key]
                        a["entry"]["oracle_chunks_total"] = (a["n"] + a["chunk"] - 1) // a["chunk"]
                        del a["cont"]
                        continue
                    key = "%s-%d/%d-way/%d" % (R.FORMAT_NAMES[a["fmt"]], a["sb"], a["ways"], a["chunk"])
                    checked[key] = oracle_check_chunks(a, args.oracle_sample)
                    if a["entry"] is not None:
                        a["entry"]["oracle_chunks_checked"] = checked[key]
                        a["entry"]["oracle_chunks_total"] = (a["n"] + a["chunk"] - 1) // a["chunk"]
                    if a.get("slots"):  # ... and every chunk of the slot container the write-once encoder left behind
                        sl = a["slots"]
                        a["entry"]["oracle_chunks_checked_slots"] = oracle_check_chunks(
                            dict(a, cont=sl["cont"], offs=sl["offs"], lens=sl["lens"], total=sl["total"], slot=sl["slot"]),
                            args.oracle_sample)
                        del sl["cont"]
                    if a.get("tight"):  # ... and of the sized-slot container
                        tg = a["tight"]
                        a["entry"]["oracle_chunks_checked_tight"] = oracle_check_chunks(
                            dict(a, cont=tg["cont"], offs=tg["offs"], lens=tg["lens"], total=tg["total"], slot=tg["slot"],
                                 worst=tg["worst"]), args.oracle_sample)
                        del tg["cont"]
                result["oracle_chunks_checked"] = checked["%s-%d/%d-way/%d" % (args.format, sb, args.ways, args.chunk)]
                result["oracle_chunks_total"] = (n + args.chunk - 1) // args.chunk
                result["oracle_chunks_checked_all"] = checked
                # ... and the other direction: the headline decoder on a container the ORACLE made
                result["decodes_oracle_container"] = decode_oracle_container(torch, R, ctx, model, arts[0], device)
                if not result["decodes_oracle_container"]:
                    all_ok = False
                result["oracle_check_s"] = round(time.perf_counter() - t_or, 2)
            except Exception as e:  # noqa: BLE001
                result["oracle_chunks_checked"] = 0
                result["oracle_check_error"] = repr(e)
                all_ok = False
            t_cb = time.perf_counter()
            for a in arts:  # the reference's own loop of every configuration's format, on this box, beside its GPU numbers
                if a["entry"] is None or a.get("adaptive"):  # (per-chunk models: the reference has one model per input, no such loop)
                    continue
                try:
                    a["entry"]["cpu_baseline"] = config_cpu_baseline(a)
                except Exception as e:  # noqa: BLE001
                    a["entry"]["cpu_baseline"] = {"value": None, "kind": "reference", "sample": "failed: %r" % (e,)}
            result["config_cpu_baselines_s"] = round(time.perf_counter() - t_cb, 2)
            try:
                result["cpu_baseline"] = cpu_baseline(d_syms, freqs, n)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference",
                                          "sample": "failed: %r" % (e,)}
        result["bit_exact_roundtrip"] = all_ok  # (after the configs and the oracle leg have had their say)
        if not all_ok:
            result["error"] = "round trip mismatch, corrupt chunk reported, or a chunk differs from the oracle"
        # The full record goes to a FILE; stdout gets ONE short line (the reference prints ~70 characters per run,
        # main_simd.cpp:267) -- the driver keeps only the last few KB of stdout, and round 4's 20 KB line was cut in two.
        details = args.details or os.path.join(ROOT, "bench_details.json")
        try:
            with open(details, "w") as fh:
                json.dump(result, fh, indent=1)
                fh.write("\n")
        except OSError as e:
            print("bench.py: could not write %s: %s" % (details, e), file=sys.stderr)
            details = None
        line = judged_line(result, os.path.relpath(details, ROOT) if details else None)
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) <= MAX_LINE_BYTES, "judged line is %d bytes" % len(text)
        if not all_ok and not args.measure:
            exit_code = 1
    if use_dist:
        # Rank 0 has just timed the reference on the host cores while the other ranks had nothing left to do: they must SLEEP
        # through that (a barrier on the RCCL backend spins a core per rank in hipStreamSynchronize, and the container's CPU
        # quota is shared) -- they block on a key of the rendezvous store (a socket read) that rank 0 sets when it is done.
        try:
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("bench_cpu_leg_done", "1")
            else:
                store.wait(["bench_cpu_leg_done"], datetime.timedelta(minutes=15))
        except Exception:  # noqa: BLE001  (no store to be had: the barrier below still lines the ranks up)
            pass
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The line is the LAST thing this process writes to stdout: RCCL prints a version banner through C stdio, which
        # sits in libc's buffer until somebody flushes it -- at exit, i.e. BEHIND a line printed earlier (seen on the GPU
        # box: the judged line followed by "RCCL version : ...").  Flush C stdio first, then print.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(text, flush=True)
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
