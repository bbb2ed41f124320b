#!/usr/bin/env python3
"""bench.py -- headline benchmark: 64-way interleaved rANS decode (word format) of
synthetic order-0 byte streams, one 1 GiB shard per GPU (BASELINE.json configs[2] / [4]).

  python bench.py                                  # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the decode hot path over the whole shard (device-resident
container -> device-resident symbols).  Per rank: generate Zipf(256, s=1) bytes on
the GPU (seed = rank+1), build the order-0 model (GPU histogram + exact
normalize_freqs), encode with the GPU encoder (setup, untimed), then W warm-up and K
timed decodes.  Shards are independent: no collective on the data path; RCCL only
carries the barriers and the 64-byte per-rank result record (weak scaling).

Rank 0 prints ONE JSON line.  `value` = decoded (uncompressed) GB/s of the whole job.
`roofline` = algorithmic bytes (compressed stream read + symbols written) of one decode
launch / its average duration measured with HIP events on the launch stream, vs the
8 TB/s HBM peak.  `cpu_baseline` = the reference's own fastest decoder (SSE4.1 8-way,
oracle/_ref) on this box's host cores over a bounded sample of the same data.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0  # measured float4 streaming copy (same guide)
MAX_CLOCK_HZ = 2.4e9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=30, help="symbols per GPU = 2^log2n (default 1 GiB)")
    ap.add_argument("--ways", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=32768, help="symbols per independent chunk stream")
    ap.add_argument("--format", default="word", choices=["word", "byte", "r64", "alias"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for dry runs)")
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="dry-run aid: every rank uses this GPU (needs --backend gloo)")
    ap.add_argument("--cpu-shard-log2", type=int, default=25, help="CPU baseline: symbols per host thread")
    return ap.parse_args()


def gen_zipf_bytes(torch, n, seed, device):
    """Zipf(K=256, s=1) bytes by inverse CDF on the GPU (SURVEY.md 8(d) distribution)."""
    w = 1.0 / torch.arange(1, 257, dtype=torch.float64, device=device)
    cdf = torch.cumsum(w / w.sum(), 0).float()
    out = torch.empty(n, dtype=torch.uint8, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    step = min(n, 1 << 26)
    for i in range(0, n, step):
        m = min(step, n - i)
        u = torch.rand(m, device=device, generator=g)
        out[i:i + m] = torch.searchsorted(cdf, u).clamp_(max=255).to(torch.uint8)
    return out


def cpu_baseline(d_syms, freqs, n, shard_log2):
    """Reference CPU path on this box (rank 0, N=1 only).  TEST/BASELINE leg: the only
    place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from _oracle import FMT_WORD, Oracle, Ref

    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    max_threads = max(1, usable)
    shard = min(1 << 22, n // max_threads)       # 4 Mi symbols per shard (fits L2/L3: CPU-friendly)
    shard -= shard % 8
    reps = max(1, (1 << 26) // shard)            # >= 64 Mi symbols (~0.08 s) of work per thread and run
    if not Ref.available():
        # port: the scalar C restatement, one thread, 64-way stream
        orc = Oracle()
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

    ref = Ref()
    import ctypes as C
    host = d_syms[:max_threads * shard].cpu().numpy()

    def enc(i):
        return ref.encode(FMT_WORD, freqs, 12, host[i * shard:(i + 1) * shard], 8)

    with ThreadPoolExecutor(min(max_threads, 64)) as ex:
        streams = list(ex.map(enc, range(max_threads)))
    offsets = np.zeros(max_threads, dtype=np.uint64)
    pos = 0
    for i, s in enumerate(streams):
        offsets[i] = pos
        pos += (s.size + 16 + 15) & ~15
    blob = np.zeros(pos + 64, dtype=np.uint8)
    for i, s in enumerate(streams):
        blob[int(offsets[i]):int(offsets[i]) + s.size] = s
    out = np.zeros(max_threads * shard, dtype=np.uint8)
    f32 = np.ascontiguousarray(freqs, dtype=np.uint32)

    def run(nthreads, nreps):
        """nthreads pthreads, one shard each, nreps passes; returns seconds per pass."""
        return ref.lib.ref_time_word_simd8(f32.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           blob.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           offsets.ctypes.data_as(C.POINTER(C.c_uint64)), nthreads, shard,
                                           out.ctypes.data_as(C.POINTER(C.c_uint8)), nthreads, nreps) / nreps

    # thread-count sweep: the box may expose more logical CPUs than it lets us run
    sweep = {}
    t = 1
    cands = []
    while t < max_threads:
        cands.append(t)
        t *= 2
    cands.append(max_threads)
    for t in cands:
        if t == 1:
            continue
        sweep[t] = t * shard / min(run(t, reps) for _ in range(2)) / 1e9
    run(max_threads, 1)
    assert np.array_equal(out, host), "reference CPU decode mismatch"
    best_threads = max(sweep, key=sweep.get) if sweep else 1
    c0 = ref.lib.ref_rdtsc()
    t_one = run(1, reps)
    clocks = (ref.lib.ref_rdtsc() - c0) / reps
    t_one = min(t_one, run(1, reps))
    single = shard / t_one / 1e9
    if not sweep or single > sweep[best_threads]:
        best_threads, best_val = 1, single
    else:
        best_val = sweep[best_threads]
    return {"value": best_val, "unit": "GB/s", "cores": best_threads, "kind": "reference",
            "sample": "%d x %d MiB shards from the start of rank 0's data, each an 8-way word stream decoded by the "
                      "reference SSE4.1 loop (main_simd.cpp:313-332), one pthread per shard, %d passes per run; "
                      "thread counts %s swept, best reported" % (best_threads, shard >> 20, reps, cands),
            "single_thread_value": single,
            "single_thread_clocks_per_symbol": clocks / shard,
            "thread_sweep_GBps": {str(k): round(v, 2) for k, v in sweep.items()},
            "host_cpus": cores, "usable_cpus": usable}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import ryg_rans_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    gpu_index = local_rank if args.all_on_device is None else args.all_on_device
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    fmt = {"word": R.FMT_WORD, "byte": R.FMT_BYTE, "r64": R.FMT_R64, "alias": R.FMT_ALIAS}[args.format]
    sb = {"word": 12, "byte": 14, "r64": 14, "alias": 16}[args.format]
    n = 1 << args.log2n

    # ---- setup (untimed): data, model, GPU encode ------------------------------
    ctx = R.Context(gpu_index)
    d_syms = gen_zipf_bytes(torch, n, seed=rank + 1, device=device)
    counts = ctx.count_freqs_device(d_syms, 256)
    freqs, _ = R.normalize_freqs(counts, 1 << sb)
    model = ctx.model(fmt, freqs, sb)
    cont, offs, lens, total = ctx.encode(model, d_syms, args.ways, args.chunk)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()

    def step():
        ctx.decode(model, cont, total, offs, lens, n, args.ways, args.chunk, d_out=out, sync=False)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()  # HIP events on the launch stream (torch's current stream)
        step()
        ev1[k].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()

    # ---- verification (after the timed region) -------------------------------------
    bad = ctx.decode_errors()
    exact = bool(torch.equal(out, d_syms))
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps

    from ryg_rans_amd.sharding import ShardRecord, aggregate, gather_records
    rec = ShardRecord(elapsed, float(n), float(total), kernel_ms, 1.0 if (exact and bad == 0) else 0.0)
    # the only payload RCCL carries: 40 bytes per rank
    records = gather_records(rec, device=device if args.backend == "nccl" else "cpu")

    if rank == 0:
        agg = aggregate(records, args.steps)
        all_ok = agg["all_ok"]
        ms_per_step = agg["ms_per_step"]
        value = agg["symbols_per_s"] / 1e9  # 1 byte per symbol
        k_s = kernel_ms * 1e-3
        achieved = (n + total) / k_s / 1e9
        result = {
            "metric": "decode GB/s (uncompressed), 64-way interleaved rANS (word format)",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
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
                "sharding": "one independent shard per GPU, no data-path collective",
            },
            "bit_exact_roundtrip": all_ok,
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": ctx.last_decode_kernel(), "kernel_ms_avg": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": n + total,
                "frac_of_measured_copy": round(achieved / HBM_COPY_GBPS, 4),
            },
            "clocks_per_symbol": {
                "gpu_aggregate_at_2.4GHz": round(k_s * MAX_CLOCK_HZ / n, 6),
                "per_wave_round_of_64": round(k_s * MAX_CLOCK_HZ / n * 64 * 8192, 1),
            },
        }
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this very command
        # (tools/profile.sh -> tools/summarize_profile.py -> profiles/<tag>_traffic.json); PMC passes
        # cannot share a process with the timed run, so the committed measurement is quoted when it
        # was taken on the same workload, else null.
        tj = os.environ.get("RANS_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "r01_traffic.json"))
        default_workload = (args.format == "word" and args.ways == 64 and args.chunk == 32768 and args.log2n == 30)
        if os.path.exists(tj) and default_workload:
            try:
                t = json.load(open(tj))
                result["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                result["roofline"]["traffic_source"] = os.path.relpath(tj, ROOT) + \
                    " (FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate --pmc passes)"
            except (OSError, ValueError):
                pass
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(d_syms, freqs, n, args.cpu_shard_log2)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference",
                                          "sample": "failed: %r" % (e,)}
        if not all_ok:
            result["error"] = "round trip mismatch or corrupt chunk reported"
        print(json.dumps(result), flush=True)
        if not all_ok:
            sys.exit(1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
