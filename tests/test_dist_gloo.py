"""The N>1 path on CPU: two processes over gloo exercise the sharding plan and the
record all-gather/aggregation bench.py uses over RCCL (no data-path collective)."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ryg_rans_amd.sharding import ShardRecord, aggregate, chunk_range, gather_records, symbol_range


def test_chunk_ranges_partition_everything():
    for nchunks in (0, 1, 7, 8, 9, 32768, 32769):
        for world in (1, 2, 3, 4, 8):
            spans = [chunk_range(nchunks, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nchunks
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert symbol_range(100001, 4096, 2, 0) == (0, 12 * 4096)
    assert symbol_range(100001, 4096, 2, 1) == (12 * 4096, 100001)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_bench(argv, env, timeout=600, launcher=None):
    """Run bench.py, return (judged line parsed from ONLY the last 4 KB of stdout, full record from --details, CompletedProcess).
    The driver keeps a few KB of stdout: whatever the test needs from the line must survive that cut."""
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        details = os.path.join(tmp, "details.json")
        cmd = (launcher or [sys.executable]) + [os.path.join(root, "bench.py")] + argv + ["--details", details]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
        if out.returncode != 0:  # (the caller asserts on the return code: hand it what the run said)
            out.stderr = (out.stderr or "")[-1500:] + "\n--- stdout tail ---\n" + (out.stdout or "")[-2500:]
            return None, None, out
        tail = out.stdout[-4096:]
        last = tail.rstrip("\n").splitlines()[-1]
        assert last.startswith("{") and len(last) < 4096, out.stdout[-500:]
        assert len([ln for ln in out.stdout.splitlines() if ln.startswith("{")]) == 1, "exactly one JSON line on stdout"
        line = json.loads(last)
        full = json.load(open(details))
    return line, full, out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r "decodes" its own shard; rank 1 is slower
    rec = ShardRecord(elapsed_s=0.010 * (rank + 1), symbols=float(1 << 20), stream_bytes=800000.0 + rank,
                      kernel_ms=0.5 + rank, ok=1.0)
    dist.barrier()
    records = gather_records(rec)
    dist.barrier()
    if rank == 0:
        q.put([r.to_list() for r in records])
    dist.destroy_process_group()


def test_two_rank_gather_and_aggregate():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    records = [ShardRecord(*r) for r in rows]
    assert [r.stream_bytes for r in records] == [800000.0, 800001.0]
    agg = aggregate(records, steps=10)
    assert agg["n_ranks"] == 2 and agg["all_ok"]
    assert agg["ms_per_step"] == pytest.approx(2.0)            # MAX over ranks: 20 ms / 10 steps
    assert agg["symbols_per_s"] == pytest.approx(2 * (1 << 20) / 0.002)  # SUM of work / max time
    bad = records + [ShardRecord(0.01, 1.0, 1.0, 1.0, 0.0)]
    assert not aggregate(bad, 10)["all_ok"]


@pytest.mark.gpu
def test_two_ranks_decode_real_shards_on_one_gpu():
    """The multi-GPU path end to end on the one GPU this box has: `torch.distributed.run` starts two ranks of
    bench.py (gloo for the 48-byte record gather, both ranks on device 0), each generates, encodes and decodes ITS
    OWN shard (seed = rank + 1) bit-exactly, and the printed line proves what it claims: `n_gpus` is the number of
    records the all-gather delivered, not an environment variable; every rank's kernel time, stream size and
    verdict is listed."""
    import sys
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port())]
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--all-on-device", "0", "--log2n", "26",
            "--prewarm-ms", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    line, d, out = _run_bench(argv, env, launcher=launcher)
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["bit_exact_roundtrip"] is True and line["scaling"] == "weak"
    pr = d["per_rank"]
    assert len(pr["kernel_ms"]) == 2 and all(v > 0 for v in pr["kernel_ms"]) and line["per_rank_kernel_ms"] == pr["kernel_ms"]
    assert len(pr["stream_bytes"]) == 2 and pr["stream_bytes"][0] != pr["stream_bytes"][1]  # different shards
    # whole-job value = symbols of both ranks / the slower rank's time
    slow = max(pr["elapsed_ms_per_step"])
    assert line["value"] == pytest.approx(2 * (1 << 26) / (slow * 1e-3) / 1e9, rel=0.02)
    # north_star: the N-GPU number stands "next to the reference CPU path timed on the same box's host cores" -- rank 0
    # attaches it at every N -- and every rank pinned a sample of ITS shard against the oracle (count summed over the records)
    assert "configs" not in line
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port")
    assert d["oracle_chunks_checked_per_rank"][0] >= 256 and d["oracle_chunks_checked_per_rank"][1] >= 256
    assert line["oracle_chunks_checked"] == sum(d["oracle_chunks_checked_per_rank"])
    assert line["oracle_chunks_total"] == 2 * ((1 << 26) // 32768)


@pytest.mark.gpu
def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` started PLAINLY (no torchrun around it, no WORLD_SIZE in the environment -- the shape
    of the driver's N = 1 command with another number) spawns its own two ranks, and the record is self-describing: it says
    it launched itself, how many GPUs were asked for and how many the node shows, and carries the job-level roofline
    fraction (algorithmic bytes of all ranks / slowest kernel / N x 8 TB/s) beside every rank's own."""
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    line, d, out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--all-on-device", "0",
                               "--log2n", "26", "--prewarm-ms", "0", "--no-cpu-baseline"], env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["bit_exact_roundtrip"] is True
    assert d["launch"]["self_launched"] is True and d["launch"]["gpus_requested"] == 2 and d["launch"]["gpus_visible"] >= 1
    pr, rl = d["per_rank"], d["roofline"]
    assert len(pr["roofline_frac"]) == 2 and all(0.0 < f < 1.0 for f in pr["roofline_frac"])
    alg = sum((1 << 26) + sb for sb in pr["stream_bytes"])
    # (one clock: the job's fraction is over ms_per_step -- the slowest rank's K timed steps --, as `value` is)
    assert rl["frac_job"] == pytest.approx(alg / (line["ms_per_step"] * 1e-3) / 1e9 / (2 * 8000.0), rel=0.01)
    assert rl["peak_job"] == 16000.0 and line["roofline"]["frac_job"] == rl["frac_job"]
    assert "cpu_baseline" not in line and "oracle_chunks_checked" not in line  # (--no-cpu-baseline: no CPU leg at all)
    # asked for more GPUs than the node has, without the dry-run aid: clamped, said so, still a valid line
    line, d, out = _run_bench(["--gpus", "64", "--steps", "2", "--warmup", "1", "--log2n", "24", "--prewarm-ms", "0",
                               "--no-configs", "--no-cpu-baseline"], env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["n_gpus"] == min(64, torch.cuda.device_count())
    assert d["launch"] == {"self_launched": True, "gpus_requested": 64, "gpus_visible": torch.cuda.device_count(),
                           "all_on_device": None}
    assert d["roofline"]["frac_job"] > 0 and line["roofline"]["frac"] > 0  # (one GPU: the job IS the launch, the line carries `frac`)


@pytest.mark.gpu
def test_bench_default_command_line_survives_the_drivers_stdout_window():
    """The driver's command at a small size, everything on (configs, CPU leg, placement probe): the judged line is the LAST
    line of stdout, parses from the last 4 KB alone, and carries value / ms_per_step / roofline / cpu_baseline (reference) /
    one row per config with its oracle verdict -- round 4's line was 20.6 KB and came back unparsed."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    line, d, out = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--log2n", "24", "--config-steps", "3"], env,
                              timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["bit_exact_roundtrip"] is True and line["headline"] is True
    rl = line["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and 0 < rl["frac"] < 1 and rl["kernel"] == "k_decode_word64"
    # one clock on the line: frac from ms_per_step (what the judge recomputes), the HIP-event figure beside it
    assert rl["frac"] == pytest.approx(rl["algorithmic_bytes_per_launch"] / (line["ms_per_step"] * 1e-3) / 8e12, rel=2e-3)
    assert rl["frac_kernel_events"] == pytest.approx(rl["algorithmic_bytes_per_launch"] / (rl["kernel_ms_avg"] * 1e-3) / 8e12, rel=2e-3)
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    rows = line["configs"]
    assert [r["name"] for r in rows] == ["C3-word64", "C2-r64x2", "C4-alias4096", "byte14", "byte12", "word8", "byte2", "word-adaptive",
                                         "byte-adaptive", "word128", "word256"]
    assert all(r["oracle_ok"] is True and r["decode_ms"] > 0 and r["enc_tight_ms"] > 0 for r in rows)
    assert all(r["encode_ms"] > 0 for r in rows if not r["name"].endswith("-adaptive"))
    assert line["oracle_chunks_checked"] == line["oracle_chunks_total"] and line["decodes_oracle_container"] is True
    # the un-probed pair: K timed steps of the same loop and clock (at this size -- 16 MiB, two steps of 0.08 ms -- the order of
    # the two is noise; at 1 GiB the canned record of tests/test_bench_cpu.py pins first pair <= headline)
    assert line["placement"]["first_pair_ms"] > 0 and line["value_first_pair"] > 0
    assert line["value_first_pair"] == pytest.approx((1 << 24) / line["placement"]["first_pair_ms_per_step"] / 1e6, rel=1e-3)
    # and the file has what the line dropped
    assert len(d["placement"]["probe_ms"]) >= 3 and len(d["configs"]) == len(rows) and "decoders" in d["cpu_baseline"]


@pytest.mark.gpu
def test_bench_placement_probe_allocates_more_when_all_candidates_look_alike():
    """The headline's placement probe (setup, untimed; profiles/r04_allocation.md): when every probed pair lies within
    --placement-spread of the fastest -- all candidates in one class of memory -- the candidates stay alive and another round
    is allocated, twice at most.  Forced here with a spread no measurement reaches: 2 + 2 + 2 outputs, 3 + 2 + 2 containers,
    the record says so, and the round trip of the chosen pair is bit-exact."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    argv = ["--steps", "2", "--warmup", "1", "--log2n", "24", "--prewarm-ms", "0", "--no-configs", "--no-cpu-baseline",
            "--placement-candidates", "2", "--placement-spread", "100"]
    line, d, out = _run_bench(argv, env)
    assert out.returncode == 0, out.stderr[-2000:]
    p = d["placement"]
    assert p["candidates"] == {"containers": 7, "outputs": 6, "extended": 2}
    assert len(p["probe_ms"]) == 7 and all(len(row) == 6 for row in p["probe_ms"])
    assert 0 <= p["chosen"][0] < 7 and 0 <= p["chosen"][1] < 6
    assert p["probe_ms_chosen"] == p["probe_ms_min"] and line["bit_exact_roundtrip"] is True
    want = {"chosen_ms": p["probe_ms_chosen"], "first_pair_ms": p["probe_ms_first_pair"], "min_ms": p["probe_ms_min"],
            "max_ms": p["probe_ms_max"], "pairs": 42, "stride_gib": p["stride_gib"]}
    assert {k: line["placement"][k] for k in want} == want
    assert p["stride_gib"] == 24  # (candidates 24 GiB apart in allocation order: profiles/r05_class_map.md)
    # the default spread on a small run: whatever the probe saw, the record carries the counts
    argv[-1] = "1.02"
    line, d, out = _run_bench(argv, env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert d["placement"]["candidates"]["extended"] in (0, 1, 2) and line["bit_exact_roundtrip"] is True


@pytest.mark.gpu
def test_bench_force_dist_runs_rccl_on_one_gpu():
    """bench.py --force-dist: torch.distributed over the nccl backend (RCCL) with a single rank -- init_process_group
    with device_id, both barriers and the on-device all_gather of the record execute on real RCCL on this box, so the
    N-GPU call path is not dead code until an 8-GPU node shows up."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    line, d, out = _run_bench(["--force-dist", "--steps", "3", "--warmup", "1", "--log2n", "26", "--prewarm-ms", "0",
                               "--no-configs", "--no-cpu-baseline"], env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["bit_exact_roundtrip"] is True and line["headline"] is True and d["knobs"] == {}
    assert d["distributed"] == {"initialised": True, "backend": "nccl", "records_gathered_on": "device (RCCL)"}


@pytest.mark.gpu
def test_eight_ranks_dry_run_on_one_gpu():
    """8-GPU readiness without an 8-GPU node (BASELINE configs[4]): the driver's SCALE command shape with
    EIGHT real ranks -- torch.distributed.run, gloo for the records, every rank on this box's one GPU (no spacers: the
    device's memory is shared) -- must deliver eight records, eight DISTINCT shards, every rank's oracle sample, the CPU
    baseline, and a last stdout line that parses from the last 4 KB and stays under the limit."""
    import sys
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port())]
    argv = ["--gpus", "8", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--all-on-device", "0", "--log2n", "24"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    line, d, out = _run_bench(argv, env, launcher=launcher, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) <= 3800 and json.loads(out.stdout[-4096:].splitlines()[-1]) == line  # (the driver's window)
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["bit_exact_roundtrip"] is True and "configs" not in line and "value_first_pair" not in line
    pr = d["per_rank"]
    assert len(pr["kernel_ms"]) == 8 and all(v > 0 for v in pr["kernel_ms"]) and line["per_rank_kernel_ms"] == pr["kernel_ms"]
    assert len(set(pr["stream_bytes"])) == 8  # eight different shards (seed = rank + 1)
    assert d["oracle_chunks_checked_per_rank"] == [d["oracle_chunks_checked_per_rank"][0]] * 8
    assert d["oracle_chunks_checked_per_rank"][0] >= 256 and line["oracle_chunks_checked"] == sum(d["oracle_chunks_checked_per_rank"])
    assert line["oracle_chunks_total"] == 8 * ((1 << 24) // 32768)
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port")
    slow = max(pr["elapsed_ms_per_step"])
    assert line["value"] == pytest.approx(8 * (1 << 24) / (slow * 1e-3) / 1e9, rel=0.02) and line["ms_per_step"] == pytest.approx(slow, rel=0.01)
    assert d["placement"].get("stride_gib", 0) == 0  # (ranks that share a device do not space their candidates)
    # ... and started plainly (`python bench.py --gpus 8`, no launcher): it spawns the eight ranks itself
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    line, d, out = _run_bench(argv + ["--no-cpu-baseline"], env, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["n_gpus"] == 8 and d["launch"]["self_launched"] is True and len(d["per_rank"]["kernel_ms"]) == 8


@pytest.mark.gpu
def test_native_example_with_eight_ranks_sharing_a_device():
    """examples/multi_gpu.cpp at world 8 on the devices this box has: eight host threads, eight contexts, eight probes with
    spacer allocations that must share the free memory, eight independent shards (--share) and ONE container split eight ways
    (--split-one); records through host memory when ranks share devices."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "multi_gpu")
    if not os.path.exists(exe):
        pytest.skip("build/multi_gpu not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for mode, want in (("--share", "one shard per rank"), ("--split-one", "one container split by chunk range")):
        out = subprocess.run([exe, mode, "8", "24", "3"], capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0 and "decode ok!" in out.stdout, (mode, out.stdout[-2000:], out.stderr[-2000:])
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert line["ranks"] == 8 and line["mode"] == want and line["bit_exact_roundtrip"] is True and line["value"] > 0
        rows = [ln.split() for ln in out.stdout.splitlines() if ln.strip() and ln.split()[0] in [str(i) for i in range(8)] and len(ln.split()) > 3]
        assert len(rows) >= 8, out.stdout
        if mode == "--share":
            assert len([ln for ln in out.stdout.splitlines() if ln.startswith("rank ") and "placement" in ln]) == 8


@pytest.mark.gpu
def test_force_dist_agrees_with_the_plain_run():
    """SCALE's N = 1 (real RCCL at world size 1: init_process_group, two barriers, the on-device all-gather) must agree with
    BENCH's N = 1 (no process group): the same timed region either way.  What the process group could add is time BETWEEN the
    launches of the timed loop -- so the test pins exactly that: ms_per_step minus the HIP-event mean of the same launches
    (0.007 ms of launch gaps) may not grow by more than 0.004 ms, i.e. 1 % of a step.  The step times themselves also carry
    each process's placement lottery (the probe narrows it to a few per cent: DESIGN 5), hence the wider bound on them."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    argv = ["--steps", "20", "--warmup", "3", "--no-configs", "--no-cpu-baseline"]
    plain, forced = [], []
    for _ in range(2):
        line, d, out = _run_bench(argv, env, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        plain.append((line["ms_per_step"], line["ms_per_step"] - line["roofline"]["kernel_ms_avg"]))
        env["MASTER_PORT"] = str(_free_port())
        line, d, out = _run_bench(argv + ["--force-dist"], env, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert d["distributed"]["backend"] == "nccl" and line["n_gpus"] == 1
        forced.append((line["ms_per_step"], line["ms_per_step"] - line["roofline"]["kernel_ms_avg"]))
    assert min(g for _, g in forced) <= min(g for _, g in plain) + 0.004, (plain, forced)
    assert min(m for m, _ in forced) == pytest.approx(min(m for m, _ in plain), rel=0.06), (plain, forced)


@pytest.mark.gpu
def test_bench_refuses_knobs_on_a_headline_run():
    """A stray RANS_AMD_* variable (another library build, an experiment knob) must not ride through the judged line:
    without --measure the run stops before it measures anything."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANS_AMD_DEBUG="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--log2n", "22"], capture_output=True,
                         text=True, timeout=300, env=env, cwd=root)
    assert out.returncode != 0 and "not a headline run" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
