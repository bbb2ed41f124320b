"""CPU-side checks of bench.py's helpers and of the committed book1 fixture (no GPU needed)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bench_generator_equals_oracle(oracle):
    """bench.gen_zipf (torch tensors, here on the CPU) is the SURVEY 8(d) generator oracle.gen_zipf implements."""
    import torch

    import bench
    for K, seed, n in ((256, 1, 200003), (4096, 1, 100000), (256, 9, 999), (4096, 2, 65537)):
        got = bench.gen_zipf(torch, n, K, 1.0, seed, "cpu").numpy()
        if got.dtype == np.int16:
            got = got.view(np.uint16)
        assert np.array_equal(got, oracle.gen_zipf(n, K=K, s=1.0, seed=seed)), (K, seed)


def test_book1_fixture_decodes_to_the_corpus(oracle):
    """tests/golden/book1_word64.bin (made by the unmodified reference) is book1: the oracle decodes it to the
    published sha256, with the committed 12-bit frequencies."""
    meta = json.load(open(os.path.join(HERE, "golden", "book1_word64.json")))
    known = json.load(open(os.path.join(HERE, "golden", "book1_golden.json")))
    stream = np.fromfile(os.path.join(HERE, "golden", meta["stream"]), dtype=np.uint8)
    assert hashlib.sha256(stream.tobytes()).hexdigest() == meta["stream_sha256"]
    want = [e for e in known["streams"] if e["fmt"] == "word" and e["n_ways"] == 64][0]
    assert stream.size == want["size"] and meta["stream_sha256"] == want["sha256"]
    om = oracle.model(np.array(meta["freqs"]["12"], dtype=np.uint32), 12)
    data = oracle.decode(meta["fmt"], om, stream, meta["n"], meta["n_ways"])
    assert hashlib.sha256(data.tobytes()).hexdigest() == known["input_sha256"]
    # and the frequencies are what the oracle's model builder makes of the corpus
    counts = oracle.count_freqs(data, 256)
    for sb in (12, 14, 16):
        f, _ = oracle.normalize(counts, 1 << sb)
        assert [int(v) for v in f] == meta["freqs"][str(sb)]


def test_index_check_rejects_a_wrong_index(oracle):
    """bench.oracle_check_chunks really compares: every chunk by default (threaded orc_compare_chunks), a sample on
    request; a flipped byte in ANY chunk or a shifted offset fails it."""
    import pytest
    import torch

    import bench
    from _oracle import FMT_WORD
    data = oracle.gen_zipf(50000, K=256, s=1.0, seed=3)
    f, _ = oracle.normalize(oracle.count_freqs(data, 256), 4096)
    om = oracle.model(f, 12)
    cont, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, 64, 4096, align=16)
    art = {"fmt": FMT_WORD, "sb": 12, "K": 256, "ways": 64, "chunk": 4096, "n": data.size, "freqs": f,
           "d_syms": torch.from_numpy(data), "cont": torch.from_numpy(cont.copy()),
           "offs": torch.from_numpy(offs.astype(np.int64)), "lens": torch.from_numpy(lens.astype(np.int32)),
           "total": int(cont.size)}
    assert bench.oracle_check_chunks(art) == len(lens)
    assert bench.oracle_check_chunks(art, sample=8) >= 5
    for chunk in range(len(lens)):  # the full comparison sees a flipped bit wherever it is
        bad = dict(art)
        c = cont.copy()
        c[int(offs[chunk]) + int(lens[chunk]) - 1] ^= 1
        bad["cont"] = torch.from_numpy(c)
        with pytest.raises(AssertionError):
            bench.oracle_check_chunks(bad)
    bad = dict(art)
    c = cont.copy()
    c[int(offs[0]) + 300] ^= 1
    bad["cont"] = torch.from_numpy(c)
    with pytest.raises(AssertionError):
        bench.oracle_check_chunks(bad, sample=8)
    bad = dict(art)
    o = offs.astype(np.int64).copy()
    o[3] += 16
    bad["offs"] = torch.from_numpy(o)
    with pytest.raises(AssertionError):
        bench.oracle_check_chunks(bad)


def test_threaded_oracle_container_equals_the_serial_one(oracle):
    """Oracle.encode_chunked_mt (ranges of whole chunks in parallel, stitched) == encode_chunked, and
    compare_container reports the first differing chunk."""
    from _oracle import FMT_ALIAS, FMT_R64, FMT_WORD
    data = oracle.gen_zipf(300001, K=256, s=1.0, seed=5)
    for fmt, sb, ways, chunk in ((FMT_WORD, 12, 64, 4096), (FMT_R64, 14, 2, 512), (FMT_ALIAS, 16, 64, 1000)):
        om = oracle.model_for(data, 256, sb, with_alias=(fmt == FMT_ALIAS))
        cont, offs, lens = oracle.encode_chunked(fmt, om, data, ways, chunk, align=16)
        c2, o2, l2 = oracle.encode_chunked_mt(fmt, om, data, ways, chunk, align=16, threads=3)
        assert np.array_equal(cont, c2) and np.array_equal(offs, o2) and np.array_equal(lens, l2)
        assert oracle.compare_container(fmt, om, data, ways, chunk, cont, offs, lens, threads=3) == (len(lens), -1)
        c3 = cont.copy()
        c3[int(offs[17]) + 5] ^= 0x80
        c3[int(offs[40]) + 5] ^= 0x80
        assert oracle.compare_container(fmt, om, data, ways, chunk, c3, offs, lens, threads=3) == (len(lens), 17)


def test_reference_loops_of_the_cpu_baseline_are_the_oracles_streams(oracle, ref):
    """bench.py's per-configuration CPU baseline runs the reference's OWN 2-way loops (main.cpp:226-280, main64.cpp:228-282,
    main_alias.cpp:353-405) and the 8-way word loop with its SSE4.1 decoder (main_simd.cpp:287-332) through
    oracle/_ref: every shard must round-trip ("decode ok!") and its stream must have exactly the size the oracle's 2-way /
    8-way stream of the same symbols has -- the timed loops are the format's loops, not look-alikes."""
    from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD
    if not ref.has_loop2():
        import pytest
        pytest.skip("oracle/_ref predates ref_time_loop2_mt")
    data = oracle.gen_zipf(4 * 100001, K=256, s=1.0, seed=11)
    for which, sb, ways in ((FMT_BYTE, 14, 2), (FMT_R64, 14, 2), (FMT_ALIAS, 16, 2), (FMT_WORD, 12, 8)):
        f, _ = oracle.normalize(oracle.count_freqs(data, 256), 1 << sb)
        om = oracle.model(f, sb, with_alias=(which == FMT_ALIAS))
        res = ref.time_loop2(which, f, sb, data, 100001, threads=4)
        for t, r in enumerate(res):
            assert r["ok"] and r["enc_s"] > 0 and r["dec_s"] > 0 and r["enc_clocks"] > 0 and r["dec_clocks"] > 0
            assert r["stream_bytes"] == oracle.encode(which, om, data[t * 100001:(t + 1) * 100001], ways).size, (which, t)
    d16 = oracle.gen_zipf(2 * 50001, K=4096, s=1.0, seed=3)
    f, _ = oracle.normalize(oracle.count_freqs(d16, 4096), 1 << 16)
    om = oracle.model(f, 16, with_alias=True)
    for t, r in enumerate(ref.time_loop2(12, f, 16, d16, 50001, threads=2)):
        assert r["ok"] and r["stream_bytes"] == oracle.encode(FMT_ALIAS, om, d16[t * 50001:(t + 1) * 50001], 2).size


def test_judged_line_is_short_and_complete():
    """bench.py's LAST stdout line is what the driver parses, and the driver keeps only the last few KB of stdout: round 4's
    20.6 KB line came back `parsed: null`.  judged_line() is a pure function of the full record; here it runs on a canned
    full record (this round's, from the GPU box: gpurun_out -> tests/golden/bench_record_r06.json, probe matrices stripped),
    on an 8-rank record and on a worst case, and must stay under bench.MAX_LINE_BYTES with every field the contract and the
    judge ask for -- and with ONE clock: value, ms_per_step, roofline.frac and value_first_pair all come from timed steps."""
    import copy

    import bench
    full = json.load(open(os.path.join(HERE, "golden", "bench_record_r06.json")))
    line = bench.judged_line(full, "bench_details.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.MAX_LINE_BYTES <= 3800, len(text)
    assert "\n" not in text and json.loads(text) == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "clocks", "configs", "details"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["config"]["workload"] == full["config"]["workload"] and "model" not in line["config"]
    rl = line["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and rl["unit"] == "GB/s" and "traffic" in rl and "traffic_source" in rl
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    # one clock: the line's frac is the algorithmic bytes over ms_per_step (what the judge recomputes), the HIP-event figure beside it
    assert abs(rl["frac"] - rl["algorithmic_bytes_per_launch"] / line["ms_per_step"] / 1e6 / 8000.0) < 2e-4
    assert abs(rl["frac_kernel_events"] - rl["algorithmic_bytes_per_launch"] / rl["kernel_ms_avg"] / 1e6 / 8000.0) < 2e-4
    assert rl["frac_kernel_events"] >= rl["frac"]  # (events bracket the launches, the steps also hold the gaps between them)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] == round(full["cpu_baseline"]["value"], 3) and cb["cores"] >= 1 and cb["sample"]
    assert "16 MiB shards" in cb["sample"]                      # the first 256 MiB, as the `configs` rows
    assert cb["port_value"] > cb["value"]                       # the AVX-512 port is extra, never `value`
    # the un-probed placement beside the chosen one: K timed steps of the same loop, no matrix
    pl = line["placement"]
    assert "probe_ms" not in pl and pl["first_pair_ms_per_step"] >= line["ms_per_step"]
    assert line["value_first_pair"] == round((1 << 30) / pl["first_pair_ms_per_step"] / 1e6, 2)
    assert line["value_first_pair"] <= line["value"] and line["frac_first_pair"] <= rl["frac"]
    rows = line["configs"]
    assert [r["name"] for r in rows] == ["C3-word64", "C2-r64x2", "C4-alias4096", "byte14", "byte12", "word8", "byte2",
                                         "word-adaptive", "byte-adaptive", "word128", "word256"]
    for r, e in zip(rows, full["configs"]):
        assert r["decode_ms"] == e["decode"]["ms_mean"] and r["decode_frac"] == e["decode"]["frac"] and r["oracle_ok"] is True
        if e.get("per_chunk_models"):  # one kernel: count + normalise + code; the container is sized from the chunks' own histograms
            assert r["enc_tight_ms"] == e["encode"]["ms_mean"] and r["enc_tight_frac"] == e["encode"]["frac"] and "encode_ms" not in r
            assert r["tight_size"] == e["encode"]["container_over_input"]
        else:
            assert r["encode_ms"] == e["encode"]["ms_mean"] and r["enc_tight_ms"] == e["encode_tight"]["ms_mean"]
            assert r["dec_tight_ms"] == e["decode_tight"]["ms_mean"] and r["tight_size"] == e["encode_tight"]["container_over_input"]
            assert r["cpu_GBps"] == e["cpu_baseline"]["value"]
    assert line["cpu_ref_cores"] == full["configs"][0]["cpu_baseline"]["cores"]
    # the probe chose the first pair itself: the headline IS the first pair
    same = copy.deepcopy(full)
    same["placement"]["chosen"] = [0, 0]
    same["placement"].pop("first_pair_ms_per_step", None)
    assert bench.judged_line(same)["value_first_pair"] == round((1 << 30) / same["ms_per_step"] / 1e6, 2)
    # a config whose oracle check did not cover every chunk is not "ok"
    broken = copy.deepcopy(full)
    broken["configs"][1]["oracle_chunks_checked"] -= 1
    assert bench.judged_line(broken)["configs"][1]["oracle_ok"] is False
    broken = copy.deepcopy(full)
    broken["configs"][7]["oracle_chunks_checked"] -= 1
    assert bench.judged_line(broken)["configs"][7]["oracle_ok"] is False
    # a measured traffic figure and its source fit as well
    tr = copy.deepcopy(full)
    tr["roofline"]["traffic"] = 2007767255.2727275
    tr["roofline"]["traffic_source"] = "profiles/r06_traffic.json (FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate --pmc passes, same kernel sources)"
    t2 = bench.judged_line(tr, "bench_details.json")
    assert t2["roofline"]["traffic_source"] == "profiles/r06_traffic.json"
    assert len(json.dumps(t2, separators=(",", ":"))) <= bench.MAX_LINE_BYTES
    # eight ranks (BASELINE configs[4]; the driver's SCALE run): no `configs`, per-rank kernel times, the summed oracle sample
    eight = copy.deepcopy(full)
    eight.pop("configs")
    eight["n_gpus"] = 8
    eight["value"] = 8 * full["value"]
    eight["per_rank"] = {"kernel_ms": [0.38123 + 0.001 * i for i in range(8)], "elapsed_ms_per_step": [0.39] * 8,
                         "stream_bytes": [841767114] * 8, "roofline_frac": [0.62] * 8}
    eight["oracle_chunks_checked"] = 8 * 266
    eight["oracle_chunks_total"] = 8 * 32768
    eight["oracle_chunks_checked_per_rank"] = [266] * 8
    eight["placement"].pop("first_pair_ms_per_step", None)
    l8 = bench.judged_line(eight, "bench_details.json")
    t8 = json.dumps(l8, separators=(",", ":"))
    assert len(t8) <= bench.MAX_LINE_BYTES and l8["n_gpus"] == 8 and len(l8["per_rank_kernel_ms"]) == 8
    assert "cpu_baseline" in l8 and "roofline" in l8 and "frac_job" in l8["roofline"] and "value_first_pair" not in l8
    assert json.loads(t8[-4096:] if len(t8) > 4096 else t8) == l8  # (parsable from the last 4 KB of stdout)
    # worst case
    worst = copy.deepcopy(full)
    worst["configs"] = (worst["configs"] * 2)[:16]
    worst["n_gpus"] = 8
    worst["per_rank"] = {"kernel_ms": [0.38123] * 8}
    worst["error"] = "round trip mismatch, corrupt chunk reported, or a chunk differs from the oracle"
    worst["knobs"] = {"RANS_AMD_LIB": "x" * 100}
    assert len(json.dumps(bench.judged_line(worst, "bench_details.json"), separators=(",", ":"))) <= bench.MAX_LINE_BYTES
