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
    full record (round 4's, renamed to this round's keys) and on a worst case (12 configs with every optional key, 8 ranks,
    an error string), and must stay under 4096 bytes with every field the contract and the judge ask for."""
    import copy

    import bench
    full = json.load(open(os.path.join(HERE, "golden", "bench_record_r04.json")))
    line = bench.judged_line(full, "bench_details.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.MAX_LINE_BYTES < 4096, len(text)
    assert "\n" not in text and json.loads(text) == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "clocks", "configs", "details"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["config"]["workload"] == full["config"]["workload"] and "model" not in line["config"]
    rl = line["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and rl["unit"] == "GB/s" and rl["traffic"] == full["roofline"]["traffic"]
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] == round(full["cpu_baseline"]["value"], 3) and cb["cores"] >= 1 and cb["sample"]
    assert cb["port_value"] > cb["value"]                       # the AVX-512 port is extra, never `value`
    # the un-probed placement beside the chosen one, and no matrix
    assert line["placement"]["first_pair_ms"] >= line["placement"]["chosen_ms"] and "probe_ms" not in line["placement"]
    assert line["value_first_pair"] == round((1 << 30) / line["placement"]["first_pair_ms"] / 1e6, 2)
    assert line["value_first_pair"] <= line["value"] * 1.02
    rows = line["configs"]
    assert [r["name"] for r in rows] == ["C3-word64", "C2-r64x2", "C4-alias4096", "byte14", "byte12", "word128", "word256"]
    for r, e in zip(rows, full["configs"]):
        assert r["decode_ms"] == e["decode"]["ms_mean"] and r["encode_ms"] == e["encode"]["ms_mean"]
        assert r["enc_slots_ms"] == e["encode_slots"]["ms_mean"] and r["oracle_ok"] is True
        assert r["cpu_ref_GBps"] == e["cpu_baseline"]["value"]
    # a config whose oracle check did not cover every chunk is not "ok"
    broken = copy.deepcopy(full)
    broken["configs"][1]["oracle_chunks_checked"] -= 1
    assert bench.judged_line(broken)["configs"][1]["oracle_ok"] is False
    # worst case
    worst = copy.deepcopy(full)
    for e in worst["configs"]:
        e["encode_tight"] = dict(e["encode_slots"], container_over_input=0.8512)
        e["decode_tight"] = e["decode_slots"]
    worst["configs"] = (worst["configs"] * 2)[:12]
    worst["n_gpus"] = 8
    worst["per_rank"] = {"kernel_ms": [0.38123] * 8}
    worst["error"] = "round trip mismatch, corrupt chunk reported, or a chunk differs from the oracle"
    worst["knobs"] = {"RANS_AMD_LIB": "x" * 100}
    assert len(json.dumps(bench.judged_line(worst, "bench_details.json"), separators=(",", ":"))) < 4096
