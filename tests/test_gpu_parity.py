"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
C ABI, against the CPU oracle on the same seeded inputs -- bit-exact both ways:

  * encode: every chunk the GPU encoder emits equals the oracle's stream for
    that chunk byte for byte (so it equals the reference's N-way loop output);
  * decode: containers produced by the ORACLE decode on the GPU to the input;
  * committed fixtures made by the unmodified reference (tests/golden/) decode
    on the GPU and are reproduced by the GPU encoder;
  * edge cases the reference exercises: n not a multiple of N (tail round),
    n < N, odd n, symbols that occur once (freq 1), 174 unused symbols,
    ragged last chunk, one-chunk == raw reference stream;
  * corruption is detected, never turns into an out-of-bounds access.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD, Oracle

HERE = os.path.dirname(os.path.abspath(__file__))

FORMATS = [(FMT_WORD, 12), (FMT_BYTE, 14), (FMT_BYTE, 16), (FMT_R64, 14), (FMT_ALIAS, 16), (FMT_ALIAS, 12)]


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    import ryg_rans_amd as R
    ctx = R.Context(0)
    yield R, ctx, torch
    ctx.close()


def _inputs(oracle):
    rng = np.random.default_rng(7)
    text_like = np.minimum(rng.geometric(0.06, 200003) - 1, 255).astype(np.uint8)
    return {
        "zipf": oracle.gen_zipf(300000, K=256, s=1.0, seed=1),
        "skew": text_like,
        "rare": np.concatenate([np.full(50000, 65, np.uint8), np.arange(256, dtype=np.uint8),
                                np.full(50001, 66, np.uint8)]),
        "two": (rng.integers(0, 2, 40000) * 255).astype(np.uint8),
    }


def _models(R, ctx, oracle, fmt, sb, data, nsyms=256):
    counts = oracle.count_freqs(data, nsyms)
    f, _ = oracle.normalize(counts, 1 << sb)
    om = oracle.model(f, sb, with_alias=(fmt == FMT_ALIAS))
    gm = ctx.model(fmt, f, sb)
    return om, gm


@pytest.mark.parametrize("fmt,sb", FORMATS)
@pytest.mark.parametrize("n_ways", [64, 256, 1, 2, 8, 33, 128, 512, 100, 200, 300, 511, 192, 320, 384, 448])
def test_single_stream_matches_oracle(gpu, oracle, fmt, sb, n_ways):
    """chunk_syms >= n: the container is the raw reference-format stream."""
    R, ctx, torch = gpu
    for name, data in _inputs(oracle).items():
        data = data[:60001]
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want = oracle.encode(fmt, om, data, n_ways)
        got = ctx.encode_host(gm, data, n_ways)
        assert got.size == want.size and np.array_equal(got, want), (name, "encode")
        out = ctx.decode_host(gm, want, data.size, n_ways)
        assert np.array_equal(out, data), (name, "decode")


@pytest.mark.parametrize("fmt,sb", FORMATS)
@pytest.mark.parametrize("n_ways,chunk_syms", [(64, 4096), (64, 5000), (256, 16384), (32, 1000), (128, 4096),
                                               (2, 512), (1, 1000), (4, 2048), (8, 4096), (2, 4095)])
def test_chunked_matches_oracle(gpu, oracle, fmt, sb, n_ways, chunk_syms):
    R, ctx, torch = gpu
    data = _inputs(oracle)["zipf"]
    om, gm = _models(R, ctx, oracle, fmt, sb, data)
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk_syms, align=16)

    # GPU decode of the oracle's container
    d_cont = torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint8)])).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
    d_out = ctx.decode(gm, d_cont, cont.size, d_offs, d_lens, data.size, n_ways, chunk_syms)
    assert np.array_equal(d_out.cpu().numpy(), data)

    # GPU encode: same index, same bytes
    d_syms = torch.from_numpy(data).cuda()
    g_cont, g_offs, g_lens, total = ctx.encode(gm, d_syms, n_ways, chunk_syms)
    assert total == cont.size
    assert np.array_equal(g_offs.cpu().numpy().astype(np.uint64), offs)
    assert np.array_equal(g_lens.cpu().numpy().astype(np.uint32), lens)
    g = g_cont.cpu().numpy()
    for c in range(len(lens)):
        a, b = int(offs[c]), int(offs[c]) + int(lens[c])
        assert np.array_equal(g[a:b], cont[a:b]), "chunk %d differs" % c

    # and the GPU decodes its own container
    d_out2 = ctx.decode(gm, g_cont, total, g_offs, g_lens, data.size, n_ways, chunk_syms)
    assert torch.equal(d_out2, d_syms)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 255, 256, 257, 4099])
def test_tiny_and_ragged_sizes(gpu, oracle, n):
    R, ctx, torch = gpu
    data = oracle.gen_zipf(5000, K=256, s=1.0, seed=3)[:n]
    for fmt, sb in FORMATS:
        if len(np.unique(data)) < 2 and (fmt == FMT_WORD or (fmt == FMT_ALIAS and sb == 16)):
            continue  # one-symbol models: the word format rejects them, the device alias tables stop at scale_bits 15
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        for n_ways in (64, 256, 3):
            want = oracle.encode(fmt, om, data, n_ways)
            assert np.array_equal(ctx.encode_host(gm, data, n_ways), want), (fmt, n_ways)
            assert np.array_equal(ctx.decode_host(gm, want, n, n_ways), data), (fmt, n_ways)


def test_alias_4096_symbols(gpu, oracle):
    """Config 4: 4096-symbol alphabet, 16-bit probabilities, u16 symbols, 64-way."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(400001, K=4096, s=1.0, seed=1)
    om, gm = _models(R, ctx, oracle, FMT_ALIAS, 16, data, nsyms=4096)
    for n_ways, chunk in ((64, 8192), (256, 400001)):
        cont, offs, lens = oracle.encode_chunked(FMT_ALIAS, om, data, n_ways, chunk, align=16)
        d_syms = torch.from_numpy(data.view(np.int16)).cuda()
        g_cont, g_offs, g_lens, total = ctx.encode(gm, d_syms, n_ways, chunk)
        assert total == cont.size
        g = g_cont.cpu().numpy()
        assert np.array_equal(g_offs.cpu().numpy().astype(np.uint64), offs)
        for c in range(len(lens)):  # bytes between chunks are alignment padding, not stream
            a, b = int(offs[c]), int(offs[c]) + int(lens[c])
            assert np.array_equal(g[a:b], cont[a:b]), "chunk %d differs" % c
        d_out = ctx.decode(gm, g_cont, total, g_offs, g_lens, data.size, n_ways, chunk)
        assert np.array_equal(d_out.cpu().numpy().view(np.uint16), data)
        # round 3: this model's tables (128 KiB of alias_remap + 32 KiB of records) fill the CU's LDS to the last byte, and the
        # kernel places its chunks itself all the same -- its mailbox lives in global memory
        assert ctx.last_encode_kernel() == ("k_encode<alias, LDS remap>", True), ctx.last_encode_kernel()


def test_reference_fixtures(gpu):
    """Streams written by the UNMODIFIED reference (tests/golden/make_golden.py)."""
    R, ctx, torch = gpu
    idx_path = os.path.join(HERE, "golden", "ref_streams.json")
    if not os.path.exists(idx_path):
        pytest.skip("fixtures not generated")
    idx = json.load(open(idx_path))
    blob = np.fromfile(os.path.join(HERE, "golden", idx["blob"]), dtype=np.uint8)
    data = blob[idx["input"][0]: idx["input"][0] + idx["input"][1]]
    for e in idx["streams"]:
        stream = blob[e["offset"]: e["offset"] + e["size"]]
        freqs = np.array(e["freqs"], dtype=np.uint32)
        gm = ctx.model(e["fmt"], freqs, e["scale_bits"])
        assert np.array_equal(ctx.decode_host(gm, stream, data.size, e["n_ways"]), data), e["name"]
        assert np.array_equal(ctx.encode_host(gm, data, e["n_ways"]), stream), e["name"]


def test_corruption_is_detected(gpu, oracle):
    R, ctx, torch = gpu
    data = oracle.gen_zipf(100000, K=256, s=1.0, seed=9)
    for fmt, sb in ((FMT_WORD, 12), (FMT_BYTE, 14), (FMT_R64, 14)):
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        good = oracle.encode(fmt, om, data, 64)
        # truncated stream
        out, rc = ctx.decode_host(gm, good[:-8], data.size, 64, check=False)
        assert rc == R.E_CORRUPT
        # flipped bit in the payload: either flagged or (rarely) decodes to other data, never crashes
        bad = good.copy()
        bad[len(bad) // 3] ^= 0x40
        out, rc = ctx.decode_host(gm, bad, data.size, 64, check=False)
        assert rc == R.E_CORRUPT or not np.array_equal(out, data)
        # wrong n
        out, rc = ctx.decode_host(gm, good, data.size - 64, 64, check=False)
        assert rc == R.E_CORRUPT


def test_model_errors(gpu, oracle):
    R, ctx, torch = gpu
    f = np.zeros(256, np.uint32)
    f[7] = 4096
    with pytest.raises(R.RansAmdError) as e:
        ctx.model(FMT_WORD, f, 12)
    assert e.value.status == R.E_MODEL
    f[7] = 4000
    with pytest.raises(R.RansAmdError):
        ctx.model(FMT_WORD, f, 12)  # does not sum to M
    # symbol with frequency 0 in the input
    f[8] = 96
    gm = ctx.model(FMT_WORD, f, 12)
    data = np.array([7, 8, 9, 7] * 100, dtype=np.uint8)
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_host(gm, data, 64)
    assert e.value.status == R.E_MODEL


def test_device_histogram(gpu, oracle):
    R, ctx, torch = gpu
    data = oracle.gen_zipf(1 << 20, K=256, s=1.0, seed=4)
    got = ctx.count_freqs_device(torch.from_numpy(data).cuda(), 256)
    assert np.array_equal(got, oracle.count_freqs(data, 256))
    data16 = oracle.gen_zipf(300001, K=4096, s=1.0, seed=4)
    got = ctx.count_freqs_device(torch.from_numpy(data16.view(np.int16)).cuda(), 4096)
    assert np.array_equal(got, oracle.count_freqs(data16, 4096))
    # ragged head / tail around the 16-byte vector body, tiny and empty inputs
    d8 = torch.from_numpy(data).cuda()
    d16 = torch.from_numpy(data16.view(np.int16)).cuda()
    for lo, hi in ((1, 1 << 20), (3, 70001), (15, 16), (5, 5), (7, 38), (16, 1 << 19)):
        assert np.array_equal(ctx.count_freqs_device(d8[lo:hi], 256), oracle.count_freqs(data[lo:hi], 256)), (lo, hi)
    for lo, hi in ((1, 300001), (3, 4), (7, 7), (5, 20), (8, 300000)):
        assert np.array_equal(ctx.count_freqs_device(d16[lo:hi], 4096),
                              oracle.count_freqs(data16[lo:hi], 4096)), (lo, hi)
    # a symbol outside the declared alphabet is an error, not a silent drop (both widths)
    small = (data & 63).copy()
    small[12345] = 64
    with pytest.raises(R.RansAmdError) as e:
        ctx.count_freqs_device(torch.from_numpy(small).cuda(), 64)
    assert e.value.status == R.E_ARG
    small[12345] = 63
    assert np.array_equal(ctx.count_freqs_device(torch.from_numpy(small).cuda(), 64), oracle.count_freqs(small, 64))
    bad16 = data16.copy()
    bad16[77] = 4096
    with pytest.raises(R.RansAmdError):
        ctx.count_freqs_device(torch.from_numpy(bad16.view(np.int16)).cuda(), 4096)


def test_full_size_roundtrip_properties(gpu):
    """BASELINE size (1 GiB): encode -> decode round trip on device, checksum of the
    output equals checksum of the input, every chunk passes its integrity check."""
    R, ctx, torch = gpu
    n = 1 << 30
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    w = 1.0 / torch.arange(1, 257, dtype=torch.float64, device="cuda")
    cdf = torch.cumsum(w / w.sum(), 0).float()
    d_syms = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 26
    for i in range(0, n, step):
        u = torch.rand(step, device="cuda", generator=g)
        d_syms[i:i + step] = torch.searchsorted(cdf, u).clamp_(max=255).to(torch.uint8)
    counts = ctx.count_freqs_device(d_syms, 256)
    assert int(counts.sum()) == n
    f, _ = R.normalize_freqs(counts, 4096)
    gm = ctx.model(FMT_WORD, f, 12)
    chunk = 32768
    cont, offs, lens, total = ctx.encode(gm, d_syms, 64, chunk)
    assert 0.5 * n < total < n
    out = ctx.decode(gm, cont, total, offs, lens, n, 64, chunk)
    assert torch.equal(out, d_syms)
    # a corrupted copy of the container is flagged
    cont2 = cont.clone()
    cont2[total // 2] ^= 1
    out2 = torch.empty_like(out)
    ctx.decode(gm, cont2, total, offs, lens, n, 64, chunk, d_out=out2, sync=False)
    assert ctx.decode_errors() >= 1 or not torch.equal(out2, d_syms)


def test_native_cpp_example(gpu):
    """examples/roundtrip.cpp: a C++ host using only the C ABI (built by __graft_entry__.build())."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "build", "roundtrip")
    if not os.path.exists(exe):
        pytest.skip("build/roundtrip not built")
    for fmt, ways in (("word", "64"), ("byte", "64"), ("r64", "2"), ("alias", "256")):
        out = subprocess.run([exe, "-", fmt, ways], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "decode ok!" in out.stdout, (fmt, out.stdout, out.stderr)


def test_native_multi_gpu_example(gpu):
    """examples/multi_gpu.cpp (built by __graft_entry__.build()): one host thread and one context per device, independent
    shards, and RCCL itself -- ncclCommInitAll + ncclAllGather of the 40-byte per-rank records, through rccl.h, no
    torch -- run on the devices this box has (one): the collective of the N-GPU run executes at least here."""
    import json
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "build", "multi_gpu")
    if not os.path.exists(exe):
        pytest.skip("build/multi_gpu not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "8", "24", "5"], capture_output=True, text=True, timeout=600, env=env)  # (8: clipped to what is visible)
    assert out.returncode == 0 and "decode ok!" in out.stdout, (out.stdout, out.stderr)
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] >= 1 and line["bit_exact_roundtrip"] is True and line["records_gathered_by"] == "ncclAllGather"
    assert line["value"] > 0 and 0.0 < line["frac_job"] < 1.0 and "roofline frac" in out.stdout
    # the caller chose its buffers through the ABI's placement probe (rans_amd_probe_placement): first pair vs chosen pair
    import re
    m = re.search(r"rank 0 placement: first pair ([0-9.]+) ms, chosen pair ([0-9.]+) ms", out.stdout)
    assert m and 0 < float(m.group(2)) <= float(m.group(1)), out.stdout
    # --split-one: ONE container split by chunk range over three ranks (they share this box's one GPU, a context each),
    # every rank holding only the bytes rans_amd_container_slice assigns it; the pieces side by side are the input
    out = subprocess.run([exe, "--split-one", "3", "24", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "decode ok!" in out.stdout, (out.stdout, out.stderr)
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["ranks"] == 3 and line["mode"] == "one container split by chunk range" and line["bit_exact_roundtrip"] is True
    rows = [ln.split() for ln in out.stdout.splitlines() if ln.strip() and ln.split()[0] in ("0", "1", "2")]
    assert len(rows) == 3 and sum(float(r[1]) for r in rows) == float(1 << 24)  # the three ranges cover the input once


def test_placement_probe_through_the_abi(gpu, oracle):
    """rans_amd_probe_placement: every (container copy, output) pair is decoded and timed, the fastest pair is named, the
    matrix is what was measured, every output holds the symbols; a corrupt copy among the candidates is reported."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(1 << 22, K=256, s=1.0, seed=31)
    om, gm = _models(R, ctx, oracle, FMT_WORD, 12, data)
    d = torch.from_numpy(data).cuda()
    cont, offs, lens, total = ctx.encode(gm, d, 64, 16384)
    conts = [cont, cont.clone()]
    outs = [torch.zeros_like(d) for _ in range(3)]
    bi, bj, ms = ctx.probe_placement(gm, conts, total, offs, lens, d.numel(), 64, 16384, outs, launches=3, sweeps=2)
    assert 0 <= bi < 2 and 0 <= bj < 3 and len(ms) == 2 and all(len(r) == 3 for r in ms)
    assert all(v > 0 for r in ms for v in r) and ms[bi][bj] == min(v for r in ms for v in r)
    assert all(torch.equal(o, d) for o in outs)
    conts[1][int(offs[5]) + 40] ^= 0x5A
    with pytest.raises(R.RansAmdError) as e:
        ctx.probe_placement(gm, conts, total, offs, lens, d.numel(), 64, 16384, outs, launches=1, sweeps=1)
    assert e.value.status == R.E_CORRUPT
    with pytest.raises(R.RansAmdError) as e:
        ctx.probe_placement(gm, [], total, offs, lens, d.numel(), 64, 16384, outs)
    assert e.value.status == R.E_ARG


def test_unaligned_buffers_and_streams(gpu, oracle):
    """Symbol buffers at odd addresses take the element-wise paths; containers must be 16-byte
    aligned (E_ARG otherwise); work on a non-default HIP stream."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(200003, K=256, s=1.0, seed=12)
    for fmt, sb in ((FMT_WORD, 12), (FMT_BYTE, 14)):
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want, offs, lens = oracle.encode_chunked(fmt, om, data, 64, 8192, align=16)
        backing = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
        for shift in (1, 2, 3, 5):
            d_syms = backing[shift:shift + data.size]
            d_syms.copy_(torch.from_numpy(data))
            assert d_syms.data_ptr() % 4 == shift % 4
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                cont, d_offs, d_lens, total = ctx.encode(gm, d_syms, 64, 8192)
                assert total == want.size
                assert np.array_equal(cont[:total].cpu().numpy()[int(offs[3]):int(offs[3]) + int(lens[3])],
                                      want[int(offs[3]):int(offs[3]) + int(lens[3])])
                out_backing = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
                d_out = out_backing[shift:shift + data.size]
                ctx.decode(gm, cont, total, d_offs, d_lens, data.size, 64, 8192, d_out=d_out)
            side.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), data), (fmt, shift)
            assert int(out_backing[:shift].sum()) == 0 and int(out_backing[shift + data.size:].sum()) == 0
        # a container that does not start on a 16-byte boundary is refused, not mis-decoded
        cont2 = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        cont2[8:8 + total] = cont[:total]
        with pytest.raises(R.RansAmdError) as e:
            ctx.decode(gm, cont2[8:], total, d_offs, d_lens, data.size, 64, 8192)
        assert e.value.status == R.E_ARG


def test_random_corruption_never_crashes(gpu, oracle):
    """Fuzz: random byte flips / truncations / index damage.  The decoder may return garbage
    symbols but must flag or survive every case (all table and window indices are masked)."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(99)
    data = oracle.gen_zipf(150000, K=256, s=1.0, seed=13)
    for fmt, sb, n_ways, chunk in ((FMT_WORD, 12, 64, 4096), (FMT_BYTE, 14, 64, 4096), (FMT_R64, 14, 2, 512),
                                   (FMT_ALIAS, 16, 128, 8192)):
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        for trial in range(12):
            bad = cont.copy()
            l2 = lens.copy()
            kind = trial % 3
            if kind == 0:
                idx = rng.integers(0, bad.size, 40)
                bad[idx] ^= rng.integers(1, 256, 40).astype(np.uint8)
            elif kind == 1:
                bad[rng.integers(0, bad.size):] = 0
            else:
                l2[rng.integers(0, l2.size, 3)] = rng.integers(0, 1 << 20, 3).astype(np.uint32)
            d_cont = torch.from_numpy(np.concatenate([bad, np.zeros(64, np.uint8)])).cuda()
            d_lens = torch.from_numpy(l2.astype(np.int32)).cuda()
            out = ctx.decode(gm, d_cont, bad.size, d_offs, d_lens, data.size, n_ways, chunk, sync=False)
            nbad = ctx.decode_errors()
            same = np.array_equal(out.cpu().numpy(), data)
            assert nbad > 0 or same, (fmt, trial)


def test_word_encoder_full_wave_path_extremes(gpu, oracle):
    """The hand-written word encoder sub-step (full waves, >= 16 rounds): every frequency class
    of its division-free update -- freq 1 (q = x - 1 identity), powers of two, the largest
    frequencies (dividend >= 2^31), a 4095/1 split -- against the oracle byte for byte, for 64-,
    128- and 256-way, and a symbol without a record anywhere in a chunk is E_MODEL."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(21)
    n = 1 << 17
    cases = []
    f = np.zeros(256, np.uint32); f[0] = 4095; f[200] = 1
    cases.append(("4095/1", f, rng.choice([0, 200], n, p=[0.999, 0.001]).astype(np.uint8)))
    f = np.zeros(256, np.uint32); f[:16] = [2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1, 1, 0, 0, 0]; f[15] = 0
    f[13] = 0; f[14] = 0
    assert f.sum() == 4096
    cases.append(("powers of two", f, rng.choice(13, n, p=f[:13] / 4096.0).astype(np.uint8)))
    f = np.ones(256, np.uint32); f[3] = 4096 - 255
    cases.append(("3841 + 255 x 1", f, rng.choice(256, n, p=f / 4096.0).astype(np.uint8)))
    # twelve bits per symbol, every symbol: 1536 bytes per sixteen rounds of a wave -- the staged words need both passes
    # of the flush (encode_wave.hip stage_flush); and the other end, a stream of almost nothing but flushed states
    rare = rng.integers(0, 255, n).astype(np.uint8)
    rare[rare >= 3] += 1
    cases.append(("255 x 1 only", f, rare))
    cases.append(("3841 only", f, np.full(n, 3, np.uint8)))
    mixed = rare.copy()
    mixed[(np.arange(n) // 2048) % 2 == 0] = 3  # bursts: pieces of every size, flushes with nothing to write
    cases.append(("bursts", f, mixed))
    f = np.zeros(256, np.uint32); f[10] = 2049; f[11] = 2047
    cases.append(("2049/2047", f, rng.integers(10, 12, n).astype(np.uint8)))
    f = np.zeros(256, np.uint32); f[:5] = [3, 5, 7, 4081 - 1365, 1365]
    cases.append(("odd", f, rng.choice(5, n, p=f[:5] / 4096.0).astype(np.uint8)))
    for name, f, data in cases:
        om = oracle.model(f, 12)
        gm = ctx.model(FMT_WORD, f, 12)
        d = torch.from_numpy(data).cuda()
        # (64-way with ragged chunks: rounds beyond the last full sixteen store their words themselves, the staging
        #  window is then primed with the 16-byte piece the write offset stands in)
        for n_ways, chunk in ((64, 8192), (128, 16384), (256, 32768), (64, n), (64, 5000), (64, 1031), (64, 2048 + 64 * 7 + 5)):
            want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, n_ways, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d, n_ways, chunk)
            assert total == want.size, (name, n_ways)
            assert np.array_equal(d_lens.cpu().numpy().astype(np.int64), lens.astype(np.int64)), (name, n_ways)
            got = cont[:total].cpu().numpy()
            for c in range(len(lens)):
                o, ln = int(offs[c]), int(lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (name, n_ways, c)
            out = ctx.decode(gm, cont, total, d_offs, d_lens, n, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy(), data), (name, n_ways)
    # a symbol the model has no record for, deep inside the full-wave part of a chunk
    name, f, data = cases[1]
    gm = ctx.model(FMT_WORD, f, 12)
    for bad_sym in (13, 255):
        bad = data.copy()
        bad[5 * 8192 + 4321] = bad_sym
        with pytest.raises(R.RansAmdError) as e:
            ctx.encode(gm, torch.from_numpy(bad).cuda(), 64, 8192)
        assert e.value.status == R.E_MODEL


def test_many_chunks_layout(gpu, oracle):
    """More than 8192 chunks: the offset scan runs on several blocks (k_layout_sums + k_layout)."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf((1 << 21) + 77, K=256, s=1.0, seed=5)
    for fmt, sb, n_ways, chunk in ((FMT_R64, 14, 2, 128), (FMT_WORD, 12, 64, 64), (FMT_BYTE, 14, 1, 96)):
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
        assert len(lens) > 2 * 8192
        cont, d_offs, d_lens, total = ctx.encode(gm, torch.from_numpy(data).cuda(), n_ways, chunk)
        assert total == want.size
        assert np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs)
        assert np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens)
        got = cont[:total].cpu().numpy()
        for c in (0, 1, 8191, 8192, 8193, 16383, 16384, len(lens) - 2, len(lens) - 1):
            o, ln = int(offs[c]), int(lens[c])
            assert np.array_equal(got[o:o + ln], want[o:o + ln]), (fmt, c)
        out = ctx.decode(gm, cont, total, d_offs, d_lens, data.size, n_ways, chunk)
        assert np.array_equal(out.cpu().numpy(), data)


@pytest.mark.parametrize("generation", ["auto", "auto+fused"])
def test_lane_kernels_both_generations(gpu, oracle, generation):
    """Narrow interleaves (N = 1, 2, 4, 8): the wave-cooperative staged kernels and -- for the shapes only they serve: chunk
    sizes that are not multiples of 16, a handful of batches -- the first generation's per-lane kernels (the named fallback;
    the option that pinned a generation was retired in round 6), every format, ragged last chunk, against the oracle byte
    for byte."""
    R, _, torch = gpu
    ctx = R.Context(0)  # (its own context: the options must not leak into the other tests)
    ctx.set_option(R.OPT_LANE_KERNELS, 0)
    with pytest.raises(R.RansAmdError) as e:  # (retired: only "automatic" is left)
        ctx.set_option(R.OPT_LANE_KERNELS, 2)
    assert e.value.status == R.E_UNSUPPORTED
    # "+fused": the staged encoders placing their chunks themselves (no k_layout / k_compact_small)
    ctx.set_option(R.OPT_LANE_FUSED_PLACEMENT, int(generation.endswith("+fused")))
    data = oracle.gen_zipf(200000 + 37, K=256, s=1.0, seed=17)
    d_syms = torch.from_numpy(data).cuda()
    seen = set()
    for fmt, sb in FORMATS:
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        for n_ways, chunk in ((2, 512), (1, 256), (4, 1024), (8, 2048), (2, 80), (2, 1000), (1, 48), (8, 64), (2, 16), (1, 32)):
            want, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d_syms, n_ways, chunk)
            assert total == want.size, (fmt, n_ways, chunk)
            assert np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens), (fmt, n_ways, chunk)
            assert np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs), (fmt, n_ways, chunk)
            seen.add(ctx.last_encode_kernel()[0])
            if generation.endswith("+fused") and chunk % 16 == 0 and ctx.last_encode_kernel()[0] == "k_encode_lanes_staged":
                assert ctx.last_encode_kernel()[1] is True
            got = cont[:total].cpu().numpy()
            for c in (0, 1, len(lens) // 2, len(lens) - 2, len(lens) - 1):
                o, ln = int(offs[c]), int(lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (fmt, n_ways, chunk, c)
            # decode the ORACLE's container (independent of the GPU encoder) and the GPU's own
            d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
            out = ctx.decode(gm, d_cont, want.size, torch.from_numpy(offs.astype(np.int64)).cuda(),
                             torch.from_numpy(lens.astype(np.int32)).cuda(), data.size, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy(), data), (fmt, n_ways, chunk)
            out = ctx.decode(gm, cont, total, d_offs, d_lens, data.size, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy(), data), (fmt, n_ways, chunk)
    assert "k_encode_lanes16" in seen, seen  # (few batches: the per-lane encoder, the named fallback)
    # ... and the staged generation, which takes over from six batches of 64 chunks per CU on: 16-symbol chunks of a larger input
    big = oracle.gen_zipf(6 * 256 * 64 * 16 + 16 * 999 + 5, K=256, s=1.0, seed=18)
    d_big = torch.from_numpy(big).cuda()
    for fmt, sb, n_ways in ((FMT_WORD, 12, 4), (FMT_BYTE, 14, 2), (FMT_R64, 14, 1), (FMT_ALIAS, 16, 4)):  # (word 8-way: encode_groups.hip)
        om, gm = _models(R, ctx, oracle, fmt, sb, big)
        want, offs, lens = oracle.encode_chunked_mt(fmt, om, big, n_ways, 16, align=16)
        cont, d_offs, d_lens, total = ctx.encode(gm, d_big, n_ways, 16)
        assert ctx.last_encode_kernel() == ("k_encode_lanes_staged", generation.endswith("+fused")), ctx.last_encode_kernel()
        assert total == want.size and np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs)
        assert np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens)
        got = cont[:total].cpu().numpy()
        for c in list(range(0, len(lens), 997)) + [len(lens) - 2, len(lens) - 1]:  # (bytes between chunks are alignment padding)
            o, ln = int(offs[c]), int(lens[c])
            assert np.array_equal(got[o:o + ln], want[o:o + ln]), (fmt, n_ways, c)
        out = ctx.decode(gm, cont, total, d_offs, d_lens, big.size, n_ways, 16)
        assert ctx.last_decode_kernel() == "k_decode_lanes_staged" and np.array_equal(out.cpu().numpy(), big)
    # unaligned symbol buffers fall back inside the library
    om, gm = _models(R, ctx, oracle, FMT_R64, 14, data)
    backing = torch.zeros(data.size + 16, dtype=torch.uint8, device="cuda")
    backing[3:3 + data.size] = d_syms
    cont, d_offs, d_lens, total = ctx.encode(gm, backing[3:3 + data.size], 2, 512)
    out_b = torch.zeros(data.size + 16, dtype=torch.uint8, device="cuda")
    ctx.decode(gm, cont, total, d_offs, d_lens, data.size, 2, 512, d_out=out_b[5:5 + data.size])
    assert np.array_equal(out_b[5:5 + data.size].cpu().numpy(), data)


@pytest.mark.parametrize("fmt,sb,n_ways,extra", [(FMT_BYTE, 14, 2, 2), (FMT_WORD, 12, 8, 2)])
def test_lane_decoder_fallback_for_huge_chunks(gpu, oracle, fmt, sb, n_ways, extra):
    """The first generation's per-lane decoder (k_decode_lanes) is the NAMED FALLBACK for what the staged decoder cannot
    take: chunks of 512 Ki symbols and more (its ring positions are 32-bit offsets inside a batch).  64 + 1 chunks of 512 Ki
    symbols in the reference's own narrow layouts: a container the ORACLE made decodes to the input, and the GPU encoder's
    container (wave encoder or per-lane encoder, whichever the shape gets) equals it chunk by chunk.  (Both layouts
    have decoders of their own for chunks of a multiple of 4 symbols, k_decode_word_groups / k_decode_byte_pairs: the
    cases here are 2 symbols off.)"""
    R, ctx, torch = gpu
    chunk = (1 << 19) + extra
    data = oracle.gen_zipf(64 * chunk + 12345, K=256, s=1.0, seed=21)
    om, gm = _models(R, ctx, oracle, fmt, sb, data)
    want, offs, lens = oracle.encode_chunked_mt(fmt, om, data, n_ways, chunk, align=16)
    d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
    out = ctx.decode(gm, d_cont, want.size, torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(),
                     data.size, n_ways, chunk)
    assert ctx.last_decode_kernel() == "k_decode_lanes", ctx.last_decode_kernel()
    assert np.array_equal(out.cpu().numpy(), data)
    cont, d_offs, d_lens, total = ctx.encode(gm, torch.from_numpy(data).cuda(), n_ways, chunk)
    assert total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens)
    got = cont[:total].cpu().numpy()
    for c in range(len(lens)):  # (bytes between chunks are alignment padding, not stream)
        o, ln = int(offs[c]), int(lens[c])
        assert np.array_equal(got[o:o + ln], want[o:o + ln]), c
    out = ctx.decode(gm, cont, total, d_offs, d_lens, data.size, n_ways, chunk)
    assert np.array_equal(out.cpu().numpy(), data)


def test_model_may_outlive_its_context(gpu, oracle):
    """Destroying a context before its models (Python GC order does that) must be harmless: the
    model frees its device tables on its own device, and no sticky HIP error is left behind for the
    next launch to trip over."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(70000, K=256, s=1.0, seed=23)
    ctx2 = R.Context(0)
    om, gm2 = _models(R, ctx2, oracle, FMT_WORD, 12, data)
    ctx2.close()
    gm2.close()
    om, gm = _models(R, ctx, oracle, FMT_WORD, 12, data)
    cont, offs, lens, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 64, 4096)
    out = ctx.decode(gm, cont, total, offs, lens, data.size, 64, 4096)
    assert np.array_equal(out.cpu().numpy(), data)


def _ref_or_none():
    from _oracle import Ref
    return Ref() if Ref.available() else None


def test_empty_input_is_n_flushed_states(gpu, oracle):
    """n == 0: the reference's loops run zero times and flush N untouched states (rans_byte.h:93-105 etc.);
    the host entry points produce and accept exactly that stream."""
    R, ctx, torch = gpu
    ref = _ref_or_none()
    data = oracle.gen_zipf(1000, K=256, s=1.0, seed=1)
    empty = np.zeros(0, np.uint8)
    for fmt, sb in FORMATS:
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        for n_ways in (1, 2, 64, 100, 512):
            want = oracle.encode(fmt, om, empty, n_ways)
            if ref is not None and n_ways <= 64:
                assert np.array_equal(ref.encode(fmt, om.freqs, sb, empty, n_ways), want)
            got = ctx.encode_host(gm, empty, n_ways)
            assert np.array_equal(got, want), (fmt, n_ways)
            assert ctx.decode_host(gm, want, 0, n_ways).size == 0
            bad = want.copy()
            bad[-1] ^= 1
            out, rc = ctx.decode_host(gm, bad, 0, n_ways, check=False)
            assert rc == R.E_CORRUPT
    # the bulk entry points: zero chunks, zero bytes
    gm = ctx.model(FMT_WORD, oracle.normalize(oracle.count_freqs(data, 256), 4096)[0], 12)
    cont, offs, lens, total = ctx.encode(gm, torch.zeros(0, dtype=torch.uint8, device="cuda"), 64, 4096)
    assert total == 0


def test_single_symbol_models(gpu, oracle):
    """freq == M (a constant input, which the reference's mains compress): inside the working range of the byte,
    alias and rans64 coders (rans_byte.h:176-178, rans64.h:169-171) -- the state never moves, the stream is the
    flushed states -- and bit-exact here; the word format's threshold wraps: E_MODEL."""
    R, ctx, torch = gpu
    ref = _ref_or_none()
    n = 70001
    for fmt, sb, nsyms, sym in ((FMT_BYTE, 14, 256, 65), (FMT_BYTE, 8, 256, 0), (FMT_BYTE, 16, 256, 255),
                                (FMT_R64, 14, 256, 7), (FMT_R64, 16, 256, 200), (FMT_ALIAS, 15, 256, 3),
                                (FMT_ALIAS, 12, 4096, 4095)):
        f = np.zeros(nsyms, np.uint32)
        f[sym] = 1 << sb
        data = np.full(n, sym, np.uint8 if nsyms <= 256 else np.uint16)
        om = oracle.model(f, sb, with_alias=(fmt == FMT_ALIAS))
        gm = ctx.model(fmt, f, sb)
        for n_ways in (1, 2, 64, 100, 256):
            want = oracle.encode(fmt, om, data, n_ways)
            if ref is not None and n_ways <= 2 and nsyms == 256:
                assert np.array_equal(ref.encode(fmt, f, sb, data, n_ways), want)
            assert np.array_equal(ctx.encode_host(gm, data, n_ways), want), (fmt, sb, n_ways)
            assert np.array_equal(ctx.decode_host(gm, want, n, n_ways), data), (fmt, sb, n_ways)
        # chunked, and a stray other symbol is still E_MODEL
        d = torch.from_numpy(data.view(np.int16) if data.dtype == np.uint16 else data).cuda()
        for n_ways, chunk in ((64, 4096), (2, 512)):
            want, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d, n_ways, chunk)
            assert total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens)
            got = cont[:total].cpu().numpy()
            for c in (0, len(lens) // 2, len(lens) - 1):
                o, ln = int(offs[c]), int(lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (fmt, sb, n_ways, c)
            out = ctx.decode(gm, cont, total, d_offs, d_lens, n, n_ways, chunk)
            assert torch.equal(out, d)
        bad = data.copy()
        bad[1234] = (sym + 1) % nsyms
        with pytest.raises(R.RansAmdError) as e:
            ctx.encode_host(gm, bad, 64)
        assert e.value.status == R.E_MODEL
    f = np.zeros(256, np.uint32)
    f[9] = 4096
    with pytest.raises(R.RansAmdError) as e:
        ctx.model(FMT_WORD, f, 12)
    assert e.value.status == R.E_MODEL
    # the device's alias records hold a half bucket's frequency in 16 bits: a 65536-wide symbol is a host-only model
    f[9] = 65536
    R.Model(None, FMT_ALIAS, f, 16)
    with pytest.raises(R.RansAmdError) as e:
        ctx.model(FMT_ALIAS, f, 16)
    assert e.value.status == R.E_UNSUPPORTED


def test_rans64_any_scale_bits(gpu, oracle):
    """rans64.h accepts scale_bits up to 31 (rans64.h:169).  7..16 go through the cum2sym decoder; 17..31 (no 2^sb
    table fits LDS) and 1..6 (small alphabets) take the search decoder / full-width encoder: same streams, checked
    against the oracle (cum2sym on the host up to 24 bits) and, at 31 bits, against the unmodified reference."""
    R, ctx, torch = gpu
    ref = _ref_or_none()
    n = 50001
    cases = [(8, 3), (16, 5), (64, 6), (256, 17), (256, 20), (256, 24), (200, 19)]
    for nsyms, sb in cases:
        data = (oracle.gen_zipf(n, K=256, s=1.0, seed=sb).astype(np.int32) % nsyms).astype(np.uint8)
        f, _ = oracle.normalize(oracle.count_freqs(data, nsyms), 1 << sb)
        om = oracle.model(f, sb)
        gm = ctx.model(FMT_R64, f, sb)
        for n_ways in (1, 2, 64, 100, 256):
            want = oracle.encode(FMT_R64, om, data, n_ways)
            assert np.array_equal(ctx.encode_host(gm, data, n_ways), want), (nsyms, sb, n_ways, "encode")
            assert np.array_equal(ctx.decode_host(gm, want, n, n_ways), data), (nsyms, sb, n_ways, "decode")
        # chunked, narrow and wide interleaves (narrow ones would take the lane kernels: the search variant has none)
        d = torch.from_numpy(data).cuda()
        for n_ways, chunk in ((2, 512), (64, 4096), (8, 1000)):
            want, offs, lens = oracle.encode_chunked(FMT_R64, om, data, n_ways, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d, n_ways, chunk)
            assert total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens), (sb, n_ways)
            got = cont[:total].cpu().numpy()
            for c in (0, len(lens) // 2, len(lens) - 1):
                o, ln = int(offs[c]), int(lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (sb, n_ways, c)
            d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
            out = ctx.decode(gm, d_cont, want.size, torch.from_numpy(offs.astype(np.int64)).cuda(),
                             torch.from_numpy(lens.astype(np.int32)).cuda(), n, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy(), data), (sb, n_ways)
    # 31 bits: the reference's encoder is the judge (its decoder would want a 2^31-entry table; so would the oracle)
    data = oracle.gen_zipf(20000, K=256, s=1.0, seed=1)
    f, _ = oracle.normalize(oracle.count_freqs(data, 256), 1 << 31)
    gm = ctx.model(FMT_R64, f, 31)
    for n_ways in (1, 2, 64):
        got = ctx.encode_host(gm, data, n_ways)
        if ref is not None:
            assert np.array_equal(got, ref.encode(FMT_R64, f, 31, data, n_ways)), n_ways
        assert np.array_equal(ctx.decode_host(gm, got, data.size, n_ways), data), n_ways
    # a corrupted wide stream is flagged like any other
    bad = got.copy()
    bad[len(bad) // 2] ^= 0x20
    out, rc = ctx.decode_host(gm, bad, data.size, 64, check=False)
    assert rc == R.E_CORRUPT or not np.array_equal(out, data)


def test_word_format_with_u16_symbols(gpu, oracle):
    """the word format over alphabets beyond 256 symbols (rans_word_sse41.h:41 fixes 256; the stream
    format -- 16-bit renormalisation, 12-bit probabilities -- does not depend on it).  Symbols are u16, the slot
    record carries a 16-bit symbol; streams equal the oracle's, whose arithmetic is the pinned word coder's."""
    R, ctx, torch = gpu
    for nsyms, seed in ((1024, 1), (4096, 2), (300, 3)):
        n = 300001
        data = (oracle.gen_zipf(n, K=4096, s=1.0, seed=seed).astype(np.uint32) % nsyms).astype(np.uint16)
        f, _ = oracle.normalize(oracle.count_freqs(data, nsyms), 4096)
        om = oracle.model(f, 12)
        gm = ctx.model(FMT_WORD, f, 12)
        assert gm.sym_bytes == 2
        for n_ways in (64, 1, 2, 100, 128):
            want = oracle.encode(FMT_WORD, om, data[:60001], n_ways)
            assert np.array_equal(ctx.encode_host(gm, data[:60001], n_ways), want), (nsyms, n_ways, "encode")
            assert np.array_equal(ctx.decode_host(gm, want, 60001, n_ways), data[:60001]), (nsyms, n_ways, "decode")
        d = torch.from_numpy(data.view(np.int16)).cuda()
        for n_ways, chunk in ((64, 8192), (128, 16384), (2, 512), (64, 5000)):
            want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, n_ways, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d, n_ways, chunk)
            assert total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens), (nsyms, n_ways)
            got = cont[:total].cpu().numpy()
            for c in (0, 1, len(lens) // 2, len(lens) - 1):
                o, ln = int(offs[c]), int(lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (nsyms, n_ways, c)
            d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
            out = ctx.decode(gm, d_cont, want.size, torch.from_numpy(offs.astype(np.int64)).cuda(),
                             torch.from_numpy(lens.astype(np.int32)).cuda(), n, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy().view(np.uint16), data), (nsyms, n_ways)
    # more symbols than slots cannot be a model
    with pytest.raises(R.RansAmdError):
        ctx.model(FMT_WORD, np.ones(8192, np.uint32), 12)


@pytest.mark.parametrize("fmt", [FMT_BYTE, FMT_WORD])
def test_per_chunk_adaptive_models(gpu, oracle, fmt):
    """one order-0 model per chunk, built as the reference builds its one model per input
    (main.cpp:139-162; the word format: main_simd.cpp:138-143, 12 bits) -- histogram and normalize_freqs per chunk on the
    GPU, tables by the coding wavefront.
    Every chunk's frequencies must be the oracle's normalize(count_freqs(chunk)), every chunk's stream the oracle's
    stream for a model of that chunk alone; the input changes statistics from chunk to chunk on purpose."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(31)
    chunk = 8192
    parts = []
    for c in range(37):  # every chunk its own distribution: skewed, two symbols, uniform, constant, text-like
        kind = c % 5
        if kind == 0:
            parts.append(np.minimum(rng.geometric(0.02 + 0.01 * c, chunk) - 1, 255))
        elif kind == 1:
            parts.append(rng.integers(0, 2, chunk) * (c + 3))
        elif kind == 2:
            parts.append(rng.integers(0, 256, chunk))
        elif kind == 3:
            parts.append(np.full(chunk, c))
        else:
            parts.append((oracle.gen_zipf(chunk, K=256, s=1.0, seed=c).astype(np.int64) + 7 * c) % 256)
    data = np.concatenate(parts).astype(np.uint8)[:37 * chunk - 1234]  # ragged last chunk
    n = data.size
    d = torch.from_numpy(data).cuda()
    for sb in ((12, 10, 8) if fmt == FMT_BYTE else (12,)):
        for n_ways in (64, 2, 128, 100, 256):
            cont, offs, lens, freqs, total = ctx.encode_adaptive(d, n_ways, chunk, sb, fmt=fmt)
            assert ctx.last_encode_kernel()[0] == ("k_encode<byte, per-chunk models>" if fmt == FMT_BYTE else "k_encode<word, per-chunk models>")
            h_freqs = freqs.cpu().numpy().view(np.uint16).reshape(-1, 256)
            h_offs = offs.cpu().numpy().astype(np.uint64)
            h_lens = lens.cpu().numpy().astype(np.uint32)
            got = cont[:total].cpu().numpy()
            nchunks = (n + chunk - 1) // chunk
            pos = 0
            for c in range(nchunks):
                part = data[c * chunk:(c + 1) * chunk]
                f, _ = oracle.normalize(oracle.count_freqs(part, 256), 1 << sb)
                assert np.array_equal(h_freqs[c].astype(np.uint32), f), (sb, n_ways, c, "model")
                want = oracle.encode(fmt, oracle.model(f, sb), part, n_ways)
                assert int(h_offs[c]) == pos and int(h_lens[c]) == want.size, (sb, n_ways, c, "index")
                assert np.array_equal(got[pos:pos + want.size], want), (sb, n_ways, c, "stream")
                pos += (want.size + 15) & ~15
            out = ctx.decode_adaptive(cont, total, offs, lens, freqs, n, n_ways, chunk, sb, fmt=fmt)
            assert ctx.last_decode_kernel() == ("k_decode<byte, per-chunk models>" if fmt == FMT_BYTE else "k_decode<word, per-chunk models>")
            assert np.array_equal(out.cpu().numpy(), data), (sb, n_ways, "decode")
    # adaptive beats one global model on this input, and a damaged frequency row is flagged, not decoded
    cont, offs, lens, freqs, total = ctx.encode_adaptive(d, 64, chunk, 12, fmt=fmt)
    gm = ctx.model_for(fmt, data, 256, 12)
    _, _, _, total_global = ctx.encode(gm, d, 64, chunk)
    if fmt == FMT_BYTE:
        assert total + freqs.numel() * 2 < total_global
    # (the word format: a chunk that holds ONE symbol value has freq = 4096 = M, and rans_word_sse41.h:85's renormalisation
    #  bound (L >> 12 << 16) * freq wraps to 0 in 32 bits -- the reference then emits a word per symbol; oracle and GPU do
    #  exactly that (the streams above are equal), so the constant chunks of this input cost 2 bytes per symbol)
    bad = freqs.clone()
    bad[5 * 256 + 3] += 1
    out = ctx.decode_adaptive(cont, total, offs, lens, bad, n, 64, chunk, 12, sync=False, fmt=fmt)
    assert ctx.decode_errors() >= 1
    for sb_bad in ((14,) if fmt == FMT_BYTE else (14, 11, 8)):  # (the word format's probabilities are 12 bits, rans_word_sse41.h:37)
        with pytest.raises(R.RansAmdError) as e:
            ctx.encode_adaptive(d, 64, chunk, sb_bad, fmt=fmt)
        assert e.value.status == R.E_UNSUPPORTED
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_adaptive(d, 64, chunk, 12, fmt=FMT_R64)
    assert e.value.status == R.E_UNSUPPORTED


def _moving_statistics(oracle, chunk, nchunks, cut=1234):
    """every chunk its own distribution: skewed, two symbols, uniform, constant, text-like; ragged last chunk"""
    rng = np.random.default_rng(31)
    parts = []
    for c in range(nchunks):
        kind = c % 5
        if kind == 0:
            parts.append(np.minimum(rng.geometric(0.02 + 0.01 * (c % 40), chunk) - 1, 255))
        elif kind == 1:
            parts.append(rng.integers(0, 2, chunk) * ((c + 3) % 256))
        elif kind == 2:
            parts.append(rng.integers(0, 256, chunk))
        elif kind == 3:
            parts.append(np.full(chunk, c % 256))
        else:
            parts.append((oracle.gen_zipf(chunk, K=256, s=1.0, seed=c).astype(np.int64) + 7 * c) % 256)
    return np.concatenate(parts).astype(np.uint8)[:nchunks * chunk - cut]


@pytest.mark.parametrize("fmt", [FMT_BYTE, FMT_WORD])
def test_per_chunk_models_in_one_kernel(gpu, oracle, fmt):
    """rans_amd_encode_adaptive_sized (round 6): the wave that codes a chunk counts it, normalises, builds its records and
    codes it into a place sized from the chunk's own histogram -- one launch.  Every chunk's row == the oracle's
    normalize(count(chunk)) and every chunk's stream == the oracle's stream under that model (main.cpp:139-162 per chunk),
    where the index puts it; the pieces lie in index order, whole 64-byte lines each, the same from run to run; the
    decoders take the container as it is; a container buffer that is too small is RANS_AMD_E_SPACE."""
    R, ctx, torch = gpu
    chunk = 8192
    data = _moving_statistics(oracle, chunk, 37)
    n = data.size
    nchunks = (n + chunk - 1) // chunk
    d = torch.from_numpy(data).cuda()
    kern = "k_encode_adaptive<byte>" if fmt == FMT_BYTE else "k_encode_adaptive<word>"
    for sb in ((12, 10, 8) if fmt == FMT_BYTE else (12,)):
        layouts = {}
        for n_ways in (64, 2, 128, 100, 256, 33, 512, 64):
            cont, offs, lens, freqs, total = ctx.encode_adaptive_sized(d, n_ways, chunk, sb, fmt=fmt)
            assert ctx.last_encode_kernel()[0] == kern
            h_offs = offs.cpu().numpy().astype(np.uint64)
            h_lens = lens.cpu().numpy().astype(np.uint32)
            assert int(h_offs[nchunks]) == total
            count, bad = oracle.compare_container_adaptive(fmt, data, n_ways, chunk, sb, cont[:total].cpu().numpy(), h_offs, h_lens,
                                                           freqs.cpu().numpy())
            assert count == nchunks and bad == -1, (sb, n_ways, bad)
            # pieces are whole 64-byte lines IN INDEX ORDER, a stream is the end of its piece, the last piece ends the container
            ends = h_offs[:nchunks] + h_lens
            assert np.all(ends % np.uint64(64) == 0) and np.all(ends[:-1] <= h_offs[1:nchunks]) and int(ends[-1]) == total
            # ... and the layout is the same from run to run
            key = (n_ways,)
            if key in layouts:
                assert np.array_equal(layouts[key], h_offs)
            layouts[key] = h_offs
            out = ctx.decode_adaptive(cont, total, offs, lens, freqs, n, n_ways, chunk, sb, fmt=fmt)
            assert np.array_equal(out.cpu().numpy(), data), (sb, n_ways, "decode")
    # the register-resident forms of the kernel (full chunks of 4 / 8 / 16 Ki symbols live in registers between the count and
    # the coding pass; the ragged last chunk takes the two-pass form inside the same kernel) and a chunk size without one
    big = _moving_statistics(oracle, 4096, 90, cut=3000)
    d_big = torch.from_numpy(big).cuda()
    for ch in (4096, 16384, 12288):
        cont, offs, lens, freqs, total = ctx.encode_adaptive_sized(d_big, 64, ch, 12, fmt=fmt)
        cnt, bad = oracle.compare_container_adaptive(fmt, big, 64, ch, 12, cont[:total].cpu().numpy(), offs.cpu().numpy(), lens.cpu().numpy(),
                                                     freqs.cpu().numpy())
        assert bad == -1 and cnt == (big.size + ch - 1) // ch, (ch, bad)
        out = ctx.decode_adaptive(cont, total, offs, lens, freqs, big.size, 64, ch, 12, fmt=fmt)
        assert np.array_equal(out.cpu().numpy(), big), ch
    # the same rows and streams as the three-launch path (rans_amd_encode_adaptive_fmt)
    c0, o0, l0, f0, t0 = ctx.encode_adaptive(d, 64, chunk, 12, fmt=fmt)
    c1, o1, l1, f1, t1 = ctx.encode_adaptive_sized(d, 64, chunk, 12, fmt=fmt)
    assert torch.equal(f0, f1) and torch.equal(l0, l1)
    h0, h1, ho0, ho1, hl = c0.cpu().numpy(), c1.cpu().numpy(), o0.cpu().numpy(), o1.cpu().numpy(), l0.cpu().numpy()
    for c in range(nchunks):
        assert np.array_equal(h0[int(ho0[c]):int(ho0[c]) + int(hl[c])], h1[int(ho1[c]):int(ho1[c]) + int(hl[c])]), c
    # the container is about as large as the streams: bound = stream + 1.1 % of the symbols (word) + states + a line or two
    streams = int(hl[:nchunks].astype(np.int64).sum())
    assert t1 <= streams + 0.02 * n + nchunks * 192, (t1, streams)
    # unaligned input (the coders read bytes), a one-chunk input, an empty one
    du = torch.empty(n + 1, dtype=torch.uint8, device="cuda")[1:]
    du.copy_(d)
    cont, offs, lens, freqs, total = ctx.encode_adaptive_sized(du, 64, chunk, 12, fmt=fmt)
    _, bad = oracle.compare_container_adaptive(fmt, data, 64, chunk, 12, cont[:total].cpu().numpy(), offs.cpu().numpy(),
                                               lens.cpu().numpy(), freqs.cpu().numpy())
    assert bad == -1
    cont, offs, lens, freqs, total = ctx.encode_adaptive_sized(d[:5000], 64, chunk, 12, fmt=fmt)
    _, bad = oracle.compare_container_adaptive(fmt, data[:5000], 64, chunk, 12, cont[:total].cpu().numpy(), offs.cpu().numpy(),
                                               lens.cpu().numpy(), freqs.cpu().numpy())
    assert bad == -1
    cont, offs, lens, freqs, total = ctx.encode_adaptive_sized(d[:0], 64, chunk, 12, fmt=fmt)
    assert total == 0
    # too small a buffer: RANS_AMD_E_SPACE, from the call and -- asynchronously -- from rans_amd_encode_status
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_adaptive_sized(d, 64, chunk, 12, fmt=fmt, cap=t1 // 2)
    assert e.value.status == R.E_SPACE
    small = torch.empty(t1 // 2, dtype=torch.uint8, device="cuda")
    ctx.encode_adaptive_sized(d, 64, chunk, 12, fmt=fmt, d_out=small, sync=False)
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_status()
    assert e.value.status == R.E_SPACE
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_adaptive_sized(d, 64, chunk, 12, fmt=FMT_R64)
    assert e.value.status == R.E_UNSUPPORTED
    # a healthy call right behind a failed one starts clean
    ctx.encode_adaptive_sized(d, 64, chunk, 12, fmt=fmt, sync=False)
    ctx.encode_status()


@pytest.mark.parametrize("fmt", [FMT_BYTE, FMT_WORD])
def test_per_chunk_model_decoder_on_an_oracle_made_container(gpu, oracle, fmt):
    """The per-chunk-model decoders fed a container (streams AND rows) the ORACLE made: nothing they read was written by a
    GPU encoder."""
    R, ctx, torch = gpu
    chunk = 8192
    data = _moving_statistics(oracle, chunk, 23, cut=777)
    for n_ways in (64, 2, 128):
        cont, offs, lens, rows = oracle.encode_chunked_adaptive(fmt, data, n_ways, chunk, 12)
        d_cont = torch.zeros(cont.size + 64, dtype=torch.uint8, device="cuda")
        d_cont[:cont.size] = torch.from_numpy(cont).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        d_rows = torch.from_numpy(rows.reshape(-1).view(np.int16)).cuda()
        out = ctx.decode_adaptive(d_cont, cont.size, d_offs, d_lens, d_rows, data.size, n_ways, chunk, 12, fmt=fmt)
        assert np.array_equal(out.cpu().numpy(), data), n_ways


def test_per_chunk_models_in_one_kernel_inside_a_hip_graph(gpu, oracle):
    """The one-launch per-chunk-model encode captured into a hipGraph and replayed on other data."""
    R, ctx, torch = gpu
    chunk, n_ways = 8192, 64
    datas = [_moving_statistics(oracle, chunk, 20, cut=0), oracle.gen_zipf(20 * chunk, K=256, s=1.0, seed=5),
             np.random.default_rng(3).integers(0, 256, 20 * chunk).astype(np.uint8)]
    d_syms = torch.from_numpy(datas[0]).cuda().clone()
    cap = R.lib().rans_amd_encode_adaptive_sized_bound(FMT_WORD, d_syms.numel(), n_ways, chunk)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    offs = torch.zeros(21, dtype=torch.int64, device="cuda")
    lens = torch.zeros(20, dtype=torch.int32, device="cuda")
    rows = torch.zeros(20 * 256, dtype=torch.int16, device="cuda")
    ctx.encode_adaptive_sized(d_syms, n_ways, chunk, 12, fmt=FMT_WORD, d_out=out, d_offsets=offs, d_lengths=lens, d_freqs=rows)
    g = torch.cuda.CUDAGraph()
    stream = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=stream):
        ctx.encode_adaptive_sized(d_syms, n_ways, chunk, 12, fmt=FMT_WORD, d_out=out, d_offsets=offs, d_lengths=lens, d_freqs=rows,
                                  sync=False)
    for d in datas:
        d_syms.copy_(torch.from_numpy(d).cuda())
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ctx.encode_status()
        total = int(offs[-1].item())
        _, bad = oracle.compare_container_adaptive(FMT_WORD, d, n_ways, chunk, 12, out[:total].cpu().numpy(), offs.cpu().numpy(),
                                                   lens.cpu().numpy(), rows.cpu().numpy())
        assert bad == -1


def test_rans64_two_way_lane_kernel(gpu, oracle):
    """BASELINE config 2's layout (the reference's 2-way rans64 loop, main64.cpp:228-282) through the dedicated
    lane-per-chunk decoder k_decode_lanes_r64x2: every scale_bits with a cum2sym table, chunk sizes of one and many
    64-symbol trips, a ragged last chunk (second launch), few and many batches, alphabets that make a stream consume
    the most bytes per symbol a valid stream can (symbols of probability 2^-scale_bits), the ORACLE's container
    and the GPU's own; then damaged containers: flagged, never a crash."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(5)
    zipf = oracle.gen_zipf(600000 + 333, K=256, s=1.0, seed=23)
    flat = rng.integers(0, 256, 300000).astype(np.uint8)
    heavy = np.where(rng.random(300000) < 0.6, 0, rng.integers(0, 256, 300000)).astype(np.uint8)  # one symbol above 50 %
    cases = [(zipf, 14, 512), (zipf, 14, 64), (zipf, 12, 1024), (zipf, 16, 4096), (zipf, 8, 128), (zipf[:64 * 64], 14, 64),
             (zipf[:70 * 512 + 5], 11, 512), (flat, 14, 512), (flat, 16, 192), (heavy, 14, 512), (heavy, 12, 256), (heavy, 13, 64)]
    for data, sb, chunk in cases:
        om, gm = _models(R, ctx, oracle, FMT_R64, sb, data)
        # round 3: models with scale_bits <= 14 and no frequency above 4095 are decoded from 4-byte packed slot records
        # (one gather per symbol); everything else from cum2sym + {freq, start} as before
        packed = sb <= 14 and int(gm.freqs.max()) <= 4095
        kernel = "k_decode_lanes_r64x2<packed slots>" if packed else "k_decode_lanes_r64x2"
        want, offs, lens = oracle.encode_chunked(FMT_R64, om, data, 2, chunk, align=16)
        d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        out = ctx.decode(gm, d_cont, want.size, d_offs, d_lens, data.size, 2, chunk)
        assert ctx.last_decode_kernel() == kernel, (ctx.last_decode_kernel(), sb, chunk)
        assert np.array_equal(out.cpu().numpy(), data), (sb, chunk, data.size)
        d_syms = torch.from_numpy(data).cuda()
        cont, o2, l2, total = ctx.encode(gm, d_syms, 2, chunk)
        got = cont[:total].cpu().numpy()
        assert total == want.size and np.array_equal(l2.cpu().numpy().astype(np.uint32), lens), (sb, chunk)
        for c in range(len(lens)):  # (the padding between chunks is not defined)
            o, ln = int(offs[c]), int(lens[c])
            assert np.array_equal(got[o:o + ln], want[o:o + ln]), (sb, chunk, c)
        out = ctx.decode(gm, cont, total, o2, l2, data.size, 2, chunk)
        assert np.array_equal(out.cpu().numpy(), data), (sb, chunk, "own container")
    # offsets need not ascend: the same chunks laid out back to front (and the general staged kernel on the word format)
    for fmt, sb in ((FMT_R64, 14), (FMT_WORD, 12)):
        data, chunk = zipf[:200 * 512], 512
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want, offs, lens = oracle.encode_chunked(fmt, om, data, 2, chunk, align=16)
        rev = np.zeros(want.size + 64, np.uint8)
        roffs = np.zeros_like(offs)
        pos = 0
        for c in range(len(lens) - 1, -1, -1):
            roffs[c] = pos
            rev[pos:pos + lens[c]] = want[int(offs[c]):int(offs[c]) + int(lens[c])]
            pos += (int(lens[c]) + 15) & ~15
        out = ctx.decode(gm, torch.from_numpy(rev).cuda(), pos, torch.from_numpy(roffs.astype(np.int64)).cuda(),
                         torch.from_numpy(lens.astype(np.int32)).cuda(), data.size, 2, chunk)
        assert np.array_equal(out.cpu().numpy(), data), (fmt, "reversed layout")
    # rare symbols only: 16 bits leave the state per symbol, the most a valid rans64 stream can consume
    sb = 16
    f = np.ones(256, np.uint32)
    f[0] = (1 << sb) - 255
    rare = rng.integers(1, 256, 64 * 1024).astype(np.uint8)
    om, gm = oracle.model(f, sb), ctx.model(FMT_R64, f, sb)
    want, offs, lens = oracle.encode_chunked(FMT_R64, om, rare, 2, 256, align=16)
    assert want.size > 2 * rare.size - 4096  # ~2 bytes per symbol
    out = ctx.decode(gm, torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda(), want.size,
                     torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(),
                     rare.size, 2, 256)
    assert ctx.last_decode_kernel() == "k_decode_lanes_r64x2"
    assert np.array_equal(out.cpu().numpy(), rare)
    # the same through the packed slot records (12 bits: the frequent symbol's 3841 fits the record's 12-bit field)
    sb = 12
    f = np.ones(256, np.uint32)
    f[0] = (1 << sb) - 255
    om, gm = oracle.model(f, sb), ctx.model(FMT_R64, f, sb)
    want, offs, lens = oracle.encode_chunked(FMT_R64, om, rare, 2, 256, align=16)
    out = ctx.decode(gm, torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda(), want.size,
                     torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(),
                     rare.size, 2, 256)
    assert ctx.last_decode_kernel() == "k_decode_lanes_r64x2<packed slots>"
    assert np.array_equal(out.cpu().numpy(), rare)
    # damage: flipped stream bytes, a length that lies, an offset off its 16-byte grid
    data, sb, chunk = zipf[:128 * 512], 14, 512
    om, gm = _models(R, ctx, oracle, FMT_R64, sb, data)
    want, offs, lens = oracle.encode_chunked(FMT_R64, om, data, 2, chunk, align=16)
    for kind in range(3):
        bad, o, l = want.copy(), offs.astype(np.int64).copy(), lens.astype(np.int32).copy()
        if kind == 0:
            for pos in rng.integers(0, want.size, 20):
                bad[pos] ^= 0x40
        elif kind == 1:
            l[7] += 4
        else:
            o[9] += 4
        ctx.decode(gm, torch.from_numpy(np.concatenate([bad, np.zeros(64, np.uint8)])).cuda(), want.size,
                   torch.from_numpy(o).cuda(), torch.from_numpy(l).cuda(), data.size, 2, chunk, sync=False)
        assert ctx.decode_errors() >= 1, kind


def test_word_eight_way_octet_decoder(gpu, oracle):
    """The reference's 8-way word layout (rans_word_sse41.h:151-227, main_simd.cpp:313-332) through k_decode_word_groups, eight
    chunks per wave: chunk sizes of one and many 16-round lines, sizes that leave one to three 4-round groups behind the last
    line, sizes off 32 (any multiple of 4), chunk counts that are no multiple of eight and a ragged last chunk (its octet goes round by round), chunks that
    start on any 2-byte boundary, in any order; the ORACLE's container and the GPU's own; streams that consume the most a
    valid one can (16 bytes per round) and next to nothing; damage is flagged, never a crash."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(8)
    zipf = oracle.gen_zipf(700000 + 77, K=256, s=1.0, seed=31)
    flat = rng.integers(0, 256, 260000).astype(np.uint8)
    heavy = np.where(rng.random(260000) < 0.93, 7, rng.integers(0, 256, 260000)).astype(np.uint8)  # ~0.1 byte per symbol
    cases = [(zipf, 1024), (zipf, 128), (zipf, 32), (zipf, 96), (zipf, 160), (zipf, 4000), (zipf, 16384), (zipf[:8 * 1024], 1024),
             (zipf[:71 * 256 + 9], 256), (flat, 1024), (flat, 224), (heavy, 512), (heavy, 16384),
             (zipf, 1000), (zipf, 36), (zipf, 100), (flat, 5004), (heavy, 44), (zipf, 60)]  # chunk sizes off 32: one round at a time at the end
    for data, chunk in cases:
        om, gm = _models(R, ctx, oracle, FMT_WORD, 12, data)
        want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, 8, chunk, align=16)
        d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        back = torch.full((data.size + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        out = back[128:128 + data.size]
        ctx.decode(gm, d_cont, want.size, d_offs, d_lens, data.size, 8, chunk, d_out=out)
        assert ctx.last_decode_kernel() == "k_decode_word_groups", (ctx.last_decode_kernel(), chunk)
        assert np.array_equal(out.cpu().numpy(), data), (chunk, data.size)
        host = back.cpu().numpy()
        assert (host[:128] == 0xA5).all() and (host[128 + data.size:] == 0xA5).all(), chunk  # nothing outside the output
        cont, o2, l2, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 8, chunk)
        out = ctx.decode(gm, cont, total, o2, l2, data.size, 8, chunk)
        assert ctx.last_decode_kernel() == "k_decode_word_groups" and np.array_equal(out.cpu().numpy(), data), (chunk, "own container")
    # output on a 4-byte boundary is enough; on any other the lane kernel takes over
    data, chunk = zipf[:100 * 256], 256
    om, gm = _models(R, ctx, oracle, FMT_WORD, 12, data)
    want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, 8, chunk, align=16)
    d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
    d_offs, d_lens = torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()
    for shift, kernel in ((4, "k_decode_word_groups"), (12, "k_decode_word_groups"), (2, "k_decode_lanes_staged"), (1, "k_decode_lanes_staged")):
        back = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
        ctx.decode(gm, d_cont, want.size, d_offs, d_lens, data.size, 8, chunk, d_out=back[shift:shift + data.size])
        assert ctx.last_decode_kernel() == kernel, (shift, ctx.last_decode_kernel())
        assert np.array_equal(back[shift:shift + data.size].cpu().numpy(), data), shift
    # chunks packed back to front on 2-byte boundaries (the format's unit)
    rev = np.zeros(want.size + 256, np.uint8)
    roffs = np.zeros_like(offs)
    pos = 2
    for c in range(len(lens) - 1, -1, -1):
        roffs[c] = pos
        rev[pos:pos + lens[c]] = want[int(offs[c]):int(offs[c]) + int(lens[c])]
        pos += int(lens[c]) + 2 * (c % 5)
    d_rev = torch.from_numpy(rev).cuda()
    out = ctx.decode(gm, d_rev, pos, torch.from_numpy(roffs.astype(np.int64)).cuda(), d_lens, data.size, 8, chunk)
    assert ctx.last_decode_kernel() == "k_decode_word_groups" and np.array_equal(out.cpu().numpy(), data)
    # rare symbols only (12 bits each, the most the format's 12-bit models can ask for): three of four states renormalise per
    # round on average, rounds in which all eight do (16 stream bytes, the refill invariant's bound) are common
    f = np.ones(256, np.uint32)
    f[0] = 4096 - 255
    rare = rng.integers(1, 256, 64 * 2048).astype(np.uint8)
    om, gm2 = oracle.model(f, 12), ctx.model(FMT_WORD, f, 12)
    w2, of2, le2 = oracle.encode_chunked(FMT_WORD, om, rare, 8, 2048, align=16)
    assert w2.size > 3 * rare.size // 2 - 4096
    out = ctx.decode(gm2, torch.from_numpy(np.concatenate([w2, np.zeros(64, np.uint8)])).cuda(), w2.size,
                     torch.from_numpy(of2.astype(np.int64)).cuda(), torch.from_numpy(le2.astype(np.int32)).cuda(), rare.size, 8, 2048)
    assert ctx.last_decode_kernel() == "k_decode_word_groups" and np.array_equal(out.cpu().numpy(), rare)
    # damage: flipped stream bytes, a length that lies, an odd offset, an offset beyond the container
    for kind in range(4):
        bad, o, l = want.copy(), offs.astype(np.int64).copy(), lens.astype(np.int32).copy()
        if kind == 0:
            for at in rng.integers(0, want.size, 20):
                bad[at] ^= 0x40
        elif kind == 1:
            l[7] += 2
        elif kind == 2:
            o[9] += 1
        else:
            o[11] = want.size + 1000
        ctx.decode(gm, torch.from_numpy(np.concatenate([bad, np.zeros(64, np.uint8)])).cuda(), want.size,
                   torch.from_numpy(o).cuda(), torch.from_numpy(l).cuda(), data.size, 8, chunk, sync=False)
        assert ctx.decode_errors() >= 1, kind
    # ... and random damage of every kind never hangs or faults
    for trial in range(9):
        bad, l = want.copy(), lens.astype(np.int32).copy()
        if trial % 3 == 0:
            at = rng.integers(0, bad.size, 40)
            bad[at] ^= rng.integers(1, 256, 40).astype(np.uint8)
        elif trial % 3 == 1:
            bad[rng.integers(0, bad.size):] = 0
        else:
            l[rng.integers(0, l.size, 3)] = rng.integers(0, 1 << 20, 3).astype(np.int32)
        out = ctx.decode(gm, torch.from_numpy(np.concatenate([bad, np.zeros(64, np.uint8)])).cuda(), bad.size, d_offs,
                         torch.from_numpy(l).cuda(), data.size, 8, chunk, sync=False)
        assert ctx.decode_errors() > 0 or np.array_equal(out.cpu().numpy(), data), trial


def test_word_eight_way_octet_encoder(gpu, oracle):
    """The mirror image, k_encode_word_groups (eight chunks per wave): every chunk's stream == the oracle's, in all three
    placements -- compact (k_layout + k_compact_small behind the launch), the slot layout, sized slots with and without chunks
    that do not fit (those are coded again into the overflow region) -- for models with Alverson reciprocals, with a frequency
    above 2048 (round-up reciprocals) and with byte values that have no record; chunk counts that are no multiple of eight; a
    stray symbol is RANS_AMD_E_MODEL; chunk sizes off 128 (the rounds behind the last whole line go one at a time) and a
    ragged last chunk (its octet goes round by round); chunks off 4 symbols still code through the lane kernels."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(3)
    zipf = oracle.gen_zipf(1 << 20, K=256, s=1.0, seed=5)
    flat = rng.integers(0, 256, 300000).astype(np.uint8)
    heavy = np.where(rng.random(300000) < 0.7, 3, rng.integers(0, 256, 300000)).astype(np.uint8)
    sparse = (rng.integers(0, 40, 300000) * 3).astype(np.uint8)

    def same(got, goffs, glens, want, offs, lens, what):
        assert np.array_equal(glens.astype(np.uint32), lens.astype(np.uint32)), what
        for c in range(len(lens)):
            a, b, ln = int(goffs[c]), int(offs[c]), int(lens[c])
            assert np.array_equal(got[a:a + ln], want[b:b + ln]), (what, c)

    for name, data in (("zipf", zipf), ("flat", flat), ("heavy", heavy), ("sparse", sparse)):
        for chunk, nch in ((1024, 64), (128, 100), (4096, 17), (256, 1000), (384, 9), (1000, 100), (36, 300), (100, 64), (5004, 11)):
            d = data[:chunk * nch]
            f, _ = R.normalize_freqs(np.bincount(d, minlength=256), 1 << 12)
            om, gm = oracle.model(f, 12), ctx.model(FMT_WORD, f, 12)
            want, offs, lens = oracle.encode_chunked(FMT_WORD, om, d, 8, chunk, align=16)
            d_syms = torch.from_numpy(d).cuda()
            cont, o, l, total = ctx.encode(gm, d_syms, 8, chunk)
            if nch >= 64:  # (fewer chunks: the wave encoder with its own placement)
                assert ctx.last_encode_kernel() == ("k_encode_word_groups", False), ctx.last_encode_kernel()
            assert total == want.size and np.array_equal(o.cpu().numpy().astype(np.uint64), offs)
            same(cont[:total].cpu().numpy(), o.cpu().numpy(), l.cpu().numpy(), want, offs, lens, (name, chunk, nch, "compact"))
            cont, o, l, total = ctx.encode_slots(gm, d_syms, 8, chunk)
            assert ctx.last_encode_kernel()[0] == "k_encode_word_groups", ctx.last_encode_kernel()
            same(cont.cpu().numpy(), o.cpu().numpy(), l.cpu().numpy(), want, offs, lens, (name, chunk, nch, "slots"))
            worst = R.slot_bytes(FMT_WORD, d.size, 8, chunk)
            for slot in (None, 64 * ((int(lens.mean()) - 8 + 63) // 64), 64):
                cont, o, l, total, sl = ctx.encode_sized(gm, d_syms, 8, chunk, slot=slot, overflow_chunks=nch)
                assert ctx.last_encode_kernel()[0] == "k_encode_word_groups", ctx.last_encode_kernel()
                go = o.cpu().numpy().astype(np.uint64)
                same(cont.cpu().numpy(), go, l.cpu().numpy(), want, offs, lens, (name, chunk, nch, "sized", sl))
                over = int((lens > sl).sum()) if sl < worst else 0
                assert int(go[-1]) == total == nch * min(sl, worst) + over * worst, (name, chunk, nch, sl, over)
                fits = np.nonzero(lens <= sl)[0]
                assert np.array_equal(go[fits] + lens[fits].astype(np.uint64), (fits.astype(np.uint64) + 1) * np.uint64(min(sl, worst)))
                out = ctx.decode(gm, cont, total, o, l, d.size, 8, chunk)
                assert np.array_equal(out.cpu().numpy(), d) and ctx.decode_errors() == 0, (name, chunk, nch, sl)
    # a symbol without a record, deep inside a chunk
    d = sparse[:64 * 1024].copy()
    f, _ = R.normalize_freqs(np.bincount(d, minlength=256), 1 << 12)
    gm = ctx.model(FMT_WORD, f, 12)
    d[37 * 1024 + 555] = 1
    for call in (ctx.encode, ctx.encode_slots):
        with pytest.raises(R.RansAmdError) as e:
            call(gm, torch.from_numpy(d).cuda(), 8, 1024)
        assert e.value.status == R.E_MODEL
    # a ragged last chunk: the input's last octet goes round by round, lanes without a symbol sit their rounds out
    for extra in (1, 7, 8, 9, 77, 1023):
        for nch in (64, 69):
            data = heavy[:nch * 1024 + extra] if extra == 9 else zipf[:nch * 1024 + extra]
            f, _ = R.normalize_freqs(np.bincount(data, minlength=256), 1 << 12)
            om, gm = oracle.model(f, 12), ctx.model(FMT_WORD, f, 12)
            want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, 8, 1024, align=16)
            d_syms = torch.from_numpy(data).cuda()
            cont, o, l, total = ctx.encode(gm, d_syms, 8, 1024)
            assert ctx.last_encode_kernel()[0] == "k_encode_word_groups" and total == want.size, ctx.last_encode_kernel()
            same(cont[:total].cpu().numpy(), o.cpu().numpy(), l.cpu().numpy(), want, offs, lens, ("ragged", extra, nch))
            cont, o, l, total, sl = ctx.encode_sized(gm, d_syms, 8, 1024, overflow_chunks=nch + 1)
            assert ctx.last_encode_kernel()[0] == "k_encode_word_groups"
            same(cont.cpu().numpy(), o.cpu().numpy(), l.cpu().numpy(), want, offs, lens, ("ragged, sized", extra, nch))
            out = ctx.decode(gm, cont, total, o, l, data.size, 8, 1024)
            assert np.array_equal(out.cpu().numpy(), data) and ctx.decode_errors() == 0, (extra, nch)
    # a shape for the lane kernels: chunks of 1002 symbols (the group kernels load and store dwords: multiples of 4)
    for data, chunk in ((zipf[:128 * 1002], 1002),):
        f, _ = R.normalize_freqs(np.bincount(data, minlength=256), 1 << 12)
        om, gm = oracle.model(f, 12), ctx.model(FMT_WORD, f, 12)
        want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, 8, chunk, align=16)
        cont, o, l, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 8, chunk)
        assert ctx.last_encode_kernel()[0].startswith("k_encode_lanes"), ctx.last_encode_kernel()
        same(cont[:total].cpu().numpy(), o.cpu().numpy(), l.cpu().numpy(), want, offs, lens, ("lanes", chunk))


@pytest.mark.parametrize("sb", [14, 8, 12, 16])
def test_byte_two_way_pair_decoder(gpu, oracle, sb):
    """The reference's 2-way byte layout (main.cpp:226-280) through k_decode_byte_pairs, 32 chunks per wave: chunk sizes of one
    and many 64-round lines, of an odd number of half lines and of any multiple of 4 symbols, chunk counts that are no multiple of 32 and a ragged last
    chunk (its batch goes round by round), chunks on any byte boundary and in any order, the ORACLE's container and the
    GPU's own, streams that take two bytes per state and round and streams that take next to none; damage is flagged."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(80 + sb)
    zipf = oracle.gen_zipf(700000 + 55, K=256, s=1.0, seed=33)
    flat = rng.integers(0, 256, 260000).astype(np.uint8)
    heavy = np.where(rng.random(260000) < 0.93, 7, rng.integers(0, 256, 260000)).astype(np.uint8)
    cases = [(zipf, 1024), (zipf, 128), (zipf, 64), (zipf, 192), (zipf, 4032), (zipf, 16384), (zipf[:32 * 1024], 1024),
             (zipf[:71 * 256 + 9], 256), (flat, 1024), (flat, 320), (heavy, 512), (heavy, 4096),
             (zipf, 1000), (zipf, 100), (flat, 5004), (heavy, 68), (zipf, 252)]  # sizes off 64: one round at a time at the end
    for data, chunk in cases:
        om, gm = _models(R, ctx, oracle, FMT_BYTE, sb, data)
        want, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, 2, chunk, align=16)
        d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        back = torch.full((data.size + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        out = back[128:128 + data.size]
        ctx.decode(gm, d_cont, want.size, d_offs, d_lens, data.size, 2, chunk, d_out=out)
        assert ctx.last_decode_kernel() == "k_decode_byte_pairs", (ctx.last_decode_kernel(), chunk)
        assert np.array_equal(out.cpu().numpy(), data), (chunk, data.size)
        host = back.cpu().numpy()
        assert (host[:128] == 0xA5).all() and (host[128 + data.size:] == 0xA5).all(), chunk  # nothing outside the output
        cont, o2, l2, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 2, chunk)
        out = ctx.decode(gm, cont, total, o2, l2, data.size, 2, chunk)
        assert ctx.last_decode_kernel() == "k_decode_byte_pairs" and np.array_equal(out.cpu().numpy(), data), (chunk, "own container")
    # chunks packed back to front on odd byte boundaries; output 4 bytes into a line (2 bytes: the lane kernel)
    data, chunk = zipf[:100 * 256], 256
    om, gm = _models(R, ctx, oracle, FMT_BYTE, sb, data)
    want, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, 2, chunk, align=16)
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
    rev = np.zeros(want.size + 1024, np.uint8)
    roffs = np.zeros_like(offs)
    pos = 1
    for c in range(len(lens) - 1, -1, -1):
        roffs[c] = pos
        rev[pos:pos + lens[c]] = want[int(offs[c]):int(offs[c]) + int(lens[c])]
        pos += int(lens[c]) + (c % 7)
    for shift, kernel in ((4, "k_decode_byte_pairs"), (2, "k_decode_lanes")):
        back = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
        ctx.decode(gm, torch.from_numpy(rev).cuda(), pos, torch.from_numpy(roffs.astype(np.int64)).cuda(), d_lens, data.size, 2, chunk,
                   d_out=back[shift:shift + data.size])
        assert ctx.last_decode_kernel().startswith(kernel), (shift, ctx.last_decode_kernel())
        assert np.array_equal(back[shift:shift + data.size].cpu().numpy(), data), shift
    # rare symbols only: scale_bits bits per symbol, two bytes per state in most rounds at 16 bits
    f = np.ones(256, np.uint32)
    f[0] = (1 << sb) - 255
    rare = rng.integers(1, 256, 64 * 2048).astype(np.uint8)
    om2, gm2 = oracle.model(f, sb), ctx.model(FMT_BYTE, f, sb)
    w2, of2, le2 = oracle.encode_chunked(FMT_BYTE, om2, rare, 2, 2048, align=16)
    assert w2.size > rare.size * sb // 8 - 4096
    out = ctx.decode(gm2, torch.from_numpy(np.concatenate([w2, np.zeros(64, np.uint8)])).cuda(), w2.size,
                     torch.from_numpy(of2.astype(np.int64)).cuda(), torch.from_numpy(le2.astype(np.int32)).cuda(), rare.size, 2, 2048)
    assert ctx.last_decode_kernel() == "k_decode_byte_pairs" and np.array_equal(out.cpu().numpy(), rare)
    # damage: flipped stream bytes, a length that lies, an offset beyond the container; then random damage.  (Not at 8 bits:
    # 256 symbols of frequency 1 make the state a shift register -- a flipped byte is one wrong symbol and nothing else.)
    if sb < 12:
        return
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    for kind in range(3):
        bad, o, l = want.copy(), offs.astype(np.int64).copy(), lens.astype(np.int32).copy()
        if kind == 0:
            for at in rng.integers(0, want.size, 20):
                bad[at] ^= 0x40
        elif kind == 1:
            l[7] += 1
        else:
            o[11] = want.size + 1000
        ctx.decode(gm, torch.from_numpy(np.concatenate([bad, np.zeros(64, np.uint8)])).cuda(), want.size,
                   torch.from_numpy(o).cuda(), torch.from_numpy(l).cuda(), data.size, 2, chunk, sync=False)
        assert ctx.decode_errors() >= 1, kind
    for trial in range(9):
        bad, l = want.copy(), lens.astype(np.int32).copy()
        if trial % 3 == 0:
            at = rng.integers(0, bad.size, 40)
            bad[at] ^= rng.integers(1, 256, 40).astype(np.uint8)
        elif trial % 3 == 1:
            bad[rng.integers(0, bad.size):] = 0
        else:
            l[rng.integers(0, l.size, 3)] = rng.integers(0, 1 << 20, 3).astype(np.int32)
        out = ctx.decode(gm, torch.from_numpy(np.concatenate([bad, np.zeros(64, np.uint8)])).cuda(), bad.size, d_offs,
                         torch.from_numpy(l).cuda(), data.size, 2, chunk, sync=False)
        assert ctx.decode_errors() > 0 or np.array_equal(out.cpu().numpy(), data), trial


@pytest.mark.parametrize("placement", ["kernels", "fused"])
def test_rans64_two_way_lane_encoder(gpu, oracle, placement):
    """BASELINE config 2's encoder: k_encode_lanes_r64x2 (whole batches of 64 full chunks) + the staged kernel for the
    rest, with k_layout / k_compact_small behind them (the default) and placing their chunks themselves
    (RANS_AMD_OPT_LANE_FUSED_PLACEMENT: scanner wave per block, decoupled look-back over the blocks' rounds): every chunk, the
    index and the total against the oracle, for scale_bits at both ends of the cum2sym range, a ragged tail, models
    with frequency-1 symbols and a single-symbol model; a symbol outside the model is reported."""
    R, _, torch = gpu
    fused = placement == "fused"
    ctx = R.Context(0)
    ctx.set_option(R.OPT_LANE_FUSED_PLACEMENT, int(fused))
    rng = np.random.default_rng(11)
    n = 1600 * 64 * 64 + 64 * 33 + 17  # 1600 full batches of 64-symbol chunks, 33 full chunks and a ragged one behind
    zipf = oracle.gen_zipf(n, K=256, s=1.0, seed=29)
    narrow = oracle.gen_zipf(n, K=40, s=0.7, seed=31)
    cases = [(zipf, 14, 64, 256), (narrow, 7, 64, 40), (zipf, 16, 64, 256), (narrow, 12, 64, 40),
             (oracle.gen_zipf(1700 * 64 * 128 + 5, K=256, s=1.2, seed=37), 15, 128, 256)]
    for data, sb, chunk, nsyms in cases:
        om, gm = _models(R, ctx, oracle, FMT_R64, sb, data, nsyms)
        want, offs, lens = oracle.encode_chunked(FMT_R64, om, data, 2, chunk, align=16)
        cont, d_offs, d_lens, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 2, chunk)
        assert ctx.last_encode_kernel() == ("k_encode_lanes_r64x2", fused), ctx.last_encode_kernel()
        assert total == want.size, (sb, chunk)
        assert np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs), (sb, chunk)
        assert np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens), (sb, chunk)
        got = cont[:total].cpu().numpy()
        # every stream byte: compare through a mask of the bytes that belong to chunks (the padding is not defined)
        mask = np.zeros(want.size + 1, np.int32)
        np.add.at(mask, offs[:-1].astype(np.int64), 1)
        np.add.at(mask, offs[:-1].astype(np.int64) + lens.astype(np.int64), -1)
        inside = np.cumsum(mask[:-1]) > 0
        assert np.array_equal(got[inside], want[inside]), (sb, chunk)
        out = ctx.decode(gm, cont, total, d_offs, d_lens, data.size, 2, chunk)
        assert np.array_equal(out.cpu().numpy(), data), (sb, chunk)
    # frequency-1 symbols (the reciprocal is 2^64 - 1, rans64.h:173-181) and a model of one symbol (nothing is ever emitted)
    sb, chunk = 16, 64
    f = np.ones(256, np.uint32)
    f[0] = (1 << sb) - 255
    rare = rng.integers(0, 256, 1600 * 64 * 64).astype(np.uint8)
    one = np.zeros(256, np.uint32)
    one[7] = 1 << sb
    for freqs, data in ((f, rare), (one, np.full(1600 * 64 * 64 + 100, 7, np.uint8))):
        om, gm = oracle.model(freqs, sb), ctx.model(FMT_R64, freqs, sb)
        want, offs, lens = oracle.encode_chunked(FMT_R64, om, data, 2, chunk, align=16)
        cont, d_offs, d_lens, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 2, chunk)
        assert ctx.last_encode_kernel() == ("k_encode_lanes_r64x2", fused)
        assert total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens)
        assert np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs)
        got = cont[:total].cpu().numpy()
        for c in list(range(0, len(lens), 997)) + [len(lens) - 1]:
            o, ln = int(offs[c]), int(lens[c])
            assert np.array_equal(got[o:o + ln], want[o:o + ln]), c
    # a symbol the model has no frequency for
    om, gm = _models(R, ctx, oracle, FMT_R64, 12, narrow, 40)
    bad = narrow.copy()
    bad[123457] = 200
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode(gm, torch.from_numpy(bad).cuda(), 2, 64)
    assert e.value.status == R.E_MODEL


def _decode_oracle_container(R, ctx, torch, gm, want, offs, lens, n, n_ways, chunk, sync=True):
    d_cont = torch.from_numpy(np.concatenate([want, np.zeros(64, np.uint8)])).cuda()
    return ctx.decode(gm, d_cont, want.size, torch.from_numpy(offs.astype(np.int64)).cuda(),
                      torch.from_numpy(lens.astype(np.int32)).cuda(), n, n_ways, chunk, sync=sync)


def test_dual_chunk_alias_decoder(gpu, oracle):
    """SURVEY 8(f)4, "two chunks per wave" (the reference's own trick, main_simd.cpp:313-325): k_decode_dual takes 64-way
    alias containers of at least two chunks.  ORACLE-made containers over 4096 symbols (config 4: u16 symbols, own-slot
    counts as bytes), 256 symbols at 16 bits (bucket width 256: counts as u16) and small alphabets, with an even and an
    odd number of chunks, a ragged last chunk, chunk sizes that do and do not fill whole groups of rounds -- decoded by
    the dual kernel and, with the context option off, by the one-chunk-per-wave kernel; both must give the input.
    Damaged streams and index entries are counted, chunk by chunk, and never decoded out of bounds."""
    R, _, torch = gpu
    ctx = R.Context(0)
    cases = [(4096, 16, 8192, 8192 * 7 + 4097), (4096, 16, 2048, 2048 * 10), (4096, 12, 1024, 1024 * 9 + 1),
             (256, 16, 4096, 4096 * 12 + 77), (256, 12, 512, 512 * 33), (16, 8, 256, 256 * 5 + 255), (2, 9, 768, 768 * 4),
             (4096, 16, 8192 + 64, (8192 + 64) * 4), (256, 14, 4096, 4096 * 2)]
    for K, sb, chunk, n in cases:
        data = oracle.gen_zipf(n, K=K, s=1.0, seed=K + sb)
        view = data if K <= 256 else data.view(np.int16)
        om, gm = _models(R, ctx, oracle, FMT_ALIAS, sb, data, nsyms=K)
        want, offs, lens = oracle.encode_chunked(FMT_ALIAS, om, data, 64, chunk, align=16)
        ctx.set_option(R.OPT_DUAL_DECODE, 1)  # automatic: the dual kernel where the tables allow one block per CU only
        out = _decode_oracle_container(R, ctx, torch, gm, want, offs, lens, n, 64, chunk)
        assert ctx.last_decode_kernel() == ("k_decode_dual<alias>" if K >= 4096 else "k_decode<alias>"), (K, sb, chunk)
        assert np.array_equal(out.cpu().numpy(), view), (K, sb, chunk)
        ctx.set_option(R.OPT_DUAL_DECODE, 0)
        out = _decode_oracle_container(R, ctx, torch, gm, want, offs, lens, n, 64, chunk)
        assert ctx.last_decode_kernel() == "k_decode<alias>", (K, sb, chunk)
        assert np.array_equal(out.cpu().numpy(), view), (K, sb, chunk)
        ctx.set_option(R.OPT_DUAL_DECODE, 2)  # whenever the tables fit: every model of this test
        out = _decode_oracle_container(R, ctx, torch, gm, want, offs, lens, n, 64, chunk)
        assert ctx.last_decode_kernel() == "k_decode_dual<alias>", (K, sb, chunk)
        assert np.array_equal(out.cpu().numpy(), view), (K, sb, chunk)
        # the GPU encoder's container through the dual decoder as well
        cont, d_offs, d_lens, total = ctx.encode(gm, torch.from_numpy(view).cuda(), 64, chunk)
        assert total == want.size
        out = ctx.decode(gm, cont, total, d_offs, d_lens, n, 64, chunk)
        assert ctx.last_decode_kernel() == "k_decode_dual<alias>"
        assert np.array_equal(out.cpu().numpy(), view), (K, sb, chunk)
    # damage: a flipped stream byte in chunks 2 and 5 (one in each half of a pair), a misaligned index entry, a length
    # beyond the container -- each is counted, the rest of the container still decodes
    K, sb, chunk, n = 4096, 16, 4096, 4096 * 8
    data = oracle.gen_zipf(n, K=K, s=1.0, seed=5)
    om, gm = _models(R, ctx, oracle, FMT_ALIAS, sb, data, nsyms=K)
    want, offs, lens = oracle.encode_chunked(FMT_ALIAS, om, data, 64, chunk, align=16)
    bad = want.copy()
    for c in (2, 5):
        bad[int(offs[c]) + 64 * 4 + 100] ^= 0x5a
    with pytest.raises(R.RansAmdError) as e:
        _decode_oracle_container(R, ctx, torch, gm, bad, offs, lens, n, 64, chunk)
    assert e.value.status == R.E_CORRUPT
    out = _decode_oracle_container(R, ctx, torch, gm, bad, offs, lens, n, 64, chunk, sync=False)
    assert ctx.decode_errors() == 2
    got = out.cpu().numpy().view(np.uint16)
    for c in (0, 1, 3, 4, 6, 7):
        assert np.array_equal(got[c * chunk:(c + 1) * chunk], data[c * chunk:(c + 1) * chunk]), c
    offs2, lens2 = offs.copy(), lens.copy()
    offs2[3] += 4            # not 16-byte aligned: rejected
    lens2[6] = want.size     # runs past the container: rejected
    out = _decode_oracle_container(R, ctx, torch, gm, want, offs2, lens2, n, 64, chunk, sync=False)
    assert ctx.decode_errors() == 2
    got = out.cpu().numpy().view(np.uint16)
    for c in (0, 1, 2, 4, 5, 7):  # the partner of a rejected chunk is decoded on its own
        assert np.array_equal(got[c * chunk:(c + 1) * chunk], data[c * chunk:(c + 1) * chunk]), c
    ctx.close()


def test_headline_configuration_runs_the_headline_kernel(gpu, oracle):
    """A dispatch regression from k_decode_word64 to the general k_decode<word> would keep every parity test green and
    cost 2.5 %: the 64-way word format with u8 symbols and aligned output must run the hand-scheduled kernel."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(32768 * 5 + 100, K=256, s=1.0, seed=2)
    om, gm = _models(R, ctx, oracle, FMT_WORD, 12, data)
    for chunk in (32768, 16384, 4096):
        want, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, 64, chunk, align=16)
        out = _decode_oracle_container(R, ctx, torch, gm, want, offs, lens, data.size, 64, chunk)
        assert ctx.last_decode_kernel() == "k_decode_word64", ctx.last_decode_kernel()
        assert np.array_equal(out.cpu().numpy(), data)


def test_fused_encoder_scratch_ring(gpu, oracle):
    """Context option RANS_AMD_OPT_ENC_SCRATCH_RING: the fused wave encoders code into a ring of scratch slots per coding
    wave (kernels.h kEncRingSlots), reused once the block's copier has drained them -- taken when there are more chunks
    than ring slots.  Forty thousand small chunks make every wave go round its ring several times; the container must be
    the oracle's, byte for byte, index and all.  (The 1 GiB shape runs in test_gpu_scale.)"""
    R, _, torch = gpu
    ctx = R.Context(0)
    ctx.set_option(R.OPT_ENC_SCRATCH_RING, 1)
    for fmt, sb, chunk in ((FMT_WORD, 12, 128), (FMT_BYTE, 14, 192), (FMT_R64, 14, 64 * 3)):
        n = chunk * 40000 + 77
        data = oracle.gen_zipf(n, K=256, s=1.0, seed=chunk)
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want, offs, lens = oracle.encode_chunked_mt(fmt, om, data, 64, chunk, align=16)
        for _ in range(2):  # (a second launch reuses the context's status words and scratch)
            cont, d_offs, d_lens, total = ctx.encode(gm, torch.from_numpy(data).cuda(), 64, chunk)
            assert ctx.last_encode_kernel()[1] is True
            assert total == want.size
            assert np.array_equal(d_offs.cpu().numpy().astype(np.uint64), offs)
            assert np.array_equal(d_lens.cpu().numpy().astype(np.uint32), lens)
            assert oracle.compare_container(fmt, om, data, 64, chunk, cont[:total].cpu().numpy(), offs, lens) == (len(lens), -1)
        out = ctx.decode(gm, cont, total, d_offs, d_lens, n, 64, chunk)
        assert np.array_equal(out.cpu().numpy(), data)


def test_hip_graph_capture(gpu, oracle):
    """rans_amd_encode + rans_amd_decode captured into a hipGraph (torch.cuda.graph = hipStreamBeginCapture on the
    stream the calls are issued on): every replay codes the chunks again -- the container equals the oracle's, the
    symbols come back -- eager launches before, between and after keep working (the work-counter ring of the eager
    launches never sees a slot a replay used), and the rules of include/ryg_rans_amd.h are enforced: no host result
    pointer inside a capture, no workspace growth inside a capture."""
    import subprocess
    import sys
    R, ctx, torch = gpu
    data = oracle.gen_zipf((1 << 20) + 777, K=256, s=1.0, seed=11)
    n = data.size
    d = torch.from_numpy(data).cuda()
    for fmt, sb, ways, chunk in ((FMT_WORD, 12, 64, 4096), (FMT_BYTE, 14, 64, 8192), (FMT_R64, 14, 2, 512), (FMT_ALIAS, 16, 64, 4096),
                                 (FMT_WORD, 12, 256, 16384)):
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want, w_offs, w_lens = oracle.encode_chunked(fmt, om, data, ways, chunk, align=16)
        # once outside the capture: the workspaces exist afterwards
        cont, offs, lens, total = ctx.encode(gm, d, ways, chunk)
        assert total == want.size
        out = ctx.decode(gm, cont, total, offs, lens, n, ways, chunk)
        assert torch.equal(out, d)
        cont2, offs2, lens2, out2 = torch.zeros_like(cont), torch.zeros_like(offs), torch.zeros_like(lens), torch.zeros_like(out)
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.encode(gm, d, ways, chunk, d_out=cont2, sync=False, d_offsets=offs2, d_lengths=lens2)
            ctx.decode(gm, cont2, total, offs2, lens2, n, ways, chunk, d_out=out2, sync=False)
        assert not bool(out2.any())  # capturing ran nothing
        for rep in range(3):
            cont2.zero_(); offs2.zero_(); lens2.zero_(); out2.zero_()
            g.replay()
            torch.cuda.synchronize()
            ctx.encode_status()
            assert ctx.decode_errors() == 0
            nchunks = len(w_lens)
            assert np.array_equal(offs2.cpu().numpy()[:nchunks].astype(np.int64), w_offs[:nchunks].astype(np.int64)), (fmt, ways, rep)
            assert np.array_equal(lens2.cpu().numpy()[:nchunks].astype(np.int64), w_lens.astype(np.int64)), (fmt, ways, rep)
            assert int(offs2[nchunks].item()) == total
            got = cont2[:total].cpu().numpy()
            for c in range(len(w_lens)):
                o, ln = int(w_offs[c]), int(w_lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (fmt, ways, rep, c)
            assert torch.equal(out2, d), (fmt, ways, rep)
            # an eager decode between the replays (the ring moves on; the graph's slot is its own)
            out.zero_()
            ctx.decode(gm, cont2, total, offs2, lens2, n, ways, chunk, d_out=out)
            assert torch.equal(out, d)
        # more eager launches than the ring has slots, then one more replay
        for _ in range(70):
            ctx.decode(gm, cont, total, offs, lens, n, ways, chunk, d_out=out, sync=False)
        out2.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert ctx.decode_errors() == 0 and torch.equal(out2, d) and torch.equal(out, d)
        del g
    # the rules, in a process of its own (a refused call inside a capture must leave this one alone)
    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import ryg_rans_amd as R
ctx = R.Context(0)
f = np.zeros(256, np.uint32); f[:4] = [1024, 1024, 1024, 1024]
m = ctx.model(R.FMT_WORD, f, 12)
d = torch.randint(0, 4, (1 << 18,), dtype=torch.uint8, device="cuda")
cap = R.encode_bound(R.FMT_WORD, d.numel(), 64, 4096) + 16
cont = torch.zeros(cap, dtype=torch.uint8, device="cuda")
offs = torch.zeros(65, dtype=torch.int64, device="cuda"); lens = torch.zeros(64, dtype=torch.int32, device="cuda")
codes = []
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    try:  # no warm-up call: the scratch would have to be allocated inside the capture
        ctx.encode(m, d, 64, 4096, d_out=cont, sync=False, d_offsets=offs, d_lengths=lens)
        codes.append(0)
    except R.RansAmdError as e:
        codes.append(e.status)
ctx.encode(m, d, 64, 4096, d_out=cont, d_offsets=offs, d_lengths=lens)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s):
    try:  # a host result pointer inside a capture
        ctx.encode(m, d, 64, 4096, d_out=cont, sync=True, d_offsets=offs, d_lengths=lens)
        codes.append(0)
    except R.RansAmdError as e:
        codes.append(e.status)
    ctx.encode(m, d, 64, 4096, d_out=cont, sync=False, d_offsets=offs, d_lengths=lens)
g2.replay(); torch.cuda.synchronize(); ctx.encode_status()
print("codes", codes, R.E_ARG)
assert codes == [R.E_ARG, R.E_ARG]
print("capture rules ok")
""" % os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "capture rules ok" in r.stdout, r.stdout + r.stderr


def test_group_kernels_inside_a_hip_graph(gpu, oracle):
    """The kernels of decode_groups.hip / encode_groups.hip hand their octets and batches out through claim counters: captured
    launches get counters of their own, zeroed by a node in front of the kernel, so that every replay starts from zero -- the
    compact and the sized encode of the 8-way word layout and both group decoders, three replays each, against the oracle."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(512 * 1024, K=256, s=1.0, seed=12)
    n = data.size
    d = torch.from_numpy(data).cuda()
    for fmt, sb, ways, chunk, enc_kernel, dec_kernel in ((FMT_WORD, 12, 8, 1024, "k_encode_word_groups", "k_decode_word_groups"),
                                                         (FMT_BYTE, 14, 2, 1024, "k_encode_lanes", "k_decode_byte_pairs")):
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        want, w_offs, w_lens = oracle.encode_chunked(fmt, om, data, ways, chunk, align=16)
        nchunks = len(w_lens)
        cont, offs, lens, total = ctx.encode(gm, d, ways, chunk)  # once outside the capture: the workspaces exist afterwards
        assert total == want.size and ctx.last_encode_kernel()[0].startswith(enc_kernel), ctx.last_encode_kernel()
        s_cont, s_offs, s_lens, s_total, slot = ctx.encode_sized(gm, d, ways, chunk)
        out = ctx.decode(gm, s_cont, s_total, s_offs, s_lens, n, ways, chunk)
        assert torch.equal(out, d) and ctx.last_decode_kernel() == dec_kernel, ctx.last_decode_kernel()
        cont2, offs2, lens2 = torch.zeros_like(cont), torch.zeros_like(offs), torch.zeros_like(lens)
        s_cont2, s_offs2, s_lens2 = torch.zeros_like(s_cont), torch.zeros_like(s_offs), torch.zeros_like(s_lens)
        out2, out3 = torch.zeros_like(out), torch.zeros_like(out)
        stream = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            ctx.encode(gm, d, ways, chunk, d_out=cont2, sync=False, d_offsets=offs2, d_lengths=lens2)
            ctx.decode(gm, cont2, total, offs2, lens2, n, ways, chunk, d_out=out2, sync=False)
            ctx.encode_sized(gm, d, ways, chunk, slot=slot, d_out=s_cont2, sync=False, d_offsets=s_offs2, d_lengths=s_lens2)
            ctx.decode(gm, s_cont2, s_total, s_offs2, s_lens2, n, ways, chunk, d_out=out3, sync=False)
        for rep in range(3):
            for t in (cont2, offs2, lens2, s_cont2, s_offs2, s_lens2, out2, out3):
                t.zero_()
            g.replay()
            torch.cuda.synchronize()
            ctx.encode_status()
            assert ctx.decode_errors() == 0
            assert np.array_equal(offs2.cpu().numpy().astype(np.uint64)[:nchunks], w_offs[:nchunks]) and int(offs2[nchunks].item()) == total
            assert np.array_equal(lens2.cpu().numpy().astype(np.uint32), w_lens) and np.array_equal(s_lens2.cpu().numpy().astype(np.uint32), w_lens)
            got, sgot, so = cont2[:total].cpu().numpy(), s_cont2.cpu().numpy(), s_offs2.cpu().numpy()
            for c in range(nchunks):
                o, ln, a = int(w_offs[c]), int(w_lens[c]), int(so[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (fmt, rep, c)
                assert np.array_equal(sgot[a:a + ln], want[o:o + ln]), (fmt, rep, c, "sized")
            assert torch.equal(out2, d) and torch.equal(out3, d), (fmt, rep)
            ctx.decode(gm, cont2, total, offs2, lens2, n, ways, chunk, d_out=out)  # an eager launch between the replays
            assert torch.equal(out, d)
        del g


def test_byte_encoder_staging_extremes(gpu, oracle):
    """The byte encoder's staged sub-step (enc_byte_full_staged, models up to 15 bits) at the edges of its window: every
    symbol the rarest one (15 bits each: 1920 bytes per sixteen rounds of a wave, the 2 KiB window nearly full, both passes
    of the flush), nothing but the commonest (flushes with nothing to write), bursts of both, ragged chunks whose tail
    rounds store for themselves -- and the 16-bit model next to it, whose window is flushed every eight rounds (17 x 64
    bytes; sixteen rounds would be 33 x 64, more than the window)."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(23)
    n = 1 << 17
    for sb in (15, 14, 9, 16):
        M = 1 << sb
        f = np.ones(256, np.uint32)
        f[3] = M - 255
        rare = rng.integers(0, 255, n).astype(np.uint8)
        rare[rare >= 3] += 1
        common = np.full(n, 3, np.uint8)
        bursts = rare.copy()
        bursts[(np.arange(n) // 3000) % 2 == 0] = 3
        om = oracle.model(f, sb)
        gm = ctx.model(FMT_BYTE, f, sb)
        for name, data in (("rare", rare), ("common", common), ("bursts", bursts)):
            d = torch.from_numpy(data).cuda()
            for n_ways, chunk in ((64, 8192), (64, n), (64, 5000), (64, 1031), (64, 2048 + 64 * 7 + 5), (128, 16384)):
                want, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, n_ways, chunk, align=16)
                cont, d_offs, d_lens, total = ctx.encode(gm, d, n_ways, chunk)
                assert total == want.size, (sb, name, n_ways, chunk)
                assert np.array_equal(d_lens.cpu().numpy().astype(np.int64), lens.astype(np.int64)), (sb, name, n_ways, chunk)
                got = cont[:total].cpu().numpy()
                for c in range(len(lens)):
                    o, ln = int(offs[c]), int(lens[c])
                    assert np.array_equal(got[o:o + ln], want[o:o + ln]), (sb, name, n_ways, chunk, c)
                out = ctx.decode(gm, cont, total, d_offs, d_lens, n, n_ways, chunk)
                assert np.array_equal(out.cpu().numpy(), data), (sb, name, n_ways, chunk)


def test_byte_encoder_sparse_model_and_stray_symbols(gpu, oracle):
    """The mirrored byte sub-step comes in two variants: models in which every byte value has a frequency skip the search
    for symbols without a record (the dense models of the test above), all others OR-accumulate a marker of the records they
    touch.  Here the other kind -- 100 of 256 symbols, frequencies of every size class -- in both layouts against the
    oracle, and a stray symbol deep inside the full-wave part of a chunk: RANS_AMD_E_MODEL from either entry point."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(29)
    n = 1 << 17
    for sb in (12, 14):
        M = 1 << sb
        f = np.zeros(256, np.uint32)
        live = np.sort(rng.choice(256, 100, replace=False))
        w = rng.integers(1, 50, 100).astype(np.float64) ** 2
        q = np.maximum(1, np.floor(w / w.sum() * (M - 100)).astype(np.int64))
        q[0] += M - q.sum()
        f[live] = q
        assert f.sum() == M and (f[live] > 0).all()
        data = rng.choice(live, n, p=f[live] / float(M)).astype(np.uint8)
        om, gm = oracle.model(f, sb), ctx.model(FMT_BYTE, f, sb)
        d = torch.from_numpy(data).cuda()
        for chunk in (8192, 5000):
            want, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, 64, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d, 64, chunk)
            s_cont, s_offs, s_lens, s_total = ctx.encode_slots(gm, d, 64, chunk)
            assert total == want.size and np.array_equal(d_lens.cpu().numpy().astype(np.int64), lens.astype(np.int64))
            assert np.array_equal(s_lens.cpu().numpy().astype(np.int64), lens.astype(np.int64))
            got, sgot, so = cont[:total].cpu().numpy(), s_cont.cpu().numpy(), s_offs.cpu().numpy()
            for c in range(len(lens)):
                o, ln, a = int(offs[c]), int(lens[c]), int(so[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (sb, chunk, c)
                assert np.array_equal(sgot[a:a + ln], want[o:o + ln]), (sb, chunk, c)
            out = ctx.decode(gm, s_cont, s_total, s_offs, s_lens, n, 64, chunk)
            assert np.array_equal(out.cpu().numpy(), data), (sb, chunk)
        stray = [v for v in range(256) if f[v] == 0]
        for bad_sym in (stray[0], stray[-1]):
            bad = data.copy()
            bad[3 * 8192 + 4321] = bad_sym
            for call in (ctx.encode, ctx.encode_slots):
                with pytest.raises(R.RansAmdError) as e:
                    call(gm, torch.from_numpy(bad).cuda(), 64, 8192)
                assert e.value.status == R.E_MODEL, (sb, bad_sym)


@pytest.mark.parametrize("fmt,sb", [(FMT_WORD, 12), (FMT_BYTE, 14), (FMT_BYTE, 16)])
def test_staged_encoders_with_the_three_kernel_placement(gpu, oracle, fmt, sb):
    """RANS_AMD_OPT_FUSED_PLACEMENT = 0 (k_encode + k_layout + k_compact): the coding kernel is the same template with
    16-wave blocks and no copier -- its LDS staging windows sit behind the same tables -- and must write the same bytes."""
    R, _, torch = gpu
    ctx = R.Context(0)
    ctx.set_option(R.OPT_FUSED_PLACEMENT, 0)
    try:
        data = oracle.gen_zipf(700001, K=256, s=1.1, seed=31)
        om, gm = _models(R, ctx, oracle, fmt, sb, data)
        d = torch.from_numpy(data).cuda()
        for n_ways, chunk in ((64, 4096), (64, 5000), (64, 32768), (128, 8192)):
            want, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
            cont, d_offs, d_lens, total = ctx.encode(gm, d, n_ways, chunk)
            assert ctx.last_encode_kernel()[1] is False  # not the fused placement
            assert total == want.size
            got = cont[:total].cpu().numpy()
            for c in range(len(lens)):
                o, ln = int(offs[c]), int(lens[c])
                assert np.array_equal(got[o:o + ln], want[o:o + ln]), (n_ways, chunk, c)
            out = ctx.decode(gm, cont, total, d_offs, d_lens, data.size, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy(), data)
    finally:
        ctx.close()


@pytest.mark.parametrize("sb", [8, 10, 12])
def test_byte_format_slot_record_decoder(gpu, oracle, sb):
    """Byte format with scale_bits <= 12: the wave-per-chunk decoders read ONE fused 8-byte record per symbol
    ({freq | sym << 24, slot - start}, rans_byte.h:125-128 + 291-298 in the form of rans_word_sse41.h:123-131) instead of
    cum2sym + {freq, start}: containers made by the ORACLE decode to the input for every lane count, ragged chunks, skewed
    and one-symbol models; at 13 bits and more, and for the lane-per-chunk interleaves, the two-gather tables stay."""
    R, ctx, torch = gpu
    rng = np.random.default_rng(sb)
    inputs = {"zipf": oracle.gen_zipf(250003, K=256, s=1.0, seed=2),
              "two": (rng.integers(0, 2, 90001) * 255).astype(np.uint8),
              "one": np.full(70000, 7, np.uint8)}
    for name, data in inputs.items():
        om, gm = _models(R, ctx, oracle, FMT_BYTE, sb, data)
        for n_ways, chunk in ((64, 4096), (64, 5000), (128, 8192), (256, 16384), (33, 1000), (512, 32768)):
            cont, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, n_ways, chunk, align=16)
            d_cont = torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint8)])).cuda()
            d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
            d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
            out = ctx.decode(gm, d_cont, cont.size, d_offs, d_lens, data.size, n_ways, chunk)
            assert np.array_equal(out.cpu().numpy(), data), (name, n_ways, chunk)
            assert ctx.last_decode_kernel() == "k_decode<byte, slot records>", ctx.last_decode_kernel()
            # a flipped stream byte is caught by the same integrity check
            bad = d_cont.clone()
            bad[int(offs[1]) + int(lens[1]) // 2] ^= 0x55
            out2 = torch.empty_like(out)
            ctx.decode(gm, bad, cont.size, d_offs, d_lens, data.size, n_ways, chunk, d_out=out2, sync=False)
            assert ctx.decode_errors() >= 1 or not np.array_equal(out2.cpu().numpy(), data)
    data = inputs["zipf"]
    om, gm = _models(R, ctx, oracle, FMT_BYTE, 13, data)
    cont, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, 64, 4096, align=16)
    out = ctx.decode(gm, torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint8)])).cuda(), cont.size,
                     torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(), data.size, 64, 4096)
    assert np.array_equal(out.cpu().numpy(), data) and ctx.last_decode_kernel() == "k_decode<byte>"
    om, gm = _models(R, ctx, oracle, FMT_BYTE, sb, data)
    cont, offs, lens = oracle.encode_chunked(FMT_BYTE, om, data, 2, 512, align=16)
    out = ctx.decode(gm, torch.from_numpy(np.concatenate([cont, np.zeros(64, np.uint8)])).cuda(), cont.size,
                     torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(), data.size, 2, 512)
    assert np.array_equal(out.cpu().numpy(), data) and ctx.last_decode_kernel() == "k_decode_byte_pairs"  # (keeps cum2sym + records)
