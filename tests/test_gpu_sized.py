"""Sized slots (rans_amd_encode_slots_sized, -m gpu): one trip through HBM AND a container about as large as the compact one.

The reference sizes its encoder's buffer from the input, not from the worst case (main_simd.cpp:145: n + n/8 + 128).  Here a
slot is rans_amd_tight_slot_bytes() -- the model's expected chunk stream and a little -- and a chunk that does not fit is
abandoned by its coder and coded again by a second launch into a worst-case slot behind the sized ones.  Checked against
the CPU oracle:

  * every chunk's bytes == the oracle's stream of that chunk, WHEREVER it lies (its own slot or the overflow region), for
    every format and kernel family (wave encoders staged and unstaged, 64..512 lanes, narrow interleaves on the wave
    encoder and on the lane encoders, the 2-way rans64 kernel, u16 symbols, the 4096-symbol alias model);
  * offsets: a chunk that fits ends at its slot's end; an overflowed one lies in the k-th worst-case slot behind the
    sized ones; offsets[n_chunks] = n_chunks * slot + overflowed * W;
  * forced overflow: a stretch of uniform-random bytes inside a skewed model, slots of 64 bytes (EVERY chunk overflows),
    an overflow region that is too small (RANS_AMD_E_SPACE), a slot at or above the worst case (plain slot layout);
  * the container decodes as it is with every decoder family, and compacts to the oracle's compact container.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    import ryg_rans_amd as R
    ctx = R.Context(0)
    yield R, ctx, torch
    ctx.close()


def _models(ctx, oracle, fmt, sb, data, nsyms=256):
    f, _ = oracle.normalize(oracle.count_freqs(data, nsyms), 1 << sb)
    return oracle.model(f, sb, with_alias=(fmt == FMT_ALIAS)), ctx.model(fmt, f, sb)


def _dev(torch, data):
    return torch.from_numpy(data.view(np.int16) if data.dtype == np.uint16 else data).cuda()


def _check_sized(R, ctx, torch, oracle, fmt, om, gm, data, n_ways, chunk, slot=None, overflow_chunks=None, want_kernel=None):
    """Encode with sized slots; compare every chunk with the oracle; returns (overflowed chunk numbers, slot, total)."""
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    nchunks = len(lens)
    d_syms = _dev(torch, data)
    worst = R.slot_bytes(fmt, data.size, n_ways, chunk)
    if overflow_chunks is None:
        overflow_chunks = nchunks
    g_cont, g_offs, g_lens, total, slot = ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, overflow_chunks=overflow_chunks)
    assert slot % 64 == 0
    if want_kernel:
        assert ctx.last_encode_kernel()[0] == want_kernel, ctx.last_encode_kernel()
    go, gl = g_offs.cpu().numpy().astype(np.uint64), g_lens.cpu().numpy().astype(np.uint32)
    assert np.array_equal(gl, lens), "lengths differ from the oracle's"
    g = g_cont.cpu().numpy()
    if slot >= worst:  # nothing can overflow: the plain slot layout
        assert total == nchunks * worst
        slot = worst
    in_slot = go[:-1] < np.uint64(nchunks * slot)
    over = np.nonzero(~in_slot)[0]
    # a chunk that fits ends at its slot's end; an overflowed one ends at the end of a worst-case slot of the region
    ends = go[:-1] + gl
    assert np.array_equal(ends[in_slot], (np.nonzero(in_slot)[0].astype(np.uint64) + 1) * np.uint64(slot))
    if over.size:
        k = (ends[over] - np.uint64(nchunks * slot))
        assert np.all(k % np.uint64(worst) == 0)
        assert sorted((k // np.uint64(worst)).tolist()) == list(range(1, over.size + 1)), "overflow slots are not 0..k-1, one each"
    assert int(go[-1]) == total == nchunks * slot + over.size * worst
    # the chunks that overflowed are the ones that cannot fit (the staged coders check exactly; the others by the round's
    # worst case: a chunk may overflow although its stream would have fitted, never the other way round)
    assert np.all(lens[in_slot] <= slot), "a chunk longer than its slot was left in it"
    for c in range(nchunks):
        a, b = int(go[c]), int(go[c]) + int(gl[c])
        assert np.array_equal(g[a:b], cont[int(offs[c]):int(offs[c]) + int(lens[c])]), "chunk %d differs from the oracle" % c
    out = ctx.decode(gm, g_cont, total, g_offs, g_lens, data.size, n_ways, chunk)
    assert torch.equal(out, d_syms)
    d_dst, d_doffs, ctotal = ctx.compact(g_cont, total, g_offs, g_lens, nchunks)
    assert ctotal == cont.size and np.array_equal(d_doffs.cpu().numpy().astype(np.uint64), offs)
    c2 = d_dst.cpu().numpy()
    for c in range(nchunks):
        a, b = int(offs[c]), int(offs[c]) + int(lens[c])
        assert np.array_equal(c2[a:b], cont[a:b]), "compacted chunk %d differs" % c
    return over, slot, total


@pytest.mark.parametrize("fmt,sb", [(FMT_WORD, 12), (FMT_BYTE, 14), (FMT_BYTE, 16), (FMT_BYTE, 12), (FMT_R64, 14), (FMT_ALIAS, 16),
                                    (FMT_ALIAS, 12)])
@pytest.mark.parametrize("n_ways,chunk", [(64, 4096), (64, 5000), (64, 16384), (256, 16384), (128, 4096), (512, 8192), (33, 1000),
                                          (2, 512), (1, 1000), (4, 2048), (8, 4096), (2, 4095)])
def test_sized_slots_every_chunk_matches_oracle(gpu, oracle, fmt, sb, n_ways, chunk):
    """The model's own data in tight slots: (almost) nothing overflows, the container is about the compact one's size."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(300000, K=256, s=1.0, seed=1)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    over, slot, total = _check_sized(R, ctx, torch, oracle, fmt, om, gm, data, n_ways, chunk)
    nchunks = (data.size + chunk - 1) // chunk
    # wide interleaves of short chunks are dominated by the flushed states and the conservative round checks; everything
    # else stays near the compact size
    if n_ways <= 64 and chunk >= 4096:
        assert over.size <= max(1, nchunks // 16), (over.size, nchunks)


def test_sized_slots_of_the_bench_configs(gpu, oracle):
    """The shapes bench.py times: coding kernels, and container <= 0.90 x input for Zipf(256) word / byte at 32 Ki chunks."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(48 * 32768 + 777, K=256, s=1.0, seed=1)
    for fmt, sb, kern in ((FMT_WORD, 12, "k_encode<word>"), (FMT_BYTE, 14, "k_encode<byte>"), (FMT_BYTE, 12, "k_encode<byte>")):
        om, gm = _models(ctx, oracle, fmt, sb, data)
        over, slot, total = _check_sized(R, ctx, torch, oracle, fmt, om, gm, data, 64, 32768, want_kernel=kern)
        assert over.size == 0 and total <= 0.90 * data.size, (fmt, slot, total, data.size)
        over, slot, total = _check_sized(R, ctx, torch, oracle, fmt, om, gm, data, 64, 16384, want_kernel=kern)
        assert over.size <= 1 and total <= 0.93 * data.size, (fmt, slot, total, data.size)
    d2 = oracle.gen_zipf(512 * 64 * 300 + 100, K=256, s=1.0, seed=2)  # >= one batch of 64 chunks per CU + a ragged tail
    om, gm = _models(ctx, oracle, FMT_R64, 14, d2)
    over, slot, total = _check_sized(R, ctx, torch, oracle, FMT_R64, om, gm, d2, 2, 512, want_kernel="k_encode_lanes_r64x2")
    assert total <= 1.15 * d2.size, (slot, total, d2.size, over.size)
    d16 = oracle.gen_zipf(40 * 16384 + 5, K=4096, s=1.0, seed=1)
    om, gm = _models(ctx, oracle, FMT_ALIAS, 16, d16, nsyms=4096)
    over, slot, total = _check_sized(R, ctx, torch, oracle, FMT_ALIAS, om, gm, d16, 64, 16384, want_kernel="k_encode<alias, LDS remap>")
    assert total <= 0.70 * d16.size * 2, (slot, total, over.size)


@pytest.mark.parametrize("fmt,sb,n_ways,chunk", [(FMT_WORD, 12, 64, 8192), (FMT_BYTE, 14, 64, 8192), (FMT_BYTE, 16, 64, 8192),
                                                 (FMT_ALIAS, 16, 64, 8192), (FMT_ALIAS, 12, 64, 8192), (FMT_R64, 14, 64, 8192),
                                                 (FMT_WORD, 12, 256, 8192), (FMT_R64, 14, 2, 512), (FMT_BYTE, 14, 4, 2048),
                                                 (FMT_WORD, 12, 8, 1024)])
def test_forced_overflow_inside_a_skewed_model(gpu, oracle, fmt, sb, n_ways, chunk):
    """A stretch of uniform-random bytes (8 bits per symbol) inside Zipf data, coded with the model of the whole: the chunks
    of the stretch do not fit slots sized for the model's entropy, are abandoned and coded again behind the slots -- and
    every chunk still equals the oracle's stream."""
    R, ctx, torch = gpu
    n = (64 * 300 + 7) * 512 if n_ways == 2 else 60 * chunk + 321
    data = oracle.gen_zipf(n, K=256, s=1.0, seed=4).copy()
    rng = np.random.default_rng(5)
    lo, hi = 7 * chunk + 100, 12 * chunk + 50  # chunks 7..12 hold random bytes (7 and 12 partly)
    data[lo:hi] = rng.integers(0, 256, hi - lo).astype(np.uint8)
    data[40 * chunk:41 * chunk] = rng.integers(0, 256, chunk).astype(np.uint8)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    over, slot, total = _check_sized(R, ctx, torch, oracle, fmt, om, gm, data, n_ways, chunk)
    _, _, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    must = {c for c in range(len(lens)) if int(lens[c]) > slot}  # (what cannot fit MUST have moved; the check is conservative beyond)
    assert must <= set(over.tolist()), (must, over)
    if n_ways <= 64:  # (wide interleaves of short chunks: states and the round's margin dominate the slot, random bytes may fit)
        assert set(range(8, 12)) | {40} <= must, (must, slot)
    if over.size == 0:
        return
    # ... and with room for exactly that many: still fine; for one fewer: RANS_AMD_E_SPACE
    d_syms = _dev(torch, data)
    ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, overflow_chunks=over.size)
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, overflow_chunks=over.size - 1)
    assert e.value.status == R.E_SPACE
    # asynchronously: the verdict comes from rans_amd_encode_status
    out = torch.empty(R.encode_sized_bound(fmt, data.size, n_ways, chunk, slot, over.size - 1), dtype=torch.uint8, device="cuda")
    ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, d_out=out, sync=False)
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_status()
    assert e.value.status == R.E_SPACE
    # a healthy call right behind starts clean
    ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, overflow_chunks=over.size, sync=False)
    ctx.encode_status()


@pytest.mark.parametrize("fmt,sb,n_ways,chunk", [(FMT_WORD, 12, 64, 4096), (FMT_BYTE, 14, 64, 4096), (FMT_R64, 14, 2, 512),
                                                 (FMT_ALIAS, 16, 128, 4096), (FMT_R64, 14, 1, 700)])
def test_every_chunk_overflows_a_64_byte_slot(gpu, oracle, fmt, sb, n_ways, chunk):
    R, ctx, torch = gpu
    n = (64 * 280 + 3) * 512 if (n_ways == 2 and chunk == 512) else 50 * chunk + 17
    data = oracle.gen_zipf(n, K=256, s=1.0, seed=6)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    over, slot, total = _check_sized(R, ctx, torch, oracle, fmt, om, gm, data, n_ways, chunk, slot=64)
    nchunks = (n + chunk - 1) // chunk
    assert over.size >= nchunks - 1  # (a ragged last chunk of a few symbols may fit 64 bytes)


def test_sized_argument_checks_and_the_worst_case_slot(gpu, oracle):
    R, ctx, torch = gpu
    data = oracle.gen_zipf(100000, K=256, s=1.0, seed=7)
    om, gm = _models(ctx, oracle, FMT_WORD, 12, data)
    d_syms = _dev(torch, data)
    worst = R.slot_bytes(FMT_WORD, data.size, 64, 4096)
    tight = ctx.tight_slot_bytes(gm, 64, 4096)
    assert 64 <= tight < worst and tight % 64 == 0
    # the estimate is the model's entropy: 4096 symbols at H bits + 2 % + 256 bytes of states + 4 sigma + a line
    assert 0.75 * 4096 < tight < 0.95 * 4096, tight
    for bad in (0, 1, 100, tight + 1):
        with pytest.raises(R.RansAmdError) as e:
            ctx.encode_sized(gm, d_syms, 64, 4096, slot=bad)
        assert e.value.status == R.E_ARG
    # out_cap below the slots: known up front
    small = torch.empty(25 * tight - 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_sized(gm, d_syms, 64, 4096, slot=tight, d_out=small)
    assert e.value.status == R.E_SPACE
    # a slot at or above the worst case is the plain slot layout
    for s in (worst, worst + 64, 10 * worst):
        over, slot, total = _check_sized(R, ctx, torch, oracle, FMT_WORD, om, gm, data, 64, 4096, slot=s)
        assert over.size == 0 and total == 25 * worst
    assert R.encode_sized_bound(FMT_WORD, data.size, 64, 4096, tight, 3) == 25 * tight + 3 * worst
    assert R.encode_sized_bound(FMT_WORD, data.size, 64, 4096, worst, 3) == 25 * worst
    # no symbols: an empty index
    empty = torch.empty(0, dtype=torch.uint8, device="cuda")
    c, o, l, total, slot = ctx.encode_sized(gm, empty, 64, 4096, slot=tight)
    assert total == 0 and int(o[0]) == 0


def test_sized_slots_u16_symbols_and_the_4096_symbol_alias_model(gpu, oracle):
    R, ctx, torch = gpu
    d16 = oracle.gen_zipf(200001, K=4096, s=1.0, seed=3)
    for fmt, sb, n_ways, chunk in ((FMT_ALIAS, 16, 64, 8192), (FMT_ALIAS, 16, 128, 8192), (FMT_WORD, 12, 64, 8192), (FMT_ALIAS, 16, 2, 512)):
        om, gm = _models(ctx, oracle, fmt, sb, d16, nsyms=4096)
        _check_sized(R, ctx, torch, oracle, fmt, om, gm, d16, n_ways, chunk)
    # forced overflow with u16 symbols: a stretch of uniform symbols
    d = d16.copy()
    rng = np.random.default_rng(8)
    d[5 * 8192:8 * 8192] = rng.integers(0, 4096, 3 * 8192).astype(np.uint16)
    om, gm = _models(ctx, oracle, FMT_ALIAS, 16, d, nsyms=4096)
    over, slot, total = _check_sized(R, ctx, torch, oracle, FMT_ALIAS, om, gm, d, 64, 8192)
    assert {5, 6, 7} <= set(over.tolist())


@pytest.mark.parametrize("sb", [5, 20, 31])
def test_sized_slots_rans64_search_variant(gpu, oracle, sb):
    """rans64 outside 7..16 probability bits (the full-width encoder, symbols found by search in the decoder): sized slots,
    the model's own size and a forced overflow."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(150000, K=20, s=1.0, seed=9)
    om, gm = _models(ctx, oracle, FMT_R64, sb, data, nsyms=20)
    for n_ways, chunk in ((64, 4096), (2, 512), (100, 3000)):
        _check_sized(R, ctx, torch, oracle, FMT_R64, om, gm, data, n_ways, chunk)
        over, _, _ = _check_sized(R, ctx, torch, oracle, FMT_R64, om, gm, data, n_ways, chunk, slot=64)
        assert over.size >= (data.size + chunk - 1) // chunk - 1


def test_sized_slots_inside_a_hip_graph(gpu, oracle):
    """Both launches (the coders and the redo pass) captured into one hipGraph and replayed on other data."""
    R, ctx, torch = gpu
    chunk, n_ways = 8192, 64
    datas = []
    for seed in (1, 2, 3):
        d = oracle.gen_zipf(40 * chunk, K=256, s=1.0, seed=seed).copy()
        if seed != 1:  # the replays meet an incompressible chunk the captured run did not have
            d[seed * chunk:(seed + 1) * chunk] = np.random.default_rng(seed).integers(0, 256, chunk).astype(np.uint8)
        datas.append(d)
    f, _ = oracle.normalize(oracle.count_freqs(datas[0], 256), 4096)
    om, gm = oracle.model(f, 12), ctx.model(FMT_WORD, f, 12)
    slot = ctx.tight_slot_bytes(gm, n_ways, chunk)
    d_syms = _dev(torch, datas[0]).clone()
    cap = R.encode_sized_bound(FMT_WORD, d_syms.numel(), n_ways, chunk, slot, 4)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    offs = torch.zeros(41, dtype=torch.int64, device="cuda")
    lens = torch.zeros(40, dtype=torch.int32, device="cuda")
    ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, d_out=out, d_offsets=offs, d_lengths=lens)  # warm-up: allocations
    g = torch.cuda.CUDAGraph()
    stream = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=stream):
        ctx.encode_sized(gm, d_syms, n_ways, chunk, slot=slot, d_out=out, d_offsets=offs, d_lengths=lens, sync=False)
    for d in datas:
        d_syms.copy_(torch.from_numpy(d).cuda())
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ctx.encode_status()
        cont, o, l = oracle.encode_chunked(FMT_WORD, om, d, n_ways, chunk, align=16)
        go, gl, gc = offs.cpu().numpy(), lens.cpu().numpy(), out.cpu().numpy()
        assert np.array_equal(gl.astype(np.uint32), l)
        for c in range(40):
            assert np.array_equal(gc[int(go[c]):int(go[c]) + int(gl[c])], cont[int(o[c]):int(o[c]) + int(l[c])]), c
