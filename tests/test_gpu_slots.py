"""Slot layout (rans_amd_encode_slots), chunk offsets on any unit boundary, rans_amd_container_compact (-m gpu).

The reference hands its encoder the END of a buffer and finds the stream at [ptr after the flush, buffer end)
(rans_byte.h:22-26, main.cpp:176-188).  rans_amd_encode_slots does that once per chunk: chunk c's stream is the last
lengths[c] bytes of slot c -- written once, never moved.  Checked here, against the CPU oracle:

  * every chunk's bytes in its slot == the oracle's stream of that chunk, for every format and kernel family (wave
    encoders of 64..512 lanes, lane encoders of 1..8, the 2-way rans64 kernel, u16 symbols, the 4096-symbol alias
    model), offsets[c] == (c + 1) * slot - lengths[c];
  * the slot container decodes as it is (its chunk starts are not 16-byte aligned) with every decoder family;
  * rans_amd_container_compact of it == the oracle's compact container == rans_amd_encode's output, byte for byte;
  * decoders take hand-made indexes whose chunks start on every residue modulo 16 (reference streams packed
    back to back, in reverse order, with odd gaps).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD

UNIT = {FMT_BYTE: 1, FMT_ALIAS: 1, FMT_WORD: 2, FMT_R64: 4}


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


def _check_slots(R, ctx, torch, oracle, fmt, om, gm, data, n_ways, chunk, want_kernel=None):
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    nchunks = len(lens)
    d_syms = _dev(torch, data)
    slot = R.slot_bytes(fmt, data.size, n_ways, chunk)
    assert slot % 64 == 0 and slot >= R.chunk_bound(fmt, min(chunk, data.size), n_ways)
    g_cont, g_offs, g_lens, total = ctx.encode_slots(gm, d_syms, n_ways, chunk)
    assert total == nchunks * slot == R.encode_slots_bound(fmt, data.size, n_ways, chunk)
    assert ctx.last_encode_placement() == 2
    if want_kernel:
        assert ctx.last_encode_kernel()[0] == want_kernel, ctx.last_encode_kernel()
    go, gl = g_offs.cpu().numpy().astype(np.uint64), g_lens.cpu().numpy().astype(np.uint32)
    assert np.array_equal(gl, lens)
    assert np.array_equal(go[:-1], (np.arange(nchunks, dtype=np.uint64) + 1) * np.uint64(slot) - gl)
    assert int(go[-1]) == nchunks * slot
    g = g_cont.cpu().numpy()
    for c in range(nchunks):
        a, b = int(go[c]), int(go[c]) + int(gl[c])
        assert b == (c + 1) * slot
        assert np.array_equal(g[a:b], cont[int(offs[c]):int(offs[c]) + int(lens[c])]), "chunk %d differs from the oracle" % c
    # the slot container decodes as it is
    out = ctx.decode(gm, g_cont, total, g_offs, g_lens, data.size, n_ways, chunk)
    assert torch.equal(out, d_syms)
    # compaction == the oracle's container == rans_amd_encode's
    d_dst, d_doffs, ctotal = ctx.compact(g_cont, total, g_offs, g_lens, nchunks)
    assert ctotal == cont.size
    assert np.array_equal(d_doffs.cpu().numpy().astype(np.uint64), offs)
    c2 = d_dst.cpu().numpy()
    for c in range(nchunks):
        a, b = int(offs[c]), int(offs[c]) + int(lens[c])
        assert np.array_equal(c2[a:b], cont[a:b]), "compacted chunk %d differs" % c
    e_cont, e_offs, e_lens, etotal = ctx.encode(gm, d_syms, n_ways, chunk)
    assert etotal == ctotal and torch.equal(e_offs, d_doffs)
    return g_cont, g_offs, g_lens, total


@pytest.mark.parametrize("fmt,sb", [(FMT_WORD, 12), (FMT_BYTE, 14), (FMT_BYTE, 16), (FMT_R64, 14), (FMT_ALIAS, 16), (FMT_ALIAS, 12)])
@pytest.mark.parametrize("n_ways,chunk", [(64, 4096), (64, 5000), (256, 16384), (128, 4096), (512, 8192), (33, 1000),
                                          (2, 512), (1, 1000), (4, 2048), (8, 4096), (2, 4095)])
def test_slot_layout_every_chunk_matches_oracle(gpu, oracle, fmt, sb, n_ways, chunk):
    R, ctx, torch = gpu
    data = oracle.gen_zipf(300000, K=256, s=1.0, seed=1)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    _check_slots(R, ctx, torch, oracle, fmt, om, gm, data, n_ways, chunk)


def test_slot_layout_kernels_of_the_bench_configs(gpu, oracle):
    """The four shapes bench.py times, small: which coding kernel ran, and that no placement kernel did."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(64 * 16384 + 777, K=256, s=1.0, seed=1)
    om, gm = _models(ctx, oracle, FMT_WORD, 12, data)
    _check_slots(R, ctx, torch, oracle, FMT_WORD, om, gm, data, 64, 16384, "k_encode<word>")
    om, gm = _models(ctx, oracle, FMT_BYTE, 14, data)
    _check_slots(R, ctx, torch, oracle, FMT_BYTE, om, gm, data, 64, 16384, "k_encode<byte>")
    d2 = oracle.gen_zipf(512 * 64 * 300 + 100, K=256, s=1.0, seed=2)  # >= one batch of 64 chunks per CU + a ragged tail
    om, gm = _models(ctx, oracle, FMT_R64, 14, d2)
    _check_slots(R, ctx, torch, oracle, FMT_R64, om, gm, d2, 2, 512, "k_encode_lanes_r64x2")
    d4 = oracle.gen_zipf(24 * 16384 + 5, K=4096, s=1.0, seed=1)
    om, gm = _models(ctx, oracle, FMT_ALIAS, 16, d4, nsyms=4096)
    _check_slots(R, ctx, torch, oracle, FMT_ALIAS, om, gm, d4, 64, 16384, "k_encode<alias, LDS remap>")


def test_slot_layout_word_u16_symbols_and_lane_generations(gpu, oracle):
    R, ctx, torch = gpu
    d16 = oracle.gen_zipf(150001, K=1000, s=1.0, seed=5)
    f, _ = oracle.normalize(oracle.count_freqs(d16, 1024), 4096)
    om, gm = oracle.model(f, 12), ctx.model(FMT_WORD, f, 12)
    _check_slots(R, ctx, torch, oracle, FMT_WORD, om, gm, d16, 64, 8192)
    data = oracle.gen_zipf(200000, K=256, s=1.0, seed=9)
    for chunk in (512, 500, 64):  # (the per-lane encoder for few batches / chunk sizes off 16, the staged one for many)
        for fmt, sb in ((FMT_BYTE, 14), (FMT_WORD, 12), (FMT_R64, 14)):
            om, gm = _models(ctx, oracle, fmt, sb, data)
            _check_slots(R, ctx, torch, oracle, fmt, om, gm, data, 2, chunk)


def test_slot_layout_capacity_is_checked_up_front(gpu, oracle):
    R, ctx, torch = gpu
    data = oracle.gen_zipf(100000, K=256, s=1.0, seed=1)
    om, gm = _models(ctx, oracle, FMT_WORD, 12, data)
    need = R.encode_slots_bound(FMT_WORD, data.size, 64, 4096)
    small = torch.empty(need - 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_slots(gm, _dev(torch, data), 64, 4096, d_out=small)
    assert e.value.status == R.E_SPACE
    # an empty input: no chunks, offsets[0] = 0
    empty = torch.empty(0, dtype=torch.uint8, device="cuda")
    _, offs, _, total = ctx.encode_slots(gm, empty, 64, 4096)
    assert total == 0 and int(offs[0]) == 0


@pytest.mark.parametrize("fmt,sb,n_ways,chunk", [(FMT_WORD, 12, 64, 4096), (FMT_WORD, 12, 128, 4096), (FMT_BYTE, 14, 64, 4096),
                                                 (FMT_BYTE, 14, 33, 1000), (FMT_R64, 14, 64, 4096), (FMT_ALIAS, 16, 64, 4096),
                                                 (FMT_R64, 14, 2, 512), (FMT_WORD, 12, 2, 512), (FMT_BYTE, 14, 4, 1024),
                                                 (FMT_R64, 14, 1, 1000), (FMT_ALIAS, 12, 8, 2048)])
def test_decoders_take_chunks_on_every_unit_boundary(gpu, oracle, fmt, sb, n_ways, chunk):
    """Reference streams packed back to back (no padding), in reverse order, and with a sliding gap: chunk starts hit
    every residue modulo 16 the format's unit allows.  Every decoder family (option sweeps as well)."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(300000, K=256, s=1.0, seed=4)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    nchunks, unit = len(lens), UNIT[fmt]
    streams = [cont[int(offs[c]):int(offs[c]) + int(lens[c])] for c in range(nchunks)]
    layouts = []
    # 1. back to back
    o = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))])[:-1]
    layouts.append(("packed", o))
    # 2. reverse order, packed
    r = np.zeros(nchunks, dtype=np.uint64)
    at = 0
    for c in range(nchunks - 1, -1, -1):
        r[c] = at
        at += int(lens[c])
    layouts.append(("reversed", r))
    # 3. a gap of (c mod 16) units in front of chunk c
    g = np.zeros(nchunks, dtype=np.uint64)
    at = 0
    for c in range(nchunks):
        at += unit * (c % 16)
        g[c] = at
        at += int(lens[c])
    layouts.append(("gaps", g))
    seen = set()
    for name, lo in layouts:
        size = int(max(int(lo[c]) + int(lens[c]) for c in range(nchunks)))
        buf = np.full(size + 64, 0xA5, dtype=np.uint8)
        for c in range(nchunks):
            buf[int(lo[c]):int(lo[c]) + int(lens[c])] = streams[c]
        seen |= {int(v) % 16 for v in lo}
        d_cont = torch.from_numpy(buf).cuda()
        d_offs = torch.from_numpy(lo.astype(np.int64)).cuda()
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        sweeps = [(None, None)]
        if n_ways <= 8:
            sweeps = [(None, None)]
        elif fmt == FMT_ALIAS:
            sweeps = [(R.OPT_DUAL_DECODE, v) for v in (1, 0, 2)]
        for opt, val in sweeps:
            if opt is not None:
                ctx.set_option(opt, val)
            try:
                out = ctx.decode(gm, d_cont, size, d_offs, d_lens, data.size, n_ways, chunk)
                assert np.array_equal(out.cpu().numpy(), data), (name, opt, val, ctx.last_decode_kernel())
            finally:
                if opt is not None:
                    ctx.set_option(opt, 1 if opt == R.OPT_DUAL_DECODE else 0)
        # an offset off the unit grid is refused chunk by chunk, never read
        if unit > 1:
            bad = lo.copy()
            bad[1] += 1
            with pytest.raises(R.RansAmdError) as e:
                ctx.decode(gm, d_cont, size, torch.from_numpy(bad.astype(np.int64)).cuda(), d_lens, data.size, n_ways, chunk)
            assert e.value.status == R.E_CORRUPT
    assert len(seen) >= 16 // unit - 1, seen


def test_compact_a_chunk_range_and_a_reordered_index(gpu, oracle):
    """rans_amd_container_compact takes any source index: a sub-range of a container, chunks in another order."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(200000, K=256, s=1.0, seed=6)
    om, gm = _models(ctx, oracle, FMT_WORD, 12, data)
    chunk, n_ways = 4096, 64
    cont, offs, lens = oracle.encode_chunked(FMT_WORD, om, data, n_ways, chunk, align=16)
    nchunks = len(lens)
    d_cont = torch.from_numpy(np.concatenate([cont, np.zeros(16, np.uint8)])).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
    lo, hi = 5, nchunks - 3
    d_dst, d_doffs, total = ctx.compact(d_cont, cont.size, d_offs[lo:], d_lens[lo:], hi - lo)
    want_offs = R.offsets_from_lengths(lens[lo:hi])
    assert np.array_equal(d_doffs.cpu().numpy().astype(np.uint64), want_offs) and total == int(want_offs[-1])
    got = d_dst.cpu().numpy()
    for c in range(lo, hi):
        a = int(want_offs[c - lo])
        assert np.array_equal(got[a:a + int(lens[c])], cont[int(offs[c]):int(offs[c]) + int(lens[c])])
    out = ctx.decode(gm, d_dst, total, d_doffs, d_lens[lo:], (hi - lo) * chunk, n_ways, chunk)
    assert np.array_equal(out.cpu().numpy(), data[lo * chunk:hi * chunk])
    # too small a destination is reported, nothing is copied
    tiny = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    with pytest.raises(R.RansAmdError) as e:
        ctx.compact(d_cont, cont.size, d_offs, d_lens, nchunks, d_dst=tiny)
    assert e.value.status == R.E_SPACE and int(tiny.sum()) == 0
    # ... and that verdict was REPORTED with the synchronous return: an encode that follows and its status are clean
    g_cont, g_offs, g_lens, g_total = ctx.encode(gm, torch.from_numpy(data).cuda(), n_ways, chunk)
    ctx.encode_status()
    assert g_total == cont.size
    # an ASYNCHRONOUS compaction that fails, an asynchronous encode queued behind it, then the status call: the compaction's
    # verdict must still be there (every encode used to wipe it) -- and it is reported once
    dd = torch.zeros(nchunks + 1, dtype=torch.int64, device="cuda")
    ctx.compact(d_cont, cont.size, d_offs, d_lens, nchunks, d_dst=tiny, sync=False, d_dst_offsets=dd)
    ctx.encode(gm, torch.from_numpy(data).cuda(), n_ways, chunk, sync=False)
    with pytest.raises(R.RansAmdError) as e:
        ctx.encode_status()
    assert e.value.status == R.E_SPACE
    ctx.encode_status()


@pytest.mark.parametrize("fmt,sb,n_ways,chunk,kernel", [(FMT_R64, 14, 2, 512, "k_decode_lanes_r64x2"), (FMT_WORD, 12, 4, 1024, "k_decode_lanes_staged"),
                                                         (FMT_BYTE, 14, 8, 2000, "k_decode_lanes_staged")])
def test_lane_decoders_take_chunks_more_than_1_gib_apart(gpu, oracle, fmt, sb, n_ways, chunk, kernel):
    """The lane decoders address a batch of 64 chunks through 32-bit offsets from its lowest chunk; chunks 1 GiB or more
    above it (an overflowed chunk of a large sized-slot container, a scattered index) are decoded in another trip over the
    same batch -- not counted as corrupt.  A 2.5 GiB buffer, chunks of different batches moved 1.1 and
    2.25 GiB up, one batch with two far chunks more than 1 GiB apart from each other as well."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf(200 * chunk + 17, K=256, s=1.0, seed=14)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    nchunks = len(lens)
    big = torch.zeros((5 << 29) + 4096, dtype=torch.uint8, device="cuda")
    big[:cont.size] = torch.from_numpy(cont).cuda()
    o2 = offs.astype(np.int64).copy()
    for c, where in ((3, (9 << 27) + 64), (70, (18 << 27) + 16 * 7), (71, (9 << 27) + 8192), (nchunks - 1, (5 << 29) - 256)):
        a, ln = int(offs[c]), int(lens[c])
        big[where:where + ln] = torch.from_numpy(cont[a:a + ln].copy()).cuda()
        big[a:a + ln] = 0xA5  # (the old place holds garbage now)
        o2[c] = where
    d_offs, d_lens = torch.from_numpy(o2).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()
    out = ctx.decode(gm, big, big.numel(), d_offs, d_lens, data.size, n_ways, chunk)
    assert ctx.last_decode_kernel().startswith(kernel), ctx.last_decode_kernel()
    assert np.array_equal(out.cpu().numpy(), data)
    # a malformed entry among them is still one bad chunk, counted once
    o3 = o2.copy()
    o3[70] = big.numel() + 64
    with pytest.raises(R.RansAmdError) as e:
        ctx.decode(gm, big, big.numel(), torch.from_numpy(o3).cuda(), d_lens, data.size, n_ways, chunk)
    assert e.value.status == R.E_CORRUPT
    del big


@pytest.mark.parametrize("n_ways,chunk", [(64, 4096), (2, 512)])  # k_compact (a wave per chunk) and k_compact_small (16 lanes)
def test_compact_rejects_an_index_that_leaves_the_source(gpu, oracle, n_ways, chunk):
    """The source index of rans_amd_container_compact is data (it may come from a file): an (offset, length) pair outside
    [0, src_bytes) is reported as RANS_AMD_E_CORRUPT, that chunk is neither read nor written, the others are copied."""
    R, ctx, torch = gpu
    fmt, sb = (FMT_WORD, 12) if n_ways == 64 else (FMT_R64, 14)
    data = oracle.gen_zipf(150000 if n_ways == 64 else 4200 * 512, K=256, s=1.0, seed=12)  # (>= 4096 chunks: k_compact_small)
    om, gm = _models(ctx, oracle, fmt, sb, data)
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    nchunks = len(lens)
    d_cont = torch.from_numpy(cont.copy()).cuda()  # exactly src_bytes long: nothing behind it belongs to the caller
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
    want_offs = R.offsets_from_lengths(lens)
    for victim, off, ln in ((3, int(cont.size) - 8, None), (nchunks - 1, 1 << 40, None), (0, int(cont.size) + 1, None),
                            (7, None, int(cont.size))):
        o2, l2 = offs.astype(np.int64).copy(), lens.astype(np.int32).copy()
        if off is not None:
            o2[victim] = off
        if ln is not None:
            l2[victim] = ln
        w2 = R.offsets_from_lengths(l2.astype(np.uint32))
        dst = torch.full((int(w2[-1]) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        with pytest.raises(R.RansAmdError) as e:
            ctx.compact(d_cont, cont.size, torch.from_numpy(o2).cuda(), torch.from_numpy(l2).cuda(), nchunks, d_dst=dst)
        assert e.value.status == R.E_CORRUPT, (victim, off, ln)
        got = dst.cpu().numpy()
        for c in range(nchunks):
            a, b = int(w2[c]), int(w2[c]) + int(l2[c])
            if c == victim:
                assert np.all(got[a:b] == 0xEE), "the rejected chunk was written"
            else:
                assert np.array_equal(got[a:b], cont[int(offs[c]):int(offs[c]) + int(lens[c])]), c
    # a healthy index right behind: the verdict is per call
    d_dst, d_doffs, total = ctx.compact(d_cont, cont.size, torch.from_numpy(offs.astype(np.int64)).cuda(), d_lens, nchunks)
    assert total == int(want_offs[-1])


@pytest.mark.parametrize("fmt,sb,n_ways,chunk", [(FMT_WORD, 12, 64, 4096), (FMT_R64, 14, 2, 512), (FMT_ALIAS, 16, 64, 4096)])
@pytest.mark.parametrize("world", [2, 3])
def test_one_container_split_over_contexts(gpu, oracle, fmt, sb, n_ways, chunk, world):
    """SURVEY 8(e)'s general rule on real data: ONE oracle-made container, shard g = chunks sharding.chunk_range(C, G, g);
    every "rank" (its own context, its own model) holds ONLY the bytes rans_amd_container_slice assigns it, decodes its
    chunk range through the plain ABI call, and the pieces laid side by side are the input.  Also the variant without a
    slice: the whole container resident, d_offsets + lo / d_lengths + lo."""
    R, ctx0, torch = gpu
    from ryg_rans_amd.sharding import chunk_range, symbol_range
    data = oracle.gen_zipf(300000 + 123, K=256, s=1.0, seed=8)
    f, _ = oracle.normalize(oracle.count_freqs(data, 256), 1 << sb)
    om = oracle.model(f, sb, with_alias=(fmt == FMT_ALIAS))
    cont, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    nchunks, n = len(lens), data.size
    d_full = torch.from_numpy(np.concatenate([cont, np.zeros(16, np.uint8)])).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
    out_sliced = torch.zeros(n, dtype=torch.uint8, device="cuda")
    out_ranged = torch.zeros(n, dtype=torch.uint8, device="cuda")
    covered = 0
    for g in range(world):
        ctx = R.Context(0)  # one context per rank (here: all on the one GPU of the box)
        gm = ctx.model(fmt, f, sb)
        lo, hi = chunk_range(nchunks, world, g)
        first, last = symbol_range(n, chunk, world, g)
        assert first == lo * chunk and last == min(n, hi * chunk)
        b, e, rebased = R.container_slice(offs, lens, lo, hi)
        assert b % 16 == 0 and b <= int(offs[lo]) and e == int(offs[hi - 1]) + int(lens[hi - 1])
        assert np.array_equal(rebased[:-1], offs[lo:hi] - np.uint64(b)) and int(rebased[-1]) == e - b
        part = torch.from_numpy(np.concatenate([cont[b:e], np.zeros(16, np.uint8)])).cuda()  # ONLY this rank's bytes
        d_reb = torch.from_numpy(rebased.astype(np.int64)).cuda()
        ctx.decode(gm, part, e - b, d_reb, d_lens[lo:hi], last - first, n_ways, chunk, d_out=out_sliced[first:last])
        ctx.decode(gm, d_full, cont.size, d_offs[lo:], d_lens[lo:], last - first, n_ways, chunk, d_out=out_ranged[first:last])
        covered += last - first
        ctx.close()
    assert covered == n
    assert np.array_equal(out_sliced.cpu().numpy(), data) and np.array_equal(out_ranged.cpu().numpy(), data)


def test_slot_encoder_and_compaction_inside_a_hip_graph(gpu, oracle):
    """rans_amd_encode_slots, rans_amd_container_compact and rans_amd_decode captured into one hipGraph (the rules of
    rans_amd_encode apply: one eager call first, no host result pointer): every replay leaves the oracle's chunks in
    their slots, the oracle's compact container behind the compaction, and the symbols in the output."""
    R, ctx, torch = gpu
    data = oracle.gen_zipf((1 << 19) + 333, K=256, s=1.0, seed=21)
    n = data.size
    d = torch.from_numpy(data).cuda()
    for fmt, sb, ways, chunk in ((FMT_WORD, 12, 64, 4096), (FMT_R64, 14, 2, 512), (FMT_BYTE, 12, 128, 8192)):
        om, gm = _models(ctx, oracle, fmt, sb, data)
        want, w_offs, w_lens = oracle.encode_chunked(fmt, om, data, ways, chunk, align=16)
        nchunks = len(w_lens)
        s_cont, s_offs, s_lens, s_total = ctx.encode_slots(gm, d, ways, chunk)                 # eager once: workspaces exist
        c_cont, c_offs, c_total = ctx.compact(s_cont, s_total, s_offs, s_lens, nchunks)
        out = ctx.decode(gm, s_cont, s_total, s_offs, s_lens, n, ways, chunk)
        assert c_total == want.size and torch.equal(out, d)
        s2, so2, sl2 = torch.zeros_like(s_cont), torch.zeros_like(s_offs), torch.zeros_like(s_lens)
        c2, co2, out2 = torch.zeros_like(c_cont), torch.zeros_like(c_offs), torch.zeros_like(out)
        stream = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            ctx.encode_slots(gm, d, ways, chunk, d_out=s2, sync=False, d_offsets=so2, d_lengths=sl2)
            ctx.compact(s2, s_total, so2, sl2, nchunks, d_dst=c2, sync=False, d_dst_offsets=co2)
            ctx.decode(gm, s2, s_total, so2, sl2, n, ways, chunk, d_out=out2, sync=False)
        assert not bool(out2.any())
        for rep in range(2):
            s2.zero_(); so2.zero_(); sl2.zero_(); c2.zero_(); out2.zero_()
            g.replay()
            torch.cuda.synchronize()
            ctx.encode_status()
            assert ctx.decode_errors() == 0 and torch.equal(out2, d), (fmt, rep)
            assert torch.equal(sl2, s_lens) and torch.equal(so2, s_offs)
            assert np.array_equal(co2.cpu().numpy().astype(np.uint64), w_offs)
            got = c2.cpu().numpy()
            for c in range(nchunks):
                a = int(w_offs[c])
                assert np.array_equal(got[a:a + int(w_lens[c])], want[a:a + int(w_lens[c])]), (fmt, rep, c)
        del g


def test_placement_watchdog_scales_with_the_chunk():
    """every wait of the fused placement gives up after a wall-clock limit; one huge chunk of a low-way stream
    keeps its coding wave busy for seconds, and the copier that waits for it must not call that a protocol failure.  The
    limit is half a minute PLUS a microsecond per symbol and lane of the call's largest chunk.  Shown with the measure
    build (its knobs shrink the base to 20 ms): a 1-way chunk of 4 Mi symbols (~0.5 s of coding) passes with the per-chunk
    share and fails with 'placement protocol timed out' without it -- the share, not the base, lets it through."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "ryg_rans_amd", "lib", "libryg_rans_amd_measure.so")
    if not os.path.exists(lib):
        pytest.skip("measure build not present (make -C ryg_rans_amd/csrc measure)")
    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import ryg_rans_amd as R
ctx = R.Context(0)
f = np.zeros(256, np.uint32); f[:4] = [1024, 1024, 1024, 1024]
m = ctx.model(R.FMT_WORD, f, 12)
d = torch.randint(0, 4, (1 << 22,), dtype=torch.uint8, device="cuda")
try:
    cont, offs, lens, total = ctx.encode(m, d, 1, 1 << 22)          # ONE chunk, one lane: the fused wave encoder
    out = ctx.decode(m, cont, total, offs, lens, d.numel(), 1, 1 << 22)
    print("RESULT ok" if torch.equal(out, d) and ctx.last_encode_kernel()[1] else "RESULT mismatch")
except R.RansAmdError as e:
    print("RESULT status %%d %%s" %% (e.status, e))
""" % (root,)
    env = dict(os.environ, RANS_AMD_LIB=lib, RANS_AMD_WATCHDOG_BASE_MS="20")
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
    assert "RESULT ok" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
    env["RANS_AMD_WATCHDOG_NO_SCALE"] = "1"
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
    assert "RESULT status 6" in out.stdout and "timed out" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
