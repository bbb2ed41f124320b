"""Serialised container (rans_amd_container_pack / _parse): CPU tests on oracle-made payloads,
GPU test for the full encode -> file -> decode path."""
import numpy as np
import pytest

import ryg_rans_amd as R
from _oracle import FMT_BYTE, FMT_R64, FMT_WORD


def _oracle_container(oracle, fmt, sb, data, n_ways, chunk):
    f, _ = oracle.normalize(oracle.count_freqs(data, 256), 1 << sb)
    om = oracle.model(f, sb)
    payload, offs, lens = oracle.encode_chunked(fmt, om, data, n_ways, chunk, align=16)
    return f, om, payload, offs, lens


@pytest.mark.parametrize("fmt,sb", [(FMT_WORD, 12), (FMT_BYTE, 14), (FMT_R64, 14)])
def test_pack_parse_roundtrip(oracle, fmt, sb):
    data = oracle.gen_zipf(50001, K=256, s=1.0, seed=6)
    f, om, payload, offs, lens = _oracle_container(oracle, fmt, sb, data, 64, 4096)
    blob = R.pack_container(fmt, f, sb, data.size, 64, 4096, lens, payload)
    assert blob[:8].tobytes() == b"RANSAMD1" and blob.size % 1 == 0
    info, f2, l2, p2 = R.parse_container(blob)
    assert (info.format, info.scale_bits, info.nsyms, info.n_ways, info.chunk_syms, info.sym_bytes) == \
        (fmt, sb, 256, 64, 4096, 1)
    assert info.n_symbols == data.size and info.n_chunks == lens.size and info.payload_bytes == payload.size
    assert np.array_equal(f2, f) and np.array_equal(l2, lens) and np.array_equal(p2, payload)
    assert np.array_equal(R.offsets_from_lengths(l2), offs)
    # what was stored is decodable by the oracle chunk by chunk: plain reference streams inside
    out = oracle.decode_chunked(fmt, om, p2, R.offsets_from_lengths(l2), l2, data.size, 64, 4096)
    assert np.array_equal(out, data)


def test_parse_rejects_damage(oracle):
    data = oracle.gen_zipf(20000, K=256, s=1.0, seed=7)
    f, om, payload, offs, lens = _oracle_container(oracle, FMT_WORD, 12, data, 64, 4096)
    blob = R.pack_container(FMT_WORD, f, 12, data.size, 64, 4096, lens, payload)
    for pos in (0, 9, 20, 90, 80 + 4 * 256 + 2):  # magic, version, format, a frequency, a length
        bad = blob.copy()
        bad[pos] ^= 0x5a
        with pytest.raises(R.RansAmdError) as e:
            R.parse_container(bad)
        assert e.value.status == R.E_CORRUPT, pos
    with pytest.raises(R.RansAmdError):
        R.parse_container(blob[:-1])   # truncated payload
    with pytest.raises(R.RansAmdError):
        R.parse_container(blob[:40])   # truncated header
    with pytest.raises(R.RansAmdError):
        R.pack_container(FMT_WORD, f, 12, data.size, 64, 4096, lens[:-1], payload)  # index does not match
    with pytest.raises(R.RansAmdError):
        g = f.copy(); g[0] += 1
        R.pack_container(FMT_WORD, g, 12, data.size, 64, 4096, lens, payload)       # model does not sum to M


def test_empty_container():
    f = np.zeros(256, np.uint32); f[0] = 4000; f[1] = 96
    blob = R.pack_container(FMT_WORD, f, 12, 0, 64, 4096, np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    info, f2, l2, p2 = R.parse_container(blob)
    assert info.n_chunks == 0 and info.payload_bytes == 0 and l2.size == 0 and np.array_equal(f2, f)


@pytest.mark.gpu
def test_gpu_encode_file_decode(tmp_path, oracle):
    import torch
    ctx = R.Context(0)
    data = oracle.gen_zipf(1 << 20, K=256, s=1.0, seed=8)
    d_syms = torch.from_numpy(data).cuda()
    freqs, _ = R.normalize_freqs(ctx.count_freqs_device(d_syms, 256), 4096)
    m = ctx.model(FMT_WORD, freqs, 12)
    cont, offs, lens, total = ctx.encode(m, d_syms, 64, 32768)
    blob = R.pack_container(FMT_WORD, freqs, 12, data.size, 64, 32768, lens.cpu().numpy().astype(np.uint32),
                            cont[:total].cpu().numpy())
    path = tmp_path / "zipf.rans"
    blob.tofile(path)

    # a fresh reader: nothing but the file
    info, f2, l2, p2 = R.parse_container(np.fromfile(path, dtype=np.uint8))
    m2 = ctx.model(info.format, f2, info.scale_bits)
    d_cont = torch.from_numpy(np.concatenate([p2, np.zeros(64, np.uint8)])).cuda()
    d_offs = torch.from_numpy(R.offsets_from_lengths(l2).astype(np.int64)).cuda()
    d_lens = torch.from_numpy(l2.astype(np.int32)).cuda()
    out = ctx.decode(m2, d_cont, info.payload_bytes, d_offs, d_lens, info.n_symbols, info.n_ways, info.chunk_syms)
    assert np.array_equal(out.cpu().numpy(), data)
    # and the oracle agrees that the file holds reference-format streams
    om = oracle.model(f2, 12)
    assert np.array_equal(oracle.decode_chunked(FMT_WORD, om, p2, R.offsets_from_lengths(l2), l2, data.size, 64, 32768),
                          data)


def _oracle_adaptive(oracle, data, sb, n_ways, chunk, fmt=FMT_BYTE):
    """Per-chunk models on the CPU: each chunk counted, normalised and encoded with ITS OWN table (byte or word format)."""
    rows, parts, lens = [], [], []
    for lo in range(0, data.size, chunk):
        piece = data[lo:lo + chunk]
        f, _ = oracle.normalize(oracle.count_freqs(piece, 256), 1 << sb)
        rows.append(f.astype(np.uint16))
        stream = oracle.encode(fmt, oracle.model(f, sb), piece, n_ways)
        lens.append(stream.size)
        pad = (-stream.size) % 16 if lo + chunk < data.size else 0
        parts.append(np.concatenate([stream, np.zeros(pad, np.uint8)]))
    payload = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return np.stack(rows) if rows else np.zeros((0, 256), np.uint16), np.array(lens, np.uint32), payload


@pytest.mark.parametrize("fmt", [FMT_BYTE, FMT_WORD])
def test_adaptive_container_roundtrip(oracle, fmt):
    rng = np.random.default_rng(3)
    # two regimes so the per-chunk models differ
    data = np.concatenate([oracle.gen_zipf(9000, K=64, s=1.2, seed=1), rng.integers(100, 256, 7001).astype(np.uint8)])
    rows, lens, payload = _oracle_adaptive(oracle, data, 12, 32, 4096, fmt)
    blob = R.pack_container_adaptive(12, data.size, 32, 4096, rows, lens, payload, fmt=fmt)
    info, f2, l2, p2 = R.parse_container_adaptive(blob)
    assert (info.format, info.scale_bits, info.nsyms, info.n_ways, info.chunk_syms, info.sym_bytes) == \
        (fmt, 12, 256, 32, 4096, 1)  # the header carries the stream format
    assert info.n_symbols == data.size and info.n_chunks == 4
    assert np.array_equal(f2, rows) and np.array_equal(l2, lens) and np.array_equal(p2, payload)
    assert not np.array_equal(f2[0], f2[3])
    # each chunk decodes with its own stored row
    offs = R.offsets_from_lengths(l2)
    for c in range(info.n_chunks):
        n_c = min(4096, data.size - c * 4096)
        got = oracle.decode(fmt, oracle.model(f2[c].astype(np.uint32), 12), p2[offs[c]:offs[c] + l2[c]], n_c, 32)
        assert np.array_equal(got, data[c * 4096:c * 4096 + n_c])
    if fmt == FMT_WORD:  # the word format's probabilities are 12 bits, per chunk as for a whole input
        with pytest.raises(R.RansAmdError):
            R.pack_container_adaptive(11, data.size, 32, 4096, rows, lens, payload, fmt=FMT_WORD)
        return
    # the two container versions do not parse as each other
    with pytest.raises(R.RansAmdError) as e:
        R.parse_container(blob)
    assert e.value.status == R.E_CORRUPT
    f, _ = oracle.normalize(oracle.count_freqs(data, 256), 4096)
    v1 = R.pack_container(FMT_BYTE, f, 12, data.size, 32, 4096, lens, payload)
    with pytest.raises(R.RansAmdError):
        R.parse_container_adaptive(v1)


def test_adaptive_container_rejects_damage(oracle):
    data = oracle.gen_zipf(10000, K=256, s=1.0, seed=2)
    rows, lens, payload = _oracle_adaptive(oracle, data, 11, 64, 4096)
    blob = R.pack_container_adaptive(11, data.size, 64, 4096, rows, lens, payload)
    for pos in (3, 8, 16, 80 + 5, 80 + 512 * 3 + 1):  # magic, version, format, a frequency, a length
        bad = blob.copy()
        bad[pos] ^= 0x21
        with pytest.raises(R.RansAmdError) as e:
            R.parse_container_adaptive(bad)
        assert e.value.status == R.E_CORRUPT, pos
    with pytest.raises(R.RansAmdError):
        R.parse_container_adaptive(blob[:-3])
    worse = rows.copy(); worse[1, 0] += 1
    with pytest.raises(R.RansAmdError) as e:
        R.pack_container_adaptive(11, data.size, 64, 4096, worse, lens, payload)   # a row that is not a model
    assert e.value.status == R.E_MODEL
    with pytest.raises(R.RansAmdError):
        R.pack_container_adaptive(14, data.size, 64, 4096, rows, lens, payload)    # u16 rows stop at 12 bits here
    blob0 = R.pack_container_adaptive(12, 0, 64, 4096, np.zeros((0, 256), np.uint16), np.zeros(0, np.uint32),
                                      np.zeros(0, np.uint8))
    info, f0, l0, p0 = R.parse_container_adaptive(blob0)
    assert info.n_chunks == 0 and f0.shape == (0, 256) and p0.size == 0


@pytest.mark.gpu
def test_gpu_adaptive_file_roundtrip(tmp_path, oracle):
    """encode_adaptive on the GPU -> version-2 file -> a fresh reader decodes on the GPU, and the oracle decodes
    sampled chunks of the file with the stored rows (reference byte-format streams inside)."""
    import torch
    ctx = R.Context(0)
    rng = np.random.default_rng(11)
    data = np.concatenate([oracle.gen_zipf(1 << 19, K=256, s=1.0, seed=8),
                           rng.integers(0, 40, (1 << 19) + 12345).astype(np.uint8)])
    d = torch.from_numpy(data).cuda()
    _gpu_adaptive_file_roundtrip(ctx, torch, tmp_path, oracle, data, d, FMT_BYTE)
    _gpu_adaptive_file_roundtrip(ctx, torch, tmp_path, oracle, data, d, FMT_WORD)


def _gpu_adaptive_file_roundtrip(ctx, torch, tmp_path, oracle, data, d, fmt):
    cont, offs, lens, freqs, total = ctx.encode_adaptive(d, 64, 32768, 12, fmt=fmt)
    nch = R.num_chunks(data.size, 32768)
    blob = R.pack_container_adaptive(12, data.size, 64, 32768, freqs.cpu().numpy().view(np.uint16)[:nch * 256],
                                     lens.cpu().numpy().astype(np.uint32)[:nch], cont[:total].cpu().numpy(), fmt=fmt)
    path = tmp_path / "mixed.rans2"
    blob.tofile(path)

    info, f2, l2, p2 = R.parse_container_adaptive(np.fromfile(path, dtype=np.uint8))
    o2 = R.offsets_from_lengths(l2)
    d_cont = torch.from_numpy(np.concatenate([p2, np.zeros(64, np.uint8)])).cuda()
    d_offs = torch.from_numpy(o2.astype(np.int64)).cuda()
    d_lens = torch.from_numpy(l2.astype(np.int32)).cuda()
    d_f = torch.from_numpy(np.ascontiguousarray(f2).view(np.int16).reshape(-1)).cuda()
    assert info.format == fmt
    out = ctx.decode_adaptive(d_cont, info.payload_bytes, d_offs, d_lens, d_f, info.n_symbols, info.n_ways,
                              info.chunk_syms, info.scale_bits, fmt=info.format)  # (the file says which coder)
    assert np.array_equal(out.cpu().numpy(), data)
    for c in (0, nch // 2, nch - 1):
        n_c = min(32768, data.size - c * 32768)
        got = oracle.decode(fmt, oracle.model(f2[c].astype(np.uint32), 12), p2[o2[c]:o2[c] + l2[c]], n_c, 64)
        assert np.array_equal(got, data[c * 32768:c * 32768 + n_c]), c


@pytest.mark.parametrize("fmt,sb", [(FMT_WORD, 12), (FMT_BYTE, 14), (FMT_R64, 14)])
def test_pack_indexed_from_any_layout(oracle, fmt, sb):
    """rans_amd_container_pack_indexed: the file written from a container in ANY layout equals, byte for byte, the file
    rans_amd_container_pack writes from the compact one -- slot layout (a stream at the END of its slot, as the reference
    leaves it in its buffer, main.cpp:176-188), scattered chunks in descending order, a sized layout with two chunks in an
    overflow region.  Index entries outside the source are refused, nothing outside is read."""
    data = oracle.gen_zipf(50001, K=256, s=1.0, seed=6)
    f, om, payload, offs, lens = _oracle_container(oracle, fmt, sb, data, 64, 4096)
    want = R.pack_container(fmt, f, sb, data.size, 64, 4096, lens, payload)
    n = lens.size
    rng = np.random.default_rng(9)
    slot = 2 * 4096 + 512
    layouts = []
    src = rng.integers(0, 256, n * slot, dtype=np.uint8)  # (garbage between the streams: none of it may reach the file)
    o = np.zeros(n + 1, dtype=np.uint64)
    for c in range(n):  # slot layout
        o[c] = (c + 1) * slot - int(lens[c])
        src[int(o[c]):int(o[c]) + int(lens[c])] = payload[int(offs[c]):int(offs[c]) + int(lens[c])]
    o[n] = n * slot
    layouts.append((src, o))
    src = rng.integers(0, 256, n * slot + 99, dtype=np.uint8)
    o = np.zeros(n + 1, dtype=np.uint64)
    at = 7  # scattered, descending, unaligned starts
    for c in reversed(range(n)):
        o[c] = at
        src[at:at + int(lens[c])] = payload[int(offs[c]):int(offs[c]) + int(lens[c])]
        at += int(lens[c]) + int(rng.integers(0, 300))
    o[n] = at
    layouts.append((src, o))
    tight = (int(lens.max()) + 63) // 64 * 64 - 128  # sized slots: the longest chunks overflow behind them
    src = rng.integers(0, 256, n * tight + n * slot, dtype=np.uint8)
    o = np.zeros(n + 1, dtype=np.uint64)
    over = 0
    for c in range(n):
        if int(lens[c]) <= tight:
            o[c] = (c + 1) * tight - int(lens[c])
        else:
            over += 1
            o[c] = n * tight + over * slot - int(lens[c])
        src[int(o[c]):int(o[c]) + int(lens[c])] = payload[int(offs[c]):int(offs[c]) + int(lens[c])]
    assert over >= 1
    o[n] = n * tight + over * slot
    layouts.append((src, o))
    for src, o in layouts:
        got = R.pack_container_indexed(fmt, f, sb, data.size, 64, 4096, o, lens, src)
        assert np.array_equal(got, want)
        info, f2, l2, p2 = R.parse_container(got)
        assert np.array_equal(oracle.decode_chunked(fmt, om, p2, R.offsets_from_lengths(l2), l2, data.size, 64, 4096), data)
    # the index is data: an entry that leaves the source is refused
    src, o = layouts[0]
    for c, delta in ((0, src.size), (n - 1, 1), (3, 1 << 62)):
        bad = o.copy()
        bad[c] += np.uint64(delta) if c != n - 1 else np.uint64(src.size - int(o[c]) - int(lens[c]) + 1)
        with pytest.raises(R.RansAmdError) as e:
            R.pack_container_indexed(fmt, f, sb, data.size, 64, 4096, bad, lens, src)
        assert e.value.status == R.E_CORRUPT, c
    with pytest.raises(R.RansAmdError):
        g = f.copy(); g[0] += 1
        R.pack_container_indexed(fmt, g, sb, data.size, 64, 4096, o, lens, src)  # model does not sum to M
    # an empty input
    blob = R.pack_container_indexed(fmt, f, sb, 0, 64, 4096, np.zeros(1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    assert R.parse_container(blob)[0].n_chunks == 0


@pytest.mark.parametrize("fmt", [FMT_BYTE, FMT_WORD])
def test_pack_indexed_adaptive(oracle, fmt):
    """... and the version-2 file (one model per chunk) from the pieces rans_amd_encode_adaptive_sized leaves: every stream
    the end of a piece of whole 64-byte lines."""
    rng = np.random.default_rng(3)
    data = np.concatenate([oracle.gen_zipf(9000, K=64, s=1.2, seed=1), rng.integers(100, 256, 7001).astype(np.uint8)])
    rows, lens, payload = _oracle_adaptive(oracle, data, 12, 32, 4096, fmt)
    want = R.pack_container_adaptive(12, data.size, 32, 4096, rows, lens, payload, fmt=fmt)
    offs = R.offsets_from_lengths(lens)
    pieces = (lens.astype(np.int64) + 200 + 63) // 64 * 64
    ends = np.cumsum(pieces)
    src = rng.integers(0, 256, int(ends[-1]), dtype=np.uint8)
    o = np.zeros(lens.size + 1, dtype=np.uint64)
    for c in range(lens.size):
        o[c] = int(ends[c]) - int(lens[c])
        src[int(o[c]):int(ends[c])] = payload[int(offs[c]):int(offs[c]) + int(lens[c])]
    o[lens.size] = ends[-1]
    got = R.pack_container_indexed(fmt, None, 12, data.size, 32, 4096, o, lens, src, chunk_freqs=rows)
    assert np.array_equal(got, want)
    bad = o.copy()
    bad[1] = np.uint64(src.size)
    with pytest.raises(R.RansAmdError) as e:
        R.pack_container_indexed(fmt, None, 12, data.size, 32, 4096, bad, lens, src, chunk_freqs=rows)
    assert e.value.status == R.E_CORRUPT
    worse = rows.copy(); worse[1, 0] += 1
    with pytest.raises(R.RansAmdError) as e:
        R.pack_container_indexed(fmt, None, 12, data.size, 32, 4096, o, lens, src, chunk_freqs=worse)
    assert e.value.status == R.E_MODEL


@pytest.mark.gpu
def test_gpu_sized_encode_to_file_without_a_compaction_pass(tmp_path, oracle):
    """The "keep it" path: sized-slot encode -> ONE D2H copy of the container -> rans_amd_container_pack_indexed
    -> file -> a fresh reader decodes it; an overflowed chunk (incompressible stretch) included.  The file equals the one made
    from the compact encoder's container.  Then the same for per-chunk models (rans_amd_encode_adaptive_sized -> version 2)."""
    import torch
    ctx = R.Context(0)
    chunk = 8192
    data = oracle.gen_zipf(60 * chunk + 321, K=256, s=1.0, seed=8).copy()
    data[7 * chunk:9 * chunk] = np.random.default_rng(5).integers(0, 256, 2 * chunk).astype(np.uint8)  # two chunks that overflow their slots
    d_syms = torch.from_numpy(data).cuda()
    freqs, _ = R.normalize_freqs(ctx.count_freqs_device(d_syms, 256), 4096)
    m = ctx.model(FMT_WORD, freqs, 12)
    t_cont, t_offs, t_lens, t_total, t_slot = ctx.encode_sized(m, d_syms, 64, chunk)
    h_offs = t_offs.cpu().numpy().astype(np.uint64)
    assert int(np.count_nonzero(h_offs[:-1] >= np.uint64(61 * t_slot))) >= 2  # (they lie in the overflow region)
    blob = R.pack_container_indexed(FMT_WORD, freqs, 12, data.size, 64, chunk, h_offs, t_lens.cpu().numpy().astype(np.uint32),
                                    t_cont[:t_total].cpu().numpy())
    # the same file as the one made from the compact encoder's container, up to the alignment padding between chunks (zeros here;
    # whatever its copier waves dragged along there)
    cont, offs, lens, total = ctx.encode(m, d_syms, 64, chunk)
    blob0 = R.pack_container(FMT_WORD, freqs, 12, data.size, 64, chunk, lens.cpu().numpy().astype(np.uint32), cont[:total].cpu().numpy())
    i0, f0, l0, p0 = R.parse_container(blob0)
    i1, f1, l1, p1 = R.parse_container(blob)
    assert blob.size == blob0.size and np.array_equal(f0, f1) and np.array_equal(l0, l1)
    oo = R.offsets_from_lengths(l1)
    for c in range(l1.size):
        assert np.array_equal(p0[int(oo[c]):int(oo[c]) + int(l1[c])], p1[int(oo[c]):int(oo[c]) + int(l1[c])]), c
    path = tmp_path / "sized.rans"
    blob.tofile(path)
    info, f2, l2, p2 = R.parse_container(np.fromfile(path, dtype=np.uint8))
    m2 = ctx.model(info.format, f2, info.scale_bits)
    d_cont = torch.from_numpy(np.concatenate([p2, np.zeros(64, np.uint8)])).cuda()
    out = ctx.decode(m2, d_cont, info.payload_bytes, torch.from_numpy(R.offsets_from_lengths(l2).astype(np.int64)).cuda(),
                     torch.from_numpy(l2.astype(np.int32)).cuda(), info.n_symbols, info.n_ways, info.chunk_syms)
    assert np.array_equal(out.cpu().numpy(), data)
    for fmt in (FMT_BYTE, FMT_WORD):
        c1, o1, l1, r1, t1 = ctx.encode_adaptive_sized(d_syms, 64, chunk, 12, fmt=fmt)
        blob = R.pack_container_indexed(fmt, None, 12, data.size, 64, chunk, o1.cpu().numpy().astype(np.uint64),
                                        l1.cpu().numpy().astype(np.uint32), c1[:t1].cpu().numpy(), chunk_freqs=r1.cpu().numpy())
        c0, o0, l0, r0, t0 = ctx.encode_adaptive(d_syms, 64, chunk, 12, fmt=fmt)
        nch = R.num_chunks(data.size, chunk)
        blob0 = R.pack_container_adaptive(12, data.size, 64, chunk, r0.cpu().numpy().view(np.uint16)[:nch * 256],
                                          l0.cpu().numpy().astype(np.uint32)[:nch], c0[:t0].cpu().numpy(), fmt=fmt)
        i0, f0, ll0, p0 = R.parse_container_adaptive(blob0)
        info, f2, l2, p2 = R.parse_container_adaptive(blob)
        assert blob.size == blob0.size and np.array_equal(f0, f2) and np.array_equal(ll0, l2)
        oo = R.offsets_from_lengths(l2)
        for c in range(nch):  # (the bytes between chunks are alignment padding: zeros here, whatever the device buffer held there)
            assert np.array_equal(p0[int(oo[c]):int(oo[c]) + int(l2[c])], p2[int(oo[c]):int(oo[c]) + int(l2[c])]), c
        d_cont = torch.from_numpy(np.concatenate([p2, np.zeros(64, np.uint8)])).cuda()
        out = ctx.decode_adaptive(d_cont, info.payload_bytes, torch.from_numpy(R.offsets_from_lengths(l2).astype(np.int64)).cuda(),
                                  torch.from_numpy(l2.astype(np.int32)).cuda(),
                                  torch.from_numpy(np.ascontiguousarray(f2).view(np.int16).reshape(-1)).cuda(), info.n_symbols,
                                  info.n_ways, info.chunk_syms, info.scale_bits, fmt=info.format)
        assert np.array_equal(out.cpu().numpy(), data)


def test_container_slice_arithmetic():
    """rans_amd_container_slice (host arrays): the byte hull of a chunk range, 16-byte aligned at its start, and offsets
    rebased to it -- for compact, slot-layout and descending indexes; bad ranges are refused."""
    import ryg_rans_amd as R
    rng = np.random.default_rng(5)
    lens = rng.integers(20, 5000, 40).astype(np.uint32)
    compact = R.offsets_from_lengths(lens)[:-1]
    slot = np.uint64(8192)
    slots = (np.arange(40, dtype=np.uint64) + np.uint64(1)) * slot - lens
    descending = compact[::-1].copy()
    for offs, ls in ((compact, lens), (slots, lens), (descending, lens[::-1].copy())):
        for lo, hi in ((0, 40), (0, 1), (7, 23), (39, 40), (12, 12)):
            b, e, reb = R.container_slice(offs, ls, lo, hi)
            if lo == hi:
                assert b == e and reb.size == 1 and reb[0] == 0
                continue
            assert b % 16 == 0 and b == int(offs[lo:hi].min()) & ~15
            assert e == int((offs[lo:hi] + ls[lo:hi]).max())
            assert np.array_equal(reb[:-1], offs[lo:hi] - np.uint64(b)) and int(reb[-1]) == e - b
    with pytest.raises(R.RansAmdError):
        R.container_slice(compact, lens, 5, 41)
    with pytest.raises(R.RansAmdError):
        R.container_slice(compact, lens, 9, 3)
