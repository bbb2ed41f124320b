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
