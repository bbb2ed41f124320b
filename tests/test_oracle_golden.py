"""Pins the CPU oracle (oracle/rans_oracle.c) before anything is checked against it.

1. book1 known answers: the five stream sizes published in the reference README
   and the SHA-256 of streams/tables made by the unmodified reference
   (tests/golden/book1_golden.json).
2. oracle == oracle/_ref (the reference headers compiled where they lie) on
   seeded random inputs, all four formats, N in {1,2,3,8,64,100}.
3. committed fixtures under tests/golden/ (made by make_golden.py through _ref)
   so the pin also holds on boxes without /root/reference.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD

HERE = os.path.dirname(os.path.abspath(__file__))
FMT = {"byte": FMT_BYTE, "word": FMT_WORD, "r64": FMT_R64, "alias": FMT_ALIAS}
GOLD = json.load(open(os.path.join(HERE, "golden", "book1_golden.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_book1_identity(book1):
    assert book1.size == GOLD["input_size"]
    assert sha(book1) == GOLD["input_sha256"]


@pytest.mark.parametrize("entry", GOLD["streams"], ids=lambda e: "%s-N%d" % (e["fmt"], e["n_ways"]))
def test_book1_streams(oracle, book1, entry):
    fmt, sb, N = FMT[entry["fmt"]], entry["scale_bits"], entry["n_ways"]
    model = oracle.model_for(book1, 256, sb, with_alias=(fmt == FMT_ALIAS))
    stream = oracle.encode(fmt, model, book1, N)
    assert stream.size == entry["size"]
    if entry["sha256"]:
        assert sha(stream) == entry["sha256"]
    assert np.array_equal(oracle.decode(fmt, model, stream, book1.size, N), book1)


def test_book1_tables(oracle, book1):
    want = {t["name"]: t for t in GOLD["tables"]}
    counts = oracle.count_freqs(book1, 256)
    for bits in (12, 14, 16):
        _, cum = oracle.normalize(counts, 1 << bits)
        assert sha(cum) == want["cum_freqs_%d" % (1 << bits)]["sha256"]
    f12, cum12 = oracle.normalize(counts, 4096)
    # RansWordTables image: slots {u16 freq, u16 bias}[4096] then slot2sym u8[4096]
    m12 = oracle.model(f12, 12)
    c2s = m12.table("cum2sym", 4096)
    slots = np.zeros((4096, 2), np.uint16)
    slots[:, 0] = f12[c2s]
    slots[:, 1] = np.arange(4096, dtype=np.uint32) - cum12[c2s]
    image = slots.tobytes() + c2s.astype(np.uint8).tobytes()
    assert hashlib.sha256(image).hexdigest() == want["word_tables"]["sha256"]
    f16, _ = oracle.normalize(counts, 65536)
    m16 = oracle.model(f16, 16, with_alias=True)
    assert sha(m16.table("divider", 256)) == want["alias_divider"]["sha256"]
    assert sha(m16.table("slot_adjust", 512)) == want["alias_slot_adjust"]["sha256"]
    assert sha(m16.table("slot_freqs", 512)) == want["alias_slot_freqs"]["sha256"]
    assert sha(m16.table("sym_id", 512).astype(np.uint8)) == want["alias_sym_id"]["sha256"]
    assert sha(m16.table("alias_remap", 65536)) == want["alias_remap"]["sha256"]


def _random_inputs():
    rng = np.random.default_rng(1234)
    yield "uniform", rng.integers(0, 256, 20011, dtype=np.uint8)
    yield "skewed", np.minimum(rng.geometric(0.08, 30000) - 1, 255).astype(np.uint8)
    yield "two-symbol", rng.integers(0, 2, 5000, dtype=np.uint8) * 200
    yield "rare", np.concatenate([np.zeros(40000, np.uint8), np.arange(256, dtype=np.uint8)])
    yield "tiny", np.array([7, 7, 9], dtype=np.uint8)


@pytest.mark.parametrize("fmt,sb", [(FMT_BYTE, 14), (FMT_BYTE, 16), (FMT_BYTE, 9), (FMT_WORD, 12), (FMT_R64, 14),
                                    (FMT_R64, 22), (FMT_ALIAS, 16), (FMT_ALIAS, 10)])
def test_oracle_equals_reference(oracle, ref, fmt, sb):
    for name, data in _random_inputs():
        f_ref, cum_ref = ref.build_model(data, 1 << sb)
        f_orc, cum_orc = oracle.normalize(oracle.count_freqs(data, 256), 1 << sb)
        assert np.array_equal(f_ref, f_orc) and np.array_equal(cum_ref, cum_orc), name
        if f_orc.max() == (1 << sb) and fmt == FMT_WORD:
            continue  # one-symbol model: outside the reference's range
        model = oracle.model(f_orc, sb, with_alias=(fmt == FMT_ALIAS))
        for N in (1, 2, 3, 8, 64, 100):
            s_orc = oracle.encode(fmt, model, data, N)
            s_ref = ref.encode(fmt, f_orc, sb, data, N)
            assert np.array_equal(s_orc, s_ref), (name, N)
            d_ref, rc = ref.decode(fmt, f_orc, sb, s_orc, data.size, N)
            assert rc == 0 and np.array_equal(d_ref, data), (name, N)
            assert np.array_equal(oracle.decode(fmt, model, s_ref, data.size, N), data), (name, N)


def test_alias_tables_equal_reference(oracle, ref):
    for name, data in _random_inputs():
        f, _ = oracle.normalize(oracle.count_freqs(data, 256), 65536)
        m = oracle.model(f, 16, with_alias=True)
        d, adj, sf, sid, remap = ref.alias_tables(f, 16)
        assert np.array_equal(m.table("divider", 256), d), name
        assert np.array_equal(m.table("slot_adjust", 512), adj), name
        assert np.array_equal(m.table("slot_freqs", 512), sf), name
        assert np.array_equal(m.table("sym_id", 512), sid.astype(np.uint32)), name
        assert np.array_equal(m.table("alias_remap", 65536), remap), name


def test_alias_4096_symbols_equal_reference(oracle, ref):
    """Config 4 model: 4096-symbol alphabet, 16-bit probabilities."""
    data = oracle.gen_zipf(200000, K=4096, s=1.0, seed=3)
    f_ref, cum_ref = ref.build_model12(data, 65536)
    f, cum = oracle.normalize(oracle.count_freqs(data, 4096), 65536)
    assert np.array_equal(f, f_ref) and np.array_equal(cum, cum_ref)
    m = oracle.model(f, 16, with_alias=True)
    d, adj, sf, sid, remap = ref.alias_tables12(f, 16)
    assert np.array_equal(m.table("divider", 4096), d)
    assert np.array_equal(m.table("slot_adjust", 8192), adj)
    assert np.array_equal(m.table("slot_freqs", 8192), sf)
    assert np.array_equal(m.table("sym_id", 8192), sid.astype(np.uint32))
    assert np.array_equal(m.table("alias_remap", 65536), remap)
    for N in (1, 2, 64):
        s = oracle.encode(FMT_ALIAS, m, data, N)
        assert np.array_equal(s, ref.encode_alias12(f, 16, data, N))
        out, rc = ref.decode_alias12(f, 16, s, data.size, N)
        assert rc == 0 and np.array_equal(out, data)
        assert np.array_equal(oracle.decode(FMT_ALIAS, m, s, data.size, N, dtype=np.uint16), data)


def test_simd8_path_consumes_oracle_stream(oracle, ref):
    """The reference's SSE4.1 8-way decoder reads the oracle's 8-way word stream."""
    data = oracle.gen_zipf(100003, K=256, s=1.0, seed=1)
    m = oracle.model_for(data, 256, 12)
    s = oracle.encode(FMT_WORD, m, data, 8)
    assert np.array_equal(ref.decode_word_simd8(m.freqs, s, data.size), data)


def test_decode_detects_corruption(oracle):
    data = oracle.gen_zipf(5000, K=256, s=1.0, seed=2)
    for fmt, sb in ((FMT_BYTE, 14), (FMT_WORD, 12), (FMT_R64, 14)):
        m = oracle.model_for(data, 256, sb)
        s = oracle.encode(fmt, m, data, 64)
        _, rc = oracle.decode(fmt, m, s[:-4], data.size, 64, check=False)
        assert rc != 0
        bad = s.copy()
        bad[len(bad) // 2] ^= 0x10
        out, rc = oracle.decode(fmt, m, bad, data.size, 64, check=False)
        assert rc != 0 or not np.array_equal(out, data)


def test_chunked_roundtrip(oracle):
    data = oracle.gen_zipf(70001, K=256, s=1.0, seed=5)
    for fmt, sb in ((FMT_BYTE, 14), (FMT_WORD, 12), (FMT_R64, 14)):
        m = oracle.model_for(data, 256, sb)
        cont, offs, lens = oracle.encode_chunked(fmt, m, data, 64, 4096, align=16)
        assert all(int(o) % 16 == 0 for o in offs[:-1])
        assert np.array_equal(oracle.decode_chunked(fmt, m, cont, offs, lens, data.size, 64, 4096), data)
        # chunk c alone is a plain reference-format stream
        c = 3
        one = cont[int(offs[c]):int(offs[c]) + int(lens[c])]
        assert np.array_equal(one, oracle.encode(fmt, m, data[c * 4096:(c + 1) * 4096], 64))
