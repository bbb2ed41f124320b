"""Property tests on the CPU side (hypothesis): the three implementations of the model builder
(product host code, oracle, unmodified reference) agree on arbitrary histograms, and the
oracle's streams equal the reference's for arbitrary inputs and lane counts."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import ryg_rans_amd as R
from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@st.composite
def histograms(draw):
    present = draw(st.integers(2, 256))
    idx = draw(st.permutations(list(range(256))))[:present]
    counts = np.zeros(256, np.uint32)
    kind = draw(st.sampled_from(["flat", "skew", "spiky"]))
    for j, i in enumerate(idx):
        if kind == "flat":
            counts[i] = draw(st.integers(1, 50))
        elif kind == "skew":
            counts[i] = max(1, 1_000_000 >> min(j, 19))
        else:
            counts[i] = draw(st.sampled_from([1, 1, 1, 2, 3, 1000, 100000]))
    return counts


@settings(max_examples=80, **COMMON)
@given(counts=histograms(), bits=st.sampled_from([8, 10, 12, 14, 16]))
def test_normalize_three_way_agreement(oracle, ref, counts, bits):
    target = 1 << bits
    f_o, c_o = oracle.normalize(counts, target)
    f_r, c_r = ref.normalize(counts, target)
    f_p, c_p = R.normalize_freqs(counts, target)
    assert np.array_equal(f_o, f_r) and np.array_equal(c_o, c_r)
    assert np.array_equal(f_p, f_r) and np.array_equal(c_p, c_r)
    assert int(f_p.sum()) == target and np.all((f_p > 0) == (counts > 0))


@settings(max_examples=40, **COMMON)
@given(counts=histograms())
def test_alias_tables_three_way_agreement(oracle, ref, counts):
    f, _ = oracle.normalize(counts, 65536)
    if f.max() == 65536:
        return
    d, adj, sf, sid, remap = ref.alias_tables(f, 16)
    om = oracle.model(f, 16, with_alias=True)
    pm = R.Model(None, FMT_ALIAS, f, 16)
    assert np.array_equal(om.table("divider", 256), d)
    assert np.array_equal(pm.table(R.TAB_ALIAS_DIVIDER, np.uint32), d)
    assert np.array_equal(pm.table(R.TAB_ALIAS_SLOT_ADJUST, np.uint32), adj)
    assert np.array_equal(pm.table(R.TAB_ALIAS_SLOT_FREQS, np.uint32), sf)
    assert np.array_equal(pm.table(R.TAB_ALIAS_SYM_ID, np.uint8), sid)
    assert np.array_equal(pm.table(R.TAB_ALIAS_REMAP, np.uint32), remap)
    assert np.array_equal(om.table("alias_remap", 65536), remap)


@settings(max_examples=60, **COMMON)
@given(data=st.binary(min_size=2, max_size=3000), n_ways=st.integers(1, 130),
       fmt_sb=st.sampled_from([(FMT_BYTE, 14), (FMT_BYTE, 16), (FMT_BYTE, 8), (FMT_WORD, 12), (FMT_R64, 14),
                               (FMT_R64, 20), (FMT_ALIAS, 16), (FMT_ALIAS, 9)]))
def test_oracle_streams_equal_reference(oracle, ref, data, n_ways, fmt_sb):
    fmt, sb = fmt_sb
    arr = np.frombuffer(data, dtype=np.uint8)
    if len(np.unique(arr)) < 2:
        return  # one-symbol model (freq == M) is outside the reference's range
    f, _ = oracle.normalize(oracle.count_freqs(arr, 256), 1 << sb)
    model = oracle.model(f, sb, with_alias=(fmt == FMT_ALIAS))
    s_o = oracle.encode(fmt, model, arr, n_ways)
    s_r = ref.encode(fmt, f, sb, arr, n_ways)
    assert np.array_equal(s_o, s_r)
    assert np.array_equal(oracle.decode(fmt, model, s_r, arr.size, n_ways), arr)
    back, rc = ref.decode(fmt, f, sb, s_o, arr.size, n_ways)
    assert rc == 0 and np.array_equal(back, arr)
