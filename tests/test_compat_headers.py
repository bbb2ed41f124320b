"""include/ryg_rans_amd/compat/: the per-symbol RansEnc*/RansDec*/Rans64*/RansWord* API,
re-provided for host code.  Checked (CPU only) against the oracle byte for byte, and --
where /root/reference exists -- by building the reference's own mains UNCHANGED against
these headers and comparing their output with the README's known answers."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from _oracle import FMT_BYTE, FMT_R64, FMT_WORD

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
COMPAT = os.path.join(ROOT, "include", "ryg_rans_amd", "compat")
BUILD = os.path.join(ROOT, "build", "compat")


@pytest.fixture(scope="module")
def drv():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libcompat_driver.so")
    subprocess.run(["g++", "-O2", "-msse4.1", "-shared", "-fPIC", "-I", COMPAT, "-o", so,
                    os.path.join(HERE, "compat_driver.cpp")], check=True)
    lib = C.CDLL(so)
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    lib.compat_encode.argtypes = [C.c_int, u32p, u32p, C.c_uint32, u8p, C.c_size_t, C.c_uint32, u8p, C.c_size_t,
                                  C.POINTER(C.c_size_t)]
    lib.compat_decode.argtypes = [C.c_int, u32p, u32p, C.c_uint32, u8p, C.c_size_t, C.c_size_t, C.c_uint32, u8p]
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def test_struct_layouts(drv):
    assert [drv.compat_sizeof(i) for i in range(7)] == [16, 4, 24, 8, 4, 20480, 16]


@pytest.mark.parametrize("fmt,cfmt,sb", [(FMT_BYTE, 0, 14), (FMT_BYTE, 4, 16), (FMT_BYTE, 0, 9), (FMT_WORD, 1, 12),
                                         (FMT_R64, 2, 14), (FMT_R64, 2, 20)])
def test_streams_equal_oracle(drv, oracle, fmt, cfmt, sb):
    rng = np.random.default_rng(3)
    inputs = [oracle.gen_zipf(50003, K=256, s=1.0, seed=4),
              np.concatenate([np.zeros(9000, np.uint8), np.arange(256, dtype=np.uint8)]),
              rng.integers(0, 256, 777, dtype=np.uint8)]
    for data in inputs:
        f, cum = oracle.normalize(oracle.count_freqs(data, 256), 1 << sb)
        model = oracle.model(f, sb)
        for N in (1, 2, 8, 64, 5):
            want = oracle.encode(fmt, model, data, N)
            cap = (data.size * 4 + N * 8 + 64) & ~7
            buf = np.zeros(cap + 16, np.uint8)
            out_len = C.c_size_t(0)
            assert drv.compat_encode(cfmt, _p(f, C.c_uint32), _p(cum, C.c_uint32), sb, _p(data, C.c_uint8), data.size,
                                     N, _p(buf, C.c_uint8), cap, C.byref(out_len)) == 0
            got = buf[cap - out_len.value:cap]
            assert np.array_equal(got, want), (N,)
            if cfmt == 4:
                continue
            padded = np.concatenate([want, np.zeros(16, np.uint8)])
            out = np.zeros(data.size, np.uint8)
            assert drv.compat_decode(cfmt, _p(f, C.c_uint32), _p(cum, C.c_uint32), sb, _p(padded, C.c_uint8),
                                     want.size, data.size, N, _p(out, C.c_uint8)) == 0
            assert np.array_equal(out, data), (N,)


def test_alias_header_equals_oracle_and_library(drv, oracle):
    """rans_alias_compat.h (the reference keeps its alias coder inside main_alias.cpp:147-267; here it is a header for any
    power-of-two alphabet): streams equal the oracle's byte for byte -- 256 symbols and the 4096-symbol u16 alphabet of
    config 4, several scale_bits and interleaves -- and the tables equal the ones the product library builds."""
    import ryg_rans_amd as R
    from _oracle import FMT_ALIAS
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    drv.compat_alias_encode.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_size_t, C.c_uint32, u8p,
                                        C.c_size_t, C.POINTER(C.c_size_t)]
    drv.compat_alias_decode.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32, u8p, C.c_size_t, C.c_size_t, C.c_uint32,
                                        C.c_void_p, C.c_int]
    drv.compat_alias_tables.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32, u32p]
    rng = np.random.default_rng(17)
    cases = [(256, 16, oracle.gen_zipf(60001, K=256, s=1.0, seed=2)),
             (256, 12, np.minimum(rng.geometric(0.05, 30000) - 1, 255).astype(np.uint8)),
             (256, 8, rng.integers(0, 256, 5000, dtype=np.uint8)),
             (16, 10, rng.integers(0, 16, 7001).astype(np.uint8)),
             (4096, 16, oracle.gen_zipf(90003, K=4096, s=1.0, seed=3)),
             (4096, 13, rng.integers(0, 4096, 40000).astype(np.uint16))]
    for nsyms, sb, data in cases:
        f, cum = oracle.normalize(oracle.count_freqs(data, nsyms), 1 << sb)
        f = np.ascontiguousarray(f, np.uint32)
        cum = np.ascontiguousarray(cum, np.uint32)
        om = oracle.model(f, sb, with_alias=True)
        # tables: header == library (host-only model, no GPU needed)
        M = 1 << sb
        tabs = np.zeros(7 * nsyms + M, np.uint32)
        assert drv.compat_alias_tables(_p(f, C.c_uint32), _p(cum, C.c_uint32), nsyms, sb, _p(tabs, C.c_uint32)) == 0
        lm = R.Model(None, FMT_ALIAS, f, sb)
        sym_dtype = np.uint8 if nsyms <= 256 else np.uint16
        parts = [lm.table(R.TAB_ALIAS_DIVIDER, np.uint32), lm.table(R.TAB_ALIAS_SLOT_ADJUST, np.uint32),
                 lm.table(R.TAB_ALIAS_SLOT_FREQS, np.uint32), lm.table(R.TAB_ALIAS_SYM_ID, sym_dtype).astype(np.uint32),
                 lm.table(R.TAB_ALIAS_REMAP, np.uint32)]
        assert np.array_equal(tabs, np.concatenate(parts)), (nsyms, sb)
        for N in (1, 2, 64, 7):
            want = oracle.encode(FMT_ALIAS, om, data, N)
            cap = (data.size * 4 + N * 8 + 64) & ~7
            buf = np.zeros(cap + 16, np.uint8)
            out_len = C.c_size_t(0)
            assert drv.compat_alias_encode(_p(f, C.c_uint32), _p(cum, C.c_uint32), nsyms, sb, data.ctypes.data,
                                           data.dtype.itemsize, data.size, N, _p(buf, C.c_uint8), cap, C.byref(out_len)) == 0
            assert np.array_equal(buf[cap - out_len.value:cap], want), (nsyms, sb, N)
            padded = np.concatenate([want, np.zeros(16, np.uint8)])
            out = np.zeros(data.size, data.dtype)
            assert drv.compat_alias_decode(_p(f, C.c_uint32), _p(cum, C.c_uint32), nsyms, sb, _p(padded, C.c_uint8), want.size,
                                           data.size, N, out.ctypes.data, data.dtype.itemsize) == 0
            assert np.array_equal(out, data), (nsyms, sb, N)
    # what the builder refuses: an alphabet that is no power of two, frequencies that do not add up
    f = np.array([1000, 1000, 2096], np.uint32)
    cum = np.array([0, 1000, 2000, 4096], np.uint32)
    t = np.zeros(64, np.uint32)
    assert drv.compat_alias_tables(_p(f, C.c_uint32), _p(cum, C.c_uint32), 3, 12, _p(t, C.c_uint32)) == 2
    f = np.array([1000, 1000, 1000, 1000], np.uint32)
    cum = np.array([0, 1000, 2000, 3000, 4000], np.uint32)
    assert drv.compat_alias_tables(_p(f, C.c_uint32), _p(cum, C.c_uint32), 4, 12, _p(t, C.c_uint32)) == 2


def test_reference_mains_build_unchanged_against_compat_headers(book1):
    """The drop-in claim for the per-symbol API: the reference's four sample programs, fed to
    the compiler from stdin (so their own directory is not on the include path), compile
    against OUR headers and reproduce the published sizes (README:48,62,82,96,110)."""
    want = {"main": ["rANS: 435113 bytes", "interleaved rANS: 435117 bytes"],
            "main64": ["rANS: 435116 bytes", "interleaved rANS: 435120 bytes"],
            "main_simd": ["rANS: 435604 bytes", "interleaved rANS: 435606 bytes", "SIMD rANS: 435626 bytes"],
            "main_alias": ["rANS: 435059 bytes", "interleaved rANS: 435063 bytes"]}
    os.makedirs(BUILD, exist_ok=True)
    for name, lines in want.items():
        exe = os.path.join(BUILD, name)
        with open("/root/reference/%s.cpp" % name, "rb") as src:
            subprocess.run(["g++", "-x", "c++", "-O2", "-msse4.1", "-w", "-I", COMPAT, "-o", exe, "-"], stdin=src,
                           check=True)
        out = subprocess.run([exe], cwd="/root/reference", capture_output=True, text=True, check=True).stdout
        for line in lines:
            assert line in out, (name, line)
        assert out.count("decode ok!") == len(lines) and "ERROR" not in out


def _avx2_driver():
    if "avx2" not in open("/proc/cpuinfo").read():
        pytest.skip("host CPU has no AVX2")
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libcompat_avx2_driver.so")
    subprocess.run(["g++", "-O3", "-mavx2", "-shared", "-fPIC", "-I", COMPAT, "-o", so,
                    os.path.join(HERE, "compat_avx2_driver.cpp")], check=True)
    lib = C.CDLL(so)
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    lib.compat_decode_avx2.argtypes = [u32p, u32p, u8p, C.c_size_t, C.c_size_t, u8p]
    lib.compat_time_word8.argtypes = [C.c_int, u32p, u32p, u8p, C.c_size_t, C.c_size_t, u8p, C.c_int]
    lib.compat_time_word8.restype = C.c_double
    return lib


def test_avx2_word_decoder_extension(oracle):
    """rans_word_avx2.h (8 lanes in one vector) decodes the reference's 8-way word streams: every
    tail length, skewed / flat / two-symbol sources, and the cursor ends exactly on the stream end."""
    lib = _avx2_driver()
    rng = np.random.default_rng(5)
    sources = [oracle.gen_zipf(100003, K=256, s=1.0, seed=6), rng.integers(0, 256, 4099, dtype=np.uint8),
               (rng.integers(0, 2, 30001) * 200).astype(np.uint8),
               np.minimum(rng.geometric(0.3, 50000) - 1, 255).astype(np.uint8)]
    for data in sources:
        f, cum = oracle.normalize(oracle.count_freqs(data, 256), 1 << 12)
        model = oracle.model(f, 12)
        for n in (data.size, data.size - 1, data.size - 5, 8, 9, 15, 16, 1):
            d = data[:n]
            want = oracle.encode(FMT_WORD, model, d, 8)
            padded = np.concatenate([want, np.zeros(32, np.uint8)])
            out = np.zeros(n, np.uint8)
            assert lib.compat_decode_avx2(_p(f, C.c_uint32), _p(cum, C.c_uint32), _p(padded, C.c_uint8), want.size,
                                          n, _p(out, C.c_uint8)) == 0, n
            assert np.array_equal(out, d), n


def test_avx512_word_decoder_extension(oracle):
    """rans_word_avx512.h (16 lanes per vector, vpexpandd renormalisation, one gather per 16 symbols from a 4-byte packed
    slot record) decodes 32-way word streams made by the oracle -- i.e. by the reference's encoder loop with 8 -> 32
    lanes -- for lengths with and without a tail round, and on models with frequency-1 symbols."""
    from _oracle import FMT_WORD, HostSimd
    hs = HostSimd()
    if not hs.has_avx512():
        pytest.skip("host CPU has no AVX-512")
    rng = np.random.default_rng(5)
    for n in (1, 31, 32, 33, 1000, 65536 + 17, 300001):
        data = oracle.gen_zipf(n, K=256, s=1.1, seed=n)
        f, _ = oracle.normalize(oracle.count_freqs(np.concatenate([data, np.arange(256, dtype=np.uint8)]), 256), 4096)
        om = oracle.model(f, 12)
        stream = oracle.encode(FMT_WORD, om, data, 32)
        assert np.array_equal(hs.decode_word_avx512x2(f, stream, n), data), n
    data = rng.integers(0, 256, 50000).astype(np.uint8)  # flat model: every frequency 16
    f, _ = oracle.normalize(oracle.count_freqs(data, 256), 4096)
    stream = oracle.encode(FMT_WORD, oracle.model(f, 12), data, 32)
    assert np.array_equal(hs.decode_word_avx512x2(f, stream, data.size), data)
    # the pinning order lists every usable CPU once
    order = hs.cpu_order()
    assert sorted(order) == sorted(os.sched_getaffinity(0))
