"""ctypes bindings for the CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

`Oracle`  -> oracle/librans_oracle.so  (plain-C restatement, always available)
`Ref`     -> oracle/_ref/libryg_ref.so (unmodified reference, prebuilt; optional)

Nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librans_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libryg_ref.so")

FMT_BYTE, FMT_WORD, FMT_R64, FMT_ALIAS = 0, 1, 2, 3
FMT_NAMES = {FMT_BYTE: "byte", FMT_WORD: "word", FMT_R64: "r64", FMT_ALIAS: "alias"}

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def _ptr(a, typ=C.c_void_p):
    return a.ctypes.data_as(typ)


def build_checkers():
    """(Re)build the oracle (and the reference .so when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class OrcModel(C.Structure):
    _fields_ = [
        ("nsyms", C.c_uint32), ("log2nsyms", C.c_uint32), ("scale_bits", C.c_uint32),
        ("freqs", u32p), ("cum", u32p), ("cum2sym", u32p),
        ("divider", u32p), ("slot_adjust", u32p), ("slot_freqs", u32p), ("sym_id", u32p),
        ("alias_remap", u32p),
    ]


class Model:
    """Owning wrapper around orc_model*."""

    def __init__(self, lib, norm_freqs, scale_bits, with_alias=False):
        self.lib = lib
        f = np.ascontiguousarray(norm_freqs, dtype=np.uint32)
        self.freqs = f
        self.nsyms = int(f.size)
        self.scale_bits = int(scale_bits)
        self.ptr = lib.orc_model_create(_ptr(f, u32p), self.nsyms, self.scale_bits, int(with_alias))
        if not self.ptr:
            raise ValueError("orc_model_create failed (bad model)")

    def __del__(self):
        if getattr(self, "ptr", None):
            self.lib.orc_model_destroy(self.ptr)
            self.ptr = None

    def table(self, name, count):
        return np.ctypeslib.as_array(getattr(self.ptr.contents, name), shape=(count,)).copy()


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_checkers()
        lib = C.CDLL(ORACLE_SO)
        lib.orc_count_freqs.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, u32p]
        lib.orc_count_freqs.restype = None
        lib.orc_normalize_freqs.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32]
        lib.orc_model_create.argtypes = [u32p, C.c_uint32, C.c_uint32, C.c_int]
        lib.orc_model_create.restype = C.POINTER(OrcModel)
        lib.orc_model_destroy.argtypes = [C.POINTER(OrcModel)]
        lib.orc_model_destroy.restype = None
        lib.orc_stream_bound.argtypes = [C.c_int, C.c_size_t, C.c_uint32]
        lib.orc_stream_bound.restype = C.c_size_t
        lib.orc_encode.argtypes = [C.c_int, C.POINTER(OrcModel), C.c_void_p, C.c_size_t, C.c_int, C.c_uint32,
                                   u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.orc_decode.argtypes = [C.c_int, C.POINTER(OrcModel), u8p, C.c_size_t, C.c_size_t, C.c_int,
                                   C.c_uint32, C.c_void_p]
        lib.orc_encode_chunked.argtypes = [C.c_int, C.POINTER(OrcModel), C.c_void_p, C.c_size_t, C.c_int,
                                           C.c_uint32, C.c_size_t, C.c_size_t, u8p, C.c_size_t, u64p, u32p,
                                           C.POINTER(C.c_size_t)]
        lib.orc_decode_chunked.argtypes = [C.c_int, C.POINTER(OrcModel), u8p, u64p, u32p, C.c_size_t, C.c_int,
                                           C.c_uint32, C.c_size_t, C.c_void_p]
        lib.orc_compare_chunks.argtypes = [C.c_int, C.POINTER(OrcModel), C.c_void_p, C.c_size_t, C.c_int, C.c_uint32,
                                           C.c_size_t, C.c_uint64, C.c_uint64, u8p, u64p, u32p]
        lib.orc_compare_chunks.restype = C.c_int64
        lib.orc_gen_zipf.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_double, C.c_uint64]
        lib.orc_gen_zipf.restype = None
        u16p = C.POINTER(C.c_uint16)
        lib.orc_compare_chunks_adaptive.argtypes = [C.c_int, u8p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint64,
                                                    C.c_uint64, u8p, u64p, u32p, u16p]
        lib.orc_compare_chunks_adaptive.restype = C.c_int64
        lib.orc_encode_chunks_adaptive.argtypes = [C.c_int, u8p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint64,
                                                   C.c_uint64, u8p, C.c_size_t, u32p, u16p]
        self.lib = lib

    # ---- model
    def count_freqs(self, syms, nsyms):
        syms = np.ascontiguousarray(syms)
        out = np.zeros(nsyms, dtype=np.uint32)
        self.lib.orc_count_freqs(_ptr(syms), syms.size, syms.dtype.itemsize, nsyms, _ptr(out, u32p))
        return out

    def normalize(self, counts, target_total):
        f = np.array(counts, dtype=np.uint32, copy=True)
        cum = np.zeros(f.size + 1, dtype=np.uint32)
        rc = self.lib.orc_normalize_freqs(_ptr(f, u32p), _ptr(cum, u32p), f.size, target_total)
        if rc:
            raise ValueError("orc_normalize_freqs rc=%d" % rc)
        return f, cum

    def model(self, norm_freqs, scale_bits, with_alias=False):
        return Model(self.lib, norm_freqs, scale_bits, with_alias)

    def model_for(self, syms, nsyms, scale_bits, with_alias=False):
        f, _ = self.normalize(self.count_freqs(syms, nsyms), 1 << scale_bits)
        return self.model(f, scale_bits, with_alias)

    # ---- single stream
    def encode(self, fmt, model, syms, n_ways):
        syms = np.ascontiguousarray(syms)
        cap = int(self.lib.orc_stream_bound(fmt, syms.size, n_ways))
        cap = (cap + 7) & ~7
        buf = np.zeros(cap, dtype=np.uint8)
        out_len = C.c_size_t(0)
        rc = self.lib.orc_encode(fmt, model.ptr, _ptr(syms), syms.size, syms.dtype.itemsize, n_ways,
                                 _ptr(buf, u8p), cap, C.byref(out_len))
        if rc:
            raise ValueError("orc_encode rc=%d" % rc)
        return buf[cap - out_len.value:].copy()

    def decode(self, fmt, model, stream, n, n_ways, dtype=np.uint8, check=True):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.zeros(n, dtype=dtype)
        rc = self.lib.orc_decode(fmt, model.ptr, _ptr(stream, u8p), stream.size, n, out.dtype.itemsize, n_ways,
                                 _ptr(out))
        if check and rc:
            raise ValueError("orc_decode rc=%d" % rc)
        return (out, rc) if not check else out

    # ---- chunked
    def encode_chunked(self, fmt, model, syms, n_ways, chunk_syms, align=16):
        syms = np.ascontiguousarray(syms)
        n = syms.size
        nchunks = max(1, (n + chunk_syms - 1) // chunk_syms) if n else 0
        per = int(self.lib.orc_stream_bound(fmt, min(chunk_syms, n), n_ways)) + align
        cap = per * max(nchunks, 1)
        out = np.zeros(cap, dtype=np.uint8)
        offs = np.zeros(nchunks + 1, dtype=np.uint64)
        lens = np.zeros(max(nchunks, 1), dtype=np.uint32)
        total = C.c_size_t(0)
        rc = self.lib.orc_encode_chunked(fmt, model.ptr, _ptr(syms), n, syms.dtype.itemsize, n_ways, chunk_syms,
                                         align, _ptr(out, u8p), cap, _ptr(offs, u64p), _ptr(lens, u32p),
                                         C.byref(total))
        if rc:
            raise ValueError("orc_encode_chunked rc=%d" % rc)
        return out[:total.value].copy(), offs, lens[:nchunks]

    def decode_chunked(self, fmt, model, container, offs, lens, n, n_ways, chunk_syms, dtype=np.uint8):
        container = np.ascontiguousarray(container, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = np.zeros(n, dtype=dtype)
        rc = self.lib.orc_decode_chunked(fmt, model.ptr, _ptr(container, u8p), _ptr(offs, u64p), _ptr(lens, u32p),
                                         n, out.dtype.itemsize, n_ways, chunk_syms, _ptr(out))
        if rc:
            raise ValueError("orc_decode_chunked rc=%d" % rc)
        return out

    # ---- whole containers, threaded over the host cores (ctypes releases the GIL; the C functions share nothing)
    @staticmethod
    def host_threads(limit=64):
        try:
            return max(1, min(limit, len(os.sched_getaffinity(0))))
        except AttributeError:
            return max(1, min(limit, os.cpu_count() or 1))

    def compare_container(self, fmt, model, syms, n_ways, chunk_syms, container, offs, lens, threads=None):
        """EVERY chunk of `container` (index offs / lens) against this oracle's stream for the same symbols.
        Returns (chunks compared, index of the first differing chunk or -1)."""
        from concurrent.futures import ThreadPoolExecutor
        syms = np.ascontiguousarray(syms)
        container = np.ascontiguousarray(container, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = syms.size
        nchunks = (n + chunk_syms - 1) // chunk_syms
        assert lens.size >= nchunks and offs.size >= nchunks
        assert nchunks == 0 or int(offs[nchunks - 1]) + int(lens[nchunks - 1]) <= container.size
        threads = threads or self.host_threads()
        per = max(1, (nchunks + threads * 4 - 1) // (threads * 4))
        ranges = [(c, min(nchunks, c + per)) for c in range(0, nchunks, per)]

        def run(rg):
            return int(self.lib.orc_compare_chunks(fmt, model.ptr, _ptr(syms), n, syms.dtype.itemsize, n_ways, chunk_syms,
                                                   rg[0], rg[1], _ptr(container, u8p), _ptr(offs, u64p), _ptr(lens, u32p)))
        with ThreadPoolExecutor(threads) as ex:
            res = [r for r in ex.map(run, ranges) if r != -1]
        return nchunks, (min(res) if res else -1)

    def compare_container_adaptive(self, fmt, syms, n_ways, chunk_syms, scale_bits, container, offs, lens, rows, threads=None):
        """Per-chunk models: EVERY chunk's row == normalize(count(chunk)) and EVERY chunk's stream == this oracle's stream of the
        chunk under its own model, wherever the index puts it.  Returns (chunks compared, -1 or (chunk, "row" | "stream"))."""
        from concurrent.futures import ThreadPoolExecutor
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        container = np.ascontiguousarray(container, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        rows = np.ascontiguousarray(rows).view(np.uint16).reshape(-1)
        n = syms.size
        nchunks = (n + chunk_syms - 1) // chunk_syms
        assert lens.size >= nchunks and offs.size >= nchunks and rows.size >= nchunks * 256
        assert nchunks == 0 or int((offs[:nchunks] + lens[:nchunks]).max()) <= container.size
        threads = threads or self.host_threads()
        per = max(1, (nchunks + threads * 4 - 1) // (threads * 4))
        ranges = [(c, min(nchunks, c + per)) for c in range(0, nchunks, per)]
        u16p = C.POINTER(C.c_uint16)

        def run(rg):
            return int(self.lib.orc_compare_chunks_adaptive(fmt, _ptr(syms, u8p), n, n_ways, chunk_syms, scale_bits, rg[0], rg[1],
                                                            _ptr(container, u8p), _ptr(offs, u64p), _ptr(lens, u32p),
                                                            _ptr(rows, u16p)))
        with ThreadPoolExecutor(threads) as ex:
            res = [r for r in ex.map(run, ranges) if r != -1]
        if not res:
            return nchunks, -1
        assert min(res) >= 0, "orc_compare_chunks_adaptive: bad argument"
        r = min(res)
        return nchunks, (r >> 1, "stream" if r & 1 else "row")

    def encode_chunked_adaptive(self, fmt, syms, n_ways, chunk_syms, scale_bits, align=16, threads=None):
        """The oracle's own per-chunk-model container: (container, offs[nchunks + 1], lens, rows u16[nchunks, 256]), chunk c
        at the sum of the aligned lengths before it."""
        from concurrent.futures import ThreadPoolExecutor
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        n = syms.size
        nchunks = (n + chunk_syms - 1) // chunk_syms
        slot = (int(self.lib.orc_stream_bound(fmt, min(chunk_syms, n), n_ways)) + 15) & ~15
        threads = threads or self.host_threads()
        per = max(1, (nchunks + threads * 4 - 1) // (threads * 4))
        ranges = [(c, min(nchunks, c + per)) for c in range(0, nchunks, per)]
        lens = np.zeros(nchunks, dtype=np.uint32)
        rows = np.zeros((nchunks, 256), dtype=np.uint16)
        u16p = C.POINTER(C.c_uint16)

        def run(rg):
            tmp = np.zeros((rg[1] - rg[0]) * slot, dtype=np.uint8)
            rc = self.lib.orc_encode_chunks_adaptive(fmt, _ptr(syms, u8p), n, n_ways, chunk_syms, scale_bits, rg[0], rg[1],
                                                     _ptr(tmp, u8p), slot, _ptr(lens[rg[0]:rg[1]], u32p),
                                                     _ptr(rows[rg[0]:rg[1]], u16p))
            assert rc == 0, "orc_encode_chunks_adaptive rc=%d" % rc
            return tmp
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(run, ranges))
        aligned = (lens.astype(np.uint64) + np.uint64(align - 1)) & ~np.uint64(align - 1)
        offs = np.zeros(nchunks + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(aligned)
        if nchunks:
            offs[nchunks] = offs[nchunks - 1] + np.uint64(lens[nchunks - 1])
        out = np.zeros(int(offs[nchunks]) + 16, dtype=np.uint8)
        for rg, tmp in zip(ranges, parts):
            for c in range(rg[0], rg[1]):
                e = (c - rg[0] + 1) * slot
                out[int(offs[c]):int(offs[c]) + int(lens[c])] = tmp[e - int(lens[c]):e]
        return out[:int(offs[nchunks])], offs, lens, rows

    def encode_chunked_mt(self, fmt, model, syms, n_ways, chunk_syms, align=16, threads=None):
        """encode_chunked over ranges of whole chunks in parallel, stitched: the same container, offsets and lengths."""
        from concurrent.futures import ThreadPoolExecutor
        syms = np.ascontiguousarray(syms)
        n = syms.size
        nchunks = (n + chunk_syms - 1) // chunk_syms
        threads = threads or self.host_threads()
        per = max(1, (nchunks + threads * 2 - 1) // (threads * 2))
        starts = list(range(0, nchunks, per))
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(lambda c: self.encode_chunked(fmt, model, syms[c * chunk_syms:(c + per) * chunk_syms],
                                                              n_ways, chunk_syms, align), starts))
        offs = np.zeros(nchunks + 1, dtype=np.uint64)
        lens = np.zeros(nchunks, dtype=np.uint32)
        pos = 0
        bases = []
        for c, (cont, o, ln) in zip(starts, parts):
            pos = (pos + align - 1) // align * align
            bases.append(pos)
            offs[c:c + ln.size] = o[:ln.size] + np.uint64(pos)
            lens[c:c + ln.size] = ln
            pos += cont.size
        offs[nchunks] = pos
        out = np.zeros(pos, dtype=np.uint8)
        for b, (cont, _, _) in zip(bases, parts):
            out[b:b + cont.size] = cont
        return out, offs, lens

    def gen_zipf(self, n, K=256, s=1.0, seed=1):
        dtype = np.uint8 if K <= 256 else np.uint16
        out = np.zeros(n, dtype=dtype)
        self.lib.orc_gen_zipf(_ptr(out), n, out.dtype.itemsize, K, float(s), seed)
        return out


HOST_SIMD_SO = os.path.join(ORACLE_DIR, "libhost_simd.so")


class HostSimd:
    """oracle/host_simd.cpp: pinned-thread timing harness for CPU decoders + the AVX-512 word decoder of
    include/ryg_rans_amd/compat/rans_word_avx512.h (bench.py's cpu_baseline leg and its tests only)."""

    @staticmethod
    def available():
        return os.path.exists(HOST_SIMD_SO)

    def __init__(self):
        if not os.path.exists(HOST_SIMD_SO):
            build_checkers()
        lib = C.CDLL(HOST_SIMD_SO)
        lib.host_has_avx512.restype = C.c_int
        lib.host_decode_word_avx512x2.argtypes = [u32p, u8p, C.c_size_t, u8p]
        lib.host_cpu_order.argtypes = [C.POINTER(C.c_int), C.c_int]
        lib.host_time_threads.argtypes = [C.c_void_p, u32p, u8p, u64p, C.c_uint32, C.c_size_t, u8p, C.c_uint32, C.c_uint32,
                                          C.c_int]
        lib.host_time_threads.restype = C.c_double
        self.lib = lib

    def has_avx512(self):
        return bool(self.lib.host_has_avx512())

    def cpu_order(self):
        buf = (C.c_int * 4096)()
        n = self.lib.host_cpu_order(buf, 4096)
        return [int(buf[i]) for i in range(n)]

    def physical_cores(self):
        """Number of distinct cores among the CPUs this process may use (the first block of cpu_order)."""
        seen = set()
        for c in self.cpu_order():
            try:
                first = int(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read()
                            .replace("-", ",").split(",")[0])
            except (OSError, ValueError):
                first = c
            seen.add(first)
        return max(1, len(seen))

    def decode_word_avx512x2(self, freqs, stream, n):
        """A 32-way word stream by two 16-lane vectors; the stream gets its 32 bytes of padding here."""
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        padded = np.concatenate([np.ascontiguousarray(stream, dtype=np.uint8), np.zeros(64, np.uint8)])
        out = np.zeros(n, dtype=np.uint8)
        rc = self.lib.host_decode_word_avx512x2(_ptr(f, u32p), _ptr(padded, u8p), n, _ptr(out, u8p))
        if rc:
            raise ValueError("host_decode_word_avx512x2 rc=%d" % rc)
        return out


class RefLoop2Times(C.Structure):
    """ref_loop2_times (oracle/ref_driver.cpp)"""
    _fields_ = [("enc_s", C.c_double), ("dec_s", C.c_double), ("enc_clocks", C.c_uint64), ("dec_clocks", C.c_uint64),
                ("stream_bytes", C.c_uint64), ("ok", C.c_int32), ("pad", C.c_int32)]


class Ref:
    """The unmodified reference behind a C ABI (oracle/ref_driver.cpp)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        lib = C.CDLL(REF_SO)
        lib.ref_build_model_u8.argtypes = [u8p, C.c_size_t, C.c_uint32, u32p, u32p]
        lib.ref_normalize_u8.argtypes = [u32p, u32p, C.c_uint32]
        lib.ref_alias_tables_u8.argtypes = [u32p, u32p, u32p, u32p, u8p, u32p]
        lib.ref_word_tables_u8.argtypes = [u32p, u8p]
        lib.ref_encode_u8.argtypes = [C.c_int, u32p, C.c_uint32, u8p, C.c_size_t, C.c_uint32, u8p, C.c_size_t,
                                      C.POINTER(C.c_size_t)]
        lib.ref_decode_u8.argtypes = [C.c_int, u32p, C.c_uint32, u8p, C.c_size_t, C.c_size_t, C.c_uint32, u8p]
        lib.ref_decode_word_simd8.argtypes = [u32p, u8p, C.c_size_t, u8p]
        lib.ref_time_word_simd8.argtypes = [u32p, u8p, u64p, C.c_uint32, C.c_size_t, u8p, C.c_uint32, C.c_uint32]
        lib.ref_time_word_simd8.restype = C.c_double
        lib.ref_rdtsc.restype = C.c_uint64
        lib.ref12_build_model.argtypes = [u16p, C.c_size_t, C.c_uint32, u32p, u32p]
        lib.ref12_alias_tables.argtypes = [u32p, u32p, u32p, u32p, u16p, u32p]
        lib.ref12_encode_alias.argtypes = [u32p, C.c_uint32, u16p, C.c_size_t, C.c_uint32, u8p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        lib.ref12_decode_alias.argtypes = [u32p, C.c_uint32, u8p, C.c_size_t, C.c_size_t, C.c_uint32, u16p]
        if hasattr(lib, "ref_time_loop2_mt"):  # (a _ref built before round 4 lacks it)
            lib.ref_time_loop2_mt.argtypes = [C.c_int, u32p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_int),
                                              C.c_int, C.POINTER(RefLoop2Times)]
        self.lib = lib

    def has_loop2(self):
        return hasattr(self.lib, "ref_time_loop2_mt")

    def time_loop2(self, which, freqs, scale_bits, data, n_per, threads=1, cpus=()):
        """The reference's own 2-way loop of a format exactly as its main times it (main.cpp:226-280, main64.cpp:228-282,
        main_alias.cpp:353-405; which = FMT_BYTE / FMT_R64 / FMT_ALIAS over bytes, 12 = the 4096-symbol alias model over
        u16 symbols): `threads` shards of n_per symbols from the start of `data`, one (pinned) thread each, encode pass then
        decode pass.  Returns one dict per thread: seconds and rdtsc clocks of each pass, stream bytes, ok (= decode ok!)."""
        data = np.ascontiguousarray(data, dtype=np.uint16 if which == 12 else np.uint8)
        assert data.size >= threads * n_per
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        times = (RefLoop2Times * threads)()
        cpu_arr = (C.c_int * max(1, len(cpus)))(*cpus) if cpus else (C.c_int * 1)(0)
        rc = self.lib.ref_time_loop2_mt(which, _ptr(f, u32p), scale_bits, data.ctypes.data, n_per, threads, cpu_arr, len(cpus), times)
        assert rc == 0, "ref_time_loop2_mt failed (%d)" % rc
        return [{"enc_s": t.enc_s, "dec_s": t.dec_s, "enc_clocks": t.enc_clocks, "dec_clocks": t.dec_clocks,
                 "stream_bytes": t.stream_bytes, "ok": bool(t.ok)} for t in times]

    def build_model(self, data, target_total):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        f = np.zeros(256, dtype=np.uint32)
        cum = np.zeros(257, dtype=np.uint32)
        self.lib.ref_build_model_u8(_ptr(data, u8p), data.size, target_total, _ptr(f, u32p), _ptr(cum, u32p))
        return f, cum

    def normalize(self, counts, target_total):
        f = np.array(counts, dtype=np.uint32, copy=True)
        assert f.size == 256
        cum = np.zeros(257, dtype=np.uint32)
        self.lib.ref_normalize_u8(_ptr(f, u32p), _ptr(cum, u32p), target_total)
        return f, cum

    def alias_tables(self, freqs, scale_bits):
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        d = np.zeros(256, np.uint32); adj = np.zeros(512, np.uint32); sf = np.zeros(512, np.uint32)
        sid = np.zeros(512, np.uint8); remap = np.zeros(1 << scale_bits, np.uint32)
        self.lib.ref_alias_tables_u8(_ptr(f, u32p), _ptr(d, u32p), _ptr(adj, u32p), _ptr(sf, u32p), _ptr(sid, u8p),
                                     _ptr(remap, u32p))
        return d, adj, sf, sid, remap

    def word_tables(self, freqs):
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        img = np.zeros(20480, np.uint8)
        self.lib.ref_word_tables_u8(_ptr(f, u32p), _ptr(img, u8p))
        return img

    def encode(self, fmt, freqs, scale_bits, data, n_ways):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        cap = (data.size * 4 + n_ways * 8 + 64 + 7) & ~7
        buf = np.zeros(cap, np.uint8)
        out_len = C.c_size_t(0)
        rc = self.lib.ref_encode_u8(fmt, _ptr(f, u32p), scale_bits, _ptr(data, u8p), data.size, n_ways,
                                    _ptr(buf, u8p), cap, C.byref(out_len))
        assert rc == 0
        return buf[cap - out_len.value:].copy()

    def decode(self, fmt, freqs, scale_bits, stream, n, n_ways):
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        s = np.zeros(len(stream) + 16, np.uint8)
        s[:len(stream)] = stream
        out = np.zeros(n, np.uint8)
        rc = self.lib.ref_decode_u8(fmt, _ptr(f, u32p), scale_bits, _ptr(s, u8p), len(stream), n, n_ways,
                                    _ptr(out, u8p))
        return out, rc

    def decode_word_simd8(self, freqs, stream, n):
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        s = np.zeros(len(stream) + 16, np.uint8)
        s[:len(stream)] = stream
        out = np.zeros(n, np.uint8)
        self.lib.ref_decode_word_simd8(_ptr(f, u32p), _ptr(s, u8p), n, _ptr(out, u8p))
        return out

    # 4096-symbol alias variant
    def build_model12(self, data, target_total):
        data = np.ascontiguousarray(data, dtype=np.uint16)
        f = np.zeros(4096, np.uint32); cum = np.zeros(4097, np.uint32)
        self.lib.ref12_build_model(_ptr(data, u16p), data.size, target_total, _ptr(f, u32p), _ptr(cum, u32p))
        return f, cum

    def alias_tables12(self, freqs, scale_bits):
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        d = np.zeros(4096, np.uint32); adj = np.zeros(8192, np.uint32); sf = np.zeros(8192, np.uint32)
        sid = np.zeros(8192, np.uint16); remap = np.zeros(1 << scale_bits, np.uint32)
        self.lib.ref12_alias_tables(_ptr(f, u32p), _ptr(d, u32p), _ptr(adj, u32p), _ptr(sf, u32p), _ptr(sid, u16p),
                                    _ptr(remap, u32p))
        return d, adj, sf, sid, remap

    def encode_alias12(self, freqs, scale_bits, data, n_ways):
        data = np.ascontiguousarray(data, dtype=np.uint16)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        cap = (data.size * 4 + n_ways * 8 + 64 + 7) & ~7
        buf = np.zeros(cap, np.uint8)
        out_len = C.c_size_t(0)
        rc = self.lib.ref12_encode_alias(_ptr(f, u32p), scale_bits, _ptr(data, u16p), data.size, n_ways,
                                         _ptr(buf, u8p), cap, C.byref(out_len))
        assert rc == 0
        return buf[cap - out_len.value:].copy()

    def decode_alias12(self, freqs, scale_bits, stream, n, n_ways):
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        s = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.zeros(n, np.uint16)
        rc = self.lib.ref12_decode_alias(_ptr(f, u32p), scale_bits, _ptr(s, u8p), s.size, n, n_ways, _ptr(out, u16p))
        return out, rc
