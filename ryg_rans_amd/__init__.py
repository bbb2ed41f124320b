"""ryg_rans_amd -- MI355X-native interleaved rANS coder (Python test/bench driver).

This package is a thin ctypes binding of the C ABI in include/ryg_rans_amd.h
(built from ryg_rans_amd/csrc into ryg_rans_amd/lib/libryg_rans_amd.so).  All
encode/decode work happens in the hand-written HIP kernels behind that ABI; there
is no Python or CPU implementation of the hot path here, and importing the
package fails loudly when the shared object is missing.

PyTorch is only plumbing: device buffers are torch tensors whose data_ptr() is
handed to the C ABI, and the current torch stream is the launch stream.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RANS_AMD_LIB: load an alternative build of the same library (kernel experiments)
LIB_PATH = os.environ.get("RANS_AMD_LIB") or os.path.join(_HERE, "lib", "libryg_rans_amd.so")

FMT_BYTE, FMT_WORD, FMT_R64, FMT_ALIAS = 0, 1, 2, 3
FORMAT_NAMES = {FMT_BYTE: "byte", FMT_WORD: "word", FMT_R64: "r64", FMT_ALIAS: "alias"}

OPT_LANE_KERNELS, OPT_LANE_FUSED_PLACEMENT, OPT_FUSED_PLACEMENT, OPT_DUAL_DECODE, OPT_ENC_SCRATCH_RING = range(5)

OK, E_ARG, E_MODEL, E_SPACE, E_CORRUPT, E_UNSUPPORTED, E_HIP, E_NOMEM = range(8)

(TAB_FREQS, TAB_CUM_FREQS, TAB_CUM2SYM, TAB_WORD_SLOTS, TAB_ALIAS_DIVIDER, TAB_ALIAS_SLOT_ADJUST,
 TAB_ALIAS_SLOT_FREQS, TAB_ALIAS_SYM_ID, TAB_ALIAS_REMAP, TAB_ENC_SYMBOLS, TAB_DEC_SYMBOLS) = range(11)

# every symbol include/ryg_rans_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "rans_amd_version", "rans_amd_build_flags", "rans_amd_status_string", "rans_amd_last_error", "rans_amd_device_count",
    "rans_amd_ctx_create", "rans_amd_ctx_destroy", "rans_amd_ctx_trim", "rans_amd_ctx_set_option",
    "rans_amd_count_freqs_host", "rans_amd_count_freqs", "rans_amd_normalize_freqs",
    "rans_amd_model_create", "rans_amd_model_destroy", "rans_amd_model_format", "rans_amd_model_scale_bits",
    "rans_amd_model_nsyms", "rans_amd_model_sym_bytes", "rans_amd_model_table",
    "rans_amd_num_chunks", "rans_amd_chunk_bound", "rans_amd_encode_bound", "rans_amd_ways_supported",
    "rans_amd_encode", "rans_amd_encode_status", "rans_amd_decode", "rans_amd_decode_errors",
    "rans_amd_encode_slots", "rans_amd_slot_bytes", "rans_amd_encode_slots_bound", "rans_amd_container_compact",
    "rans_amd_container_slice",
    "rans_amd_encode_slots_sized", "rans_amd_tight_slot_bytes", "rans_amd_encode_sized_bound", "rans_amd_probe_placement",
    "rans_amd_encode_host", "rans_amd_decode_host",
    "rans_amd_set_timing", "rans_amd_last_kernel_ms", "rans_amd_last_decode_kernel", "rans_amd_last_encode_kernel", "rans_amd_last_wave_clocks",
    "rans_amd_launch_spans",
    "rans_amd_chunk_freqs_bytes", "rans_amd_encode_adaptive", "rans_amd_decode_adaptive",
    "rans_amd_encode_adaptive_fmt", "rans_amd_decode_adaptive_fmt",
    "rans_amd_encode_adaptive_sized", "rans_amd_encode_adaptive_sized_bound",
    "rans_amd_container_bytes_adaptive", "rans_amd_container_pack_adaptive", "rans_amd_container_parse_adaptive",
    "rans_amd_offsets_from_lengths", "rans_amd_container_bytes", "rans_amd_container_pack",
    "rans_amd_packed_payload_bytes", "rans_amd_container_pack_indexed", "rans_amd_container_pack_indexed_adaptive",
    "rans_amd_container_parse", "rans_amd_encode_workspace_bytes", "rans_amd_build_model_o0",
]


class ContainerInfo(C.Structure):
    """rans_amd_container_info"""
    _fields_ = [("format", C.c_uint32), ("scale_bits", C.c_uint32), ("nsyms", C.c_uint32), ("n_ways", C.c_uint32),
                ("chunk_syms", C.c_uint32), ("sym_bytes", C.c_uint32), ("n_symbols", C.c_uint64),
                ("n_chunks", C.c_uint64), ("payload_bytes", C.c_uint64)]


class WaveClocks(C.Structure):
    """rans_amd_wave_clocks"""
    _fields_ = [("waves", C.c_uint64), ("rounds", C.c_uint64), ("shader_cycles", C.c_uint64),
                ("sclk_hz", C.c_double), ("kernel_ticks_ms", C.c_double)]


class RansAmdError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s: status %d (%s)%s" % (where, status, _status_string(status),
                                                   (": " + detail) if detail else ""))


def _load():
    # torch ships its own libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's).
    # Whoever loads first wins; two HIP runtimes in one process see no devices from the
    # second one.  Load torch's first so that our library binds to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ryg_rans_amd: %s is missing -- build it with `make -C ryg_rans_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback implementation" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    sig = {
        "rans_amd_version": (i32, []),
        "rans_amd_build_flags": (u32, []),
        "rans_amd_status_string": (C.c_char_p, [i32]),
        "rans_amd_last_error": (C.c_char_p, []),
        "rans_amd_device_count": (i32, []),
        "rans_amd_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "rans_amd_ctx_destroy": (i32, [vp]),
        "rans_amd_ctx_trim": (i32, [vp]),
        "rans_amd_ctx_set_option": (i32, [vp, i32, i32]),
        "rans_amd_encode_status": (i32, [vp, vp]),
        "rans_amd_count_freqs_host": (i32, [vp, u64, i32, u32, u32p]),
        "rans_amd_count_freqs": (i32, [vp, vp, u64, i32, u32, u32p, vp]),
        "rans_amd_normalize_freqs": (i32, [u32p, u32p, u32, u32]),
        "rans_amd_model_create": (i32, [vp, i32, u32p, u32, u32, C.POINTER(vp)]),
        "rans_amd_model_destroy": (i32, [vp]),
        "rans_amd_model_format": (i32, [vp]),
        "rans_amd_model_scale_bits": (u32, [vp]),
        "rans_amd_model_nsyms": (u32, [vp]),
        "rans_amd_model_sym_bytes": (i32, [vp]),
        "rans_amd_model_table": (i32, [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "rans_amd_num_chunks": (u64, [u64, u32]),
        "rans_amd_chunk_bound": (u64, [i32, u32, u32]),
        "rans_amd_encode_bound": (u64, [i32, u64, u32, u32]),
        "rans_amd_ways_supported": (i32, [i32, u32]),
        "rans_amd_encode": (i32, [vp, vp, vp, u64, u32, u32, vp, u64, vp, vp, u64p, vp]),
        "rans_amd_encode_slots": (i32, [vp, vp, vp, u64, u32, u32, vp, u64, vp, vp, u64p, vp]),
        "rans_amd_slot_bytes": (u64, [i32, u64, u32, u32]),
        "rans_amd_encode_slots_sized": (i32, [vp, vp, vp, u64, u32, u32, u64, vp, u64, vp, vp, u64p, vp]),
        "rans_amd_tight_slot_bytes": (u64, [vp, u32, u32]),
        "rans_amd_encode_sized_bound": (u64, [i32, u64, u32, u32, u64, u64]),
        "rans_amd_encode_slots_bound": (u64, [i32, u64, u32, u32]),
        "rans_amd_container_compact": (i32, [vp, vp, u64, vp, vp, u64, vp, u64, vp, u64p, vp]),
        "rans_amd_container_slice": (i32, [u64p, u32p, u64, u64, u64, u64p, u64p, u64p]),
        "rans_amd_decode": (i32, [vp, vp, vp, u64, vp, vp, u64, u32, u32, vp, u64p, vp]),
        "rans_amd_decode_errors": (i32, [vp, u64p, vp]),
        "rans_amd_probe_placement": (i32, [vp, vp, C.POINTER(vp), u32, u64, vp, vp, u64, u32, u32, C.POINTER(vp), u32, u32, u32,
                                         u32p, u32p, C.POINTER(C.c_float), vp]),
        "rans_amd_encode_host": (i32, [vp, vp, vp, u64, u32, vp, u64, u64p]),
        "rans_amd_decode_host": (i32, [vp, vp, vp, u64, u64, u32, vp]),
        "rans_amd_set_timing": (i32, [vp, i32]),
        "rans_amd_last_kernel_ms": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "rans_amd_last_decode_kernel": (C.c_char_p, [vp]),
        "rans_amd_last_encode_kernel": (C.c_char_p, [vp, C.POINTER(C.c_int)]),
        "rans_amd_last_wave_clocks": (i32, [vp, C.POINTER(WaveClocks)]),
        "rans_amd_launch_spans": (i32, [vp, u32, C.POINTER(C.c_double), vp]),
        "rans_amd_chunk_freqs_bytes": (u64, [u64, u32]),
        "rans_amd_encode_adaptive": (i32, [vp, vp, u64, u32, u32, u32, vp, u64, vp, vp, vp, u64p, vp]),
        "rans_amd_decode_adaptive": (i32, [vp, vp, u64, vp, vp, vp, u64, u32, u32, u32, vp, u64p, vp]),
        "rans_amd_encode_adaptive_fmt": (i32, [vp, i32, vp, u64, u32, u32, u32, vp, u64, vp, vp, vp, u64p, vp]),
        "rans_amd_decode_adaptive_fmt": (i32, [vp, i32, vp, u64, vp, vp, vp, u64, u32, u32, u32, vp, u64p, vp]),
        "rans_amd_encode_adaptive_sized": (i32, [vp, i32, vp, u64, u32, u32, u32, vp, u64, vp, vp, vp, u64p, vp]),
        "rans_amd_encode_adaptive_sized_bound": (u64, [i32, u64, u32, u32]),
        "rans_amd_encode_workspace_bytes": (u64, [i32, u64, u32, u32]),
        "rans_amd_build_model_o0": (i32, [vp, i32, vp, u64, i32, u32, u32, u32p, C.POINTER(vp), vp]),
        "rans_amd_offsets_from_lengths": (i32, [u32p, u64, u64p]),
        "rans_amd_container_bytes": (u64, [C.POINTER(ContainerInfo)]),
        "rans_amd_container_pack": (i32, [C.POINTER(ContainerInfo), u32p, u32p, vp, vp, u64, u64p]),
        "rans_amd_container_parse": (i32, [vp, u64, C.POINTER(ContainerInfo), C.POINTER(u32p), C.POINTER(u32p),
                                         C.POINTER(vp)]),
        "rans_amd_packed_payload_bytes": (u64, [u32p, u64]),
        "rans_amd_container_pack_indexed": (i32, [C.POINTER(ContainerInfo), u32p, u64p, u32p, vp, u64, vp, u64, u64p]),
        "rans_amd_container_pack_indexed_adaptive": (i32, [C.POINTER(ContainerInfo), vp, u64p, u32p, vp, u64, vp, u64, u64p]),
        "rans_amd_container_bytes_adaptive": (u64, [C.POINTER(ContainerInfo)]),
        "rans_amd_container_pack_adaptive": (i32, [C.POINTER(ContainerInfo), vp, u32p, vp, vp, u64, u64p]),
        "rans_amd_container_parse_adaptive": (i32, [vp, u64, C.POINTER(ContainerInfo), C.POINTER(vp), C.POINTER(u32p),
                                                  C.POINTER(vp)]),
    }
    for name, (res, args) in sig.items():
        if not hasattr(lib, name) and os.environ.get("RANS_AMD_LIB"):
            continue  # an older experimental build (A/B runs) may lack the newest entry points
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load()


def _status_string(status):
    return _lib.rans_amd_status_string(status).decode()


def _check(status, where):
    if status != OK:
        raise RansAmdError(status, where, _lib.rans_amd_last_error().decode())


def lib():
    """The raw ctypes handle (for ABI-level tests)."""
    return _lib


def device_count():
    return int(_lib.rans_amd_device_count())


# ---- model building on the host (SymbolStats, main.cpp:49-129) -----------------

def count_freqs(syms, nsyms):
    syms = np.ascontiguousarray(syms)
    assert syms.dtype in (np.uint8, np.uint16)
    out = np.zeros(nsyms, dtype=np.uint32)
    _check(_lib.rans_amd_count_freqs_host(syms.ctypes.data, syms.size, syms.dtype.itemsize, nsyms,
                                          out.ctypes.data_as(C.POINTER(C.c_uint32))), "count_freqs")
    return out


def normalize_freqs(counts, target_total):
    f = np.array(counts, dtype=np.uint32, copy=True)
    cum = np.zeros(f.size + 1, dtype=np.uint32)
    _check(_lib.rans_amd_normalize_freqs(f.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         cum.ctypes.data_as(C.POINTER(C.c_uint32)), f.size, target_total),
           "normalize_freqs")
    return f, cum


def num_chunks(n, chunk_syms):
    return int(_lib.rans_amd_num_chunks(n, chunk_syms))


def chunk_bound(fmt, chunk_syms, n_ways):
    return int(_lib.rans_amd_chunk_bound(fmt, chunk_syms, n_ways))


def encode_bound(fmt, n, n_ways, chunk_syms):
    return int(_lib.rans_amd_encode_bound(fmt, n, n_ways, chunk_syms))


def slot_bytes(fmt, n, n_ways, chunk_syms):
    return int(_lib.rans_amd_slot_bytes(fmt, n, n_ways, chunk_syms))


def encode_slots_bound(fmt, n, n_ways, chunk_syms):
    return int(_lib.rans_amd_encode_slots_bound(fmt, n, n_ways, chunk_syms))


def encode_sized_bound(fmt, n, n_ways, chunk_syms, slot, overflow_chunks):
    return int(_lib.rans_amd_encode_sized_bound(fmt, n, n_ways, chunk_syms, slot, overflow_chunks))


def ways_supported(fmt, n_ways):
    return bool(_lib.rans_amd_ways_supported(fmt, n_ways))


class Context:
    """One per (process, GPU).  Raises RansAmdError(E_HIP) when no GPU is usable."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(_lib.rans_amd_ctx_create(device, C.byref(self._h)), "ctx_create")
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.rans_amd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim(self):
        """rans_amd_ctx_trim: frees the workspaces the context keeps between calls (they come back on demand)."""
        _check(_lib.rans_amd_ctx_trim(self._h), "ctx_trim")

    def set_option(self, option, value):
        """rans_amd_ctx_set_option: kernel-family choices (all produce the same bytes)."""
        _check(_lib.rans_amd_ctx_set_option(self._h, int(option), int(value)), "ctx_set_option")

    # -- measurement
    def set_timing(self, on=True):
        """True/1: HIP events around every launch; 2: per-wave clocks as well (slower, synchronising)."""
        _check(_lib.rans_amd_set_timing(self._h, int(on)), "set_timing")

    def launch_spans(self, count):
        """First-wave-start to last-wave-end (ms) of the last `count` decode launches, oldest first."""
        out = (C.c_double * count)()
        _check(_lib.rans_amd_launch_spans(self._h, count, out, _torch_stream()), "launch_spans")
        return [float(v) for v in out]

    def last_wave_clocks(self):
        """Per-wave clocks of the last decode launched under set_timing(2), as a dict."""
        wc = WaveClocks()
        _check(_lib.rans_amd_last_wave_clocks(self._h, C.byref(wc)), "last_wave_clocks")
        return {"waves": wc.waves, "rounds": wc.rounds, "shader_cycles": wc.shader_cycles, "sclk_hz": wc.sclk_hz,
                "kernel_ticks_ms": wc.kernel_ticks_ms}

    def last_kernel_ms(self):
        d, e = C.c_float(-1), C.c_float(-1)
        _check(_lib.rans_amd_last_kernel_ms(self._h, C.byref(d), C.byref(e)), "last_kernel_ms")
        return d.value, e.value

    def last_decode_kernel(self):
        return _lib.rans_amd_last_decode_kernel(self._h).decode()

    def last_encode_kernel(self):
        """(name of the coding kernel of the last encode, True if no layout / compaction kernels ran behind it)"""
        fused = C.c_int(0)
        name = _lib.rans_amd_last_encode_kernel(self._h, C.byref(fused)).decode()
        return name, bool(fused.value)

    def last_encode_placement(self):
        """0: layout + compaction kernels behind the coder, 1: the coder placed its chunks itself, 2: slot layout (nothing moved)"""
        fused = C.c_int(0)
        _lib.rans_amd_last_encode_kernel(self._h, C.byref(fused))
        return int(fused.value)

    # -- model
    def model(self, fmt, norm_freqs, scale_bits):
        return Model(self, fmt, norm_freqs, scale_bits)

    def model_for(self, fmt, syms, nsyms, scale_bits):
        """count_freqs + normalize_freqs + table build, like every reference main does."""
        f, _ = normalize_freqs(count_freqs(syms, nsyms), 1 << scale_bits)
        return Model(self, fmt, f, scale_bits)

    def count_freqs_device(self, d_syms, nsyms):
        """Histogram of a device tensor (uint8 / int16-viewed-uint16) on the GPU."""
        out = np.zeros(nsyms, dtype=np.uint32)
        _check(_lib.rans_amd_count_freqs(self._h, d_syms.data_ptr(), d_syms.numel(), d_syms.element_size(), nsyms,
                                         out.ctypes.data_as(C.POINTER(C.c_uint32)), _torch_stream()),
               "count_freqs(device)")
        return out

    # -- raw single-stream convenience (host buffers, reference stream layout)
    def encode_host(self, model, syms, n_ways):
        syms = np.ascontiguousarray(syms)
        cap = chunk_bound(model.fmt, max(int(syms.size), 1), n_ways) + 16
        buf = np.zeros(cap, dtype=np.uint8)
        out_len = C.c_uint64(0)
        _check(_lib.rans_amd_encode_host(self._h, model._h, syms.ctypes.data if syms.size else None, syms.size, n_ways,
                                         buf.ctypes.data,
                                         cap, C.byref(out_len)), "encode_host")
        return buf[cap - out_len.value:].copy()

    def decode_host(self, model, stream, n, n_ways, check=True):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.zeros(n, dtype=np.uint8 if model.sym_bytes == 1 else np.uint16)
        rc = _lib.rans_amd_decode_host(self._h, model._h, stream.ctypes.data, stream.size, n, n_ways,
                                       out.ctypes.data)
        if check:
            _check(rc, "decode_host")
            return out
        return out, rc

    # -- bulk, device-resident (torch tensors)
    def encode(self, model, d_syms, n_ways, chunk_syms, d_out=None, sync=True, d_offsets=None, d_lengths=None):
        """Returns (d_container, d_offsets, d_lengths, total_bytes).  d_out / d_offsets / d_lengths may be
        passed in (timed loops: no allocation between launches)."""
        import torch
        n = d_syms.numel()
        nchunks = num_chunks(n, chunk_syms)
        cap = encode_bound(model.fmt, n, n_ways, chunk_syms) + 16
        dev = d_syms.device
        if d_out is None:
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        if d_offsets is None:
            d_offsets = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
        if d_lengths is None:
            d_lengths = torch.zeros(max(nchunks, 1), dtype=torch.int32, device=dev)
        total = C.c_uint64(0)
        _check(_lib.rans_amd_encode(self._h, model._h, d_syms.data_ptr(), n, n_ways, chunk_syms, d_out.data_ptr(),
                                    d_out.numel(), d_offsets.data_ptr(), d_lengths.data_ptr(),
                                    C.byref(total) if sync else None, _torch_stream()), "encode")
        return d_out, d_offsets, d_lengths, (total.value if sync else None)

    def encode_slots(self, model, d_syms, n_ways, chunk_syms, d_out=None, sync=True, d_offsets=None, d_lengths=None):
        """rans_amd_encode_slots: every chunk stays in its slot of slot_bytes() bytes (its stream ends at the slot's end).
        Returns (d_container, d_offsets, d_lengths, total_bytes = n_chunks * slot)."""
        import torch
        n = d_syms.numel()
        nchunks = num_chunks(n, chunk_syms)
        cap = encode_slots_bound(model.fmt, n, n_ways, chunk_syms)
        dev = d_syms.device
        if d_out is None:
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        if d_offsets is None:
            d_offsets = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
        if d_lengths is None:
            d_lengths = torch.zeros(max(nchunks, 1), dtype=torch.int32, device=dev)
        total = C.c_uint64(0)
        _check(_lib.rans_amd_encode_slots(self._h, model._h, d_syms.data_ptr(), n, n_ways, chunk_syms, d_out.data_ptr(),
                                          d_out.numel(), d_offsets.data_ptr(), d_lengths.data_ptr(),
                                          C.byref(total) if sync else None, _torch_stream()), "encode_slots")
        return d_out, d_offsets, d_lengths, (total.value if sync else None)

    def tight_slot_bytes(self, model, n_ways, chunk_syms):
        """rans_amd_tight_slot_bytes: the slot rans_amd_encode_slots_sized is meant to run with -- the model's expected
        chunk stream + 2 % + the flushed states + four standard deviations + 16 bytes, in whole 64-byte lines."""
        return int(_lib.rans_amd_tight_slot_bytes(model._h, n_ways, chunk_syms))

    def encode_sized(self, model, d_syms, n_ways, chunk_syms, slot=None, overflow_chunks=None, d_out=None, sync=True,
                     d_offsets=None, d_lengths=None):
        """rans_amd_encode_slots_sized: slots of `slot` bytes (default tight_slot_bytes()), chunks that do not fit coded
        again into worst-case slots behind them (room for `overflow_chunks` of those; default 1/64 of the chunks + 4).
        Returns (d_container, d_offsets, d_lengths, total_bytes, slot)."""
        import torch
        n = d_syms.numel()
        nchunks = num_chunks(n, chunk_syms)
        if slot is None:
            slot = self.tight_slot_bytes(model, n_ways, chunk_syms)
        if overflow_chunks is None:
            overflow_chunks = nchunks // 64 + 4
        dev = d_syms.device
        if d_out is None:
            d_out = torch.empty(encode_sized_bound(model.fmt, n, n_ways, chunk_syms, slot, overflow_chunks), dtype=torch.uint8,
                                device=dev)
        if d_offsets is None:
            d_offsets = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
        if d_lengths is None:
            d_lengths = torch.zeros(max(nchunks, 1), dtype=torch.int32, device=dev)
        total = C.c_uint64(0)
        _check(_lib.rans_amd_encode_slots_sized(self._h, model._h, d_syms.data_ptr(), n, n_ways, chunk_syms, slot,
                                                d_out.data_ptr(), d_out.numel(), d_offsets.data_ptr(), d_lengths.data_ptr(),
                                                C.byref(total) if sync else None, _torch_stream()), "encode_slots_sized")
        return d_out, d_offsets, d_lengths, (total.value if sync else None), slot

    def compact(self, d_src, src_bytes, d_src_offsets, d_lengths, n_chunks, d_dst=None, sync=True, d_dst_offsets=None):
        """rans_amd_container_compact: -> (d_dst, d_dst_offsets, total_bytes)."""
        import torch
        dev = d_src.device
        if d_dst is None:
            cap = int(((d_lengths[:n_chunks].to(torch.int64) + 15) & ~15).sum().item()) + 16 if n_chunks else 16
            d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)
        if d_dst_offsets is None:
            d_dst_offsets = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
        total = C.c_uint64(0)
        _check(_lib.rans_amd_container_compact(self._h, d_src.data_ptr(), src_bytes, d_src_offsets.data_ptr(),
                                               d_lengths.data_ptr(), n_chunks, d_dst.data_ptr(), d_dst.numel(),
                                               d_dst_offsets.data_ptr(), C.byref(total) if sync else None, _torch_stream()),
               "container_compact")
        return d_dst, d_dst_offsets, (total.value if sync else None)

    def decode(self, model, d_container, container_bytes, d_offsets, d_lengths, n, n_ways, chunk_syms, d_out=None,
               sync=True):
        import torch
        if d_out is None:
            d_out = torch.empty(n, dtype=torch.uint8 if model.sym_bytes == 1 else torch.int16,
                                device=d_container.device)
        bad = C.c_uint64(0)
        _check(_lib.rans_amd_decode(self._h, model._h, d_container.data_ptr(), container_bytes, d_offsets.data_ptr(),
                                    d_lengths.data_ptr(), n, n_ways, chunk_syms, d_out.data_ptr(),
                                    C.byref(bad) if sync else None, _torch_stream()), "decode")
        return d_out

    def probe_placement(self, model, d_containers, container_bytes, d_offsets, d_lengths, n, n_ways, chunk_syms, d_outs,
                        launches=6, sweeps=2):
        """rans_amd_probe_placement: -> (best container index, best output index, matrix[container][output] of mean ms)."""
        conts = (C.c_void_p * len(d_containers))(*[t.data_ptr() for t in d_containers])
        outs = (C.c_void_p * len(d_outs))(*[t.data_ptr() for t in d_outs])
        bi, bj = C.c_uint32(0), C.c_uint32(0)
        ms = (C.c_float * (len(d_containers) * len(d_outs)))()
        _check(_lib.rans_amd_probe_placement(self._h, model._h, conts, len(d_containers), container_bytes, d_offsets.data_ptr(),
                                             d_lengths.data_ptr(), n, n_ways, chunk_syms, outs, len(d_outs), launches, sweeps,
                                             C.byref(bi), C.byref(bj), ms, _torch_stream()), "probe_placement")
        k = len(d_outs)
        return bi.value, bj.value, [[ms[i * k + j] for j in range(k)] for i in range(len(d_containers))]

    # -- one model per chunk (byte format, 256 symbols, scale_bits 8..12)
    def encode_adaptive(self, d_syms, n_ways, chunk_syms, scale_bits, sync=True, fmt=FMT_BYTE):
        """Returns (d_container, d_offsets, d_lengths, d_chunk_freqs, total_bytes).  fmt: FMT_BYTE or FMT_WORD (scale_bits 12)."""
        import torch
        n = d_syms.numel()
        nchunks = num_chunks(n, chunk_syms)
        cap = encode_bound(fmt, n, n_ways, chunk_syms) + 16
        dev = d_syms.device
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_offsets = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
        d_lengths = torch.zeros(max(nchunks, 1), dtype=torch.int32, device=dev)
        d_freqs = torch.zeros(max(nchunks, 1) * 256, dtype=torch.int16, device=dev)
        assert int(_lib.rans_amd_chunk_freqs_bytes(n, chunk_syms)) == nchunks * 512
        total = C.c_uint64(0)
        if fmt == FMT_BYTE:  # (the entry point older callers bind)
            rc = _lib.rans_amd_encode_adaptive(self._h, d_syms.data_ptr(), n, n_ways, chunk_syms, scale_bits, d_out.data_ptr(),
                                               d_out.numel(), d_offsets.data_ptr(), d_lengths.data_ptr(), d_freqs.data_ptr(),
                                               C.byref(total) if sync else None, _torch_stream())
        else:
            rc = _lib.rans_amd_encode_adaptive_fmt(self._h, fmt, d_syms.data_ptr(), n, n_ways, chunk_syms, scale_bits,
                                                   d_out.data_ptr(), d_out.numel(), d_offsets.data_ptr(), d_lengths.data_ptr(),
                                                   d_freqs.data_ptr(), C.byref(total) if sync else None, _torch_stream())
        _check(rc, "encode_adaptive")
        return d_out, d_offsets, d_lengths, d_freqs, (total.value if sync else None)

    def encode_adaptive_sized(self, d_syms, n_ways, chunk_syms, scale_bits, fmt=FMT_BYTE, cap=None, d_out=None, sync=True,
                              d_offsets=None, d_lengths=None, d_freqs=None):
        """rans_amd_encode_adaptive_sized: count + normalise + code in one kernel; chunk c's stream ends where its piece --
        sized from the chunk's own histogram, placed behind the pieces before it -- ends.  cap: bytes of the container
        buffer (default: the bound no input can exceed).  Returns (d_container, d_offsets, d_lengths, d_chunk_freqs, total_bytes)."""
        import torch
        n = d_syms.numel()
        nchunks = num_chunks(n, chunk_syms)
        dev = d_syms.device
        if d_out is None:
            if cap is None:
                cap = int(_lib.rans_amd_encode_adaptive_sized_bound(fmt, n, n_ways, chunk_syms))
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        if d_offsets is None:
            d_offsets = torch.zeros(nchunks + 1, dtype=torch.int64, device=dev)
        if d_lengths is None:
            d_lengths = torch.zeros(max(nchunks, 1), dtype=torch.int32, device=dev)
        if d_freqs is None:
            d_freqs = torch.zeros(max(nchunks, 1) * 256, dtype=torch.int16, device=dev)
        total = C.c_uint64(0)
        _check(_lib.rans_amd_encode_adaptive_sized(self._h, fmt, d_syms.data_ptr(), n, n_ways, chunk_syms, scale_bits,
                                                   d_out.data_ptr(), d_out.numel(), d_offsets.data_ptr(), d_lengths.data_ptr(),
                                                   d_freqs.data_ptr(), C.byref(total) if sync else None, _torch_stream()),
               "encode_adaptive_sized")
        return d_out, d_offsets, d_lengths, d_freqs, (total.value if sync else None)

    def decode_adaptive(self, d_container, container_bytes, d_offsets, d_lengths, d_freqs, n, n_ways, chunk_syms,
                        scale_bits, d_out=None, sync=True, fmt=FMT_BYTE):
        import torch
        if d_out is None:
            d_out = torch.empty(n, dtype=torch.uint8, device=d_container.device)
        bad = C.c_uint64(0)
        if fmt == FMT_BYTE:
            rc = _lib.rans_amd_decode_adaptive(self._h, d_container.data_ptr(), container_bytes, d_offsets.data_ptr(),
                                               d_lengths.data_ptr(), d_freqs.data_ptr(), n, n_ways, chunk_syms, scale_bits,
                                               d_out.data_ptr(), C.byref(bad) if sync else None, _torch_stream())
        else:
            rc = _lib.rans_amd_decode_adaptive_fmt(self._h, fmt, d_container.data_ptr(), container_bytes, d_offsets.data_ptr(),
                                                   d_lengths.data_ptr(), d_freqs.data_ptr(), n, n_ways, chunk_syms, scale_bits,
                                                   d_out.data_ptr(), C.byref(bad) if sync else None, _torch_stream())
        _check(rc, "decode_adaptive")
        return d_out

    def encode_status(self):
        """rans_amd_encode_status: raises what the last asynchronous (sync=False) or graph-replayed encode ended with."""
        _check(_lib.rans_amd_encode_status(self._h, _torch_stream()), "encode_status")

    def decode_errors(self):
        bad = C.c_uint64(0)
        rc = _lib.rans_amd_decode_errors(self._h, C.byref(bad), _torch_stream())
        if rc not in (OK, E_CORRUPT):
            _check(rc, "decode_errors")
        return bad.value


class Model:
    def __init__(self, ctx, fmt, norm_freqs, scale_bits):
        f = np.ascontiguousarray(norm_freqs, dtype=np.uint32)
        self._h = C.c_void_p()
        self._ctx = ctx  # keep the context alive
        _check(_lib.rans_amd_model_create(ctx._h if ctx is not None else None, fmt, f.ctypes.data_as(C.POINTER(C.c_uint32)), f.size, scale_bits,
                                          C.byref(self._h)), "model_create")
        self.fmt = fmt
        self.freqs = f
        self.nsyms = int(f.size)
        self.scale_bits = int(scale_bits)
        self.sym_bytes = int(_lib.rans_amd_model_sym_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.rans_amd_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self, which, dtype=np.uint8):
        size = C.c_size_t(0)
        _check(_lib.rans_amd_model_table(self._h, which, None, 0, C.byref(size)), "model_table(size)")
        buf = np.zeros(size.value, dtype=np.uint8)
        _check(_lib.rans_amd_model_table(self._h, which, buf.ctypes.data, buf.size, C.byref(size)), "model_table")
        return buf.view(dtype)


# ---- container file format (include/ryg_rans_amd.h "container file format") -------------

def offsets_from_lengths(lengths):
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    offs = np.zeros(lengths.size + 1, dtype=np.uint64)
    _check(_lib.rans_amd_offsets_from_lengths(lengths.ctypes.data_as(C.POINTER(C.c_uint32)), lengths.size,
                                              offs.ctypes.data_as(C.POINTER(C.c_uint64))), "offsets_from_lengths")
    return offs


def container_slice(offsets, lengths, lo, hi):
    """rans_amd_container_slice: chunk range [lo, hi) of an index (host arrays) -> (byte_begin, byte_end, rebased offsets
    [hi - lo + 1]): the bytes a rank must hold and where its chunks lie in them."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    out = np.zeros(max(1, hi - lo + 1), dtype=np.uint64)
    b, e = C.c_uint64(0), C.c_uint64(0)
    _check(_lib.rans_amd_container_slice(offsets.ctypes.data_as(C.POINTER(C.c_uint64)), lengths.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         lengths.size, lo, hi, C.byref(b), C.byref(e), out.ctypes.data_as(C.POINTER(C.c_uint64))),
           "container_slice")
    return b.value, e.value, out


def pack_container(fmt, norm_freqs, scale_bits, n_symbols, n_ways, chunk_syms, lengths, payload):
    """Serialise model + index + payload (host numpy arrays) into one self-describing uint8 array."""
    f = np.ascontiguousarray(norm_freqs, dtype=np.uint32)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    info = ContainerInfo(fmt, scale_bits, f.size, n_ways, chunk_syms, 1 if f.size <= 256 else 2, n_symbols,
                         num_chunks(n_symbols, chunk_syms), payload.size)
    if lengths.size != info.n_chunks:
        raise RansAmdError(E_ARG, "container_pack", "len(lengths) != number of chunks")
    total = int(_lib.rans_amd_container_bytes(C.byref(info)))
    if total == 0:
        raise RansAmdError(E_ARG, "container_bytes", "inconsistent container description")
    out = np.zeros(total, dtype=np.uint8)
    wrote = C.c_uint64(0)
    _check(_lib.rans_amd_container_pack(C.byref(info), f.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        lengths.ctypes.data_as(C.POINTER(C.c_uint32)), payload.ctypes.data,
                                        out.ctypes.data, out.size, C.byref(wrote)), "container_pack")
    return out[:wrote.value]


def pack_container_indexed(fmt, norm_freqs, scale_bits, n_symbols, n_ways, chunk_syms, offsets, lengths, payload, chunk_freqs=None):
    """rans_amd_container_pack_indexed[_adaptive]: the file of a container in ANY layout -- `payload` is the host copy of the
    device container, chunk c its lengths[c] bytes at offsets[c].  chunk_freqs (u16[n_chunks * 256]): the version-2 file."""
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    nchunks = num_chunks(n_symbols, chunk_syms)
    if lengths.size != nchunks or offsets.size < nchunks:
        raise RansAmdError(E_ARG, "container_pack_indexed", "index does not match the number of chunks")
    packed = int(_lib.rans_amd_packed_payload_bytes(lengths.ctypes.data_as(C.POINTER(C.c_uint32)), nchunks))
    wrote = C.c_uint64(0)
    if chunk_freqs is None:
        f = np.ascontiguousarray(norm_freqs, dtype=np.uint32)
        info = ContainerInfo(fmt, scale_bits, f.size, n_ways, chunk_syms, 1 if f.size <= 256 else 2, n_symbols, nchunks, packed)
        total = int(_lib.rans_amd_container_bytes(C.byref(info)))
        if total == 0:
            raise RansAmdError(E_ARG, "container_bytes", "inconsistent container description")
        out = np.zeros(total, dtype=np.uint8)
        _check(_lib.rans_amd_container_pack_indexed(C.byref(info), f.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                    lengths.ctypes.data_as(C.POINTER(C.c_uint32)), payload.ctypes.data, payload.size,
                                                    out.ctypes.data, out.size, C.byref(wrote)), "container_pack_indexed")
    else:
        cf = np.ascontiguousarray(chunk_freqs).view(np.uint16).reshape(-1)
        info = ContainerInfo(fmt, scale_bits, 256, n_ways, chunk_syms, 1, n_symbols, nchunks, packed)
        total = int(_lib.rans_amd_container_bytes_adaptive(C.byref(info)))
        if total == 0 or cf.size != nchunks * 256:
            raise RansAmdError(E_ARG, "container_bytes_adaptive", "inconsistent container description")
        out = np.zeros(total, dtype=np.uint8)
        _check(_lib.rans_amd_container_pack_indexed_adaptive(C.byref(info), cf.ctypes.data, offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                             lengths.ctypes.data_as(C.POINTER(C.c_uint32)), payload.ctypes.data,
                                                             payload.size, out.ctypes.data, out.size, C.byref(wrote)),
               "container_pack_indexed_adaptive")
    return out[:wrote.value]


def parse_container(blob):
    """-> (ContainerInfo, freqs, lengths, payload) as numpy views into `blob` (validated)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    info = ContainerInfo()
    pf, pl, pp = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.c_void_p()
    _check(_lib.rans_amd_container_parse(blob.ctypes.data, blob.size, C.byref(info), C.byref(pf), C.byref(pl),
                                         C.byref(pp)), "container_parse")
    base = blob.ctypes.data
    f_off = C.addressof(pf.contents) - base
    l_off = f_off + 4 * info.nsyms
    p_off = pp.value - base
    freqs = blob[f_off:f_off + 4 * info.nsyms].view(np.uint32)
    lengths = blob[l_off:l_off + 4 * info.n_chunks].view(np.uint32)
    payload = blob[p_off:p_off + info.payload_bytes]
    return info, freqs, lengths, payload


def pack_container_adaptive(scale_bits, n_symbols, n_ways, chunk_syms, chunk_freqs, lengths, payload, fmt=FMT_BYTE):
    """Version-2 container (one model per chunk): chunk_freqs is u16[n_chunks * 256]; fmt FMT_BYTE or FMT_WORD (12 bits)."""
    cf = np.ascontiguousarray(chunk_freqs, dtype=np.uint16)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    info = ContainerInfo(fmt, scale_bits, 256, n_ways, chunk_syms, 1, n_symbols, num_chunks(n_symbols, chunk_syms),
                         payload.size)
    if lengths.size != info.n_chunks or cf.size != info.n_chunks * 256:
        raise RansAmdError(E_ARG, "container_pack_adaptive", "lengths / chunk_freqs do not match the number of chunks")
    total = int(_lib.rans_amd_container_bytes_adaptive(C.byref(info)))
    if total == 0:
        raise RansAmdError(E_ARG, "container_bytes_adaptive", "inconsistent container description")
    out = np.zeros(total, dtype=np.uint8)
    wrote = C.c_uint64(0)
    _check(_lib.rans_amd_container_pack_adaptive(C.byref(info), cf.ctypes.data, lengths.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 payload.ctypes.data, out.ctypes.data, out.size, C.byref(wrote)),
           "container_pack_adaptive")
    return out[:wrote.value]


def parse_container_adaptive(blob):
    """-> (ContainerInfo, chunk_freqs u16[n_chunks, 256], lengths, payload) as numpy views into `blob`."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    info = ContainerInfo()
    pf, pl, pp = C.c_void_p(), C.POINTER(C.c_uint32)(), C.c_void_p()
    _check(_lib.rans_amd_container_parse_adaptive(blob.ctypes.data, blob.size, C.byref(info), C.byref(pf), C.byref(pl),
                                                  C.byref(pp)), "container_parse_adaptive")
    base = blob.ctypes.data
    f_off = pf.value - base
    l_off = f_off + 512 * info.n_chunks
    p_off = pp.value - base
    freqs = blob[f_off:f_off + 512 * info.n_chunks].view(np.uint16).reshape(-1, 256)
    lengths = blob[l_off:l_off + 4 * info.n_chunks].view(np.uint32)
    payload = blob[p_off:p_off + info.payload_bytes]
    return info, freqs, lengths, payload


def _torch_stream():
    import torch
    if not torch.cuda.is_available():
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
