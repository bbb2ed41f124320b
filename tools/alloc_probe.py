"""Where the headline decode's +-6 % comes from: the same kernel over the same bytes with the container and the output in
different ALLOCATIONS and at different places inside ONE allocation (profiles/r04_allocation.md).

    python tools/alloc_probe.py time [--rounds 3]         # timing table of every placement, interleaved rounds
    python tools/alloc_probe.py pmc  [--launches 6]       # the same placements, `launches` decodes each, in a fixed order
                                                          # (run under rocprofv3 --pmc ...; prints the dispatch order)

Placements:
  torch<i>    container and output in their own torch.empty allocations (what bench.py did until round 3)
  arena+<s>   ONE 6 GiB allocation: container at its start, output 2 GiB further plus a shift of s bytes
  hip<i>      raw hipMalloc pairs (no caching allocator in between)
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from bench import gen_zipf  # noqa: E402


def hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise OSError("libamdhip64 not found")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["time", "pmc", "matrix"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--shifts", default="0,256,4096,65536,1048576,2097152,33554432,34603008,1073741824")
    ap.add_argument("--torch-pairs", type=int, default=6)
    ap.add_argument("--hip-pairs", type=int, default=4)
    ap.add_argument("--ext-flags", default="", help="comma-separated hipExtMallocWithFlags flags to try, e.g. 0x4,0x1,0x3")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    n = 1 << a.log2n
    d = gen_zipf(torch, n, 256, 1.0, 1, dev)
    f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 4096)
    m = ctx.model(R.FMT_WORD, f, 12)
    cont, offs, lens, total = ctx.encode(m, d, 64, a.chunk)
    nb = (total + 4095) & ~4095
    lib = R.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def decode(cptr, optr):
        rc = lib.rans_amd_decode(ctx._h, m._h, C.c_void_p(cptr), total, C.c_void_p(offs.data_ptr()), C.c_void_p(lens.data_ptr()), n, 64,
                                 a.chunk, C.c_void_p(optr), None, stream)
        assert rc == 0, rc

    places, keep = [], []
    for i in range(a.torch_pairs):
        c = torch.empty(nb, dtype=torch.uint8, device=dev)
        c.copy_(cont[:nb])
        o = torch.empty(n, dtype=torch.uint8, device=dev)
        keep += [c, o]
        places.append(("torch%d" % i, c.data_ptr(), o.data_ptr()))
    arena = torch.empty(6 << 30, dtype=torch.uint8, device=dev)
    keep.append(arena)
    arena[:nb].copy_(cont[:nb])
    for s in [int(v) for v in a.shifts.split(",")]:
        places.append(("arena+%d" % s, arena.data_ptr(), arena.data_ptr() + (2 << 30) + s))
    h = hip()
    h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for i in range(a.hip_pairs):
        pc, po = C.c_void_p(), C.c_void_p()
        assert h.hipMalloc(C.byref(pc), nb + 4096) == 0 and h.hipMalloc(C.byref(po), n + 4096) == 0
        assert h.hipMemcpy(pc, C.c_void_p(cont.data_ptr()), nb, 3) == 0  # hipMemcpyDeviceToDevice
        places.append(("hip%d" % i, pc.value, po.value))
    if a.ext_flags:
        # hipExtMallocWithFlags: 0x4 contiguous, 0x1 fine-grained, 0x3 uncached (hip_runtime_api.h:879-891): does a flag
        # choose the memory class?  (one pair per flag, and mixed pairs with plain hipMalloc)
        h.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        for fl in [int(v, 0) for v in a.ext_flags.split(",")]:
            pc, po = C.c_void_p(), C.c_void_p()
            if h.hipExtMallocWithFlags(C.byref(pc), nb + 4096, fl) != 0 or h.hipExtMallocWithFlags(C.byref(po), n + 4096, fl) != 0:
                print("  hipExtMallocWithFlags(%#x) refused" % fl, flush=True)
                continue
            assert h.hipMemcpy(pc, C.c_void_p(cont.data_ptr()), nb, 3) == 0
            places.append(("ext%#x" % fl, pc.value, po.value))
            places.append(("ext%#x-out" % fl, places[0][1], po.value))   # torch0's container, this output
            places.append(("ext%#x-cont" % fl, pc.value, places[0][2]))  # this container, torch0's output
    print("placements:", flush=True)
    for name, c, o in places:
        print("  %-16s cont %#x  out %#x  (out - cont) mod 2 MiB = %d KiB, mod 1 GiB = %d MiB" % (
            name, c, o, ((o - c) % (2 << 20)) >> 10, ((o - c) % (1 << 30)) >> 20), flush=True)

    # correctness of every placement once
    chk = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, c, o in places:
        decode(c, o)
        torch.cuda.synchronize()
        assert h.hipMemcpy(C.c_void_p(chk.data_ptr()), C.c_void_p(o), n, 3) == 0
        assert torch.equal(chk, d), name
    assert ctx.decode_errors() == 0

    def timed(c, o, launches):
        for _ in range(3):
            decode(c, o)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for e0, e1 in ev:
            e0.record()
            decode(c, o)
            e1.record()
        torch.cuda.synchronize()
        ms = [e0.elapsed_time(e1) for e0, e1 in ev]
        return sum(ms) / len(ms)

    # settle the clocks
    for _ in range(200):
        decode(places[0][1], places[0][2])
    torch.cuda.synchronize()
    if a.mode == "matrix":
        # containers x outputs of the torch / hip placements: is the speed a property of the container's allocation, of
        # the output's, or of the pair?
        names = [p[0] for p in places if not p[0].startswith("arena")]
        cs = [p[1] for p in places if not p[0].startswith("arena")]
        os_ = [p[2] for p in places if not p[0].startswith("arena")]
        print("rows = container of placement, columns = output of placement; ms per decode (8 launches), two sweeps")
        print("%-8s " % "" + " ".join("%-7s" % nm for nm in names))
        for sweep in range(2):
            for i, c in enumerate(cs):
                print("%-8s " % names[i] + " ".join("%.4f " % timed(c, o, 8) for o in os_), flush=True)
            print()
        return
    if a.mode == "time":
        rows = {name: [] for name, _, _ in places}
        for r in range(a.rounds):
            for name, c, o in places:
                rows[name].append(timed(c, o, 12))
        for name, _, _ in places:
            v = rows[name]
            print("%-16s %s   mean %.4f ms  frac %.4f" % (name, " ".join("%.4f" % x for x in v), sum(v) / len(v),
                                                         (n + total) / (sum(v) / len(v)) / 1e6 / 8000.0), flush=True)
    else:
        print("dispatch order: %d decodes per placement, in the order of the list above (k_decode_word64 dispatches %d.. of the "
              "process; the first %d are setup)" % (a.launches, 0, 0), flush=True)
        torch.cuda.synchronize()
        print("PMC-SEQUENCE-BEGIN", flush=True)
        for name, c, o in places:
            ms = timed(c, o, a.launches)  # (3 untimed + `launches` timed dispatches per placement)
            print("PMC-PLACEMENT %s launches %d ms_under_profiler %.4f" % (name, 3 + a.launches, ms), flush=True)
        print("PMC-SEQUENCE-END", flush=True)


if __name__ == "__main__":
    main()
