"""Buffer placement probe (bench.py setup, tools/; not part of the C ABI).

MI355X device memory is not uniform for a kernel that streams one buffer in while it streams another out: allocations fall
into (at least) two classes, and the SAME decode over the SAME bytes is 4-6 % faster when its container and its output
lie in DIFFERENT classes than when they share one (profiles/r04_allocation.md: a 12 x 12 container x output matrix shows
the XOR pattern; the counters that move are the memory-side read latency -- TCC_EA0_RDREQ_LEVEL, TCP_TCC_READ_REQ_LATENCY
-- not address translation: UTCL1 misses are 1e-4 of the requests and anti-correlated).  Which class an allocation gets
is the driver's choice (its position in the physical address space); hipMalloc has no knob for it.  What a caller that
owns its buffers for a while CAN do is what this module does: allocate a few candidates, time the real call on each
pair for a few launches, keep the fastest pair, free the rest.

Round 5 mapped the classes over 240 GiB of allocation order (tools/class_map.py, profiles/r05_class_map.md): they are WINDOWS
of 30-60 GiB, a container collides (+5-6 %) with the outputs of its own window and of one or two others -- a third of all
pairs -- and buffers allocated one after the other lie in the same window, so the pair a fresh process gets first is more
often slow than not, and two dozen candidates allocated in a row can all be.  Candidates therefore have to be FAR apart in
allocation order: `spaced()` below holds spacer allocations between them while they are allocated.
"""


def spaced(torch, make, k, device, first=None, stride_bytes=24 << 30, reserve_bytes=48 << 30):
    """k buffers -- `first` (if given) and make() for the rest -- about `stride_bytes` apart in allocation order.  Returns
    (buffers, spacers): keep `spacers` alive until every candidate of the probe is allocated, then drop them (they are never
    touched).  The stride shrinks so that the spacers leave `reserve_bytes` of the device's free memory alone."""
    oom = getattr(torch, "OutOfMemoryError", None) or torch.cuda.OutOfMemoryError  # (older PyTorch has only the cuda one)
    bufs = [first if first is not None else make()]
    size = bufs[0].numel() * bufs[0].element_size()
    free = torch.cuda.mem_get_info(device)[0]
    gap = min(stride_bytes, max(0, (free - reserve_bytes) // max(k - 1, 1))) - size
    spacers = []
    for _ in range(k - 1):
        if gap >= (1 << 30):
            try:
                spacers.append(torch.empty(gap, dtype=torch.uint8, device=device))
            except oom:  # (somebody else took the memory meanwhile: the rest in a row)
                gap = 0
        try:
            bufs.append(make())
        except oom:  # (the candidates matter, the spacing does not: from here on in a row -- the caller sees an empty
            spacers = []  #  `spacers` and reports a stride of 0)
            gap = 0
            torch.cuda.empty_cache()
            bufs.append(make())
    return bufs, spacers


def time_launches(torch, fn, launches=6, warm=2):
    """Mean milliseconds of `launches` back-to-back calls of fn (HIP events on the current stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / launches


def choose_pair(torch, run, firsts, seconds, launches=6, sweeps=2):
    """run(first, second) launches the call on a pair of buffers.  Times every pair `sweeps` times (interleaved, so that a
    drift of the clocks does not favour a pair) and returns (i, j, matrix) with matrix[i][j] = mean ms, (i, j) the fastest."""
    acc = [[0.0] * len(seconds) for _ in firsts]
    for _ in range(sweeps):
        for i, a in enumerate(firsts):
            for j, b in enumerate(seconds):
                acc[i][j] += time_launches(torch, lambda: run(a, b), launches)
    matrix = [[v / sweeps for v in row] for row in acc]
    best = min(((matrix[i][j], i, j) for i in range(len(firsts)) for j in range(len(seconds))))
    return best[1], best[2], matrix


def choose_one(torch, run, candidates, launches=6, sweeps=2):
    """The same for one buffer that varies (the other side fixed inside `run`): returns (index, [ms per candidate])."""
    acc = [0.0] * len(candidates)
    for _ in range(sweeps):
        for i, c in enumerate(candidates):
            acc[i] += time_launches(torch, lambda: run(c), launches)
    ms = [v / sweeps for v in acc]
    return min(range(len(ms)), key=ms.__getitem__), ms
