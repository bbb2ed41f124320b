"""Buffer placement probe (bench.py setup, tools/; not part of the C ABI).

MI355X device memory is not uniform for a kernel that streams one buffer in while it streams another out: allocations fall
into (at least) two classes, and the SAME decode over the SAME bytes is 4-6 % faster when its container and its output
lie in DIFFERENT classes than when they share one (profiles/r04_allocation.md: a 12 x 12 container x output matrix shows
the XOR pattern; the counters that move are the memory-side read latency -- TCC_EA0_RDREQ_LEVEL, TCP_TCC_READ_REQ_LATENCY
-- not address translation: UTCL1 misses are 1e-4 of the requests and anti-correlated).  Which class an allocation gets
is the driver's choice (its position in the physical address space); hipMalloc has no knob for it.  What a caller that
owns its buffers for a while CAN do is what this module does: allocate a few candidates, time the real call on each
pair for a few launches, keep the fastest pair, free the rest.
"""


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
