"""Multi-GPU sharding of the decode path (one process per GPU, torch.distributed).

The path partitions trivially: chunks are independent streams, so a container is
split into contiguous chunk ranges (or, for the headline benchmark, every rank
owns a whole shard) and NO payload crosses xGMI.  The only communication is one
all-gather of a small per-rank record, which doubles as the end barrier.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from dataclasses import dataclass


def chunk_range(nchunks, world, rank):
    """Contiguous chunk slice [lo, hi) owned by `rank`: shard g = chunks [g*C/G, (g+1)*C/G)."""
    lo = (nchunks * rank) // world
    hi = (nchunks * (rank + 1)) // world
    return lo, hi


def symbol_range(n, chunk_syms, world, rank):
    """Symbol slice [first, last) covered by rank's chunk range."""
    nchunks = (n + chunk_syms - 1) // chunk_syms
    lo, hi = chunk_range(nchunks, world, rank)
    return min(n, lo * chunk_syms), min(n, hi * chunk_syms)


@dataclass
class ShardRecord:
    elapsed_s: float      # wall time of the K timed steps on this rank
    symbols: float        # symbols decoded per step on this rank
    stream_bytes: float   # compressed bytes read per step
    kernel_ms: float      # average decode-kernel duration (HIP events)
    ok: float             # 1.0 when the round trip was bit exact and no chunk was flagged
    oracle_chunks: float = 0.0  # chunks of this rank's shard compared byte for byte with the CPU oracle (bench.py, N > 1)

    def to_list(self):
        return [self.elapsed_s, self.symbols, self.stream_bytes, self.kernel_ms, self.ok, self.oracle_chunks]


def gather_records(rec, device="cpu", force=False):
    """all_gather of one ShardRecord per rank (6 doubles = 48 bytes each).  force: run the collective in a one-rank
    group as well (bench.py --force-dist: the RCCL call path of the N-GPU run on one GPU)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(rec.to_list(), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        rows = torch.stack(parts).cpu().tolist()
    else:
        rows = [t.cpu().tolist()]
    return [ShardRecord(*r) for r in rows]


def aggregate(records, steps):
    """Whole-job figures: time = MAX over ranks, work = SUM over ranks."""
    max_elapsed = max(r.elapsed_s for r in records)
    total_syms = sum(r.symbols for r in records)
    return {
        "ms_per_step": max_elapsed / steps * 1e3,
        "symbols_per_s": total_syms / (max_elapsed / steps),
        "all_ok": all(r.ok == 1.0 for r in records),
        "n_ranks": len(records),
    }
