"""The N>1 path on CPU: two processes over gloo exercise the sharding plan and the
record all-gather/aggregation bench.py uses over RCCL (no data-path collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ryg_rans_amd.sharding import ShardRecord, aggregate, chunk_range, gather_records, symbol_range


def test_chunk_ranges_partition_everything():
    for nchunks in (0, 1, 7, 8, 9, 32768, 32769):
        for world in (1, 2, 3, 4, 8):
            spans = [chunk_range(nchunks, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nchunks
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert symbol_range(100001, 4096, 2, 0) == (0, 12 * 4096)
    assert symbol_range(100001, 4096, 2, 1) == (12 * 4096, 100001)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r "decodes" its own shard; rank 1 is slower
    rec = ShardRecord(elapsed_s=0.010 * (rank + 1), symbols=float(1 << 20), stream_bytes=800000.0 + rank,
                      kernel_ms=0.5 + rank, ok=1.0)
    dist.barrier()
    records = gather_records(rec)
    dist.barrier()
    if rank == 0:
        q.put([r.to_list() for r in records])
    dist.destroy_process_group()


def test_two_rank_gather_and_aggregate():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    records = [ShardRecord(*r) for r in rows]
    assert [r.stream_bytes for r in records] == [800000.0, 800001.0]
    agg = aggregate(records, steps=10)
    assert agg["n_ranks"] == 2 and agg["all_ok"]
    assert agg["ms_per_step"] == pytest.approx(2.0)            # MAX over ranks: 20 ms / 10 steps
    assert agg["symbols_per_s"] == pytest.approx(2 * (1 << 20) / 0.002)  # SUM of work / max time
    bad = records + [ShardRecord(0.01, 1.0, 1.0, 1.0, 0.0)]
    assert not aggregate(bad, 10)["all_ok"]
