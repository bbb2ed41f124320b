"""GPU parity at the BASELINE sizes (run with -m gpu on an MI355X).

The small-size parity tests compare every byte with the oracle; here the inputs are the full configurations
of BASELINE.json -- C3: 1 GiB, word format, 64-way; C2: 256 MiB, rans64, 2-way; C4: 512 Mi u16 symbols, alias
tables over 4096 symbols, 64-way -- and what is compared with the CPU oracle is the WHOLE container the GPU encoder
produced: every chunk is re-encoded by the oracle (threaded over the host cores, oracle/rans_oracle.c
orc_compare_chunks) and must have the oracle's length and bytes, and the whole index must be the prefix sum of its
16-byte aligned lengths (bench.oracle_check_chunks, the same check bench.py reports as `oracle_chunks_checked`) -- the
reference compares all bytes too (main_simd.cpp:340-343).  The decoders are then fed a container the ORACLE made from
the same symbols (Oracle.encode_chunked_mt), so that nothing they read was written by the GPU encoder.  A symmetric
encoder/decoder bug, in any chunk, at any size, cannot pass this.

book1 goes through the GPU as well: the committed 64-way word stream, written by the
unmodified reference, is decoded on the GPU to the corpus (sha256 pinned), and the GPU encoder reproduces all
thirteen appendix B streams.
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD

HERE = os.path.dirname(os.path.abspath(__file__))
FMT = {"byte": FMT_BYTE, "word": FMT_WORD, "r64": FMT_R64, "alias": FMT_ALIAS}


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    import ryg_rans_amd as R
    ctx = R.Context(0)
    yield R, ctx, torch
    ctx.close()


def test_bench_generator_equals_oracle_on_gpu(gpu, oracle):
    """bench.gen_zipf on the GPU (splitmix64 in int64 tensors, float64 inverse CDF) makes the very bytes
    Oracle.gen_zipf makes on the CPU, so the bench buffers can be regenerated and checked anywhere."""
    R, ctx, torch = gpu
    import bench
    for K, seed, n in ((256, 1, (1 << 20) + 3), (4096, 1, 300001), (256, 5, 70001)):
        got = bench.gen_zipf(torch, n, K, 1.0, seed, "cuda").cpu().numpy()
        if got.dtype == np.int16:
            got = got.view(np.uint16)
        assert np.array_equal(got, oracle.gen_zipf(n, K=K, s=1.0, seed=seed)), (K, seed)
    # the generator is counter based: the head of a 1 GiB buffer equals a short run from the same seed
    big = bench.gen_zipf(torch, 1 << 30, 256, 1.0, 1, "cuda")
    assert np.array_equal(big[:1 << 16].cpu().numpy(), oracle.gen_zipf(1 << 16, K=256, s=1.0, seed=1))


@pytest.mark.parametrize("name,fmt,sb,K,ways,chunk,log2n", [
    ("C3", FMT_WORD, 12, 256, 64, 32768, 30),
    ("C3-16k-chunks", FMT_WORD, 12, 256, 64, 16384, 30),
    ("C2", FMT_R64, 14, 256, 2, 512, 28),
    ("C4", FMT_ALIAS, 16, 4096, 64, 32768, 29),
    ("byte-64", FMT_BYTE, 14, 256, 64, 32768, 30),
    ("C4-16k-chunks", FMT_ALIAS, 16, 4096, 64, 16384, 29),
    ("byte-64-16k-chunks", FMT_BYTE, 14, 256, 64, 16384, 30),
    # the wider interleaves and the 64-way rans64 coder at a quarter of the size (several rounds of every persistent grid)
    ("word-256-way", FMT_WORD, 12, 256, 256, 32768, 28),
    ("word-128-way", FMT_WORD, 12, 256, 128, 16384, 28),
    ("r64-64-way", FMT_R64, 14, 256, 64, 16384, 28),
    ("alias256-64-way", FMT_ALIAS, 16, 256, 64, 16384, 28),
])
def test_full_size_every_chunk_equals_oracle(gpu, oracle, name, fmt, sb, K, ways, chunk, log2n):
    R, ctx, torch = gpu
    import bench
    n = 1 << log2n
    d_syms = bench.gen_zipf(torch, n, K, 1.0, 1, "cuda")
    counts = ctx.count_freqs_device(d_syms, K)
    assert int(counts.sum()) == n
    freqs, _ = R.normalize_freqs(counts, 1 << sb)
    gm = ctx.model(fmt, freqs, sb)
    cont, offs, lens, total = ctx.encode(gm, d_syms, ways, chunk)
    art = {"fmt": fmt, "sb": sb, "K": K, "ways": ways, "chunk": chunk, "n": n, "freqs": freqs, "d_syms": d_syms,
           "cont": cont, "offs": offs, "lens": lens, "total": total}
    # 1. EVERY chunk of the GPU encoder's container == the oracle's stream for it; the index == prefix sums
    assert bench.oracle_check_chunks(art) == (n + chunk - 1) // chunk
    out = ctx.decode(gm, cont, total, offs, lens, n, ways, chunk)
    assert torch.equal(out, d_syms)
    if name.startswith("C3"):  # (both chunk sizes: 32 Ki is bench.py's headline, 16 Ki its `configs` entries)
        assert ctx.last_decode_kernel() == "k_decode_word64"
    if name.startswith("C4"):
        assert ctx.last_decode_kernel() == "k_decode_dual<alias>"
    del out
    # 2. the decoder on a container made by the ORACLE alone (host, threaded), not by the GPU encoder
    assert bench.decode_oracle_container(torch, R, ctx, gm, art, "cuda")
    # 3. the encoder with its scratch ring (RANS_AMD_OPT_ENC_SCRATCH_RING; wave-per-chunk encoders): the same container
    if ways == 64 and fmt != FMT_ALIAS:
        ctx2 = R.Context(0)
        ctx2.set_option(R.OPT_ENC_SCRATCH_RING, 1)
        gm2 = ctx2.model(fmt, freqs, sb)
        cont_r, offs_r, lens_r, total_r = ctx2.encode(gm2, d_syms, ways, chunk)
        assert total_r == total and torch.equal(offs_r, offs) and torch.equal(lens_r, lens)
        art_r = dict(art, cont=cont_r)
        assert bench.oracle_check_chunks(art_r) == (n + chunk - 1) // chunk
        del cont_r, gm2
        ctx2.close()
    # 4. the slot layout (rans_amd_encode_slots: every chunk written once, where it was coded): every chunk == the
    #    oracle's stream again, index == (c + 1) * slot - length, the slot container decodes as it is, and its compaction
    #    is the container of step 1
    s_cont, s_offs, s_lens, s_total = ctx.encode_slots(gm, d_syms, ways, chunk)
    assert ctx.last_encode_placement() == 2 and torch.equal(s_lens, lens)
    art_s = dict(art, cont=s_cont, offs=s_offs, lens=s_lens, total=s_total, slot=R.slot_bytes(fmt, n, ways, chunk))
    assert bench.oracle_check_chunks(art_s) == (n + chunk - 1) // chunk
    out = ctx.decode(gm, s_cont, s_total, s_offs, s_lens, n, ways, chunk)
    assert torch.equal(out, d_syms)
    del out
    c_cont, c_offs, c_total = ctx.compact(s_cont, s_total, s_offs, s_lens, lens.numel())
    assert c_total == total and torch.equal(c_offs, offs)
    assert bench.oracle_check_chunks(dict(art, cont=c_cont, offs=c_offs)) == (n + chunk - 1) // chunk
    del s_cont, c_cont
    # 4b. SIZED slots (rans_amd_encode_slots_sized, slot = rans_amd_tight_slot_bytes()): every chunk == the oracle's stream
    #     wherever it lies (its slot or the overflow region), the index follows the layout's rule, the container is about
    #     the compact one's size and decodes as it is.  Then the same with slots of HALF that size: every chunk overflows
    #     into the region behind them (the redo launch codes the whole shard) and still equals the oracle's.
    nchunks = (n + chunk - 1) // chunk
    worst = R.slot_bytes(fmt, n, ways, chunk)
    t_cont, t_offs, t_lens, t_total, t_slot = ctx.encode_sized(gm, d_syms, ways, chunk)
    assert torch.equal(t_lens, lens) and t_slot < worst
    assert t_total <= 1.35 * total, (t_total, total)  # (config 2's 512-symbol chunks in whole 64-byte lines: 1.23 x)
    assert bench.oracle_check_chunks(dict(art, cont=t_cont, offs=t_offs, lens=t_lens, total=t_total, slot=t_slot, worst=worst)) == nchunks
    out = ctx.decode(gm, t_cont, t_total, t_offs, t_lens, n, ways, chunk)
    assert torch.equal(out, d_syms)
    del out, t_cont
    half = max(64, (t_slot // 2) & ~63)
    h_cont, h_offs, h_lens, h_total, _ = ctx.encode_sized(gm, d_syms, ways, chunk, slot=half, overflow_chunks=nchunks)
    assert torch.equal(h_lens, lens) and h_total >= nchunks * half + (nchunks - 1) * worst  # (a ragged last chunk may fit)
    assert bench.oracle_check_chunks(dict(art, cont=h_cont, offs=h_offs, lens=h_lens, total=h_total, slot=half, worst=worst)) == nchunks
    out = ctx.decode(gm, h_cont, h_total, h_offs, h_lens, n, ways, chunk)
    assert torch.equal(out, d_syms)
    del out, h_cont
    # 5. a corrupted chunk is flagged (or at least does not decode to the input)
    bad = cont.clone()
    bad[int(offs[0].item()) + int(lens[0].item()) // 2] ^= 0x10
    out2 = torch.empty_like(d_syms)
    ctx.decode(gm, bad, total, offs, lens, n, ways, chunk, d_out=out2, sync=False)
    assert ctx.decode_errors() >= 1 or not torch.equal(out2, d_syms)


@pytest.mark.parametrize("fmt,sb,chunk", [(FMT_WORD, 12, 16384), (FMT_BYTE, 12, 16384), (FMT_WORD, 12, 32768), (FMT_BYTE, 10, 32768)])
def test_full_size_per_chunk_models_every_chunk_equals_oracle(gpu, oracle, fmt, sb, chunk):
    """Per-chunk models at the BASELINE size (parity was pinned on 37 chunks): 2^30 bytes whose
    statistics move every 4 Mi symbols, coded by the one-kernel encoder (rans_amd_encode_adaptive_sized) and by the
    three-launch path -- EVERY chunk's row == the oracle's normalize(count(chunk)) and EVERY chunk's stream == the oracle's
    stream of that chunk under its own model (main.cpp:139-162 with the chunk as the input); the pieces lie in index order;
    the decoder takes the container as it is and -- independent of any GPU encoder -- a container the ORACLE made."""
    R, ctx, torch = gpu
    import bench
    n = 1 << 30
    d_syms = bench.gen_moving(torch, n, 1, "cuda")
    # (the generator on the GPU == the oracle's on the CPU, group by group)
    for g in (0, 5, 255):
        part = oracle.gen_zipf(1 << 22, K=(256, 64, 16, 4)[g % 4], s=(0.5, 1.0, 1.5, 2.0)[(g // 4) % 4], seed=1 + g)
        assert np.array_equal(d_syms[g << 22:(g + 1) << 22].cpu().numpy(), ((part.astype(np.int32) + 37 * g) % 256).astype(np.uint8))
    h_syms = d_syms.cpu().numpy()
    nchunks = n // chunk
    cont, offs, lens, rows, total = ctx.encode_adaptive_sized(d_syms, 64, chunk, sb, fmt=fmt)
    assert ctx.last_encode_kernel()[0] == ("k_encode_adaptive<word>" if fmt == FMT_WORD else "k_encode_adaptive<byte>")
    h_offs, h_lens = offs.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
    h_rows = rows.cpu().numpy()
    count, bad = oracle.compare_container_adaptive(fmt, h_syms, 64, chunk, sb, cont[:total].cpu().numpy(), h_offs, h_lens, h_rows)
    assert count == nchunks and bad == -1, bad
    ends = h_offs[:nchunks] + h_lens
    assert np.all(ends % np.uint64(64) == 0) and np.all(ends[:-1] <= h_offs[1:nchunks]) and int(ends[-1]) == total == int(h_offs[nchunks])
    streams = int(h_lens.astype(np.int64).sum())
    # (the bound's slack per chunk: 1.1 % of its symbols in the word format, the 2^-12 of the f32 sum, a line, rounding to lines)
    assert total <= streams + nchunks * (chunk // 80 + 512), (total, streams)
    out = ctx.decode_adaptive(cont, total, offs, lens, rows, n, 64, chunk, sb, fmt=fmt)
    assert torch.equal(out, d_syms)
    del out
    # the three-launch path: the same rows and the same streams (its container: the compact layout)
    c0, o0, l0, r0, t0 = ctx.encode_adaptive(d_syms, 64, chunk, sb, fmt=fmt)
    assert torch.equal(r0, rows) and torch.equal(l0, lens)
    count, bad = oracle.compare_container_adaptive(fmt, h_syms, 64, chunk, sb, c0[:t0].cpu().numpy(), o0.cpu().numpy(), h_lens, h_rows)
    assert count == nchunks and bad == -1, bad
    del c0, cont
    # the decoder on a container (streams AND rows) made by the oracle alone
    o_cont, o_offs, o_lens, o_rows = oracle.encode_chunked_adaptive(fmt, h_syms, 64, chunk, sb)
    assert np.array_equal(o_lens, h_lens) and np.array_equal(o_rows.reshape(-1), h_rows.view(np.uint16))
    d_cont = torch.zeros(o_cont.size + 64, dtype=torch.uint8, device="cuda")
    d_cont[:o_cont.size] = torch.from_numpy(o_cont).cuda()
    out = ctx.decode_adaptive(d_cont, o_cont.size, torch.from_numpy(o_offs.astype(np.int64)).cuda(),
                              torch.from_numpy(o_lens.astype(np.int32)).cuda(), torch.from_numpy(o_rows.reshape(-1).view(np.int16)).cuda(),
                              n, 64, chunk, sb, fmt=fmt)
    assert torch.equal(out, d_syms)


@pytest.mark.parametrize("name,fmt,sb,ways,chunk,n", [
    ("word-64-way", FMT_WORD, 12, 64, 32768, 5 << 30),                 # 5 Gi symbols: 163 840 chunks, 32 768 of them past 2^32
    ("r64-2-way", FMT_R64, 14, 2, 512, (1 << 32) + (1 << 20) + 77),    # config 2's shape: 8 390 657 chunks (> 2^23), a ragged last one
    ("word-8-way", FMT_WORD, 12, 8, 1024, (5 << 30) + 77),             # the group kernels (8 chunks per wave): 5 242 881 chunks, a ragged last one
])
def test_more_than_2_to_the_32_symbols(gpu, oracle, name, fmt, sb, ways, chunk, n):
    """n >= 2^32: the ABI takes uint64_t n and the kernels form 64-bit bases; here they meet symbols, chunks
    and offsets beyond 32 bits.  Sized-slot encode, decode, round trip; an oracle sample of 1024 chunks plus the first and the
    LAST ones (past 2^32 symbols) compared byte for byte; the compact encoder's index agrees with the sized one's lengths."""
    R, ctx, torch = gpu
    import bench
    d_syms = bench.gen_zipf(torch, n, 256, 1.0, 1, "cuda")
    # the model builder's counters are the reference's 32-bit ones (main.cpp:49-57): a histogram of the whole input is REFUSED,
    # not wrapped -- the first 32-bit limit this test found, now a documented one; the model comes from the first GiB
    with pytest.raises(R.RansAmdError) as e:
        ctx.count_freqs_device(d_syms, 256)
    assert e.value.status == R.E_UNSUPPORTED
    counts = ctx.count_freqs_device(d_syms[:1 << 30], 256)
    freqs, _ = R.normalize_freqs(counts, 1 << sb)
    gm = ctx.model(fmt, freqs, sb)
    nchunks = (n + chunk - 1) // chunk
    t_cont, t_offs, t_lens, t_total, t_slot = ctx.encode_sized(gm, d_syms, ways, chunk)
    worst = R.slot_bytes(fmt, n, ways, chunk)
    assert t_total > (1 << 32) and int(t_offs[-1].item()) == t_total
    art = {"fmt": fmt, "sb": sb, "K": 256, "ways": ways, "chunk": chunk, "n": n, "freqs": freqs, "d_syms": d_syms,
           "cont": t_cont, "offs": t_offs, "lens": t_lens, "total": t_total, "slot": t_slot, "worst": worst}
    assert bench.oracle_check_chunks(art, sample=1024) >= 1024
    # the chunks whose symbols lie past 2^32, one by one (the last 64 and the ragged last)
    om = oracle.model(freqs, sb)
    h_offs = t_offs[nchunks - 64:nchunks].cpu().numpy()
    h_lens = t_lens[nchunks - 64:nchunks].cpu().numpy()
    for i, c in enumerate(range(nchunks - 64, nchunks)):
        lo, hi = c * chunk, min(n, (c + 1) * chunk)
        assert lo >= (1 << 32)
        ref = oracle.encode(fmt, om, d_syms[lo:hi].cpu().numpy(), ways)
        a, ln = int(h_offs[i]), int(h_lens[i])
        assert ln == ref.size and np.array_equal(t_cont[a:a + ln].cpu().numpy(), ref), c
    out = ctx.decode(gm, t_cont, t_total, t_offs, t_lens, n, ways, chunk)
    assert torch.equal(out, d_syms)
    del out, t_cont
    # the compact layout (fused placement / lane encoders + offset scan over > 2^23 chunks): same lengths, prefix-sum index
    cont, offs, lens, total = ctx.encode(gm, d_syms, ways, chunk)
    assert torch.equal(lens, t_lens)
    aligned = (lens.to(torch.int64) + 15) & ~15
    want = torch.zeros(nchunks + 1, dtype=torch.int64, device="cuda")
    want[1:] = torch.cumsum(aligned, 0)
    want[nchunks] = want[nchunks - 1] + lens[nchunks - 1]
    assert torch.equal(offs, want) and total == int(want[-1].item())
    out = ctx.decode(gm, cont, total, offs, lens, n, ways, chunk)
    assert torch.equal(out, d_syms)


def test_per_chunk_models_past_2_to_the_32_symbols(gpu, oracle):
    """The one-kernel per-chunk-model encoder beyond 32 bits: 2^32 + 3 x 16384 + 77 symbols of moving statistics in 262 148
    chunks (the register-resident form, a ragged last chunk in the two-pass form, look-back sums beyond 2^32 bytes) -- every
    chunk's row and stream == the oracle's, the pieces in index order, decoded back."""
    R, ctx, torch = gpu
    import bench
    n, chunk = (1 << 32) + 3 * 16384 + 77, 16384
    d_syms = bench.gen_moving(torch, n, 3, "cuda")
    nchunks = (n + chunk - 1) // chunk
    cont, offs, lens, rows, total = ctx.encode_adaptive_sized(d_syms, 64, chunk, 12, fmt=FMT_WORD, cap=n + (n >> 3))
    h_offs, h_lens = offs.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
    ends = h_offs[:nchunks] + h_lens
    assert np.all(ends % np.uint64(64) == 0) and np.all(ends[:-1] <= h_offs[1:nchunks]) and int(ends[-1]) == total == int(h_offs[nchunks])
    count, bad = oracle.compare_container_adaptive(FMT_WORD, d_syms.cpu().numpy(), 64, chunk, 12, cont[:total].cpu().numpy(), h_offs, h_lens,
                                                   rows.cpu().numpy())
    assert count == nchunks and bad == -1, bad
    out = ctx.decode_adaptive(cont, total, offs, lens, rows, n, 64, chunk, 12, fmt=FMT_WORD)
    assert torch.equal(out, d_syms)


def test_book1_appendix_b_on_gpu(gpu):
    """SURVEY appendix B through the HIP path: decode the reference-made 64-way word stream of book1, then
    re-encode book1 into every pinned stream (sizes from the README, SHA-256 from the unmodified reference)."""
    R, ctx, torch = gpu
    meta = json.load(open(os.path.join(HERE, "golden", "book1_word64.json")))
    known = json.load(open(os.path.join(HERE, "golden", "book1_golden.json")))
    stream = np.fromfile(os.path.join(HERE, "golden", meta["stream"]), dtype=np.uint8)
    assert hashlib.sha256(stream.tobytes()).hexdigest() == meta["stream_sha256"]
    f12 = np.array(meta["freqs"]["12"], dtype=np.uint32)
    gm = ctx.model(FMT_WORD, f12, 12)
    book1 = ctx.decode_host(gm, stream, meta["n"], 64)
    assert book1.size == known["input_size"]
    assert hashlib.sha256(book1.tobytes()).hexdigest() == known["input_sha256"]
    # the model builder on the GPU histogram reproduces the reference's normalised frequencies
    counts = ctx.count_freqs_device(torch.from_numpy(book1).cuda(), 256)
    for sb in (12, 14, 16):
        f, _ = R.normalize_freqs(counts, 1 << sb)
        assert [int(v) for v in f] == meta["freqs"][str(sb)], sb
    for e in known["streams"]:
        fmt, sb, N = FMT[e["fmt"]], e["scale_bits"], e["n_ways"]
        m = ctx.model(fmt, np.array(meta["freqs"][str(sb)], dtype=np.uint32), sb)
        s = ctx.encode_host(m, book1, N)
        assert s.size == e["size"], (e["fmt"], N)
        if e["sha256"]:  # (entries without a hash carry the size only)
            assert hashlib.sha256(s.tobytes()).hexdigest() == e["sha256"], (e["fmt"], N)
        assert np.array_equal(ctx.decode_host(m, s, book1.size, N), book1), (e["fmt"], N)
    # chunked (what the bulk path does): the corpus as 24 chunks of 32 Ki symbols, decoded back
    d = torch.from_numpy(book1).cuda()
    cont, offs, lens, total = ctx.encode(gm, d, 64, 32768)
    out = ctx.decode(gm, cont, total, offs, lens, book1.size, 64, 32768)
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == known["input_sha256"]
