// decode_wave.hip -- the wave-per-chunk decoder (the headline kernel) of the hand-written gfx950
// (CDNA4, wave64) kernels for interleaved rANS; encode_wave.hip, lanes.hip and container_kernels.hip
// hold the rest.
//
// Mapping of the reference's hot loops onto the GPU
// --------------------------------------------------
// The reference decodes an N-way interleaved stream with N rANS states that take
// turns: every "round" each state decodes one symbol (table lookup + one
// multiply-add, no stream access), then the states, in ascending lane order,
// pull the renormalisation units they need from ONE shared cursor
// (main.cpp:259-280, main_simd.cpp:313-332).  The SSE4.1 decoder does this for
// 4 lanes with movemask + pshufb (rans_word_sse41.h:182-227).  Here:
//
//   * one wavefront owns one chunk (an independent N-way stream), N = 64*K:
//     lane l holds states l, l+64, ..., i.e. K states per lane;
//   * the symbol lookup table lives in LDS, shared by the 16 waves of a block;
//   * "which lanes renormalise" is a 64-bit ballot; a lane's position in the
//     stream is popcount(ballot & lanes_below) (v_mbcnt), so the wave consumes
//     popcount(ballot) consecutive units per sub-step: stream I/O is dense and
//     in order by construction;
//   * the compressed stream is pulled through a per-wave LDS window ("ring")
//     in aligned 1 KiB blocks (16 B per lane, one global_load_dwordx4 per
//     block), prefetched one block ahead in registers;
//   * decoded bytes are transposed in registers across 4 rounds (v_perm_b32 +
//     quad DPP) so a store instruction writes 256 contiguous bytes per wave.
//
// No MFMA: the work is integer, table-driven and serial per state.
// The encoder is the exact mirror (symbols visited last to first, units pushed
// downwards); it writes every chunk into a worst-case scratch slot, then a
// layout pass (prefix sum of sizes) and a compaction pass build the container.
//
// Bit-exactness: the arithmetic below is the reference's (file:line cited at
// each step); only its *scheduling* across lanes is new.

#include "decode_common.hpp"

namespace rans_amd {

namespace {

// ---------------------------------------------------------------------------
// Word format, 64-way, full waves: FOUR rounds as one hand-scheduled instruction sequence
// (rans_word_sse41.h:123-141 / :151-227 for 64 lanes; the D step and the renormalisation are
// the ones of dec_step / renorm_word_full).  Per round 9 VALU + 2 LDS + 6 SALU:
//   v_and, v_lshlrev            slot = x & 4095 -> LDS byte address of the slot record
//   ds_read_b64                 {freq | sym << 24, bias}
//   v_lshrrev, v_mad_u32_u24    x = freq * (x >> 12) + bias
//   v_perm                      the symbol joins the three others of this lane (rounds 1..3)
//   v_cmpx .. v_perm            renormalisation, see renorm_word_full
// The transposition and the store of the PREVIOUS group's symbols (two DPP + v_perm pairs and a
// buffer_store_dword) sit in round 0, where the wave would otherwise wait for the slot record, and where
// the wait states a DPP operand needs after a VALU write (2) are filled by instructions that have to be
// issued anyway.  pa: in = the previous group's four symbols of this lane, out = this group's.
// Temporaries are fixed registers (v56..v62) because the halves of a 64-bit asm operand cannot be named; the
// store's data register (v62) is written nowhere else, so that nothing has to wait for the store to have read it
// (with the data in v60, round 1's ds_read_b64 v[60:61] stalled every group: +15 % kernel time).
// ---------------------------------------------------------------------------
#define RANS_WORD_RENORM                                   \
    "v_cmpx_gt_u32_e32 vcc, %[lim], %[x]\n\t"              \
    "s_bcnt1_i32_b64 %[cnt], vcc\n\t"                      \
    "s_nop 0\n\t"                                          \
    "v_mbcnt_lo_u32_b32 v56, vcc_lo, 0\n\t"                \
    "v_mbcnt_hi_u32_b32 v56, vcc_hi, v56\n\t"              \
    "v_lshl_add_u32 v56, v56, 1, %[cur]\n\t"               \
    "ds_read_u16 v57, v56\n\t"                             \
    "s_lshl1_add_u32 %[cur], %[cnt], %[cur]\n\t"           \
    "s_waitcnt lgkmcnt(0)\n\t"                             \
    "v_perm_b32 %[x], %[x], v57, %[selm]\n\t"              \
    "s_mov_b64 exec, -1\n\t"
#define RANS_WORD_LOOKUP(E)                                \
    "v_and_b32_e32 v56, %[m12], %[x]\n\t"                  \
    "v_lshlrev_b32_e32 v56, 3, v56\n\t"                    \
    "ds_read_b64 " E ", v56\n\t"                           \
    "v_lshrrev_b32_e32 v57, 12, %[x]\n\t"

template <bool kStorePrev>
__device__ __forceinline__ void decode_group_word(uint32_t &x, uint32_t &pa, uint32_t &cur, uint32_t m12,
                                                  uint32_t k65536, uint32_t sel1, uint32_t sel2, uint32_t selA,
                                                  uint32_t selB, uint32_t selC, u32x4 orsrc, uint32_t out_lane_off,
                                                  uint32_t osoff_prev)
{
    uint32_t cnt;
    if constexpr (kStorePrev) {
        asm volatile(
            // ---- round 0 (+ the previous group's transposition and store)
            RANS_WORD_LOOKUP("v[58:59]")
            "v_mov_b32_dpp v60, %[pa] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_perm_b32 v61, v60, %[pa], %[sel1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v58, v57, v59\n\t"
            "v_mov_b32_dpp v60, v61 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_perm_b32 v62, v60, v61, %[sel2]\n\t"
            "buffer_store_dword v62, %[ooff], %[orsrc], %[osoff] offen" RANS_STORE_MODS "\n\t"
            RANS_WORD_RENORM
            // ---- round 1
            RANS_WORD_LOOKUP("v[60:61]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v60, v57, v61\n\t"
            "v_perm_b32 %[pa], v60, v58, %[selA]\n\t"
            RANS_WORD_RENORM
            // ---- round 2
            RANS_WORD_LOOKUP("v[58:59]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v58, v57, v59\n\t"
            "v_perm_b32 %[pa], v58, %[pa], %[selB]\n\t"
            RANS_WORD_RENORM
            // ---- round 3
            RANS_WORD_LOOKUP("v[60:61]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v60, v57, v61\n\t"
            "v_perm_b32 %[pa], v60, %[pa], %[selC]\n\t"
            RANS_WORD_RENORM
            : [x] "+v"(x), [pa] "+v"(pa), [cur] "+s"(cur), [cnt] "=&s"(cnt)
            : [m12] "v"(m12), [lim] "v"(k65536), [sel1] "v"(sel1), [sel2] "v"(sel2), [selA] "v"(selA), [selB] "v"(selB),
              [selC] "v"(selC), [selm] "s"(0x05040100u), [orsrc] "s"(orsrc), [ooff] "v"(out_lane_off),
              [osoff] "s"(osoff_prev)
            : "vcc", "scc", "memory", "v56", "v57", "v58", "v59", "v60", "v61", "v62"
            );
        // (The store is deliberately inside the sequence, in round 0.  The compiler does not see it, so the
        // s_waitcnt vmcnt(0) it puts in front of a window refill -- which happens after round 3 -- also waits
        // for this store; by then it is ~1000 cycles old and done.  A store the compiler knows about makes it
        // wait with vmcnt(1) at the top of EVERY group, i.e. for the window prefetch right after each refill.)
    } else {
        asm volatile(
            RANS_WORD_LOOKUP("v[58:59]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v58, v57, v59\n\t"
            RANS_WORD_RENORM
            RANS_WORD_LOOKUP("v[60:61]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v60, v57, v61\n\t"
            "v_perm_b32 %[pa], v60, v58, %[selA]\n\t"
            RANS_WORD_RENORM
            RANS_WORD_LOOKUP("v[58:59]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v58, v57, v59\n\t"
            "v_perm_b32 %[pa], v58, %[pa], %[selB]\n\t"
            RANS_WORD_RENORM
            RANS_WORD_LOOKUP("v[60:61]")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mad_u32_u24 %[x], v60, v57, v61\n\t"
            "v_perm_b32 %[pa], v60, %[pa], %[selC]\n\t"
            RANS_WORD_RENORM
            : [x] "+v"(x), [pa] "+v"(pa), [cur] "+s"(cur), [cnt] "=&s"(cnt)
            : [m12] "v"(m12), [lim] "v"(k65536), [selA] "v"(selA), [selB] "v"(selB), [selC] "v"(selC),
              [selm] "s"(0x05040100u)
            : "vcc", "scc", "memory", "v56", "v57", "v58", "v59", "v60", "v61", "v62"
            );
    }
}
#undef RANS_WORD_RENORM
#undef RANS_WORD_LOOKUP


template <int FMT, int K, int OUT>
__global__ void __launch_bounds__(kDecBlockThreads, (K <= 2 ? 8 : 4)) k_decode(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // wave clocks (rans_amd_set_timing(ctx, 2) / RANS_AMD_TRACE): constant 100 MHz clock and shader clock
    const unsigned long long t_start = (p.trace || p.span) ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? __builtin_readcyclecounter() : 0ull;
    uint32_t rounds_done = 0;

    // ---- stage the tables into LDS (once per block) ----------------------
    // (per-chunk models: no shared tables -- "table 0" is the waves' own regions, filled per chunk below)
    const uint32_t t0_bytes = kIsAdaptive<FMT> ? (blockDim.x >> 6) * kAdaptDecWaveLds : (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = kIsAdaptive<FMT> ? 0u : (p.table1_bytes + 15u) & ~15u;
    if constexpr (!kIsAdaptive<FMT>) {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
        const uint4 *g1 = reinterpret_cast<const uint4 *>(p.table1);
        uint4 *l1 = reinterpret_cast<uint4 *>(smem + t0_bytes);
        for (uint32_t i = threadIdx.x; i < t1_bytes / 16u; i += blockDim.x)
            l1[i] = g1[i];
    }
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;

    DecTables<FMT> T;
    if constexpr (kIsAdaptive<FMT>) // cum2sym[M] then {freq, start}[256] of the chunk in hand, this wave's own
        T.init(smem + wave * kAdaptDecWaveLds, smem + wave * kAdaptDecWaveLds + (1u << kAdaptMaxScaleBits), p.scale_bits,
               p.log2nsyms);
    else
        T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);
    if (!lds_starts_at_zero(smem)) { // cannot happen without static LDS; never decode on a wrong assumption
        if (threadIdx.x == 0)
            atomicAdd(p.err_count, 1ull << 32);
        return;
    }

    uint8_t *ring = smem + t0_bytes + t1_bytes + wave * kRingStride;
    const uint32_t N = (OUT != OUT_SLOW) ? 64u * K : p.n_ways;
    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);

    // per-lane constants of the output transpose
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t out_lane_off = (lane & 3u) * N + (lane & ~3u);

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u;
    if (p.span_reset && blockIdx.x == 0 && threadIdx.x < 2)
        p.span_reset[threadIdx.x] = 0ull;
    // Chunks are handed out dynamically.  The SIMD arbitrates VALU issue by wave age, so the
    // waves of the older of a CU's two workgroups run ~20 % faster than the younger ones
    // (measured: 314 vs 372 us for the same work); with a static split the kernel lasts as long
    // as the slowest wave while the SIMDs drain.  One atomic per chunk on a single word tops
    // out near 88 claims/us, so there are kWorkPools counters on separate cache lines; pool =
    // blockIdx % 8 (= the XCD, as dispatched today; only speed depends on that) owns the chunks
    // c with c % 8 == pool.
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t chunk_v = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    for (;;) {
        if (p.work_counter) {
            uint32_t got = 0;
            if (lane == 0)
                got = atomicAdd(p.work_counter + pool * kWorkPoolStride, 1u);
            chunk_v = (uint64_t)uniform(got) * npools + pool;
        }
        if (chunk_v >= p.nchunks)
            break;
        // everything derived from the chunk index is wave-uniform; say so explicitly
        // so it lives in SGPRs and the loop control below is scalar
        const uint64_t chunk = uniform64(chunk_v);
        chunk_v += total_waves; // static stride when there is no counter
        const uint64_t off = uniform64(p.offsets[chunk]);
        const uint32_t len = uniform(p.lengths[chunk]);
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        const uint64_t src = cbase + off;
        uint8_t RANS_GLOBAL *dst = reinterpret_cast<uint8_t RANS_GLOBAL *>(
            reinterpret_cast<uint64_t>(p.out) + first * p.sym_bytes);

        // off and len come from the caller's index: compare without forming off + len (which can wrap).  A chunk may
        // start at any multiple of the format's unit (compact containers start theirs on 16 bytes, the slot layout of
        // rans_amd_encode_slots ENDS them there): the window fetches whole 16-byte granules from the one that holds the
        // chunk's first byte, `skip` bytes into it.
        bool ok = ((off & (Tr::kUnit - 1u)) == 0) && (len >= N * Tr::kStateBytes) && (off <= p.container_bytes) &&
                  (len <= p.container_bytes - off);
        if (!ok) { // wave-uniform
            if (lane == 0)
                atomicAdd(p.err_count, 1ull);
            continue;
        }

        if constexpr (kIsAdaptive<FMT>) { // this chunk's model -> this wave's tables (main.cpp:139-162 / main_simd.cpp:138-143 per chunk)
            if (!adapt_build_dec(p.chunk_freqs + chunk * 256u, p.scale_bits, lane, const_cast<uint8_t *>(T.t0),
                                 reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(T.t1)))) {
                if (lane == 0)
                    atomicAdd(p.err_count, 1ull);
                continue;
            }
        }

        // ---- initial states: lane 0's first (RansDecInit order, main.cpp:261-262)
        state_t x[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = k * 64u + lane;
            x[k] = Tr::kL;
            if (idx < N) {
                if constexpr (kIsR64<FMT>) {
                    const u32x2 v = *(reinterpret_cast<const u32x2 RANS_GLOBAL *>(src) + idx);
                    x[k] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                } else {
                    x[k] = *(reinterpret_cast<const uint32_t RANS_GLOBAL *>(src) + idx);
                }
            }
        }

        StreamWindow W;
        // fetch nothing beyond this chunk's own stream (rounded up to the 16-byte granule) nor beyond the
        // container's last granule
        const uint32_t skip = (uint32_t)off & 15u;
        const uint64_t room = ((p.container_bytes + 15u) & ~uint64_t(15)) - (off - skip);
        const uint32_t climit = (skip + len + 15u) & ~15u;
        W.open(ring, src - skip, skip + N * Tr::kStateBytes, climit < room ? climit : (uint32_t)room, lane);

        const uint32_t rounds = uniform(nsym / N);
        rounds_done += rounds;
        const uint32_t tail = uniform(nsym - rounds * N);
        uint32_t r = 0;
        // sub-steps between two window checkpoints: at most kMaxAdvance bytes are consumed
        constexpr int kCheckEvery = kIsR64<FMT> ? 2 : 4;

        if constexpr (OUT == OUT_FAST16) {
            // ---- pairs of full rounds, u16 symbols: lane 2i ends up with round r's symbols of
            // lanes 2i,2i+1 and lane 2i+1 with round r+1's: one dword store per lane and pair
            const uint32_t pairs = rounds >> 1;
            uint8_t RANS_GLOBAL *gdst = dst;
            const uint32_t sel16 = (lane & 1u) ? 0x03020706u : 0x05040100u;
            const uint32_t lane_off16 = ((lane & 1u) * N + (lane & ~1u)) * 2u;
            const uint32_t k2p23 = (1u << 23) + (lane >> 6), k2p15 = (1u << 15) + (lane >> 6);
            const uint32_t k65536w = 0x10000u + (lane >> 6);
            for (uint32_t g = 0; g < pairs; ++g) {
                uint32_t acc[K];
#pragma unroll
                for (int J = 0; J < 2; ++J) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint32_t s16 = dec_step<FMT>(T, x[k]) & 0xffffu;
                        acc[k] = J == 0 ? s16 : (acc[k] | (s16 << 16));
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if ((J * K + k) % kCheckEvery == 0)
                            W.checkpoint(lane);
                        if constexpr (kIsByteStream<FMT>)
                            renorm_byte_full(x[k], W.cur, k2p23, k2p15);
                        else if constexpr (FMT == FMT_WORD16)
                            renorm_word_full(x[k], W.cur, k65536w);
                        else
                            W.consume(dec_renorm<FMT>(W, x[k], true));
                    }
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t o = quad_perm<1, 0, 3, 2>(acc[k]);
                    const uint32_t v = __builtin_amdgcn_perm(o, acc[k], sel16);
                    __builtin_nontemporal_store(v, reinterpret_cast<uint32_t RANS_GLOBAL *>(gdst + (lane_off16 + k * 128u)));
                }
                gdst += 4u * N;
            }
            r = pairs << 1;
        } else if constexpr (OUT != OUT_SLOW) {
            // ---- groups of 4 full rounds, symbols transposed in registers ----
            const uint32_t groups = rounds >> 2;
            // symbol stores go through a descriptor of the chunk's output with the running offset in an SGPR
            // (soffset): no 64-bit VALU pointer arithmetic in the loop
            const rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(reinterpret_cast<uint64_t>(dst)), 0, nsym, kRsrcFlags);
            uint32_t osoff = 0;
            const uint32_t k65536 = 0x10000u + (lane >> 6); // VGPRs holding the renorm limits (lane < 64)
            const uint32_t k2p23 = (1u << 23) + (lane >> 6), k2p15 = (1u << 15) + (lane >> 6);
            for (uint32_t g = 0; g < groups; ++g) {
                uint32_t acc[K];
#define RANS_ROUND(J)                                                              \
    _Pragma("unroll") for (int k = 0; k < K; ++k)                                  \
        acc[k] = acc_symbol<Tr::kSymByte, J>(dec_step<FMT>(T, x[k]), acc[k]);      \
    _Pragma("unroll") for (int k = 0; k < K; ++k) {                                \
        if ((J * K + k) % kCheckEvery == 0)                                        \
            W.checkpoint(lane);                                                    \
        if constexpr (FMT == FMT_WORD || FMT == FMT_WORDA)                         \
            renorm_word_full(x[k], W.cur, k65536);                                 \
        else if constexpr (kIsByteStream<FMT>)                                     \
            renorm_byte_full(x[k], W.cur, k2p23, k2p15);                           \
        else                                                                       \
            W.consume(dec_renorm<FMT>(W, x[k], true));                             \
    }
                RANS_ROUND(0)
                RANS_ROUND(1)
                RANS_ROUND(2)
                RANS_ROUND(3)
#undef RANS_ROUND
                // One state per lane: one more refill check here, in FRONT of the group's stores.  A refill waits for the prefetched
                // block with s_waitcnt vmcnt(0) -- the compiler cannot tell what else is in flight -- and the check at the top of the
                // next group comes right behind these stores: it would wait for them as well (per-chunk models, word: 0.692 ->
                // 0.673 ms; byte: 0.755 -> 0.748; with two states per lane, whose checks sit inside the group, it costs 2-6 %)
                if constexpr (K == 1)
                    W.checkpoint(lane);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t v = quad_transpose(acc[k], sel1, sel2);
                    __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, out_lane_off + k * 64u, osoff, kAuxStore);
                }
                osoff += 4u * N;
            }
            r = groups << 2;
        }

        // ---- remaining full rounds and the partial tail round: element stores
        for (; r <= rounds; ++r) {
            const uint32_t cnt = (r < rounds) ? N : tail;
            if (cnt == 0)
                break;
            uint8_t RANS_GLOBAL *rdst = dst + (uint64_t)r * N * p.sym_bytes;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t idx = k * 64u + lane;
                if (idx < cnt) {
                    uint32_t s = dec_step<FMT>(T, x[k]);
                    if constexpr (Tr::kSymByte == 3)
                        s >>= 24;
                    if (p.sym_bytes == 1)
                        rdst[idx] = (uint8_t)s;
                    else
                        reinterpret_cast<uint16_t RANS_GLOBAL *>(rdst)[idx] = (uint16_t)s;
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t idx = k * 64u + lane;
                W.checkpoint(lane);
                W.consume(dec_renorm<FMT>(W, x[k], idx < cnt));
            }
        }

        // ---- integrity: every state back at L, cursor exactly at the end ----
        bool good = true;
#pragma unroll
        for (int k = 0; k < K; ++k)
            good = good && (x[k] == Tr::kL);
        const bool all_good = __builtin_amdgcn_ballot_w64(!good) == 0 && W.position() == skip + len;
        if (!all_good && lane == 0)
            atomicAdd(p.err_count, 1ull);
    }
    if (p.trace && lane == 0) { // per wave: start / end on the 100 MHz clock, XCD, shader cycles spent, rounds decoded
        unsigned long long *t = p.trace + (uint64_t)kTraceWords * ((uint64_t)blockIdx.x * waves_per_block + wave);
        t[0] = t_start;
        t[1] = wall_clock64();
        t[2] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
        t[3] = __builtin_readcyclecounter() - c_start;
        t[4] = rounds_done;
    }
    record_span(p, t_start, smem);
}

// ---------------------------------------------------------------------------
// k_decode_word64 -- the headline configuration on its own: word format, 64-way, u8 symbols, 4-byte aligned
// output (main_simd.cpp:313-332 for 64 lanes).  Same decoding as k_decode<FMT_WORD, 1, OUT_FAST8>, four rounds as ONE asm sequence,
// plus the hand-over between chunks taken off the critical path: a wave that stops to claim a chunk, read
// its index entry, then its initial states and first stream blocks sits through three dependent memory
// round trips (~3 us, and every SIMD has 8 waves doing that once per chunk).  Here the three steps of the
// NEXT chunk are issued while the last rounds of the current one are decoded:
//     kClaimAhead  groups before the end   atomic on the chunk counter                          (claim)
//     kDataAhead                           offsets[next], lengths[next] through the scalar cache,
//                                          then initial states and stream blocks 0 and 1         (data, 9 VGPRs)
// so the switch itself is two LDS block writes.  Claims are made late on purpose: a claimed chunk is work
// committed to this wave, and the slow (young) waves of a SIMD should not sit on chunks at the end.
// ---------------------------------------------------------------------------
constexpr uint32_t kClaimAhead = 10, kDataAhead = 5; // in groups of 4 rounds

// buffer descriptor of one 64-way word chunk's stream (StreamWindow::stream_rsrc): what may be fetched is the
// chunk's length rounded up to the 16-byte granule, clipped to the container's last granule
// (a chunk may start anywhere on a 2-byte boundary: the descriptor starts at the 16-byte granule that holds its first
//  byte, `off & 15` bytes further down -- see k_decode)
__device__ __forceinline__ rsrc_t chunk_rsrc(uint64_t cbase, uint64_t cbytes16, uint64_t off, uint32_t len)
{
    // off and len are wave-uniform (scalar loads); saying so here keeps the descriptor in SGPRs whatever the
    // compiler concluded about the loop-carried copies (a descriptor in VGPRs turns every fetch into a waterfall loop)
    off = uniform64(off);
    len = uniform(len);
    const uint32_t skip = (uint32_t)off & 15u;
    const uint64_t room = cbytes16 - (off - skip);
    const uint32_t climit = (skip + len + 15u) & ~15u;
    return StreamWindow::stream_rsrc(cbase + (off - skip), skip + 64u * 4u, climit < room ? climit : (uint32_t)room);
}

// (eight waves per SIMD at 64 VGPRs with a few spills; a spill-free 72-VGPR build in blocks of 14 waves -- seven per SIMD -- was
//  1.47 x slower: profiles/r04_word64_launch_bounds.md)
constexpr int kWord64Threads = kDecBlockThreads;
__global__ void __launch_bounds__(kWord64Threads, 8) k_decode_word64(const DecParams p)
{
    using Tr = FmtTraits<FMT_WORD>;
    constexpr uint32_t N = 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned long long t_start = (p.trace || p.span) ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? __builtin_readcyclecounter() : 0ull;
    uint32_t rounds_done = 0;

    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
    }
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    DecTables<FMT_WORD> T;
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);
    if (!lds_starts_at_zero(smem)) {
        if (threadIdx.x == 0)
            atomicAdd(p.err_count, 1ull << 32);
        return;
    }
    uint8_t *ring = smem + t0_bytes + wave * kRingStride;
    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t cbytes16 = (p.container_bytes + 15u) & ~uint64_t(15);

    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t out_lane_off = (lane & 3u) * N + (lane & ~3u);
    uint32_t selA = 0x03020703u, selB = 0x03070100u, selC = 0x07020100u; // acc_symbol<3, 1..3>
    uint32_t k65536 = 0x10000u, m12 = 0xfffu;
    asm volatile("v_mov_b32 %0, %0" : "+v"(selA)); // opaque: live in VGPRs, never rematerialised in the loop
    asm volatile("v_mov_b32 %0, %0" : "+v"(selB));
    asm volatile("v_mov_b32 %0, %0" : "+v"(selC));
    asm volatile("v_mov_b32 %0, %0" : "+v"(k65536));
    asm volatile("v_mov_b32 %0, %0" : "+v"(m12));

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u;
    if (p.span_reset && blockIdx.x == 0 && threadIdx.x < 2)
        p.span_reset[threadIdx.x] = 0ull;

    // ---- chunk hand-out (see k_decode): one counter per XCD, or static striding without counters
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t static_next = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    uint32_t claimed_v = 0; // lane 0: what the atomic returned

    // ---- the three steps of taking a chunk; state of the chunk being taken: n_* ----------------------
    uint64_t n_idx = 0;               // chunk index
    uint64_t n_off = 0;               // its index entry: scalar loads, i.e. SGPRs from the start
    uint32_t n_len = 0;
    uint32_t n_x = 0;                 // initial state of this lane (RansDecInit order: lane l's state is the l-th)
    u32x4 n_b0 = {0u, 0u, 0u, 0u}, n_b1 = n_b0; // stream blocks 0 and 1
    bool n_any = false, n_ok = false; // a chunk was claimed / its index entry passed validation and data is on its way

#define RANS_STEP_CLAIM()                                                                   \
    do {                                                                                    \
        if (p.work_counter && lane == 0)                                                    \
            claimed_v = atomicAdd(p.work_counter + pool * kWorkPoolStride, 1u);             \
    } while (0)
    // The index entry comes through the scalar cache (s_load_dwordx2 / s_load_dword + wait, ~0.2 us on an L2
    // hit, once per chunk): it lands in SGPRs, no VGPR is tied up.  off and len come from the caller: compare
    // without forming off + len (which can wrap); fetch nothing beyond the chunk's own stream (16-byte granule)
    // nor beyond the container's last granule.
#define RANS_STEP_DATA()                                                                    \
    do {                                                                                    \
        if (p.work_counter) {                                                               \
            n_idx = (uint64_t)uniform(claimed_v) * npools + pool;                           \
        } else {                                                                            \
            n_idx = uniform64(static_next);                                                 \
            static_next += total_waves;                                                     \
        }                                                                                   \
        n_any = n_idx < p.nchunks;                                                          \
        n_ok = false;                                                                       \
        if (n_any) {                                                                        \
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" \
                         : "=&s"(n_off), "=&s"(n_len)                                       \
                         : "s"(uniform64(reinterpret_cast<uint64_t>(p.offsets + n_idx))),   \
                           "s"(uniform64(reinterpret_cast<uint64_t>(p.lengths + n_idx)))    \
                         : "memory");                                                       \
            n_off = uniform64(n_off); /* (asm results count as divergent: say what they are) */ \
            n_len = uniform(n_len);                                                         \
            n_ok = ((n_off & 1u) == 0) && (n_len >= N * Tr::kStateBytes) && (n_off <= p.container_bytes) && \
                   (n_len <= p.container_bytes - n_off);                                    \
            if (n_ok) {                                                                     \
                n_x = *(reinterpret_cast<const uint32_t RANS_GLOBAL *>(cbase + n_off) + lane); \
                StreamWindow::prefetch(chunk_rsrc(cbase, cbytes16, n_off, n_len), lane, n_b0, n_b1); \
            } else if (lane == 0) {                                                         \
                atomicAdd(p.err_count, 1ull);                                               \
            }                                                                               \
        }                                                                                   \
    } while (0)

    // the first chunk: all three in a row (and again after a chunk whose index entry was rejected)
    for (;;) {
        RANS_STEP_CLAIM();
        RANS_STEP_DATA();
        if (n_ok || !n_any)
            break;
    }
    while (n_ok) {
        // ---- the chunk that was being taken becomes the current one ----
        // (everything about the chunk is wave-uniform by construction; saying so keeps it, and the descriptors
        // built from it, in SGPRs whatever the compiler concluded about the loop-carried n_* copies)
        const uint64_t c_idx = uniform64(n_idx);
        const uint32_t c_skip = uniform((uint32_t)n_off & 15u);
        const uint32_t c_len = uniform(n_len) + c_skip; // (where the cursor must end up, counted from the first granule)
        StreamWindow W;
        W.install(ring, chunk_rsrc(cbase, cbytes16, n_off, n_len), c_skip + N * Tr::kStateBytes, lane, n_b0, n_b1);
        uint32_t x = n_x;
        const uint64_t first_sym = c_idx * p.chunk_syms;
        const uint32_t nsym = uniform((uint32_t)((p.n - first_sym) < p.chunk_syms ? (p.n - first_sym) : p.chunk_syms));
        uint8_t RANS_GLOBAL *dst = reinterpret_cast<uint8_t RANS_GLOBAL *>(reinterpret_cast<uint64_t>(p.out) + first_sym);
        const uint32_t rounds = uniform(nsym / N);
        const uint32_t tail = uniform(nsym - rounds * N);
        const uint32_t groups = rounds >> 2;
        rounds_done += rounds;
        n_any = false;
        n_ok = false;
        uint32_t stage = 0; // steps of the next chunk already issued

        if (groups) {
            const uint64_t dsta = reinterpret_cast<uint64_t>(dst);
            const u32x4 orsrc = {uniform((uint32_t)dsta), uniform((uint32_t)(dsta >> 32)) & 0xffffu,
                                 uniform(nsym), kRsrcFlags};
            uint32_t pa = 0;
            W.marker_rsrc = marker_rsrc(p, blockIdx.x * waves_per_block + wave);
            W.template checkpoint<true>(lane);
            decode_group_word<false>(x, pa, W.cur, m12, k65536, sel1, sel2, selA, selB, selC, orsrc, out_lane_off, 0u);
            // groups 1 .. groups-1 store their predecessor's symbols; osoff = 256 * (group - 1).  One scalar
            // compare per group watches for the point where the next step of the hand-over is due.
            const uint32_t oend = (groups - 1u) * 256u;
            uint32_t trigger = groups > kClaimAhead ? (groups - kClaimAhead) * 256u : 0u;
            for (uint32_t osoff = 0; osoff != oend; osoff += 256u) {
                if (osoff == trigger) {
                    if (stage == 0) {
                        RANS_STEP_CLAIM();
                        trigger += (kClaimAhead - kDataAhead) * 256u;
                    } else {
                        RANS_STEP_DATA();
                        trigger = ~0u;
                    }
                    ++stage;
                }
                W.template checkpoint<true>(lane);
                decode_group_word<true>(x, pa, W.cur, m12, k65536, sel1, sel2, selA, selB, selC, orsrc, out_lane_off, osoff);
            }
            const uint32_t v = quad_transpose(pa, sel1, sel2);
            __builtin_nontemporal_store(v, reinterpret_cast<uint32_t RANS_GLOBAL *>(dst + oend + out_lane_off));
        }
        // short chunks: whatever step of the hand-over has not been issued yet
        if (stage < 1)
            RANS_STEP_CLAIM();
        if (stage < 2)
            RANS_STEP_DATA();

        // ---- remaining full rounds and the partial tail round: element stores
        for (uint32_t r = groups << 2; r <= rounds; ++r) {
            const uint32_t cnt = (r < rounds) ? N : tail;
            if (cnt == 0)
                break;
            if (lane < cnt) {
                const uint32_t sy = dec_step<FMT_WORD>(T, x);
                dst[(uint64_t)r * N + lane] = (uint8_t)(sy >> 24);
            }
            W.checkpoint(lane);
            W.consume(dec_renorm<FMT_WORD>(W, x, lane < cnt));
        }

        // ---- integrity: every state back at L, cursor exactly at the end ----
        const bool all_good = __builtin_amdgcn_ballot_w64(x != Tr::kL) == 0 && W.position() == c_len;
        if (!all_good && lane == 0)
            atomicAdd(p.err_count, 1ull);

        // the next chunk's index entry was rejected (already counted): look further, one step after the other
        while (n_any && !n_ok) {
            RANS_STEP_CLAIM();
            RANS_STEP_DATA();
        }
    }
#undef RANS_STEP_CLAIM
#undef RANS_STEP_DATA
    if (p.trace && lane == 0) {
        unsigned long long *t = p.trace + (uint64_t)kTraceWords * ((uint64_t)blockIdx.x * waves_per_block + wave);
        t[0] = t_start;
        t[1] = wall_clock64();
        t[2] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
        t[3] = __builtin_readcyclecounter() - c_start;
        t[4] = rounds_done;
    }
    record_span(p, t_start, smem);
}

hipError_t launch_decode_word64(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const uint32_t t0 = (p.table0_bytes + 15u) & ~15u;
    const uint32_t waves = kWord64Threads / 64;
    const size_t lds = (size_t)t0 + (size_t)waves * kRingStride;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_decode_word64), 160 * 1024, lds_ok); e != hipSuccess)
        return e;
    const uint64_t want = (p.nchunks + waves - 1) / waves;
    const uint64_t cap = (uint64_t)num_cus * 2u;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    if (name)
        *name = "k_decode_word64";
    RANS_LAUNCH(k_decode_word64, dim3(grid), dim3(kWord64Threads), lds, stream, p);
    return hipGetLastError();
}

template <int FMT, int K, int OUT>
hipError_t launch_decode_t(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    // per-chunk models: every wave owns its tables (5 KiB) and window, nothing is shared -- workgroups of FOUR waves, five of
    // them per CU: 20 waves where one 16-wave workgroup held 16, and one wave per SIMD from every workgroup (one- and two-wave
    // workgroups spread unevenly over the CUs when the grid does not fill them: 0.84 / 0.80 ms against 0.73 for the word
    // format, profiles/r06_adaptive_decoder.md)
    constexpr uint32_t kAdaptDecThreads = 256;
    const uint32_t threads = kIsAdaptive<FMT> ? kAdaptDecThreads : kDecBlockThreads;
    const uint32_t waves = threads / 64;
    const uint32_t t0 = kIsAdaptive<FMT> ? waves * kAdaptDecWaveLds : (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1 = kIsAdaptive<FMT> ? 0u : (p.table1_bytes + 15u) & ~15u;
    if (kIsAdaptive<FMT> && (!p.chunk_freqs || p.scale_bits > kAdaptMaxScaleBits || p.scale_bits < 8 || (FMT == FMT_WORDA && p.scale_bits != 12)))
        return hipErrorInvalidValue;
    const size_t lds = (size_t)t0 + t1 + (size_t)waves * kRingStride;
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    auto kern = k_decode<FMT, K, OUT>;
    static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 160 * 1024, lds_ok); e != hipSuccess)
        return e;
    int blocks_per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
    if (kIsAdaptive<FMT> && threads < kDecBlockThreads) { // small workgroups: what the LDS allows, within 32 waves per CU
        blocks_per_cu = (int)((160 * 1024) / ((lds + 255) & ~(size_t)255));
        blocks_per_cu = blocks_per_cu * (int)waves > 32 ? 32 / (int)waves : blocks_per_cu;
    }
    uint64_t want = (p.nchunks + waves - 1) / waves;
    uint64_t cap = (uint64_t)num_cus * blocks_per_cu;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    if (name)
        *name = FMT == FMT_WORD ? "k_decode<word>" : FMT == FMT_BYTE ? "k_decode<byte>" : FMT == FMT_BYTEF ? "k_decode<byte, slot records>"
                : FMT == FMT_R64 ? "k_decode<r64>" : FMT == FMT_R64S ? "k_decode<r64 search>"
                : FMT == FMT_WORD16 ? "k_decode<word, u16 symbols>"
                : FMT == FMT_BYTEA ? "k_decode<byte, per-chunk models>" : FMT == FMT_WORDA ? "k_decode<word, per-chunk models>" : "k_decode<alias>";
    RANS_LAUNCH(kern, dim3(grid), dim3(threads), lds, stream, p);
    return hipGetLastError();
}

template <int FMT> hipError_t launch_decode_f(const DecParams &p, int num_cus, hipStream_t s, const char **name)
{
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.out) | (uintptr_t)p.chunk_syms) & 3u) == 0;
    const bool fast = aligned && p.sym_bytes == 1;
    if (aligned && p.sym_bytes == 2) {
        switch (p.n_ways) {
        case 64: return launch_decode_t<FMT, 1, OUT_FAST16>(p, num_cus, s, name);
        case 128: return launch_decode_t<FMT, 2, OUT_FAST16>(p, num_cus, s, name);
        case 256: return launch_decode_t<FMT, 4, OUT_FAST16>(p, num_cus, s, name);
        default: break;
        }
    }
    // (alternatives that were measured and lost -- compiler-scheduled renormalisation -2 %, output through an LDS tile -7 %,
    //  per-round byte stores -5 %, groups without the pipelined chunk hand-over, the byte format's byte stores: HISTORY.md,
    //  profiles/r04_byte_decoder_variants.log -- are no longer in the sources)
    if constexpr (FMT == FMT_WORD) {
        if (fast && p.n_ways == 64)
            return launch_decode_word64(p, num_cus, s, name);
    }
    switch (p.n_ways) {
    case 64:
        return fast ? launch_decode_t<FMT, 1, OUT_FAST8>(p, num_cus, s, name)
                    : launch_decode_t<FMT, 1, OUT_SLOW>(p, num_cus, s, name);
    case 128:
        return fast ? launch_decode_t<FMT, 2, OUT_FAST8>(p, num_cus, s, name)
                    : launch_decode_t<FMT, 2, OUT_SLOW>(p, num_cus, s, name);
    case 256:
        return fast ? launch_decode_t<FMT, 4, OUT_FAST8>(p, num_cus, s, name)
                    : launch_decode_t<FMT, 4, OUT_SLOW>(p, num_cus, s, name);
    case 512:
        return fast ? launch_decode_t<FMT, 8, OUT_FAST8>(p, num_cus, s, name)
                    : launch_decode_t<FMT, 8, OUT_SLOW>(p, num_cus, s, name);
    default:
        // any other lane count: K = ceil(N / 64) states per lane, the unused tail lanes idle
        if (p.n_ways >= 1 && p.n_ways < 64)
            return launch_decode_t<FMT, 1, OUT_SLOW>(p, num_cus, s, name);
        if (p.n_ways < 128)
            return launch_decode_t<FMT, 2, OUT_SLOW>(p, num_cus, s, name);
        if (p.n_ways < 256)
            return launch_decode_t<FMT, 4, OUT_SLOW>(p, num_cus, s, name);
        if (p.n_ways < 512)
            return launch_decode_t<FMT, 8, OUT_SLOW>(p, num_cus, s, name);
        return hipErrorInvalidValue;
    }
}


} // namespace

hipError_t launch_decode_wave(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    switch (format) {
    case FMT_WORD: return launch_decode_f<FMT_WORD>(p, num_cus, stream, name);
    case FMT_BYTE: return launch_decode_f<FMT_BYTE>(p, num_cus, stream, name);
    case FMT_BYTEF: return launch_decode_f<FMT_BYTEF>(p, num_cus, stream, name);
    case FMT_R64: return launch_decode_f<FMT_R64>(p, num_cus, stream, name);
    case FMT_BYTEA: return launch_decode_f<FMT_BYTEA>(p, num_cus, stream, name);
    case FMT_WORDA: return launch_decode_f<FMT_WORDA>(p, num_cus, stream, name);
    case FMT_WORD16: { // u16 symbols: paired-round stores for full waves, element stores otherwise
        const bool aligned = ((reinterpret_cast<uintptr_t>(p.out) | (uintptr_t)p.chunk_syms * 2u) & 3u) == 0;
        if (aligned && p.n_ways == 64)
            return launch_decode_t<FMT_WORD16, 1, OUT_FAST16>(p, num_cus, stream, name);
        if (aligned && p.n_ways == 128)
            return launch_decode_t<FMT_WORD16, 2, OUT_FAST16>(p, num_cus, stream, name);
        if (p.n_ways >= 1 && p.n_ways <= 64)
            return launch_decode_t<FMT_WORD16, 1, OUT_SLOW>(p, num_cus, stream, name);
        if (p.n_ways <= 128)
            return launch_decode_t<FMT_WORD16, 2, OUT_SLOW>(p, num_cus, stream, name);
        if (p.n_ways <= 256)
            return launch_decode_t<FMT_WORD16, 4, OUT_SLOW>(p, num_cus, stream, name);
        if (p.n_ways <= 512)
            return launch_decode_t<FMT_WORD16, 8, OUT_SLOW>(p, num_cus, stream, name);
        return hipErrorInvalidValue;
    }
    case FMT_R64S: // the search decoder exists in its general form only (any N, element stores)
        if (p.n_ways >= 1 && p.n_ways <= 64)
            return launch_decode_t<FMT_R64S, 1, OUT_SLOW>(p, num_cus, stream, name);
        if (p.n_ways <= 128)
            return launch_decode_t<FMT_R64S, 2, OUT_SLOW>(p, num_cus, stream, name);
        if (p.n_ways <= 256)
            return launch_decode_t<FMT_R64S, 4, OUT_SLOW>(p, num_cus, stream, name);
        if (p.n_ways <= 512)
            return launch_decode_t<FMT_R64S, 8, OUT_SLOW>(p, num_cus, stream, name);
        return hipErrorInvalidValue;
    case FMT_ALIAS: return launch_decode_f<FMT_ALIAS>(p, num_cus, stream, name);
    default: return hipErrorInvalidValue;
    }
}

} // namespace rans_amd
