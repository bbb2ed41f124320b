// decode_dual.hip -- TWO chunks per wavefront for the 64-way decoders of the byte-stream formats.
//
// The reference hides the latency of its table lookups by running two independent decoders in one loop
// (main.cpp:259-280: two scalar states; main_simd.cpp:313-325: two 4-lane SIMD decoders).  The wave-per-chunk
// decoder of decode_wave.hip relies on occupancy for that -- eight waves per SIMD -- which a model with large
// tables cannot have: the 4096-symbol alias tables (config 4) leave room for one 16-wave block per CU, four waves
// per SIMD, each a chain of two dependent LDS gathers and one stream read per round.  Here every wave owns a PAIR of
// consecutive chunks (2j, 2j + 1): two states per lane, two stream windows, the D steps of both issued back to back
// and the renormalisation reads of both in flight together -- twice the independent work per wave at the same
// occupancy.  Two more things change against k_decode<alias>:
//   * alias tables in the FMT_ALIAS2 form (device_common.hpp): 9 VALU instructions per D step instead of 11, and the
//     symbol is the low half of the record's first word;
//   * u16 symbols are stored per round (buffer_store_short, 128 contiguous bytes per wave) -- no packing of two
//     rounds, no exchange between lane pairs: these decoders are bound by VALU issue (~4.3 cycles per wave64
//     instruction), not by their stores.
// Config 4 (512 Mi u16 symbols): 0.468 -> 0.400 ms (0.445 -> 0.52 of the HBM roofline).  Where the tables are small enough
// for two blocks per CU (byte format, 256-symbol alias models) eight waves per SIMD with one chunk each beat four with
// two -- 0.517 against 0.585 ms for the byte format: the hardware interleaves eight instruction streams better than a
// fixed pairing does -- so api.cpp sends only models whose tables leave no room for a second block here.
// Tried and dropped: reading both renormalisation bytes of a lane with ONE ds_read_u16 at the lane's (odd or even) stream
// position.  gfx950 executes the unaligned LDS access correctly -- the whole GPU suite passed with it -- and four times
// slower: 1.13 ms for config 4, 2.15 ms for the byte format (profiles/r03_dual_decoder.md).  Two ds_read_u8 it is.
// Pairs whose two chunks are not both full-sized (the container's last chunk, an odd chunk count, a rejected index
// entry) go through decode_single(): the plain round loop with element stores.
#include "decode_common.hpp"

namespace rans_amd {

namespace {

// Renormalisation of the byte stream formats (rans_byte.h:307-318) for two chunks at once, full waves.  Per chunk:
// mask 1 = lanes with x < 2^23 (at least one byte), mask 2 = lanes with x < 2^15 (two); lane i's bytes are adjacent
// in the stream, at cursor + (bytes taken by the lanes below it); the first byte is the more significant one.
// Both bytes are read under the first mask (the second read of a one-byte lane is dropped by the exec mask of its
// merge, as in renorm_byte_full); the four reads of the two chunks are in flight together.
// 18 VALU + 4 LDS + 15 SALU for two chunk-rounds; s[52:59] are scratch.
__device__ __forceinline__ void renorm_byte_dual(uint32_t &xa, uint32_t &cura, uint32_t &xb, uint32_t &curb,
                                                 uint32_t k2p23, uint32_t k2p15)
{
    uint32_t ta, tb, wa, wb, c;
    uint32_t wa1, wb1;
    asm volatile("v_cmp_gt_u32_e64 s[52:53], %[l23], %[xa]\n\t"
                 "v_cmp_gt_u32_e64 s[54:55], %[l15], %[xa]\n\t"
                 "v_cmp_gt_u32_e64 s[56:57], %[l23], %[xb]\n\t"
                 "v_cmp_gt_u32_e64 s[58:59], %[l15], %[xb]\n\t"
                 "v_mbcnt_lo_u32_b32 %[ta], s52, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[ta], s53, %[ta]\n\t"
                 "v_mbcnt_lo_u32_b32 %[ta], s54, %[ta]\n\t"
                 "v_mbcnt_hi_u32_b32 %[ta], s55, %[ta]\n\t"
                 "v_add_u32_e32 %[ta], %[cura], %[ta]\n\t"
                 "v_mbcnt_lo_u32_b32 %[tb], s56, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[tb], s57, %[tb]\n\t"
                 "v_mbcnt_lo_u32_b32 %[tb], s58, %[tb]\n\t"
                 "v_mbcnt_hi_u32_b32 %[tb], s59, %[tb]\n\t"
                 "v_add_u32_e32 %[tb], %[curb], %[tb]\n\t"
                 "s_mov_b64 exec, s[52:53]\n\t"
                 "ds_read_u8 %[wa], %[ta]\n\t"
                 "ds_read_u8 %[wa1], %[ta] offset:1\n\t"
                 "s_mov_b64 exec, s[56:57]\n\t"
                 "ds_read_u8 %[wb], %[tb]\n\t"
                 "ds_read_u8 %[wb1], %[tb] offset:1\n\t"
                 "s_bcnt1_i32_b64 %[c], s[52:53]\n\t"
                 "s_add_i32 %[cura], %[cura], %[c]\n\t"
                 "s_bcnt1_i32_b64 %[c], s[54:55]\n\t"
                 "s_add_i32 %[cura], %[cura], %[c]\n\t"
                 "s_bcnt1_i32_b64 %[c], s[56:57]\n\t"
                 "s_add_i32 %[curb], %[curb], %[c]\n\t"
                 "s_bcnt1_i32_b64 %[c], s[58:59]\n\t"
                 "s_add_i32 %[curb], %[curb], %[c]\n\t"
                 "s_mov_b64 exec, s[52:53]\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t"
                 "v_lshl_or_b32 %[xa], %[xa], 8, %[wa]\n\t"
                 "s_mov_b64 exec, s[54:55]\n\t"
                 "v_lshl_or_b32 %[xa], %[xa], 8, %[wa1]\n\t"
                 "s_mov_b64 exec, s[56:57]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_lshl_or_b32 %[xb], %[xb], 8, %[wb]\n\t"
                 "s_mov_b64 exec, s[58:59]\n\t"
                 "v_lshl_or_b32 %[xb], %[xb], 8, %[wb1]\n\t"
                 "s_mov_b64 exec, -1"
                 : [xa] "+v"(xa), [xb] "+v"(xb), [ta] "=&v"(ta), [tb] "=&v"(tb), [wa] "=&v"(wa), [wb] "=&v"(wb),
                   [wa1] "=&v"(wa1), [wb1] "=&v"(wb1), [c] "=&s"(c), [cura] "+s"(cura), [curb] "+s"(curb)
                 : [l23] "v"(k2p23), [l15] "v"(k2p15)
                 : "vcc", "scc", "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59");
}

// One chunk the plain way: any symbol count, element stores (rans_byte.h:307-318 round by round).  64-way.
template <int FMT>
__device__ __forceinline__ void decode_single(const DecParams &p, const DecTables<FMT> &T, uint8_t *ring, uint64_t chunk,
                                              uint64_t off, uint32_t len, uint32_t lane, uint32_t &rounds_done)
{
    using Tr = FmtTraits<FMT>;
    constexpr uint32_t N = 64;
    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t first = chunk * p.chunk_syms;
    const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
    uint8_t RANS_GLOBAL *dst = reinterpret_cast<uint8_t RANS_GLOBAL *>(reinterpret_cast<uint64_t>(p.out) + first * p.sym_bytes);
    const uint64_t src = cbase + off;
    uint32_t x = *(reinterpret_cast<const uint32_t RANS_GLOBAL *>(src) + lane); // RansDecInit order: lane 0's state first
    StreamWindow W;
    // (a chunk may start at any byte: the window fetches whole granules from the one that holds its first byte, see k_decode)
    const uint32_t skip = (uint32_t)off & 15u;
    const uint64_t room = ((p.container_bytes + 15u) & ~uint64_t(15)) - (off - skip);
    const uint32_t climit = (skip + len + 15u) & ~15u;
    W.open(ring, src - skip, skip + N * Tr::kStateBytes, climit < room ? climit : (uint32_t)room, lane);
    const uint32_t rounds = uniform(nsym / N);
    const uint32_t tail = uniform(nsym - rounds * N);
    rounds_done += rounds;
    for (uint32_t r = 0; r <= rounds; ++r) {
        const uint32_t cnt = (r < rounds) ? N : tail;
        if (cnt == 0)
            break;
        uint8_t RANS_GLOBAL *rdst = dst + (uint64_t)r * N * p.sym_bytes;
        if (lane < cnt) {
            uint32_t s = dec_step<FMT>(T, x);
            if constexpr (Tr::kSymByte == 3)
                s >>= 24;
            if (p.sym_bytes == 1)
                rdst[lane] = (uint8_t)s;
            else
                reinterpret_cast<uint16_t RANS_GLOBAL *>(rdst)[lane] = (uint16_t)s;
        }
        W.checkpoint(lane);
        W.consume(dec_renorm<FMT>(W, x, lane < cnt));
    }
    const bool all_good = __builtin_amdgcn_ballot_w64(x != Tr::kL) == 0 && W.position() == skip + len;
    if (!all_good && lane == 0)
        atomicAdd(p.err_count, 1ull);
}

// SYM16: u16 symbols (one buffer_store_short per chunk and round); else u8 symbols, four rounds transposed in
// registers, one buffer_store_dword per chunk and group.
template <int FMT, bool SYM16>
__global__ void __launch_bounds__(kDecBlockThreads, 4) k_decode_dual(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    constexpr uint32_t N = 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned long long t_start = (p.trace || p.span) ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? __builtin_readcyclecounter() : 0ull;
    uint32_t rounds_done = 0;

    // ---- tables into LDS (once per block).  The alias forms keep table 1 (the own-slot counts, addressed by the bare
    // bucket number) at LDS address 0 and the records behind it; the byte format keeps cum2sym (table 0) first, as
    // dec_step<FMT_BYTE> expects.
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    constexpr bool kT1First = kIsAlias2<FMT>;
    uint8_t *lt0 = smem + (kT1First ? t1_bytes : 0u);
    uint8_t *lt1 = smem + (kT1First ? 0u : t0_bytes);
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(lt0);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
        const uint4 *g1 = reinterpret_cast<const uint4 *>(p.table1);
        uint4 *l1 = reinterpret_cast<uint4 *>(lt1);
        for (uint32_t i = threadIdx.x; i < t1_bytes / 16u; i += blockDim.x)
            l1[i] = g1[i];
    }
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    DecTables<FMT> T;
    T.init(lt0, lt1, p.scale_bits, p.log2nsyms);
    if (!lds_starts_at_zero(smem)) {
        if (threadIdx.x == 0)
            atomicAdd(p.err_count, 1ull << 32);
        return;
    }
    uint8_t *ring_a = smem + t0_bytes + t1_bytes + (2u * wave) * kRingStride;
    uint8_t *ring_b = ring_a + kRingStride;
    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t cbytes16 = (p.container_bytes + 15u) & ~uint64_t(15);

    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t out_lane_off = SYM16 ? lane * 2u : (lane & 3u) * N + (lane & ~3u);
    const uint32_t k2p23 = (1u << 23) + (lane >> 6), k2p15 = (1u << 15) + (lane >> 6); // VGPRs holding the limits

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u;
    if (p.span_reset && blockIdx.x == 0 && threadIdx.x < 2)
        p.span_reset[threadIdx.x] = 0ull;

    // pairs are handed out like k_decode's chunks: one counter per pool of blocks, or static striding
    const uint64_t npairs = (p.nchunks + 1u) >> 1;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t pair_v = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    // a full-sized chunk whose rounds fill whole store groups takes the fast loop
    constexpr uint32_t kGroupRounds = 4u; // (u8: four rounds per stored dword; u16: four rounds per window checkpoint)
    const bool shape_ok = (p.chunk_syms % (N * kGroupRounds)) == 0;
    for (;;) {
        if (p.work_counter) {
            uint32_t got = 0;
            if (lane == 0)
                got = atomicAdd(p.work_counter + pool * kWorkPoolStride, 1u);
            pair_v = (uint64_t)uniform(got) * npools + pool;
        }
        if (pair_v >= npairs)
            break;
        const uint64_t ca = uniform64(pair_v) * 2u, cb = ca + 1u;
        pair_v += total_waves;
        const bool has_b = cb < p.nchunks;
        const uint64_t off_a = uniform64(p.offsets[ca]);
        const uint32_t len_a = uniform(p.lengths[ca]);
        const uint64_t off_b = has_b ? uniform64(p.offsets[cb]) : 0ull;
        const uint32_t len_b = has_b ? uniform(p.lengths[cb]) : 0u;
        // off and len come from the caller's index: compare without forming off + len (which can wrap)
        const bool ok_a = (len_a >= N * Tr::kStateBytes) && (off_a <= p.container_bytes) &&
                          (len_a <= p.container_bytes - off_a);
        const bool ok_b = has_b && (len_b >= N * Tr::kStateBytes) && (off_b <= p.container_bytes) &&
                          (len_b <= p.container_bytes - off_b);
        if ((!ok_a || (has_b && !ok_b)) && lane == 0)
            atomicAdd(p.err_count, (ok_a ? 0ull : 1ull) + ((has_b && !ok_b) ? 1ull : 0ull));
        const bool full_b = has_b && (p.n - cb * p.chunk_syms) >= p.chunk_syms; // (chunk a of a pair with a chunk b is full)
        if (!(ok_a && ok_b && full_b && shape_ok)) { // wave-uniform
            if (ok_a)
                decode_single<FMT>(p, T, ring_a, ca, off_a, len_a, lane, rounds_done);
            if (ok_b)
                decode_single<FMT>(p, T, ring_b, cb, off_b, len_b, lane, rounds_done);
            continue;
        }

        // ---- both chunks, round by round --------------------------------------------------------------------------
        const uint64_t src_a = cbase + off_a, src_b = cbase + off_b;
        uint32_t xa = *(reinterpret_cast<const uint32_t RANS_GLOBAL *>(src_a) + lane); // RansDecInit order (main.cpp:261-262)
        uint32_t xb = *(reinterpret_cast<const uint32_t RANS_GLOBAL *>(src_b) + lane);
        StreamWindow Wa, Wb;
        const uint32_t skip_a = (uint32_t)off_a & 15u, skip_b = (uint32_t)off_b & 15u;
        {
            // (byte streams start anywhere: whole granules from the one that holds the chunk's first byte, see k_decode)
            const uint64_t room_a = cbytes16 - (off_a - skip_a), room_b = cbytes16 - (off_b - skip_b);
            const uint32_t cl_a = (skip_a + len_a + 15u) & ~15u, cl_b = (skip_b + len_b + 15u) & ~15u;
            Wa.open(ring_a, src_a - skip_a, skip_a + N * Tr::kStateBytes, cl_a < room_a ? cl_a : (uint32_t)room_a, lane);
            Wb.open(ring_b, src_b - skip_b, skip_b + N * Tr::kStateBytes, cl_b < room_b ? cl_b : (uint32_t)room_b, lane);
        }
        const uint32_t rounds = uniform(p.chunk_syms / N);
        rounds_done += 2u * rounds;
        const uint64_t dsta = reinterpret_cast<uint64_t>(p.out) + ca * p.chunk_syms * p.sym_bytes;
        const uint32_t chunk_bytes = p.chunk_syms * p.sym_bytes;
        // symbol stores go through descriptors of the two chunks' outputs with the running offset in an SGPR
        const rsrc_t orsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(uniform64(dsta)), 0, chunk_bytes, kRsrcFlags);
        const rsrc_t orsrc_b =
            __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(uniform64(dsta + chunk_bytes)), 0, chunk_bytes, kRsrcFlags);
        uint32_t osoff = 0;
        if constexpr (SYM16) {
            // a checkpoint every 4 rounds: at most 4 x 128 = kMaxAdvance bytes are consumed in between
            for (uint32_t r = 0; r < rounds; r += 4) {
                Wa.checkpoint(lane);
                Wb.checkpoint(lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t sa = dec_step<FMT>(T, xa);
                    const uint32_t sb = dec_step<FMT>(T, xb);
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)sa, orsrc_a, out_lane_off, osoff, kAuxStore);
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)sb, orsrc_b, out_lane_off, osoff, kAuxStore);
                    renorm_byte_dual(xa, Wa.cur, xb, Wb.cur, k2p23, k2p15);
                    osoff += 2u * N;
                }
            }
        } else {
            const uint32_t groups = rounds >> 2;
            for (uint32_t g = 0; g < groups; ++g) {
                Wa.checkpoint(lane);
                Wb.checkpoint(lane);
                uint32_t acc_a = 0, acc_b = 0;
#define RANS_DUAL_ROUND(J)                                          \
    acc_a = acc_symbol<Tr::kSymByte, J>(dec_step<FMT>(T, xa), acc_a); \
    acc_b = acc_symbol<Tr::kSymByte, J>(dec_step<FMT>(T, xb), acc_b); \
    renorm_byte_dual(xa, Wa.cur, xb, Wb.cur, k2p23, k2p15);
                RANS_DUAL_ROUND(0)
                RANS_DUAL_ROUND(1)
                RANS_DUAL_ROUND(2)
                RANS_DUAL_ROUND(3)
#undef RANS_DUAL_ROUND
                __builtin_amdgcn_raw_buffer_store_b32(quad_transpose(acc_a, sel1, sel2), orsrc_a, out_lane_off, osoff, kAuxStore);
                __builtin_amdgcn_raw_buffer_store_b32(quad_transpose(acc_b, sel1, sel2), orsrc_b, out_lane_off, osoff, kAuxStore);
                osoff += 4u * N;
            }
        }
        // ---- integrity: every state back at L, both cursors exactly at their chunk's end ----
        const bool good_a = __builtin_amdgcn_ballot_w64(xa != Tr::kL) == 0 && Wa.position() == skip_a + len_a;
        const bool good_b = __builtin_amdgcn_ballot_w64(xb != Tr::kL) == 0 && Wb.position() == skip_b + len_b;
        if ((!good_a || !good_b) && lane == 0)
            atomicAdd(p.err_count, (good_a ? 0ull : 1ull) + (good_b ? 0ull : 1ull));
    }
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

template <int FMT, bool SYM16>
hipError_t launch_dual_t(const DecParams &p, int num_cus, hipStream_t stream, const char **name, const char *kname)
{
    const uint32_t waves = kDecBlockThreads / 64;
    const uint32_t t0 = (p.table0_bytes + 15u) & ~15u, t1 = (p.table1_bytes + 15u) & ~15u;
    const size_t lds = (size_t)t0 + t1 + (size_t)waves * 2u * kRingStride;
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    auto kern = k_decode_dual<FMT, SYM16>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 160 * 1024, lds_ok); e != hipSuccess)
        return e;
    const uint64_t npairs = (p.nchunks + 1u) >> 1;
    const uint64_t want = (npairs + waves - 1) / waves;
    const uint64_t cap = (uint64_t)num_cus; // one block per CU
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    if (name)
        *name = kname;
    RANS_LAUNCH(kern, dim3(grid), dim3(kDecBlockThreads), lds, stream, p);
    return hipGetLastError();
}

} // namespace

bool decode_dual_fits(uint32_t table0_bytes, uint32_t table1_bytes)
{
    const size_t t0 = (table0_bytes + 15u) & ~15u, t1 = (table1_bytes + 15u) & ~15u;
    return t0 + t1 + (size_t)(kDecBlockThreads / 64) * 2u * kRingStride <= 160 * 1024;
}

// format: kKernelFormatAlias2 / kKernelFormatAlias2W (tables in the FMT_ALIAS2 form).
// The caller has checked n_ways == 64 and the alignment of the output.
hipError_t launch_decode_dual(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const bool sym16 = p.sym_bytes == 2;
    switch (format) {
    case FMT_ALIAS2:
        return sym16 ? launch_dual_t<FMT_ALIAS2, true>(p, num_cus, stream, name, "k_decode_dual<alias>")
                     : launch_dual_t<FMT_ALIAS2, false>(p, num_cus, stream, name, "k_decode_dual<alias>");
    case FMT_ALIAS2W:
        return sym16 ? launch_dual_t<FMT_ALIAS2W, true>(p, num_cus, stream, name, "k_decode_dual<alias>")
                     : launch_dual_t<FMT_ALIAS2W, false>(p, num_cus, stream, name, "k_decode_dual<alias>");
    default:
        return hipErrorInvalidValue;
    }
}

} // namespace rans_amd
