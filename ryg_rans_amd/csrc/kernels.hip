// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for interleaved rANS.
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

#include "kernels.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ryg_rans_amd.h"
#include "model.h"

namespace rans_amd {

namespace {

constexpr int FMT_BYTE = RANS_AMD_FMT_BYTE;
constexpr int FMT_WORD = RANS_AMD_FMT_WORD;
constexpr int FMT_R64 = RANS_AMD_FMT_R64;
constexpr int FMT_ALIAS = RANS_AMD_FMT_ALIAS;

// OUT_SLOW: element stores (any N, any alignment, u16 symbols).  OUT_FAST8: 4 rounds of u8
// symbols transposed in registers.  OUT_FAST8_NOASM: same with the compiler-scheduled renorm
// (A/B knob).  OUT_FAST8_LDS: symbols staged through a 256-byte LDS tile per wave
// (ds_write_b8 per round, one ds_read_b32 + global_store_dword per 4 rounds; K == 1 only).
// OUT_FAST16: u16 symbols, 2 rounds packed per dword and swapped between lane pairs.
// OUT_FAST8_BYTE: one global_store_byte per lane and round (64 contiguous bytes per wave), no transpose.
enum OutMode { OUT_SLOW = 0, OUT_FAST8 = 1, OUT_FAST8_NOASM = 2, OUT_FAST8_LDS = 3, OUT_FAST16 = 4, OUT_FAST8_BYTE = 5 };
constexpr uint32_t kOutTileBytes = 256;

template <int FMT> struct FmtTraits;
template <> struct FmtTraits<FMT_WORD> {
    using state_t = uint32_t;
    static constexpr uint32_t kUnit = 2, kStateBytes = 4;
    static constexpr uint32_t kL = 1u << 16; // rans_word_sse41.h:35
    static constexpr int kSymByte = 3;        // WordSlot.lo keeps the symbol in its top byte
};
template <> struct FmtTraits<FMT_BYTE> {
    using state_t = uint32_t;
    static constexpr uint32_t kUnit = 1, kStateBytes = 4;
    static constexpr uint32_t kL = 1u << 23; // rans_byte.h:50
    static constexpr int kSymByte = 0;
};
template <> struct FmtTraits<FMT_ALIAS> {
    using state_t = uint32_t;
    static constexpr uint32_t kUnit = 1, kStateBytes = 4;
    static constexpr uint32_t kL = 1u << 23;
    static constexpr int kSymByte = 0;
};
template <> struct FmtTraits<FMT_R64> {
    using state_t = uint64_t;
    static constexpr uint32_t kUnit = 4, kStateBytes = 8;
    static constexpr uint64_t kL = 1ull << 31; // rans64.h:59
    static constexpr int kSymByte = 0;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of m strictly below this lane
__device__ __forceinline__ uint32_t rank_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return (uint64_t)uniform((uint32_t)v) | ((uint64_t)uniform((uint32_t)(v >> 32)) << 32);
}

// explicit global address space: keeps loads/stores as global_* (not flat_*)
#define RANS_GLOBAL __attribute__((address_space(1)))
typedef const u32x4 RANS_GLOBAL *gvec_cptr;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// quad_perm DPP: lane i of each quad reads lane P[i]
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
}

// ---------------------------------------------------------------------------
// Stream window: a 2 KiB ring per wave in LDS, filled in aligned 1 KiB blocks
// (one global_load_dwordx4 per lane), the next block prefetched in registers.
//
//   rd    ring offset of the read cursor at the last checkpoint (< 2048)
//   adv   bytes consumed since that checkpoint
//   avail valid bytes ahead of rd at the checkpoint
//   wr    ring offset of the block that is written next (0 or 1024)
//
// checkpoint() folds adv into rd/avail and, when a block is free (avail <=
// 1024), writes the prefetched block and issues the next fetch.  Between two
// checkpoints the decoder consumes at most kMaxAdvance bytes and reads at most
// one more sub-step beyond that, so addresses never wrap between checkpoints:
// the first kRingMirror bytes of the ring are mirrored behind its end.
// Everything here is wave-uniform (SGPRs); the refill branch is a scalar branch.
// ---------------------------------------------------------------------------
constexpr uint32_t kMaxAdvance = 512;
static_assert(kRingMirror >= kMaxAdvance + 256, "mirror must cover one checkpoint interval plus one sub-step");

struct StreamWindow {
    uint8_t *ring;      // LDS, wave-private
    uint32_t ring_addr; // the same as a raw LDS byte address (for the asm path)
    uint64_t gnext;     // global address of the next 1 KiB block to fetch
    uint64_t glimit;    // 16-byte aligned end of what may be fetched for this chunk
    uint32_t rd, adv, avail, wr;
    u32x4 pre; // prefetched block (16 B per lane)

    __device__ __forceinline__ u32x4 fetch(uint32_t lane)
    {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (gnext + kRingBlock <= glimit) { // whole block readable: wave-uniform fast path
            gvec_cptr g = reinterpret_cast<gvec_cptr>(gnext);
            v = __builtin_nontemporal_load(g + lane);
        } else if (gnext + lane * 16u < glimit) {
            gvec_cptr g = reinterpret_cast<gvec_cptr>(gnext);
            v = __builtin_nontemporal_load(g + lane);
        }
        gnext += kRingBlock;
        return v;
    }
    __device__ __forceinline__ void put(uint32_t lane, uint32_t at, const u32x4 &v)
    {
        *reinterpret_cast<u32x4 *>(ring + at + lane * 16u) = v;
        if (at == 0 && lane < kRingMirror / 16u)
            *reinterpret_cast<u32x4 *>(ring + kRingBytes + lane * 16u) = v;
    }
    __device__ __forceinline__ void open(uint8_t *lds, uint64_t gaddr, uint64_t limit, uint32_t lane)
    {
        ring = lds;
        ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
        glimit = limit;
        gnext = gaddr & ~uint64_t(15);
        rd = (uint32_t)(gaddr & 15u);
        adv = 0;
        u32x4 b0 = fetch(lane);
        u32x4 b1 = fetch(lane);
        pre = fetch(lane);
        put(lane, 0, b0);
        put(lane, kRingBlock, b1);
        wr = 0;
        avail = kRingBytes - rd;
    }
    __device__ __forceinline__ void checkpoint(uint32_t lane)
    {
        rd = (rd + adv) & (kRingBytes - 1);
        avail -= adv;
        adv = 0;
        if (avail <= kRingBlock) {
            put(lane, wr, pre);
            wr ^= kRingBlock;
            avail += kRingBlock;
            pre = fetch(lane);
        }
    }
    // ring offset / LDS address of the read cursor (no wrap between checkpoints)
    __device__ __forceinline__ uint32_t cursor() const { return rd + adv; }
    __device__ __forceinline__ uint32_t cursor_addr() const { return ring_addr + rd + adv; }
    __device__ __forceinline__ void consume(uint32_t bytes) { adv += bytes; }
};

// ---------------------------------------------------------------------------
// D step: symbol lookup + state update, no stream access.
// Returns a word whose byte FmtTraits::kSymByte (u8 alphabets) or low 16 bits
// hold the symbol.
// ---------------------------------------------------------------------------
template <int FMT> struct DecTables {
    const uint8_t *t0; // LDS
    const uint8_t *t1; // LDS
    uint32_t scale_bits;
    uint32_t mask;
    uint32_t bucket_shift; // alias: scale_bits - log2(nsyms)
    uint32_t mask12v;      // 0xfff held in a VGPR (a literal operand makes v_and a 3.5-cycle op)
    // VGPR copies of mask / scale_bits / bucket_shift: a VALU and/shift with an SGPR operand issues in
    // 4.7 cycles, with VGPR operands in 2.7 (profiles/r01_ubench.log)
    uint32_t maskv, sbv, bshiftv;

    __device__ __forceinline__ void init(const uint8_t *table0, const uint8_t *table1, uint32_t sb, uint32_t log2nsyms)
    {
        t0 = table0;
        t1 = table1;
        scale_bits = sb;
        mask = (1u << sb) - 1u;
        bucket_shift = sb - log2nsyms;
        mask12v = 0xfffu;
        maskv = mask;
        sbv = sb;
        bshiftv = bucket_shift;
        asm volatile("v_mov_b32 %0, %0" : "+v"(mask12v)); // opaque: keep them in VGPRs
        asm volatile("v_mov_b32 %0, %0" : "+v"(maskv));
        asm volatile("v_mov_b32 %0, %0" : "+v"(sbv));
        asm volatile("v_mov_b32 %0, %0" : "+v"(bshiftv));
    }
};

template <int FMT>
__device__ __forceinline__ uint32_t dec_step(const DecTables<FMT> &T, typename FmtTraits<FMT>::state_t &x)
{
    if constexpr (FMT == FMT_WORD) {
        // rans_word_sse41.h:123-131 / :151-179: slot = x & 4095;
        // x = freq * (x >> 12) + bias.  freq < 2^12 and x >> 12 < 2^20, so the
        // 24-bit multiply-add is exact.
        const uint2 e = reinterpret_cast<const uint2 *>(T.t0)[x & T.mask12v];
        x = (e.x & 0xffffffu) * (x >> 12) + e.y;
        return e.x;
    } else if constexpr (FMT == FMT_BYTE) {
        // rans_byte.h:125-128 (get), :291-298 (step)
        const uint32_t cf = x & T.maskv;
        const uint32_t s = T.t0[cf];
        const uint2 r = reinterpret_cast<const uint2 *>(T.t1)[s]; // {freq, start}
        // freq <= 2^16 and x >> scale_bits < 2^23 (scale_bits >= 8): 24-bit multiply is exact
        x = (r.x & 0xffffffu) * ((x >> T.sbv) & 0xffffffu) + cf - r.y;
        return s;
    } else if constexpr (FMT == FMT_R64) {
        // rans64.h:118-121 (get), :286-292 (step)
        const uint32_t cf = (uint32_t)x & T.maskv;
        const uint32_t s = T.t0[cf];
        const uint2 r = reinterpret_cast<const uint2 *>(T.t1)[s];
        // freq * (x >> sb) + (cf - start) with x < 2^63: cf - start is in [0, freq), so it is a plain
        // 32-bit value; the high word of x >> sb is < 2^17 and freq <= 2^16, so its product is one
        // 24-bit multiply added to the high word -- one v_mad_u64_u32 instead of two plus a 64-bit
        // subtract-with-borrow
        const uint64_t xs = x >> T.scale_bits;
        const uint32_t bias = cf - r.y;
        x = (uint64_t)r.x * (uint32_t)xs + bias + ((uint64_t)__umul24(r.x, (uint32_t)(xs >> 32)) << 32);
        return s;
    } else {
        // main_alias.cpp:252-267; the subtraction wraps in 32 bits on purpose
        const uint32_t xm = x & T.maskv;
        const uint32_t bucket = xm >> T.bshiftv;
        const uint32_t div = reinterpret_cast<const uint32_t *>(T.t1)[bucket];
        // xm < div as the sign bit of the difference (both < 2^17): a compare + v_cndmask costs
        // ~27 issue cycles on gfx950, sub + shift 5.5
        const uint32_t below = (xm - div) >> 31;
        const uint32_t half = 2u * bucket + below;
        const uint2 e = reinterpret_cast<const uint2 *>(T.t0)[half]; // {freq | sym << 16, adjust}
        x = (e.x & 0xffffu) * ((x >> T.sbv) & 0xffffffu) + xm - e.y;
        return e.x >> 16;
    }
}

// ---------------------------------------------------------------------------
// Renormalisation of one sub-step (64 lanes, ascending lane order == ascending
// stream address).  `active` masks lanes that have no symbol in this round.
// Returns the bytes consumed (wave-uniform).
// ---------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ uint32_t dec_renorm(const StreamWindow &W, typename FmtTraits<FMT>::state_t &x,
                                               bool active)
{
    if constexpr (FMT == FMT_WORD) {
        // rans_word_sse41.h:134-141 / :182-227
        const bool need = active && x < (1u << 16);
        const uint64_t m = __builtin_amdgcn_ballot_w64(need);
        const uint32_t at = W.cursor() + 2u * rank_below(m);
        const uint32_t w = *reinterpret_cast<const uint16_t *>(W.ring + at);
        x = need ? ((x << 16) | w) : x;
        return 2u * (uint32_t)__builtin_popcountll(m);
    } else if constexpr (FMT == FMT_R64) {
        // rans64.h:305-316
        const bool need = active && x < (1ull << 31);
        const uint64_t m = __builtin_amdgcn_ballot_w64(need);
        const uint32_t at = W.cursor() + 4u * rank_below(m);
        const uint32_t w = *reinterpret_cast<const uint32_t *>(W.ring + at);
        x = need ? ((x << 32) | w) : x;
        return 4u * (uint32_t)__builtin_popcountll(m);
    } else {
        // rans_byte.h:307-318.  With scale_bits <= 16 a lane needs 0, 1 or 2
        // bytes: x >= 2^7 after D, and a second byte is needed iff x < 2^15.
        // The first byte read is the more significant one.
        const bool n1 = active && x < (1u << 23);
        const bool n2 = active && x < (1u << 15);
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(n1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(n2);
        const uint32_t at = W.cursor() + rank_below(m1) + rank_below(m2);
        const uint32_t b0 = W.ring[at];
        const uint32_t b1 = W.ring[at + 1];
        const uint32_t x1 = (x << 8) | b0;
        const uint32_t x2 = (x1 << 8) | b1;
        x = n2 ? x2 : (n1 ? x1 : x);
        return (uint32_t)__builtin_popcountll(m1) + (uint32_t)__builtin_popcountll(m2);
    }
}

// Hand-written renormalisation sub-step of the word format for a FULL wave (all 64
// lanes hold a state and are active): rans_word_sse41.h:134-141 for 64 lanes at once.
//   v_cmpx      lanes with x < 2^16 stay enabled; vcc = the same mask
//   v_mbcnt x2  rank of the lane among the enabled ones = its word index in the stream
//   ds_read_u16 only the enabled lanes read; v_perm merges (x << 16) | word
//   s_bcnt1     words consumed by the wave (returned)
// 5 VALU + 1 LDS + 2 SALU, no branch, no v_cndmask.  exec is restored to all ones,
// which is what it was (the caller runs this only in wave-uniform full-wave code).
__device__ __forceinline__ uint32_t renorm_word_full(uint32_t &x, uint32_t cursor_addr, uint32_t k65536)
{
    uint32_t t, w, cnt;
    // gfx940+ hazard: a VALU write of an SGPR/VCC needs 2 wait states before a VALU reads it
    // as an operand (LLVM GCNHazardRecognizer, VALUWriteSGPRVALURead); hipcc does not pad
    // inside an asm statement, hence the s_nop 1.
    asm volatile("v_cmpx_gt_u32_e32 vcc, %[lim], %[x]\n\t"
                 "s_nop 1\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "v_lshl_add_u32 %[t], %[t], 1, %[cur]\n\t"
                 "ds_read_u16 %[w], %[t]\n\t"
                 "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_perm_b32 %[x], %[x], %[w], %[sel]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [t] "=&v"(t), [w] "=&v"(w), [cnt] "=&s"(cnt)
                 : [lim] "v"(k65536), [cur] "s"(cursor_addr), [sel] "s"(0x05040100u)
                 : "vcc", "scc", "memory");
    return cnt;
}

// Same for the byte formats (rans_byte.h:307-318): a lane needs 0, 1 or 2 bytes
// (x < 2^23, x < 2^15); its offset in the stream is the sum of both masks' ranks; the
// first byte is the more significant one.  9 VALU + 2 LDS, no branch.  Returns bytes consumed.
__device__ __forceinline__ uint32_t renorm_byte_full(uint32_t &x, uint32_t cursor_addr, uint32_t k2p23, uint32_t k2p15)
{
    uint32_t t, b0, b1, c1, c2;
    uint64_t m1;
    asm volatile("v_cmp_gt_u32_e32 vcc, %[l23], %[x]\n\t"
                 "s_nop 1\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "s_mov_b64 %[m1], vcc\n\t"
                 "s_bcnt1_i32_b64 %[c1], vcc\n\t"
                 "v_cmp_gt_u32_e32 vcc, %[l15], %[x]\n\t"
                 "v_add_u32_e32 %[t], %[cur], %[t]\n\t"
                 "s_nop 0\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, %[t]\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "s_bcnt1_i32_b64 %[c2], vcc\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "ds_read_u8 %[b0], %[t]\n\t"
                 "s_mov_b64 exec, vcc\n\t"
                 "ds_read_u8 %[b1], %[t] offset:1\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "s_waitcnt lgkmcnt(1)\n\t"
                 "v_lshl_or_b32 %[x], %[x], 8, %[b0]\n\t"
                 "s_mov_b64 exec, vcc\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_lshl_or_b32 %[x], %[x], 8, %[b1]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [t] "=&v"(t), [b0] "=&v"(b0), [b1] "=&v"(b1), [c1] "=&s"(c1), [c2] "=&s"(c2),
                   [m1] "=&s"(m1)
                 : [l23] "v"(k2p23), [l15] "v"(k2p15), [cur] "s"(cursor_addr)
                 : "vcc", "scc", "memory");
    return c1 + c2;
}

// byte `kSymByte` of `raw` goes to byte J of acc, the other bytes of acc stay
template <int SYMBYTE, int J> __device__ __forceinline__ uint32_t acc_symbol(uint32_t raw, uint32_t acc)
{
    if constexpr (J == 0) {
        return raw; // fixed up by J == 1
    } else if constexpr (J == 1) {
        // byte0 <- acc[SYMBYTE] (round 0's symbol), byte1 <- raw[SYMBYTE]
        constexpr uint32_t sel = (uint32_t)SYMBYTE | ((4u + SYMBYTE) << 8) | 0x03020000u;
        return __builtin_amdgcn_perm(raw, acc, sel);
    } else {
        constexpr uint32_t ident = 0x03020100u;
        constexpr uint32_t sel = (ident & ~(0xffu << (8 * J))) | ((4u + SYMBYTE) << (8 * J));
        return __builtin_amdgcn_perm(raw, acc, sel);
    }
}

// 4x4 byte transpose inside each quad of lanes.  In: lane q of a quad holds the
// bytes of column (4j+q) for rows 0..3.  Out: lane q holds row q, columns
// 4j..4j+3, i.e. four consecutive output bytes.
__device__ __forceinline__ uint32_t quad_transpose(uint32_t v, uint32_t sel1, uint32_t sel2)
{
    // (the same shuffles through the LDS crossbar, ds_swizzle, measured 2 % slower: the LDS pipe is
    // the co-bottleneck of the decoder)
    uint32_t o = quad_perm<1, 0, 3, 2>(v);
    v = __builtin_amdgcn_perm(o, v, sel1);
    o = quad_perm<2, 3, 0, 1>(v);
    return __builtin_amdgcn_perm(o, v, sel2);
}

template <int FMT, int K, int OUT>
__global__ void __launch_bounds__(kDecBlockThreads, (K <= 2 ? 8 : 4)) k_decode(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;

    // ---- stage the tables into LDS (once per block) ----------------------
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    {
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
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);

    uint8_t *ring = smem + t0_bytes + t1_bytes + wave * kRingStride;
    uint8_t *tile = smem + t0_bytes + t1_bytes + waves_per_block * kRingStride + wave * kOutTileBytes; // OUT_FAST8_LDS
    static_assert(OUT != OUT_FAST8_LDS || K == 1, "the LDS output tile holds 4 rounds of 64 symbols");
    const uint32_t N = (OUT != OUT_SLOW) ? 64u * K : p.n_ways;
    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t glimit = (cbase + p.container_bytes + 15u) & ~uint64_t(15);

    // per-lane constants of the output transpose
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t out_lane_off = (lane & 3u) * N + (lane & ~3u);

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u;
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

        bool ok = ((off & 15u) == 0) && (len >= N * Tr::kStateBytes) && (off + len <= p.container_bytes);
        if (!ok) { // wave-uniform
            if (lane == 0)
                atomicAdd(p.err_count, 1ull);
            continue;
        }

        // ---- initial states: lane 0's first (RansDecInit order, main.cpp:261-262)
        state_t x[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = k * 64u + lane;
            x[k] = Tr::kL;
            if (idx < N) {
                if constexpr (FMT == FMT_R64) {
                    const u32x2 v = *(reinterpret_cast<const u32x2 RANS_GLOBAL *>(src) + idx);
                    x[k] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                } else {
                    x[k] = *(reinterpret_cast<const uint32_t RANS_GLOBAL *>(src) + idx);
                }
            }
        }

        StreamWindow W;
        // fetch nothing beyond this chunk's own stream (rounded up to the 16-byte granule)
        const uint64_t climit = (src + len + 15u) & ~uint64_t(15);
        W.open(ring, src + N * Tr::kStateBytes, climit < glimit ? climit : glimit, lane);
        uint32_t consumed = N * Tr::kStateBytes;

        const uint32_t rounds = uniform(nsym / N);
        const uint32_t tail = uniform(nsym - rounds * N);
        uint32_t r = 0;
        // sub-steps between two window checkpoints: at most kMaxAdvance bytes are consumed
        constexpr int kCheckEvery = (FMT == FMT_R64) ? 2 : 4;

        if constexpr (OUT == OUT_FAST16) {
            // ---- pairs of full rounds, u16 symbols: lane 2i ends up with round r's symbols of
            // lanes 2i,2i+1 and lane 2i+1 with round r+1's: one dword store per lane and pair
            const uint32_t pairs = rounds >> 1;
            uint8_t RANS_GLOBAL *gdst = dst;
            const uint32_t sel16 = (lane & 1u) ? 0x03020706u : 0x05040100u;
            const uint32_t lane_off16 = ((lane & 1u) * N + (lane & ~1u)) * 2u;
            const uint32_t k2p23 = (1u << 23) + (lane >> 6), k2p15 = (1u << 15) + (lane >> 6);
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
                        uint32_t c;
                        if constexpr (FMT == FMT_BYTE || FMT == FMT_ALIAS)
                            c = renorm_byte_full(x[k], W.cursor_addr(), k2p23, k2p15);
                        else
                            c = dec_renorm<FMT>(W, x[k], true);
                        W.consume(c);
                        consumed += c;
                    }
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t o = quad_perm<1, 0, 3, 2>(acc[k]);
                    const uint32_t v = __builtin_amdgcn_perm(o, acc[k], sel16);
                    *reinterpret_cast<uint32_t RANS_GLOBAL *>(gdst + (lane_off16 + k * 128u)) = v;
                }
                gdst += 4u * N;
            }
            r = pairs << 1;
        } else
        if constexpr (OUT != OUT_SLOW) {
            // ---- groups of 4 full rounds, symbols transposed in registers ----
            const uint32_t groups = rounds >> 2;
            uint8_t RANS_GLOBAL *gdst = dst;
            const uint32_t k65536 = 0x10000u + (lane >> 6); // VGPRs holding the renorm limits (lane < 64)
            const uint32_t k2p23 = (1u << 23) + (lane >> 6), k2p15 = (1u << 15) + (lane >> 6);
            for (uint32_t g = 0; g < groups; ++g) {
                uint32_t acc[K];
#define RANS_ROUND(J)                                                              \
    _Pragma("unroll") for (int k = 0; k < K; ++k) {                                \
        const uint32_t raw = dec_step<FMT>(T, x[k]);                               \
        if constexpr (OUT == OUT_FAST8_LDS)                                        \
            tile[J * 64 + lane] = (uint8_t)(raw >> (8 * Tr::kSymByte));            \
        else if constexpr (OUT == OUT_FAST8_BYTE)                                  \
            gdst[J * N + k * 64 + lane] = (uint8_t)(raw >> (8 * Tr::kSymByte));    \
        else                                                                       \
            acc[k] = acc_symbol<Tr::kSymByte, J>(raw, acc[k]);                     \
    }                                                                              \
    _Pragma("unroll") for (int k = 0; k < K; ++k) {                                \
        if ((J * K + k) % kCheckEvery == 0)                                        \
            W.checkpoint(lane);                                                    \
        uint32_t c;                                                                \
        if constexpr (FMT == FMT_WORD && (OUT == OUT_FAST8 || OUT == OUT_FAST8_LDS || OUT == OUT_FAST8_BYTE)) \
            c = 2u * renorm_word_full(x[k], W.cursor_addr(), k65536);              \
        else if constexpr ((FMT == FMT_BYTE || FMT == FMT_ALIAS) && (OUT == OUT_FAST8 || OUT == OUT_FAST8_BYTE)) \
            c = renorm_byte_full(x[k], W.cursor_addr(), k2p23, k2p15);             \
        else                                                                       \
            c = dec_renorm<FMT>(W, x[k], true);                                    \
        W.consume(c);                                                              \
        consumed += c;                                                             \
    }
                RANS_ROUND(0)
                RANS_ROUND(1)
                RANS_ROUND(2)
                RANS_ROUND(3)
#undef RANS_ROUND
                if constexpr (OUT == OUT_FAST8_LDS) {
                    // LDS ops of one wave execute in order: the read sees the four writes
                    const uint32_t v = reinterpret_cast<const uint32_t *>(tile)[lane];
                    *reinterpret_cast<uint32_t RANS_GLOBAL *>(gdst + lane * 4u) = v;
                } else if constexpr (OUT != OUT_FAST8_BYTE) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint32_t v = quad_transpose(acc[k], sel1, sel2);
                        *reinterpret_cast<uint32_t RANS_GLOBAL *>(gdst + (out_lane_off + k * 64u)) = v;
                    }
                }
                gdst += 4u * N;
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
                const uint32_t c = dec_renorm<FMT>(W, x[k], idx < cnt);
                W.consume(c);
                consumed += c;
            }
        }

        // ---- integrity: every state back at L, cursor exactly at the end ----
        bool good = true;
#pragma unroll
        for (int k = 0; k < K; ++k)
            good = good && (x[k] == Tr::kL);
        const bool all_good = __builtin_amdgcn_ballot_w64(!good) == 0 && consumed == len;
        if (!all_good && lane == 0)
            atomicAdd(p.err_count, 1ull);
    }
    if (p.trace && lane == 0) { // debug timeline: when did this wave start and stop, on which XCD
        unsigned long long *t = p.trace + 3ull * ((uint64_t)blockIdx.x * waves_per_block + wave);
        t[0] = t_start;
        t[1] = wall_clock64();
        t[2] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    }
}

// ===========================================================================
// Encoder: mirror image of the decoder.  One wave per chunk, symbols visited
// last round first; within a sub-step the lanes that must emit compact their
// units below the write cursor in ascending lane order (the decoder will read
// them back in exactly that order).
// ===========================================================================

template <int FMT> struct EncTables {
    const uint4 *recs; // LDS: EncRec {freq, start, rcp, remap}
    const uint32_t *alias_remap; // global
    uint32_t scale_bits;
    uint32_t nsyms;
};

// exact x / freq and x % freq from the 32-bit reciprocal floor(2^32 / freq):
// the estimate is never too large and at most 1 too small.
__device__ __forceinline__ void divmod_rcp(uint32_t x, uint32_t freq, uint32_t rcp, uint32_t &q, uint32_t &rem)
{
    q = __umulhi(x, rcp);
    // after renormalisation x < 2^(31-scale_bits) * freq (byte, scale_bits >= 8) or 2^20 * freq
    // (word), so q < 2^23 and freq <= 2^16: the 24-bit multiply (full rate) is exact
    rem = x - __umul24(q, freq);
    if (rem >= freq) {
        q += 1;
        rem -= freq;
    }
}

// 64-bit variant for rans64 (state < 2^63): Alverson reciprocal, exact for freq >= 2; freq == 1
// (rcp = 2^64 - 1, q = x - 1) is fixed by the correction step.  rec = {freq | rshift << 24, start,
// rcp lo, rcp hi} (model.cpp).
__device__ __forceinline__ void divmod_rcp64(uint64_t x, uint32_t freq, const uint4 &rec, uint64_t &q, uint64_t &rem)
{
    const uint64_t rcp = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
    q = __umul64hi(x, rcp) >> (rec.x >> 24);
    rem = x - q * freq;
    if (rem >= freq) {
        q += 1;
        rem -= freq;
    }
}

// One encoder sub-step for 64 lanes.  `wp` = write cursor (byte offset inside the
// slot, moves down, wave-uniform).
template <int FMT>
__device__ __forceinline__ void enc_substep(const EncTables<FMT> &T, typename FmtTraits<FMT>::state_t &x,
                                            uint32_t sym, bool active, uint8_t RANS_GLOBAL *slot, uint32_t &wp,
                                            bool &bad)
{
    const bool in_alphabet = sym < T.nsyms;
    const uint4 rec = T.recs[in_alphabet ? sym : 0u];
    const uint32_t freq = (FMT == FMT_R64) ? (rec.x & 0xffffffu) : rec.x, start = rec.y, rcp = rec.z;
    if (active && (!in_alphabet || freq == 0)) {
        bad = true;
        active = false;
    }

    if constexpr (FMT == FMT_WORD) {
        // rans_word_sse41.h:81-93
        const bool emit = active && x >= (freq << 20);
        const uint64_t m = __builtin_amdgcn_ballot_w64(emit);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        wp -= 2u * cnt;
        if (emit)
            *reinterpret_cast<uint16_t RANS_GLOBAL *>(slot + wp + 2u * rank_below(m)) = (uint16_t)(x & 0xffffu);
        uint32_t y = emit ? (x >> 16) : x;
        uint32_t q, rem;
        divmod_rcp(y, freq, rcp, q, rem);
        const uint32_t xn = (q << 12) + rem + start;
        x = active ? xn : x;
    } else if constexpr (FMT == FMT_R64) {
        // rans64.h:77-93
        const uint64_t x_max = ((uint64_t)freq) << (63u - T.scale_bits); // ((L >> sb) << 32) * freq
        const bool emit = active && x >= x_max;
        const uint64_t m = __builtin_amdgcn_ballot_w64(emit);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        wp -= 4u * cnt;
        if (emit)
            *reinterpret_cast<uint32_t RANS_GLOBAL *>(slot + wp + 4u * rank_below(m)) = (uint32_t)x;
        uint64_t y = emit ? (x >> 32) : x;
        uint64_t q, rem;
        divmod_rcp64(y, freq, rec, q, rem);
        const uint64_t xn = (q << T.scale_bits) + rem + start;
        x = active ? xn : x;
    } else {
        // rans_byte.h:62-74 (renorm: 0, 1 or 2 bytes for scale_bits <= 16), :83-90 (put),
        // main_alias.cpp:241-250 (alias put).  The low byte is emitted first, i.e.
        // ends up at the higher address.
        const uint32_t x_max = ((1u << 23 >> T.scale_bits) << 8) * freq;
        const bool e1 = active && x >= x_max;
        const bool e2 = e1 && (x >> 8) >= x_max;
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(e1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(e2);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m1) + (uint32_t)__builtin_popcountll(m2);
        wp -= cnt;
        const uint32_t at = wp + rank_below(m1) + rank_below(m2);
        if (e2) {
            slot[at] = (uint8_t)(x >> 8);
            slot[at + 1] = (uint8_t)x;
        } else if (e1) {
            slot[at] = (uint8_t)x;
        }
        uint32_t y = e2 ? (x >> 16) : (e1 ? (x >> 8) : x);
        uint32_t q, rem;
        divmod_rcp(y, freq, rcp, q, rem);
        uint32_t xn;
        if constexpr (FMT == FMT_ALIAS)
            xn = (q << T.scale_bits) + (active ? T.alias_remap[rem + start] : 0u);
        else
            xn = (q << T.scale_bits) + rem + start;
        x = active ? xn : x;
    }
}

// Hand-written encoder sub-step of the word format for a FULL wave (64 active lanes, symbols
// already turned into LDS addresses of their WordEncRec): rans_word_sse41.h:81-93 for 64 lanes.
//   v_add_co    carry of x + (cmpl << 20)  <=>  x >= freq << 20: the lanes that emit a word
//   s_bcnt1 ..  words emitted -> the wave's write offset moves down (these SALU ops are also the
//               wait states a VALU write of vcc needs before v_mbcnt may read it)
//   v_mbcnt x2  rank among the emitting lanes = word index (ascending lane = ascending address)
//   global_store_short + v_lshrrev under the emit mask, then exec back to all ones
//   x / freq    round-up reciprocal (model.h, WordEncRec): one v_mul_hi_u32 and four cheap ops,
//               exact, so no compare/select; x' = x + bias + q * cmpl in one v_mad_u32_u24 + add
// 15 VALU, no v_cndmask, no branch.  `wp` is the byte offset of the lowest word written so far.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void enc_word_full(uint32_t &x, const u32x3 &rec, uint32_t &wp,
                                              const uint8_t RANS_GLOBAL *slot, uint32_t &worst)
{
    uint32_t t, q, sh, cnt;
    asm volatile("v_lshlrev_b32_e32 %[t], 20, %[w]\n\t"
                 "v_max_u32_e32 %[worst], %[worst], %[w]\n\t"
                 "v_add_co_u32_e32 %[t], vcc, %[t], %[x]\n\t"
                 "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                 "s_lshl_b32 %[cnt], %[cnt], 1\n\t"
                 "s_sub_u32 %[wp], %[wp], %[cnt]\n\t"
                 "s_mov_b64 exec, vcc\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "v_lshl_add_u32 %[t], %[t], 1, %[wp]\n\t"
                 "global_store_short %[t], %[x], %[base]\n\t"
                 "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 "v_mul_hi_u32 %[q], %[x], %[m]\n\t"
                 "v_lshrrev_b32_e32 %[sh], 24, %[w]\n\t"
                 "v_sub_u32_e32 %[t], %[x], %[q]\n\t"
                 "v_lshrrev_b32_e32 %[t], 1, %[t]\n\t"
                 "v_add_u32_e32 %[q], %[q], %[t]\n\t"
                 "v_lshrrev_b32_e32 %[q], %[sh], %[q]\n\t"
                 "v_mad_u32_u24 %[q], %[q], %[w], %[x]\n\t"
                 "v_add_u32_e32 %[x], %[q], %[bias]"
                 : [x] "+v"(x), [wp] "+s"(wp), [worst] "+v"(worst), [t] "=&v"(t), [q] "=&v"(q), [sh] "=&v"(sh),
                   [cnt] "=&s"(cnt)
                 : [m] "v"(rec.x), [w] "v"(rec.y), [bias] "v"(rec.z), [base] "s"(slot)
                 : "vcc", "scc", "memory");
}

template <int FMT, int K>
__global__ void __launch_bounds__(kEncBlockThreads) k_encode(const EncParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // word format: the 256 WordEncRec of the full-wave path come first (LDS address = sym << 4),
    // the per-symbol EncRec table of the general path behind them
    constexpr uint32_t kWordRecBytes = FMT == FMT_WORD ? 256u * (uint32_t)sizeof(WordEncRec) : 0u;
    if constexpr (FMT == FMT_WORD) {
        const uint4 *g = reinterpret_cast<const uint4 *>(p.word_enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x)
            l[i] = g[i];
    }
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem + kWordRecBytes);
        for (uint32_t i = threadIdx.x; i < p.nsyms; i += blockDim.x)
            l[i] = g[i];
    }
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t N = p.n_ways; // <= 64 * K; lanes idx >= N idle

    EncTables<FMT> T;
    T.recs = reinterpret_cast<const uint4 *>(smem + kWordRecBytes);
    T.alias_remap = p.alias_remap;
    T.scale_bits = p.scale_bits;
    T.nsyms = p.nsyms;

    bool bad = false;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    // per-lane constants of the 4x4 byte transpose (same lane mapping as the decoder's stores)
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t in_lane_off = (lane & 3u) * N + (lane & ~3u);

    for (uint64_t chunk_v = (uint64_t)blockIdx.x * waves_per_block + wave; chunk_v < p.nchunks;
         chunk_v += total_waves) {
        const uint64_t chunk = uniform64(chunk_v);
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + first * p.sym_bytes;
        uint8_t RANS_GLOBAL *slot = (uint8_t RANS_GLOBAL *)p.scratch + chunk * p.slot_bytes;
        uint32_t wp = (uint32_t)p.slot_bytes;

        state_t x[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            x[k] = Tr::kL; // RansEncInit / RansWordEncInit / Rans64EncInit

        const uint32_t rounds = uniform(nsym / N);
        const uint32_t tail = uniform(nsym - rounds * N);
        // Fast input path: full waves, u8 symbols, dword-aligned rows -> symbols of 4 rounds
        // arrive as one coalesced dword per lane and are transposed in registers; loads run
        // one super-group (16 rounds) ahead of the arithmetic.
        const bool fast_in = p.sym_bytes == 1 && N == p.n_ways && (N & 63u) == 0 &&
                             ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 3u) == 0;
        // (the word path addresses its record table by raw LDS address: dynamic LDS must start at 0)
        const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem == 0u;
        const uint32_t fast_rounds = (fast_in && (FMT != FMT_WORD || lds_at_zero)) ? (rounds & ~15u) : 0u;

        // rounds from last to first; round `rounds` is the partial one
        for (uint32_t rr = rounds + 1; rr-- > fast_rounds;) {
            const uint32_t cnt = (rr < rounds) ? N : tail;
            if (cnt == 0)
                continue;
            const uint8_t RANS_GLOBAL *rsrc = src + (uint64_t)rr * N * p.sym_bytes;
            uint32_t sym[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t idx = k * 64u + lane;
                sym[k] = 0;
                if (idx < cnt)
                    sym[k] = p.sym_bytes == 1 ? (uint32_t)rsrc[idx]
                                              : (uint32_t) reinterpret_cast<const uint16_t RANS_GLOBAL *>(rsrc)[idx];
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const uint32_t idx = k * 64u + lane;
                enc_substep<FMT>(T, x[k], sym[k], idx < cnt, slot, wp, bad);
            }
        }

        uint32_t worst = 0; // word fast path: max of cmpl_sh, > 0x0fffffff iff a symbol has no record
        if (fast_rounds) {
            uint32_t rec_mask = 0xff0u;
            asm volatile("" : "+v"(rec_mask)); // keep the mask in a VGPR (a literal operand costs a slower VALU form)
            uint32_t cur[4][K], nxt[4][K];
            auto load_super = [&](uint32_t (&dstq)[4][K], uint32_t sg) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        dstq[j][k] = *reinterpret_cast<const uint32_t RANS_GLOBAL *>(
                            src + (uint64_t)(sg * 16u + j * 4u) * N + in_lane_off + k * 64u);
            };
            uint32_t sg = fast_rounds >> 4;
            load_super(cur, sg - 1);
            while (sg-- > 0) {
                if (sg > 0)
                    load_super(nxt, sg - 1);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    uint32_t t[K];
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        t[k] = quad_transpose(cur[j][k], sel1, sel2);
                if constexpr (FMT == FMT_WORD) {
                    // symbol byte J -> LDS address of its record, (sym << 4) + table offset; the record
                    // of the next sub-step is read before the current one is worked on (the asm block is
                    // a scheduling barrier for the compiler)
                    auto rec_at = [&](int step) { // step 0 = (J 3, k K-1), descending
                        const int J = 3 - step / K, k = K - 1 - step % K;
                        const uint32_t at = (J == 0 ? (t[k] << 4) : (t[k] >> (8 * J - 4))) & rec_mask;
                        return *reinterpret_cast<const __attribute__((address_space(3))) u32x3 *>((uintptr_t)at); // table at LDS address 0
                    };
                    u32x3 rec = rec_at(0);
#pragma unroll
                    for (int step = 0; step < 4 * K; ++step) {
                        const u32x3 now = rec;
                        if (step + 1 < 4 * K)
                            rec = rec_at(step + 1);
                        enc_word_full(x[K - 1 - step % K], now, wp, slot, worst);
                    }
                } else {
#pragma unroll
                    for (int J = 3; J >= 0; --J)
#pragma unroll
                        for (int k = K - 1; k >= 0; --k)
                            enc_substep<FMT>(T, x[k], (t[k] >> (8 * J)) & 0xffu, true, slot, wp, bad);
                }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        cur[j][k] = nxt[j][k];
            }
        }

        if (worst > 0x0fffffffu)
            bad = true;
        // flush: lane N-1 first, i.e. lane 0's state ends up first in memory
        // (main.cpp:244-245, main_simd.cpp:298-299)
        wp -= N * Tr::kStateBytes;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = k * 64u + lane;
            if (idx < N) {
                uint8_t RANS_GLOBAL *at = slot + wp + idx * Tr::kStateBytes;
                if constexpr (FMT == FMT_R64) {
                    reinterpret_cast<uint32_t RANS_GLOBAL *>(at)[0] = (uint32_t)x[k];
                    reinterpret_cast<uint32_t RANS_GLOBAL *>(at)[1] = (uint32_t)(x[k] >> 32);
                } else if constexpr (FMT == FMT_WORD) {
                    reinterpret_cast<uint16_t RANS_GLOBAL *>(at)[0] = (uint16_t)x[k];
                    reinterpret_cast<uint16_t RANS_GLOBAL *>(at)[1] = (uint16_t)(x[k] >> 16);
                } else {
                    at[0] = (uint8_t)x[k];
                    at[1] = (uint8_t)(x[k] >> 8);
                    at[2] = (uint8_t)(x[k] >> 16);
                    at[3] = (uint8_t)(x[k] >> 24);
                }
            }
        }
        if (lane == 0)
            p.lengths[chunk] = (uint32_t)p.slot_bytes - wp;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
}

// ===========================================================================
// Lane-per-stream kernels for narrow interleaves (N = 1, 2, 4, 8; BASELINE config 2 is
// the reference's 2-way rans64 loop, main64.cpp:224-287).  An N-way stream with N << 64
// cannot feed a wavefront, so here every LANE owns a whole chunk: its N states live in
// registers, it walks its own stream with its own pointer (the renormalisation order
// inside a chunk is the sequential reference order, no cross-lane work at all), and a
// wave decodes 64 chunks at once.  Tables are shared through LDS as before.
// ===========================================================================

// Per-lane stream window for the lane-per-stream decoder: 16 bytes of the lane's own stream
// in registers plus the next 16 prefetched, so that a lane touches global memory once per
// 16 stream bytes (a dword-per-unit walk would issue 4-16x as many scattered accesses).
// The window is consumed from its low end and shifted down after every unit -- positional
// indexing would be a chain of v_cndmask (~22 issue cycles each on gfx950) -- and the refill
// loads straight into `pre` under the lanes' exec mask, so the load issued at one refill is only
// waited for at the next one.
struct LaneWindow {
    u32x4 win, pre;
    uint64_t next;  // global address of the 16 bytes after `pre`
    uint64_t limit; // 16-byte aligned end of the readable container
    uint32_t left;  // bytes left in win (1..16)
    uint32_t used;  // stream bytes consumed so far

    template <int UNIT> __device__ __forceinline__ void shift()
    {
        if constexpr (UNIT == 4) {
            win.x = win.y;
            win.y = win.z;
            win.z = win.w;
        } else {
            win.x = __builtin_amdgcn_alignbit(win.y, win.x, 8 * UNIT);
            win.y = __builtin_amdgcn_alignbit(win.z, win.y, 8 * UNIT);
            win.z = __builtin_amdgcn_alignbit(win.w, win.z, 8 * UNIT);
            win.w >>= 8 * UNIT;
        }
    }
    __device__ __forceinline__ void refill()
    {
        win = pre;
        left = 16;
        if (next < limit) // past the container: `pre` keeps stale bytes, which only a corrupt chunk consumes
            pre = *reinterpret_cast<const u32x4 RANS_GLOBAL *>(next);
        next += 16;
    }
    template <int UNIT> __device__ __forceinline__ void open(uint64_t addr, uint64_t lim)
    {
        limit = lim;
        const uint64_t base = addr & ~uint64_t(15);
        used = 0;
        win = u32x4{0u, 0u, 0u, 0u};
        pre = win;
        if (base < limit)
            win = *reinterpret_cast<const u32x4 RANS_GLOBAL *>(base);
        if (base + 16 < limit)
            pre = *reinterpret_cast<const u32x4 RANS_GLOBAL *>(base + 16);
        next = base + 32;
        left = 16;
        for (uint32_t skip = (uint32_t)(addr & 15u); skip != 0; skip -= UNIT) { // once per chunk
            shift<UNIT>();
            left -= UNIT;
        }
    }
    template <int UNIT> __device__ __forceinline__ uint32_t take()
    {
        uint32_t w = win.x;
        if constexpr (UNIT == 2)
            w &= 0xffffu;
        else if constexpr (UNIT == 1)
            w &= 0xffu;
        shift<UNIT>();
        used += UNIT;
        left -= UNIT;
        if (left == 0)
            refill();
        return w;
    }
};

template <int FMT>
__device__ __forceinline__ void lane_renorm(typename FmtTraits<FMT>::state_t &x, LaneWindow &W, bool active)
{
    if constexpr (FMT == FMT_WORD) {
        if (active && x < (1u << 16)) // rans_word_sse41.h:134-141
            x = (x << 16) | W.take<2>();
    } else if constexpr (FMT == FMT_R64) {
        if (active && x < (1ull << 31)) // rans64.h:305-316
            x = (x << 32) | W.take<4>();
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) // rans_byte.h:307-318, at most two bytes for scale_bits <= 16
            if (active && x < (1u << 23))
                x = (x << 8) | W.take<1>();
    }
}

// ---------------------------------------------------------------------------
// Staged lane-per-stream decoder.  Per-lane 16-byte loads pull a whole memory line for every
// 16 bytes used, and the line is long evicted when the lane comes back for its next 16 bytes
// (measured: >= 4x over-fetch, the kernel sat on the fabric at ~3.3 TB/s); and a lane that
// refills on its own stalls its whole wave.  Here the WAVE refills for all of its 64 chunks at
// once, every 16 symbols: a lane's stream lives in a 128-byte ring in LDS (two 64-byte lines),
// lanes publish which line they need next, and 4 lanes fetch one chunk's line with coalesced
// 16-byte loads (4 load instructions cover 64 chunks x 64 B).  16 symbols consume at most one
// line (rans64: <= 4 B per symbol), so "at least 64 bytes ahead" before every group is all the
// invariant there is.  Ring rows are 136 bytes apart: equal positions of the 64 lanes spread over
// 32 banks.  Positions are 32-bit offsets from the line of the wave's first chunk.
// ---------------------------------------------------------------------------
constexpr uint32_t kLaneLine = 64;
constexpr uint32_t kLaneRingStride = 2 * kLaneLine + 8;
constexpr uint32_t kLaneWaveLds = 64 * kLaneRingStride + 64 * 4; // rings + one request word per lane

template <int FMT> struct LaneRing {
    const uint8_t *row; // this lane's ring in LDS
    uint32_t cur;       // read position (offset from the wave's region base)
    uint32_t used;      // stream bytes consumed

    template <int UNIT> __device__ __forceinline__ uint32_t take()
    {
        const uint8_t *at = row + (cur & (2 * kLaneLine - 1));
        cur += UNIT;
        used += UNIT;
        if constexpr (UNIT == 4)
            return *reinterpret_cast<const uint32_t *>(at);
        else if constexpr (UNIT == 2)
            return *reinterpret_cast<const uint16_t *>(at);
        else
            return *at;
    }
    __device__ __forceinline__ void renorm(typename FmtTraits<FMT>::state_t &x, bool active)
    {
        // the branch bodies hold an LDS read, so they stay exec-masked branches (no v_cndmask)
        if constexpr (FMT == FMT_WORD) {
            if (active && x < (1u << 16)) // rans_word_sse41.h:134-141
                x = (x << 16) | take<2>();
        } else if constexpr (FMT == FMT_R64) {
            if (active && x < (1ull << 31)) // rans64.h:305-316
                x = (x << 32) | take<4>();
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) // rans_byte.h:307-318, at most two bytes for scale_bits <= 16
                if (active && x < (1u << 23))
                    x = (x << 8) | take<1>();
        }
    }
};

template <int FMT, int NW>
__global__ void __launch_bounds__(1024) k_decode_lanes_staged(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    {
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

    DecTables<FMT> T;
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u; // keep the wave kernels' counter ring consistent
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    uint8_t *rings = smem + t0_bytes + t1_bytes + wave * kLaneWaveLds;
    uint32_t *req = reinterpret_cast<uint32_t *>(rings + 64u * kLaneRingStride);
    const uint32_t part = lane & 3u, grp = lane >> 2;

    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t glimit = (cbase + p.container_bytes + 15u) & ~uint64_t(15);
    const bool wide_out = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.out) | p.chunk_syms) & 15u) == 0;
    uint32_t nbad = 0;
    const uint64_t nbatches = (p.nchunks + 63u) / 64u;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t batch_v = (uint64_t)blockIdx.x * waves_per_block + wave; batch_v < nbatches; batch_v += total_waves) {
        const uint64_t batch = uniform64(batch_v);
        const uint64_t chunk = batch * 64u + lane;
        bool valid = chunk < p.nchunks;
        const uint64_t off = valid ? p.offsets[chunk] : 0;
        const uint32_t len = valid ? p.lengths[chunk] : 0;
        const uint64_t first = chunk * p.chunk_syms;
        // region base: the line of the batch's first chunk (lane 0 always holds a chunk)
        const uint64_t rb = uniform64(off) & ~uint64_t(kLaneLine - 1);
        if (valid && ((off & 15u) != 0 || len < NW * Tr::kStateBytes || off + len > p.container_bytes || off < rb ||
                      off - rb >= (1u << 30))) {
            nbad++;
            valid = false;
        }
        const uint32_t nsym = valid ? (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms) : 0u;
        uint8_t RANS_GLOBAL *dst = (uint8_t RANS_GLOBAL *)p.out + first * p.sym_bytes;

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l) { // RansDecInit order: lane 0's state first
            x[l] = Tr::kL;
            if (valid) {
                const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.container + off;
                if constexpr (FMT == FMT_R64) {
                    const u32x2 v = reinterpret_cast<const u32x2 RANS_GLOBAL *>(src)[l];
                    x[l] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                } else {
                    x[l] = reinterpret_cast<const uint32_t RANS_GLOBAL *>(src)[l];
                }
            }
        }
        LaneRing<FMT> W;
        W.row = rings + lane * kLaneRingStride;
        W.cur = (uint32_t)(off - rb) + NW * Tr::kStateBytes;
        W.used = 0;
        uint32_t ld = W.cur & ~(kLaneLine - 1u); // next line this lane has not staged yet

        // the whole wave takes part (also lanes without a chunk): lane -> which line its chunk needs,
        // then lane (4 g + part) moves 16 bytes of chunk (16 j + g)'s line, j = 0..3
        auto refill = [&]() {
            const bool need = valid && (int32_t)(ld - W.cur) < (int32_t)kLaneLine;
            req[lane] = need ? (ld | 1u) : 0u;
            if (need)
                ld += kLaneLine;
            uint32_t r[4];
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = req[16 * j + grp]; // LDS ops of one wave execute in order: sees the writes above
                v[j] = u32x4{0u, 0u, 0u, 0u};
                const uint64_t a = cbase + rb + (r[j] & ~(kLaneLine - 1u)) + part * 16u;
                if ((r[j] & 1u) && a < glimit)
                    v[j] = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(a));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r[j] & 1u) {
                    uint8_t *at = rings + (16u * j + grp) * kLaneRingStride + (r[j] & kLaneLine) + part * 16u;
                    reinterpret_cast<u32x2 *>(at)[0] = u32x2{v[j].x, v[j].y}; // rows are 8-byte aligned
                    reinterpret_cast<u32x2 *>(at)[1] = u32x2{v[j].z, v[j].w};
                }
        };

        // 16 symbols into four dwords: 16/NW rounds of NW steps + renormalisations
        auto decode16 = [&]() -> u32x4 {
            uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int rr = 0; rr < 16 / NW; ++rr) {
#pragma unroll
                for (int l = 0; l < NW; ++l) {
                    uint32_t sy = dec_step<FMT>(T, x[l]);
                    if constexpr (Tr::kSymByte == 3)
                        sy >>= 24;
                    const int pos = rr * NW + l;
                    pk[pos / 4] |= (sy & 0xffu) << (8 * (pos % 4));
                }
#pragma unroll
                for (int l = 0; l < NW; ++l)
                    W.renorm(x[l], true);
            }
            return u32x4{pk[0], pk[1], pk[2], pk[3]};
        };

        refill(); // two lines ahead to start with
        // 64 symbols per trip: four groups of 16 (a refill before each), then the lane writes its 64
        // bytes with four back-to-back 16-byte stores.  One 16-byte store per group left every line
        // dirty in L2 for four groups -- long enough to be evicted half-written: 3.2x the bytes on
        // the write side (WRITE_SIZE 0.87 GB for 0.27 GB of symbols).
        for (uint32_t i0 = 0; __builtin_amdgcn_ballot_w64(i0 < nsym) != 0; i0 += 64u) {
            const uint32_t left = i0 < nsym ? nsym - i0 : 0u;
            if (__builtin_amdgcn_ballot_w64(!(left >= 64u && wide_out) && valid) == 0) { // wave-uniform
                u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0;
                refill();
                if (left)
                    q0 = decode16();
                refill();
                if (left)
                    q1 = decode16();
                refill();
                if (left)
                    q2 = decode16();
                refill();
                if (left) {
                    q3 = decode16();
                    u32x4 RANS_GLOBAL *o = reinterpret_cast<u32x4 RANS_GLOBAL *>(dst + i0);
                    o[0] = q0;
                    o[1] = q1;
                    o[2] = q2;
                    o[3] = q3;
                }
                continue;
            }
            // ragged end of a chunk, unaligned or 16-bit output: 16 symbols at a time
            for (uint32_t g = 0; g < 4u; ++g) {
                refill();
                const uint32_t j0 = i0 + 16u * g;
                if (j0 >= nsym)
                    continue;
                const uint32_t cnt = nsym - j0 < 16u ? nsym - j0 : 16u;
                if (cnt == 16u && wide_out) {
                    const u32x4 q = decode16();
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(dst + j0) = q;
                } else {
                    for (uint32_t i = 0; i < cnt; i += NW) {
                        const uint32_t c = cnt - i < (uint32_t)NW ? cnt - i : (uint32_t)NW;
#pragma unroll
                        for (int l = 0; l < NW; ++l)
                            if ((uint32_t)l < c) {
                                uint32_t sy = dec_step<FMT>(T, x[l]);
                                if constexpr (Tr::kSymByte == 3)
                                    sy >>= 24;
                                if (p.sym_bytes == 1)
                                    dst[j0 + i + l] = (uint8_t)sy;
                                else
                                    reinterpret_cast<uint16_t RANS_GLOBAL *>(dst)[j0 + i + l] = (uint16_t)sy;
                            }
#pragma unroll
                        for (int l = 0; l < NW; ++l)
                            W.renorm(x[l], (uint32_t)l < c);
                    }
                }
            }
        }
        bool bad = false;
#pragma unroll
        for (int l = 0; l < NW; ++l)
            bad = bad || (x[l] != Tr::kL);
        if (valid && (bad || W.used + NW * Tr::kStateBytes != len))
            nbad++;
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}

template <int FMT, int NW>
__global__ void __launch_bounds__(256) k_decode_lanes(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    {
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

    DecTables<FMT> T;
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u; // keep the wave kernels' counter ring consistent
    const uint8_t RANS_GLOBAL *cbase = (const uint8_t RANS_GLOBAL *)p.container;
    const uint64_t glimit = (reinterpret_cast<uint64_t>(p.container) + p.container_bytes + 15u) & ~uint64_t(15);
    const bool wide_out = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.out) | p.chunk_syms) & 15u) == 0;
    uint32_t nbad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; chunk < p.nchunks; chunk += stride) {
        const uint64_t off = p.offsets[chunk];
        const uint32_t len = p.lengths[chunk];
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        if ((off & 15u) != 0 || len < NW * Tr::kStateBytes || off + len > p.container_bytes) {
            nbad++;
            continue;
        }
        const uint8_t RANS_GLOBAL *src = cbase + off;
        uint8_t RANS_GLOBAL *dst = (uint8_t RANS_GLOBAL *)p.out + first * p.sym_bytes;

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l) { // RansDecInit order: lane 0's state first
            if constexpr (FMT == FMT_R64) {
                const u32x2 v = reinterpret_cast<const u32x2 RANS_GLOBAL *>(src)[l];
                x[l] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            } else {
                x[l] = reinterpret_cast<const uint32_t RANS_GLOBAL *>(src)[l];
            }
        }
        LaneWindow W;
        W.open<(int)Tr::kUnit>(reinterpret_cast<uint64_t>(p.container) + off + NW * Tr::kStateBytes, glimit);
        bool bad = false;

        const uint32_t rounds = nsym / NW;
        const uint32_t tail = nsym - rounds * NW;
        uint32_t i = 0; // symbol index inside the chunk
        if (wide_out) {
            // 16 symbols per 16-byte store: 16/NW rounds per group
            constexpr int kRoundsPerGroup = 16 / NW;
            const uint32_t groups = rounds / kRoundsPerGroup;
            for (uint32_t g = 0; g < groups; ++g) {
                u32x4 pack = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int rr = 0; rr < kRoundsPerGroup; ++rr) {
#pragma unroll
                    for (int l = 0; l < NW; ++l) {
                        uint32_t sy = dec_step<FMT>(T, x[l]);
                        if constexpr (Tr::kSymByte == 3)
                            sy >>= 24;
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int pos = rr * NW + l;
                        pack[pos / 4] |= (sy & 0xffu) << (8 * (pos % 4));
                    }
#pragma unroll
                    for (int l = 0; l < NW; ++l)
                        lane_renorm<FMT>(x[l], W, true);
                }
                *reinterpret_cast<u32x4 RANS_GLOBAL *>(dst + i) = pack;
                i += 16;
            }
        }
        // remaining rounds + tail: element stores
        while (i < nsym) {
            const uint32_t cnt = nsym - i < (uint32_t)NW ? nsym - i : (uint32_t)NW;
#pragma unroll
            for (int l = 0; l < NW; ++l)
                if ((uint32_t)l < cnt) {
                    uint32_t sy = dec_step<FMT>(T, x[l]);
                    if constexpr (Tr::kSymByte == 3)
                        sy >>= 24;
                    if (p.sym_bytes == 1)
                        dst[i + l] = (uint8_t)sy;
                    else
                        reinterpret_cast<uint16_t RANS_GLOBAL *>(dst)[i + l] = (uint16_t)sy;
                }
#pragma unroll
            for (int l = 0; l < NW; ++l)
                lane_renorm<FMT>(x[l], W, (uint32_t)l < cnt);
            i += cnt;
        }
        (void)tail;
#pragma unroll
        for (int l = 0; l < NW; ++l)
            bad = bad || (x[l] != Tr::kL);
        if (bad || W.used + NW * Tr::kStateBytes != len)
            nbad++;
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}

// one symbol of the sequential reference encoder (RansEncPut / RansWordEncPut / Rans64EncPut /
// RansEncPutAlias) for a lane-private state and write pointer
template <int FMT>
__device__ __forceinline__ void lane_put(typename FmtTraits<FMT>::state_t &x, uint32_t sym, const uint4 *recs,
                                         const EncParams &p, uint8_t RANS_GLOBAL *&wp, bool &bad)
{
    const bool known = sym < p.nsyms;
    const uint4 rec = recs[known ? sym : 0u];
    const uint32_t freq = (FMT == FMT_R64) ? (rec.x & 0xffffffu) : rec.x, start = rec.y, rcp = rec.z;
    if (!known || freq == 0) {
        bad = true;
        return;
    }
    if constexpr (FMT == FMT_WORD) {
        uint32_t y = x;
        if (y >= (freq << 20)) {
            wp -= 2;
            *reinterpret_cast<uint16_t RANS_GLOBAL *>(wp) = (uint16_t)y;
            y >>= 16;
        }
        uint32_t q, rem;
        divmod_rcp(y, freq, rcp, q, rem);
        x = (q << 12) + rem + start;
    } else if constexpr (FMT == FMT_R64) {
        uint64_t y = x;
        if (y >= (((uint64_t)freq) << (63u - p.scale_bits))) {
            wp -= 4;
            *reinterpret_cast<uint32_t RANS_GLOBAL *>(wp) = (uint32_t)y;
            y >>= 32;
        }
        uint64_t q, rem;
        divmod_rcp64(y, freq, rec, q, rem);
        x = (q << p.scale_bits) + rem + start;
    } else {
        uint32_t y = x;
        const uint32_t x_max = ((1u << 23 >> p.scale_bits) << 8) * freq;
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (y >= x_max) {
                *--wp = (uint8_t)y;
                y >>= 8;
            }
        uint32_t q, rem;
        divmod_rcp(y, freq, rcp, q, rem);
        if constexpr (FMT == FMT_ALIAS)
            x = (q << p.scale_bits) + p.alias_remap[rem + start];
        else
            x = (q << p.scale_bits) + rem + start;
    }
}

// ---------------------------------------------------------------------------
// Staged lane-per-stream encoder: the mirror image of k_decode_lanes_staged.  Per-lane stores of
// every emitted unit reached HBM as partial lines (measured 6.4x the stream bytes on the write
// side) and per-lane 16-byte symbol loads pulled whole lines (3.6x).  Here
//   * symbols: the wave loads one 64-byte block of each of its 64 chunks with coalesced 16-byte
//     loads (4 lanes per chunk) into per-lane rows in LDS; every lane then walks its row from
//     the top, 16 symbols per ds_read_b128;
//   * stream: units go into a 128-byte ring per lane (two 64-byte lines, written downwards); after
//     every 16 symbols (at most 64 bytes emitted) a line that has filled up is written out by 4
//     lanes with 16-byte stores -- whole 64-byte lines, each written once.
// Slots are whole lines (api.cpp, encode_slot_bytes); what lies below the stream start inside the
// lowest line is never read.
// ---------------------------------------------------------------------------
constexpr uint32_t kEncRowStride = 80; // 64 symbol bytes, rows 16-byte aligned, 20 dwords apart
constexpr uint32_t kEncWaveLds = 64 * kEncRowStride + 64 * kLaneRingStride + 64 * 4;

template <int FMT> struct LaneOut {
    uint8_t *row;  // this lane's output ring in LDS
    uint32_t w;    // write offset inside the chunk's slot, moves down
    template <int UNIT> __device__ __forceinline__ void emit(uint32_t v)
    {
        w -= UNIT;
        uint8_t *at = row + (w & (2 * kLaneLine - 1));
        if constexpr (UNIT == 4)
            *reinterpret_cast<uint32_t *>(at) = v;
        else if constexpr (UNIT == 2)
            *reinterpret_cast<uint16_t *>(at) = (uint16_t)v;
        else
            *at = (uint8_t)v;
    }
};

// one symbol of the sequential reference encoder for a lane-private state, emitting into the ring
template <int FMT>
__device__ __forceinline__ void lane_put_staged(typename FmtTraits<FMT>::state_t &x, uint32_t sym, const uint4 *recs,
                                                const EncParams &p, LaneOut<FMT> &O, bool &bad)
{
    const bool known = sym < p.nsyms;
    const uint4 rec = recs[known ? sym : 0u];
    const uint32_t freq = (FMT == FMT_R64) ? (rec.x & 0xffffffu) : rec.x, start = rec.y, rcp = rec.z;
    if (!known || freq == 0) {
        bad = true;
        return;
    }
    if constexpr (FMT == FMT_WORD) {
        uint32_t y = x;
        if (y >= (freq << 20)) { // rans_word_sse41.h:85-89
            O.template emit<2>(y);
            y >>= 16;
        }
        uint32_t q, rem;
        divmod_rcp(y, freq, rcp, q, rem);
        x = (q << 12) + rem + start;
    } else if constexpr (FMT == FMT_R64) {
        uint64_t y = x;
        if (y >= (((uint64_t)freq) << (63u - p.scale_bits))) { // rans64.h:83-88
            O.template emit<4>((uint32_t)y);
            y >>= 32;
        }
        uint64_t q, rem;
        divmod_rcp64(y, freq, rec, q, rem);
        x = (q << p.scale_bits) + rem + start;
    } else {
        uint32_t y = x;
        const uint32_t x_max = ((1u << 23 >> p.scale_bits) << 8) * freq; // rans_byte.h:64-70
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (y >= x_max) {
                O.template emit<1>(y);
                y >>= 8;
            }
        uint32_t q, rem;
        divmod_rcp(y, freq, rcp, q, rem);
        if constexpr (FMT == FMT_ALIAS)
            x = (q << p.scale_bits) + p.alias_remap[rem + start];
        else
            x = (q << p.scale_bits) + rem + start;
    }
}

template <int FMT, int NW>
__global__ void __launch_bounds__(1024) k_encode_lanes_staged(const EncParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < p.nsyms; i += blockDim.x)
            l[i] = g[i];
    }
    __syncthreads();
    const uint4 *recs = reinterpret_cast<const uint4 *>(smem);
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    uint8_t *rows = smem + p.nsyms * (uint32_t)sizeof(EncRec) + wave * kEncWaveLds;
    uint8_t *rings = rows + 64u * kEncRowStride;
    uint32_t *req = reinterpret_cast<uint32_t *>(rings + 64u * kLaneRingStride);
    const uint32_t part = lane & 3u, grp = lane >> 2;
    const uint32_t slot_lines = (uint32_t)(p.slot_bytes / kLaneLine);

    bool bad = false;
    const uint64_t nbatches = (p.nchunks + 63u) / 64u;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t batch_v = (uint64_t)blockIdx.x * waves_per_block + wave; batch_v < nbatches; batch_v += total_waves) {
        const uint64_t chunk0 = uniform64(batch_v) * 64u;
        const uint64_t chunk = chunk0 + lane;
        const bool valid = chunk < p.nchunks;
        auto syms_of = [&](uint64_t c) -> uint32_t { // symbols in chunk c (0 past the end)
            if (c >= p.nchunks)
                return 0u;
            const uint64_t first = c * p.chunk_syms;
            return (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        };
        const uint32_t nsym = syms_of(chunk);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + chunk * (uint64_t)p.chunk_syms;
        uint8_t RANS_GLOBAL *slots0 = (uint8_t RANS_GLOBAL *)p.scratch + chunk0 * p.slot_bytes; // wave-uniform

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l)
            x[l] = Tr::kL;
        LaneOut<FMT> O;
        O.row = rings + lane * kLaneRingStride;
        O.w = (uint32_t)p.slot_bytes;
        uint32_t flushed = slot_lines; // lines [flushed, slot_lines) are in memory

        // the whole wave takes part: lanes publish the line they have filled (or, at the end, the lines
        // that hold anything), lane (4 g + part) writes 16 bytes of chunk (16 j + g)'s line
        auto flush = [&](bool final) {
            const bool need = valid && flushed != 0u &&
                              (final ? O.w < flushed * kLaneLine : O.w <= (flushed - 1u) * kLaneLine);
            req[lane] = need ? (((flushed - 1u) << 1) | 1u) : 0u;
            if (need)
                flushed -= 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = 16u * j + grp;
                const uint32_t r = req[q]; // LDS ops of one wave execute in order
                if (r & 1u) {
                    const uint32_t line = r >> 1;
                    const uint8_t *at = rings + q * kLaneRingStride + (line & 1u) * kLaneLine + part * 16u;
                    const u32x2 a = reinterpret_cast<const u32x2 *>(at)[0], b = reinterpret_cast<const u32x2 *>(at)[1];
                    u32x4 RANS_GLOBAL *o = reinterpret_cast<u32x4 RANS_GLOBAL *>(
                        slots0 + (uint64_t)q * p.slot_bytes + (uint64_t)line * kLaneLine + part * 16u);
                    *o = u32x4{a.x, a.y, b.x, b.y};
                }
            }
        };

        // symbol i belongs to state i mod NW; visit i = nsym-1 .. 0 (main.cpp:233-243).  The top
        // nsym % 16 symbols come one by one from memory, the rest through the staged rows.
        const uint32_t nsym16 = nsym & ~15u;
        for (uint32_t i = nsym; i > nsym16; --i) {
            const uint32_t sym = (uint32_t)src[i - 1];
            const uint32_t l = (i - 1) % NW;
#pragma unroll
            for (int ll = 0; ll < NW; ++ll) // static register indexing
                if ((uint32_t)ll == l)
                    lane_put_staged<FMT>(x[ll], sym, recs, p, O, bad);
        }
        flush(false);

        const uint32_t my_blocks = (nsym16 + 63u) >> 6;
        uint32_t max_blocks = my_blocks;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)max_blocks, d, 64);
            max_blocks = o > max_blocks ? o : max_blocks;
        }
        max_blocks = uniform(max_blocks);
        for (uint32_t k = max_blocks; k-- > 0;) {
            // stage block k of every chunk: 16 bytes per lane, 4 instructions for 64 chunks
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = 16u * j + grp;
                const uint32_t qsyms16 = syms_of(chunk0 + q) & ~15u;
                const uint32_t at = 64u * k + 16u * part;
                if (at + 16u <= qsyms16) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(
                        reinterpret_cast<uint64_t>(p.syms) + (chunk0 + q) * (uint64_t)p.chunk_syms + at));
                    *reinterpret_cast<u32x4 *>(rows + q * kEncRowStride + 16u * part) = v;
                }
            }
            const uint8_t *row = rows + lane * kEncRowStride;
#pragma unroll
            for (int g = 3; g >= 0; --g) {
                if (64u * k + 16u * g + 16u <= nsym16) {
                    const u32x4 cur = *reinterpret_cast<const u32x4 *>(row + 16 * g);
#pragma unroll
                    for (int j = 15; j >= 0; --j) {
                        const uint32_t sym = (cur[j >> 2] >> (8 * (j & 3))) & 0xffu;
                        lane_put_staged<FMT>(x[j % NW], sym, recs, p, O, bad);
                    }
                }
                flush(false);
            }
        }
        // flush states NW-1 .. 0 (lane 0's first in memory), then whatever the ring still holds
        if (valid) {
#pragma unroll
            for (int l = NW - 1; l >= 0; --l) {
                if constexpr (FMT == FMT_R64) {
                    O.template emit<4>((uint32_t)(x[l] >> 32));
                    O.template emit<4>((uint32_t)x[l]);
                } else if constexpr (FMT == FMT_WORD) {
                    O.template emit<2>(x[l] >> 16);
                    O.template emit<2>(x[l]);
                } else {
                    O.template emit<1>(x[l] >> 24);
                    O.template emit<1>(x[l] >> 16);
                    O.template emit<1>(x[l] >> 8);
                    O.template emit<1>(x[l]);
                }
            }
            p.lengths[chunk] = (uint32_t)p.slot_bytes - O.w;
        }
        flush(true);
        flush(true);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
}

// Lane-per-stream encoder, second generation: symbols arrive as 16-byte per-lane loads
// (one scattered access per 16 symbols instead of per symbol), one group prefetched.
template <int FMT, int NW>
__global__ void __launch_bounds__(256) k_encode_lanes16(const EncParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < p.nsyms; i += blockDim.x)
            l[i] = g[i];
    }
    __syncthreads();
    const uint4 *recs = reinterpret_cast<const uint4 *>(smem);
    const bool wide_in = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 15u) == 0;

    bool bad = false;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; chunk < p.nchunks; chunk += stride) {
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + first * p.sym_bytes;
        uint8_t RANS_GLOBAL *slot = (uint8_t RANS_GLOBAL *)p.scratch + chunk * p.slot_bytes;
        uint8_t RANS_GLOBAL *wp = slot + p.slot_bytes;

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l)
            x[l] = Tr::kL;

        // symbol i belongs to state i mod NW; visit i = nsym-1 .. 0 (main.cpp:233-243).
        // [0, fast_end) is walked in 16-symbol groups; the ragged top part one by one.
        const uint32_t fast_end = wide_in ? (nsym & ~15u) : 0u;
        for (uint32_t i = nsym; i > fast_end; --i) {
            const uint32_t sym = p.sym_bytes == 1 ? (uint32_t)src[i - 1]
                                                  : (uint32_t) reinterpret_cast<const uint16_t RANS_GLOBAL *>(src)[i - 1];
            const uint32_t l = (i - 1) % NW;
#pragma unroll
            for (int ll = 0; ll < NW; ++ll) // static register indexing
                if ((uint32_t)ll == l)
                    lane_put<FMT>(x[ll], sym, recs, p, wp, bad);
        }
        if (fast_end) {
            const u32x4 RANS_GLOBAL *g16 = reinterpret_cast<const u32x4 RANS_GLOBAL *>(src);
            uint32_t g = fast_end >> 4;
            u32x4 cur = g16[g - 1], nxt = cur;
            while (g-- > 0) {
                if (g > 0)
                    nxt = g16[g - 1];
#pragma unroll
                for (int j = 15; j >= 0; --j) {
                    const uint32_t sym = (cur[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    lane_put<FMT>(x[j % NW], sym, recs, p, wp, bad);
                }
                cur = nxt;
            }
        }
        // flush states NW-1 .. 0 (lane 0's first in memory)
#pragma unroll
        for (int l = NW - 1; l >= 0; --l) {
            wp -= Tr::kStateBytes;
            if constexpr (FMT == FMT_R64) {
                reinterpret_cast<uint32_t RANS_GLOBAL *>(wp)[0] = (uint32_t)x[l];
                reinterpret_cast<uint32_t RANS_GLOBAL *>(wp)[1] = (uint32_t)(x[l] >> 32);
            } else if constexpr (FMT == FMT_WORD) {
                reinterpret_cast<uint16_t RANS_GLOBAL *>(wp)[0] = (uint16_t)x[l];
                reinterpret_cast<uint16_t RANS_GLOBAL *>(wp)[1] = (uint16_t)(x[l] >> 16);
            } else {
                wp[0] = (uint8_t)x[l];
                wp[1] = (uint8_t)(x[l] >> 8);
                wp[2] = (uint8_t)(x[l] >> 16);
                wp[3] = (uint8_t)(x[l] >> 24);
            }
        }
        p.lengths[chunk] = (uint32_t)((slot + p.slot_bytes) - wp);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane_id() == 0)
        atomicOr(p.flags, 1u);
}

// ---------------------------------------------------------------------------
// Layout: offsets[c] = sum_{i<c} align16(lengths[i]); offsets[nchunks] = end of
// the last stream.  Every block scans kLayoutChunksPerBlock chunks; with more than one
// block (narrow interleaves produce 10^5..10^6 chunks) k_layout_sums first leaves every
// block's total in block_sums[] and k_layout adds the totals of the blocks before it.
// ---------------------------------------------------------------------------
constexpr int kLayoutPer = 8; // consecutive chunks per thread
constexpr uint32_t kLayoutChunksPerBlock = 1024 * kLayoutPer;

__device__ __forceinline__ uint64_t block_sum_1024(uint64_t v, uint64_t *wave_sum)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    if (lane_id() == 0)
        wave_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t t = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w)
        t += wave_sum[w];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(1024) k_layout_sums(const LayoutParams p)
{
    __shared__ uint64_t wave_sum[16];
    const uint64_t c0 = (uint64_t)blockIdx.x * kLayoutChunksPerBlock + (uint64_t)threadIdx.x * kLayoutPer;
    uint64_t mine = 0;
#pragma unroll
    for (int i = 0; i < kLayoutPer; ++i)
        mine += c0 + i < p.nchunks ? (((uint64_t)p.lengths[c0 + i] + 15u) & ~uint64_t(15)) : 0;
    const uint64_t total = block_sum_1024(mine, wave_sum);
    if (threadIdx.x == 0)
        p.block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_layout(const LayoutParams p)
{
    __shared__ uint64_t wave_sum[16];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    uint64_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += blockDim.x)
        part += p.block_sums[b];
    const uint64_t carry = blockIdx.x ? block_sum_1024(part, wave_sum) : 0;

    const uint64_t c0 = (uint64_t)blockIdx.x * kLayoutChunksPerBlock + (uint64_t)threadIdx.x * kLayoutPer;
    uint64_t sz[kLayoutPer];
    uint64_t mine = 0;
#pragma unroll
    for (int i = 0; i < kLayoutPer; ++i) {
        sz[i] = c0 + i < p.nchunks ? (((uint64_t)p.lengths[c0 + i] + 15u) & ~uint64_t(15)) : 0;
        mine += sz[i];
    }
    // inclusive scan of the per-thread sums inside the wave
    uint64_t v = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, 64);
        if ((int)lane >= d)
            v += o;
    }
    if (lane == 63)
        wave_sum[wave] = v;
    __syncthreads();
    uint64_t before = carry;
    for (uint32_t w = 0; w < wave; ++w)
        before += wave_sum[w];
    uint64_t at = before + v - mine;
#pragma unroll
    for (int i = 0; i < kLayoutPer; ++i) {
        const uint64_t c = c0 + i;
        if (c < p.nchunks) {
            p.offsets[c] = at;
            if (c == p.nchunks - 1) {
                p.offsets[p.nchunks] = at + p.lengths[c];
                if (at + sz[i] > p.out_cap)
                    atomicOr(p.flags, 2u);
            }
        }
        at += sz[i];
    }
    if (p.nchunks == 0 && threadIdx.x == 0)
        p.offsets[0] = 0;
}

// ---------------------------------------------------------------------------
// Compaction: chunk c's stream sits at the END of its scratch slot (arbitrary
// alignment); copy it to out + offsets[c] (16-byte aligned).  One wave per
// chunk, dword granularity; source dwords are realigned with v_alignbyte_b32.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_compact(const CompactParams p)
{
    if (*p.flags & 2u)
        return;
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t chunk = (uint64_t)blockIdx.x * waves_per_block + wave; chunk < p.nchunks; chunk += total_waves) {
        const uint32_t len = p.lengths[chunk];
        const uint8_t *src = p.scratch + (chunk + 1) * p.slot_bytes - len;
        uint32_t *dst = reinterpret_cast<uint32_t *>(p.out + p.offsets[chunk]);
        const uintptr_t sa = reinterpret_cast<uintptr_t>(src);
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(sa & ~uintptr_t(3));
        const uint32_t shift = (uint32_t)(sa & 3u);
        const uint32_t ndw = (len + 3u) >> 2;
        for (uint32_t i = lane; i < ndw; i += 64u) {
            const uint32_t lo = s4[i];
            // the dword after the slot end is never needed when shift == 0
            const uint32_t hi = shift ? s4[i + 1] : 0u;
            dst[i] = __builtin_amdgcn_alignbyte(hi, lo, shift);
        }
    }
}

// ---------------------------------------------------------------------------
// Histogram (count_freqs, main.cpp:59-66).  1 byte of HBM traffic per symbol, so the LDS
// atomic rate is what has to keep up: a skewed source (Zipf: the top symbol is 16 % of
// the input) sends ~10 lanes of every wave to the same counter, and same-bank atomics
// serialise.  u8 path: every wave owns kHistCopies private copies of the 256 counters, a
// lane uses copy (lane & 7), and the copies start 8 banks apart, so the lanes that hit
// one symbol spread over eight banks; counters of symbols >= nsyms are simply counted and
// flagged at the end (no per-symbol range check).  u16 path (alphabets up to 4096): one
// table per block.
// ---------------------------------------------------------------------------
constexpr uint32_t kHistCopies = 8;
constexpr uint32_t kHistCopyStride = 256 + 8; // dwords: copy c starts in bank 8 * c

__global__ void __launch_bounds__(256) k_histogram_u8(const void *syms, uint64_t n, uint32_t nsyms, uint32_t *hist,
                                                      uint32_t *flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *all = reinterpret_cast<uint32_t *>(smem);
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t total = waves * kHistCopies * kHistCopyStride;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x)
        all[i] = 0;
    __syncthreads();
    uint32_t *h = all + ((threadIdx.x >> 6) * kHistCopies + (threadIdx.x & (kHistCopies - 1))) * kHistCopyStride;

    const uint8_t *p = static_cast<const uint8_t *>(syms);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 16-byte loads from the first aligned address on; the ragged head and tail go bytewise
    const uint64_t head = (16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    const uint64_t nhead = head < n ? head : n;
    const uint64_t nvec = (n - nhead) / 16;
    gvec_cptr pv = reinterpret_cast<gvec_cptr>(reinterpret_cast<uintptr_t>(p + nhead));
    auto count4 = [&](uint32_t w) {
        atomicAdd(&h[w & 0xffu], 1u);
        atomicAdd(&h[(w >> 8) & 0xffu], 1u);
        atomicAdd(&h[(w >> 16) & 0xffu], 1u);
        atomicAdd(&h[w >> 24], 1u);
    };
    uint64_t i = tid;
    if (i < nvec) {
        u32x4 v = __builtin_nontemporal_load(pv + i);
        for (i += stride; i < nvec; i += stride) { // next load in flight while this one is counted
            const u32x4 nv = __builtin_nontemporal_load(pv + i);
            count4(v.x);
            count4(v.y);
            count4(v.z);
            count4(v.w);
            v = nv;
        }
        count4(v.x);
        count4(v.y);
        count4(v.z);
        count4(v.w);
    }
    for (uint64_t j = tid; j < nhead; j += stride)
        atomicAdd(&h[p[j]], 1u);
    for (uint64_t j = nhead + nvec * 16 + tid; j < n; j += stride)
        atomicAdd(&h[p[j]], 1u);
    __syncthreads();

    bool bad = false;
    for (uint32_t b = threadIdx.x; b < 256u; b += blockDim.x) {
        uint32_t sum = 0;
        for (uint32_t c = 0; c < waves * kHistCopies; ++c)
            sum += all[c * kHistCopyStride + b];
        if (sum) {
            if (b < nsyms)
                atomicAdd(&hist[b], sum);
            else
                bad = true;
        }
    }
    if (bad)
        atomicOr(flags, 1u);
}

__global__ void __launch_bounds__(256) k_histogram_u16(const void *syms, uint64_t n, uint32_t nsyms, uint32_t *hist,
                                                       uint32_t *flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    for (uint32_t i = threadIdx.x; i < nsyms; i += blockDim.x)
        h[i] = 0;
    __syncthreads();
    bool bad = false;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint16_t *p = static_cast<const uint16_t *>(syms);
    auto count1 = [&](uint32_t sym) {
        if (sym < nsyms)
            atomicAdd(&h[sym], 1u);
        else
            bad = true;
    };
    // 16-byte loads (8 symbols) from the first aligned address on
    const uint64_t head = ((16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) / 2;
    const uint64_t nhead = head < n ? head : n;
    const uint64_t nvec = (n - nhead) / 8;
    gvec_cptr pv = reinterpret_cast<gvec_cptr>(reinterpret_cast<uintptr_t>(p + nhead));
    for (uint64_t i = tid; i < nvec; i += stride) {
        const u32x4 v = __builtin_nontemporal_load(pv + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            count1(w[a] & 0xffffu);
            count1(w[a] >> 16);
        }
    }
    for (uint64_t j = tid; j < nhead; j += stride)
        count1(p[j]);
    for (uint64_t j = nhead + nvec * 8 + tid; j < n; j += stride)
        count1(p[j]);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsyms; i += blockDim.x)
        if (h[i])
            atomicAdd(&hist[i], h[i]);
    if (bad)
        atomicOr(flags, 1u);
}

// ------------------------------------------------------------------ launchers

template <int FMT, int K, int OUT>
hipError_t launch_decode_t(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const uint32_t t0 = (p.table0_bytes + 15u) & ~15u, t1 = (p.table1_bytes + 15u) & ~15u;
    const uint32_t waves = kDecBlockThreads / 64;
    const size_t lds = (size_t)t0 + t1 + (size_t)waves * kRingStride + (OUT == OUT_FAST8_LDS ? waves * kOutTileBytes : 0);
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    auto kern = k_decode<FMT, K, OUT>;
    static bool attr_set = false; // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess)
            return e;
        attr_set = true;
    }
    const int blocks_per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
    uint64_t want = (p.nchunks + waves - 1) / waves;
    uint64_t cap = (uint64_t)num_cus * blocks_per_cu;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    if (name)
        *name = FMT == FMT_WORD ? "k_decode<word>" : FMT == FMT_BYTE ? "k_decode<byte>"
                : FMT == FMT_R64 ? "k_decode<r64>" : "k_decode<alias>";
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kDecBlockThreads), lds, stream, p);
    return hipGetLastError();
}

// RANS_AMD_LANES=staged | regwin: pin the lane-per-stream kernel generation (tests, A/B runs); read at
// every launch so one process can exercise both
static int lanes_force()
{
    const char *e = getenv("RANS_AMD_LANES");
    if (!e)
        return 0;
    return e[0] == 's' ? 1 : (e[0] == 'r' ? -1 : 0);
}

template <int FMT, int NW>
hipError_t launch_decode_lanes_t(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const uint32_t t0 = (p.table0_bytes + 15u) & ~15u, t1 = (p.table1_bytes + 15u) & ~15u;
    // RANS_AMD_LANES=regwin: the per-lane register window this kernel replaced (>= 4x over-fetch)
    const int force = lanes_force();
    const bool reg_window = force < 0;
    // staged kernel: the tables are shared by the block, every wave adds kLaneWaveLds of rings, so one
    // large block per CU keeps the most waves resident (rans64, 14 bits: 15 waves; 4-wave blocks: 12)
    const size_t table_lds = (size_t)t0 + t1;
    uint32_t sw = table_lds + kLaneWaveLds <= 160 * 1024 ? (uint32_t)((160 * 1024 - table_lds) / kLaneWaveLds) : 0;
    sw = sw > 16 ? 16 : sw;
    if (const char *e = getenv("RANS_AMD_LANES_WAVES")) { // experiment knob: waves per block
        const uint32_t v = (uint32_t)atoi(e);
        if (v >= 1 && v < sw)
            sw = v;
    }
    // 64 chunks of one wave must lie within 2^30 bytes (32-bit ring positions): any sane chunk size
    const bool staged = !reg_window && sw >= 1 && (uint64_t)p.chunk_syms * 8u < (1u << 22);
    {   // One batch (64 chunks) is a long latency-bound job, so a last round with a few waves per CU
        // costs as much as a full one: take the fewest rounds the LDS allows and split the batches
        // evenly over them (16 batches per CU: 16 waves x 1 round, or 8 x 2 -- never 14 + 2).
        const uint64_t batches = (p.nchunks + 63) / 64;
        const uint64_t per_cu = (batches + (uint64_t)num_cus - 1) / (uint64_t)num_cus;
        if (sw >= 1) {
            const uint64_t rounds = (per_cu + sw - 1) / sw;
            const uint64_t even = rounds ? (per_cu + rounds - 1) / rounds : 1;
            sw = (uint32_t)(even ? even : 1);
        }
    }
    const size_t lds = staged ? table_lds + (size_t)sw * kLaneWaveLds : table_lds;
    auto kern = staged ? k_decode_lanes_staged<FMT, NW> : k_decode_lanes<FMT, NW>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[staged]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess)
            return e;
        attr_set[staged] = true;
    }
    if (staged) {
        const uint64_t batches = (p.nchunks + 63) / 64;
        const uint64_t want_blocks = (batches + sw - 1) / sw;
        const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)num_cus ? want_blocks : (uint64_t)num_cus);
        if (name)
            *name = "k_decode_lanes_staged";
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * sw), lds, stream, p);
        return hipGetLastError();
    }
    const uint64_t want = (p.nchunks + 255) / 256;
    // 256-thread blocks: up to 8 per CU (32 waves) when the tables leave room in LDS; the kernel
    // is latency-bound (per-lane scattered loads), so residency matters more than anything else
    const size_t lds_room = lds ? (160 * 1024) / lds : 8;
    const uint64_t cap = (uint64_t)num_cus * (lds_room >= 8 ? 8 : (lds_room >= 1 ? lds_room : 1));
    const uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (name)
        *name = "k_decode_lanes";
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

// narrow interleaves with enough chunks to fill wavefronts: one chunk per lane
constexpr uint64_t kLaneKernelMinChunks = 64;

template <int FMT> hipError_t launch_decode_f(const DecParams &p, int num_cus, hipStream_t s, const char **name)
{
    if (p.nchunks >= kLaneKernelMinChunks) {
        switch (p.n_ways) {
        case 1: return launch_decode_lanes_t<FMT, 1>(p, num_cus, s, name);
        case 2: return launch_decode_lanes_t<FMT, 2>(p, num_cus, s, name);
        case 4: return launch_decode_lanes_t<FMT, 4>(p, num_cus, s, name);
        case 8: return launch_decode_lanes_t<FMT, 8>(p, num_cus, s, name);
        default: break;
        }
    }
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
    // A/B knobs (word format, 64-way only): alternatives that were measured and lost, kept so the
    // measurements in DESIGN.md can be repeated.
    static const bool no_asm = getenv("RANS_AMD_NO_ASM") != nullptr;   // compiler-scheduled renorm: -2 %
    static const bool lds_out = getenv("RANS_AMD_LDS_OUT") != nullptr;  // output via an LDS tile: -7 %
    static const bool byte_out = getenv("RANS_AMD_BYTE_OUT") != nullptr; // per-round byte stores: -5 %
    if constexpr (FMT == FMT_WORD) {
        if (fast && lds_out && p.n_ways == 64)
            return launch_decode_t<FMT_WORD, 1, OUT_FAST8_LDS>(p, num_cus, s, name);
        if (fast && byte_out && p.n_ways == 64)
            return launch_decode_t<FMT_WORD, 1, OUT_FAST8_BYTE>(p, num_cus, s, name);
    }
    if (FMT == FMT_WORD && fast && no_asm) {
        switch (p.n_ways) {
        case 64: return launch_decode_t<FMT_WORD, 1, OUT_FAST8_NOASM>(p, num_cus, s, name);
        case 128: return launch_decode_t<FMT_WORD, 2, OUT_FAST8_NOASM>(p, num_cus, s, name);
        case 256: return launch_decode_t<FMT_WORD, 4, OUT_FAST8_NOASM>(p, num_cus, s, name);
        default: break;
        }
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

template <int FMT, int K> hipError_t launch_encode_t(const EncParams &p, int num_cus, hipStream_t stream)
{
    const uint32_t waves = kEncBlockThreads / 64;
    const size_t lds = (size_t)p.nsyms * sizeof(EncRec) + (FMT == FMT_WORD ? 256 * sizeof(WordEncRec) : 0);
    if (lds > 128 * 1024 || (FMT == FMT_WORD && !p.word_enc_recs))
        return hipErrorInvalidValue;
    auto kern = k_encode<FMT, K>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess)
            return e;
        attr_set = true;
    }
    uint64_t want = (p.nchunks + waves - 1) / waves;
    uint64_t cap = (uint64_t)num_cus * 8;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kEncBlockThreads), lds, stream, p);
    return hipGetLastError();
}

template <int FMT, int NW> hipError_t launch_encode_lanes_t(const EncParams &p, int num_cus, hipStream_t stream)
{
    const size_t table_lds = (size_t)p.nsyms * sizeof(EncRec);
    if (table_lds > 128 * 1024)
        return hipErrorInvalidValue;
    // staged kernel (coalesced symbol loads, whole-line stream stores): u8 symbols in 16-byte aligned
    // chunks, slots made of whole lines; RANS_AMD_LANES=regwin keeps the per-lane kernel (A/B runs), =staged forces
    // this one whatever the batch count (tests)
    const int force = lanes_force();
    const bool reg_window = force < 0;
    uint32_t sw = table_lds + kEncWaveLds <= 160 * 1024 ? (uint32_t)((160 * 1024 - table_lds) / kEncWaveLds) : 0;
    sw = sw > 16 ? 16 : sw;
    const bool staged = !reg_window && sw >= 1 && p.sym_bytes == 1 && (p.slot_bytes % kLaneLine) == 0 &&
                        ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 15u) == 0 &&
                        (reinterpret_cast<uintptr_t>(p.scratch) & 15u) == 0 &&
                        // fewer, longer batches: the per-lane kernel's many small blocks hide latency better
                        (force > 0 || (p.nchunks + 63) / 64 >= (uint64_t)num_cus * 6);
    if (staged) {
        // same split as the staged decoder: fewest rounds, batches spread evenly over them
        const uint64_t batches = (p.nchunks + 63) / 64;
        const uint64_t per_cu = (batches + (uint64_t)num_cus - 1) / (uint64_t)num_cus;
        const uint64_t rounds = (per_cu + sw - 1) / sw;
        const uint64_t even = rounds ? (per_cu + rounds - 1) / rounds : 1;
        sw = (uint32_t)(even ? even : 1);
        auto kern = k_encode_lanes_staged<FMT, NW>;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess)
                return e;
            attr_set = true;
        }
        const uint64_t want_blocks = (batches + sw - 1) / sw;
        const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)num_cus ? want_blocks : (uint64_t)num_cus);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * sw), table_lds + (size_t)sw * kEncWaveLds, stream, p);
        return hipGetLastError();
    }
    const size_t lds = table_lds;
    auto kern = k_encode_lanes16<FMT, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess)
            return e;
        attr_set = true;
    }
    const uint64_t want = (p.nchunks + 255) / 256;
    const uint64_t cap = (uint64_t)num_cus * 8;
    const uint32_t grid = (uint32_t)(want < cap ? want : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int FMT> hipError_t launch_encode_f(const EncParams &p, int num_cus, hipStream_t s)
{
    if (p.nchunks >= kLaneKernelMinChunks) {
        switch (p.n_ways) {
        case 1: return launch_encode_lanes_t<FMT, 1>(p, num_cus, s);
        case 2: return launch_encode_lanes_t<FMT, 2>(p, num_cus, s);
        case 4: return launch_encode_lanes_t<FMT, 4>(p, num_cus, s);
        case 8: return launch_encode_lanes_t<FMT, 8>(p, num_cus, s);
        default: break;
        }
    }
    // K = ceil(N / 64) states per lane; lane counts that are not a multiple of 64 leave lanes idle
    if (p.n_ways >= 1 && p.n_ways <= 64)
        return launch_encode_t<FMT, 1>(p, num_cus, s);
    if (p.n_ways <= 128)
        return launch_encode_t<FMT, 2>(p, num_cus, s);
    if (p.n_ways <= 256)
        return launch_encode_t<FMT, 4>(p, num_cus, s);
    if (p.n_ways <= 512)
        return launch_encode_t<FMT, 8>(p, num_cus, s);
    return hipErrorInvalidValue;
}

} // namespace

uint32_t layout_blocks(uint64_t nchunks)
{
    const uint64_t b = (nchunks + kLayoutChunksPerBlock - 1) / kLayoutChunksPerBlock;
    return (uint32_t)(b ? b : 1);
}

bool ways_supported(int format, uint32_t n_ways)
{
    if (format < 0 || format > 3)
        return false;
    return n_ways >= 1 && n_ways <= 512;
}

hipError_t launch_decode(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **kernel_name)
{
    switch (format) {
    case FMT_WORD: return launch_decode_f<FMT_WORD>(p, num_cus, stream, kernel_name);
    case FMT_BYTE: return launch_decode_f<FMT_BYTE>(p, num_cus, stream, kernel_name);
    case FMT_R64: return launch_decode_f<FMT_R64>(p, num_cus, stream, kernel_name);
    case FMT_ALIAS: return launch_decode_f<FMT_ALIAS>(p, num_cus, stream, kernel_name);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_encode(int format, const EncParams &p, int num_cus, hipStream_t stream)
{
    switch (format) {
    case FMT_WORD: return launch_encode_f<FMT_WORD>(p, num_cus, stream);
    case FMT_BYTE: return launch_encode_f<FMT_BYTE>(p, num_cus, stream);
    case FMT_R64: return launch_encode_f<FMT_R64>(p, num_cus, stream);
    case FMT_ALIAS: return launch_encode_f<FMT_ALIAS>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_layout(const LayoutParams &p, hipStream_t stream)
{
    const uint32_t blocks = layout_blocks(p.nchunks);
    if (blocks > 1) {
        if (!p.block_sums)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_layout_sums, dim3(blocks), dim3(1024), 0, stream, p);
    }
    hipLaunchKernelGGL(k_layout, dim3(blocks), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_compact(const CompactParams &p, int num_cus, hipStream_t stream)
{
    uint64_t want = (p.nchunks + 3) / 4;
    uint64_t cap = (uint64_t)num_cus * 8;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_histogram(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *d_hist,
                            uint32_t *d_flags, int num_cus, hipStream_t stream)
{
    const uint32_t grid = (uint32_t)num_cus * 4;
    if (sym_bytes == 1) {
        const size_t lds = (size_t)(256 / 64) * kHistCopies * kHistCopyStride * 4;
        hipLaunchKernelGGL(k_histogram_u8, dim3(grid), dim3(256), lds, stream, syms, n, nsyms, d_hist, d_flags);
    } else {
        const size_t lds = (size_t)nsyms * 4;
        hipLaunchKernelGGL(k_histogram_u16, dim3(grid), dim3(256), lds, stream, syms, n, nsyms, d_hist, d_flags);
    }
    return hipGetLastError();
}

} // namespace rans_amd
