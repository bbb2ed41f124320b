// device_common.hpp -- device-side helpers shared by the kernel translation units (formats, lane
// utilities, the D step of every format, the exact divisions of the encoders).  Included inside
// each .hip file; everything lives in an anonymous namespace.
#pragma once

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
// Internal kernel format: rans64 (RANS_AMD_FMT_R64 to the caller) for the scale_bits the cum2sym decoder
// cannot take -- 17..31, where a 2^scale_bits lookup table fits no LDS, and 1..6, where its 24-bit partial
// products do not hold.  The symbol comes from a binary search over the cumulative frequencies
// (nsyms + 1 words in LDS), the state updates use full 64 x 32 multiplies.  Same stream, any
// scale_bits rans64.h accepts (rans64.h:169: <= 31); several times slower than the table decoder.
constexpr int FMT_R64S = 4;
template <int FMT> constexpr bool kIsR64 = (FMT == FMT_R64 || FMT == FMT_R64S);
// Internal kernel format of the ENCODER: alias coding (RANS_AMD_FMT_ALIAS to the caller) with the slot
// permutation alias_remap (main_alias.cpp:63,225-228) held in LDS as u16 next to 8-byte symbol records, for the
// models where both fit (2 M + 8 nsyms <= 160 KiB: every model up to 4096 symbols at 16 bits).  The general
// alias encoder gathers alias_remap from L2: 64 random dwords per sub-step drag 64 cache lines through the L1.
constexpr int FMT_ALIAS_LDS = 5;
template <int FMT> constexpr bool kIsAlias = (FMT == FMT_ALIAS || FMT == FMT_ALIAS_LDS);
// Internal kernel format of the DECODER: the word format (RANS_AMD_FMT_WORD to the caller) over an alphabet of more
// than 256 symbols -- SURVEY 8(f)4's "16-bit-symbol word format"; rans_word_sse41.h:41 fixes 256, the stream
// format itself does not care.  Slot record {freq, bias | sym << 16}: one more v_and than the byte-symbol record.
constexpr int FMT_WORD16 = 6;
// Internal kernel format of the DECODER: byte format with one model PER CHUNK: every wave builds
// cum2sym + symbol records of its chunk in its own LDS region from the chunk's 256 normalised frequencies
// (scale_bits <= 12: 4 KiB + 2 KiB per wave), so the tables are addressed through per-wave pointers.
constexpr int FMT_BYTEA = 7;
// Internal kernel formats of the two-chunks-per-wave DECODER (decode_dual.hip) for alias models: the half-bucket
// record is {sym | (M - freq) << 16, adjust} -- the update x' = x - adjust - (M - freq) * (x >> scale_bits) needs
// neither x mod M nor a mask on the frequency, and the low half IS the symbol a 16-bit store writes -- and the divider
// is held as the bucket's own-slot count (main_alias.cpp:209 `divider[i] = i * tgt + h0`, here h0 alone), one byte per
// bucket at LDS address 0 (FMT_ALIAS2, M / nsyms <= 255) or two (FMT_ALIAS2W).
constexpr int FMT_ALIAS2 = 8;
constexpr int FMT_ALIAS2W = 9;
template <int FMT> constexpr bool kIsAlias2 = (FMT == FMT_ALIAS2 || FMT == FMT_ALIAS2W);
// Internal kernel format of the DECODER: the byte format (RANS_AMD_FMT_BYTE to the caller) with the slot table of the word
// format -- one 8-byte record {freq | sym << 24, slot - start} per cumulative slot at LDS address 8 * slot -- for models whose
// table fits beside the stream windows (scale_bits <= 13).  D step: v_and, v_lshlrev, ds_read_b64, v_lshrrev, v_mad_u32_u24
// (rans_byte.h:125-128 + :291-298 as rans_word_sse41.h:123-131 does it): one gather instead of two dependent ones.
constexpr int FMT_BYTEF = 11;
// Internal kernel format: the WORD format (rans_word_sse41.h, 12-bit probabilities, 16-bit renormalisation) with one model
// PER CHUNK (SURVEY 8(f)3 on the headline's format): the decoder's waves build cum2sym + {freq, start} of their chunk as
// FMT_BYTEA does (a 4096-slot table per wave, rans_word_sse41.h:64-72, would be 32 KiB each) -- slot = x & 4095,
// x = freq * (x >> 12) + (slot - start) is the very update of rans_word_sse41.h:123-131; the encoder's waves build the
// general path's {freq, start, reciprocal} records.
constexpr int FMT_WORDA = 12;
template <int FMT> constexpr bool kIsAdaptive = (FMT == FMT_BYTEA || FMT == FMT_WORDA); // per-chunk models: per-wave tables
template <int FMT> constexpr bool kIsByteStream = (FMT == FMT_BYTE || FMT == FMT_ALIAS || FMT == FMT_ALIAS_LDS || FMT == FMT_BYTEA ||
                                                   kIsAlias2<FMT> || FMT == FMT_BYTEF);
template <int FMT> constexpr bool kIsWord = (FMT == FMT_WORD || FMT == FMT_WORD16 || FMT == FMT_WORDA);

// OUT_SLOW: element stores (any N, any alignment, u16 symbols).  OUT_FAST8: 4 rounds of u8 symbols transposed in
// registers, one dword store per lane.  OUT_FAST16: u16 symbols, 2 rounds packed per dword and swapped between lane pairs.
// (The 64-way word decoder has a kernel of its own: decode_wave.hip k_decode_word64.)
enum OutMode { OUT_SLOW = 0, OUT_FAST8 = 1, OUT_FAST16 = 4 };

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
template <> struct FmtTraits<FMT_R64S> : FmtTraits<FMT_R64> {};
template <> struct FmtTraits<FMT_ALIAS_LDS> : FmtTraits<FMT_ALIAS> {};
template <> struct FmtTraits<FMT_BYTEA> : FmtTraits<FMT_BYTE> {};
template <> struct FmtTraits<FMT_BYTEF> : FmtTraits<FMT_BYTE> {
    static constexpr int kSymByte = 3; // the slot record keeps the symbol in the top byte of its first word
};
template <> struct FmtTraits<FMT_ALIAS2> : FmtTraits<FMT_ALIAS> {};
template <> struct FmtTraits<FMT_ALIAS2W> : FmtTraits<FMT_ALIAS> {};
template <> struct FmtTraits<FMT_WORD16> : FmtTraits<FMT_WORD> {
    static constexpr int kSymByte = 0; // dec_step returns the symbol itself
};
template <> struct FmtTraits<FMT_WORDA> : FmtTraits<FMT_WORD> {
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

// Status words of the encoders' decoupled look-back (EncParams::status): top two bits = state, the rest a byte count.
// AGGREGATE: the unit (a chunk of the wave encoders, a 64-chunk batch of the lane encoders) is coded and this is its
// own aligned size; PREFIX: this is the inclusive sum up to and including the unit, i.e. where the next one starts.
constexpr unsigned long long kStAggregate = 1ull << 62, kStPrefix = 2ull << 62, kStValue = (1ull << 62) - 1;

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return (uint64_t)uniform((uint32_t)v) | ((uint64_t)uniform((uint32_t)(v >> 32)) << 32);
}

// explicit global address space: keeps loads/stores as global_* (not flat_*)
#define RANS_GLOBAL __attribute__((address_space(1)))
typedef const u32x4 RANS_GLOBAL *gvec_cptr;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Raw LDS addressing.  Every kernel here uses dynamic LDS only, which therefore starts at LDS address
// 0; the tables that sit first are addressed by their byte offset alone, which saves the VALU add of a
// link-time base the compiler cannot fold (`v_add_u32 v, 0, v`).  lds_starts_at_zero() is checked
// once per kernel; a kernel that ever failed the check reports an error instead of decoding.
#define RANS_LDS __attribute__((address_space(3)))
__device__ __forceinline__ bool lds_starts_at_zero(const uint8_t *smem)
{
    return (uint32_t)(uintptr_t)(RANS_LDS const uint8_t *)smem == 0u;
}
__device__ __forceinline__ uint32_t lds0_u8(uint32_t byte_offset)
{
    return *reinterpret_cast<RANS_LDS const uint8_t *>((uintptr_t)byte_offset);
}

// quad_perm DPP: lane i of each quad reads lane P[i]
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
}

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
    uint32_t lmaskv;       // alias: (M / nsyms) - 1, the mask of a slot's position inside its bucket
    uint32_t log2n;        // alias: log2(nsyms)

    __device__ __forceinline__ void init(const uint8_t *table0, const uint8_t *table1, uint32_t sb, uint32_t log2nsyms)
    {
        t0 = table0;
        t1 = table1;
        scale_bits = sb;
        mask = (1u << sb) - 1u;
        bucket_shift = FMT == FMT_R64S ? log2nsyms /* log2 of the padded cum table */ : sb - log2nsyms;
        mask12v = 0xfffu;
        maskv = mask;
        sbv = sb;
        bshiftv = bucket_shift;
        log2n = log2nsyms;
        lmaskv = (1u << (bucket_shift & 31u)) - 1u;
        asm volatile("v_mov_b32 %0, %0" : "+v"(lmaskv));
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
    } else if constexpr (FMT == FMT_WORD16) {
        // the same with a 16-bit symbol in the record's second word: {freq, bias | sym << 16}
        const uint2 e = reinterpret_cast<const uint2 *>(T.t0)[x & T.mask12v];
        x = (e.x & 0xffffffu) * (x >> 12) + (e.y & 0xffffu);
        return e.y >> 16;
    } else if constexpr (FMT == FMT_BYTE) {
        // rans_byte.h:125-128 (get), :291-298 (step)
        const uint32_t cf = x & T.maskv;
        const uint32_t s = lds0_u8(cf); // cum2sym is the first table in LDS (t0 == LDS address 0)
        const uint2 r = reinterpret_cast<const uint2 *>(T.t1)[s]; // {freq, start}
        // freq <= 2^16 and x >> scale_bits < 2^23 (scale_bits >= 8): 24-bit multiply is exact
        x = (r.x & 0xffffffu) * ((x >> T.sbv) & 0xffffffu) + cf - r.y;
        return s;
    } else if constexpr (FMT == FMT_BYTEF) {
        // the fused slot record at LDS address 8 * slot (the table is the first thing in LDS): freq <= 2^13 and
        // x >> scale_bits < 2^23 (scale_bits >= 8): the 24-bit multiply-add is exact, its mask drops the symbol byte
        const u32x2 e = *reinterpret_cast<RANS_LDS const u32x2 *>((uintptr_t)((x & T.maskv) << 3));
        x = (e.x & 0xffffffu) * ((x >> T.sbv) & 0xffffffu) + e.y;
        return e.x;
    } else if constexpr (FMT == FMT_BYTEA || FMT == FMT_WORDA) {
        // the same through the wave's own table pointers (per-chunk models; the word format: scale_bits is 12 and
        // freq * (x >> 12) + (slot - start) is rans_word_sse41.h:123-131's freq * (x >> 12) + bias).  Records packed into
        // four bytes, freq | start << 16 (both <= 4096): a kilobyte per wave instead of two -- the LDS a wave needs is
        // what bounds how many of these decoders a CU holds
        const uint32_t cf = x & T.maskv;
        const uint32_t s = T.t0[cf];
        const uint32_t r = reinterpret_cast<const uint32_t *>(T.t1)[s];
        x = (r & 0xffffu) * ((x >> T.sbv) & 0xffffffu) + cf - (r >> 16);
        return s;
    } else if constexpr (FMT == FMT_R64) {
        // rans64.h:118-121 (get), :286-292 (step)
        const uint32_t cf = (uint32_t)x & T.maskv;
        const uint32_t s = lds0_u8(cf); // cum2sym is the first table in LDS (t0 == LDS address 0)
        const uint2 r = reinterpret_cast<const uint2 *>(T.t1)[s];
        // freq * (x >> sb) + (cf - start) with x < 2^63: cf - start is in [0, freq), so it is a plain
        // 32-bit value; the high word of x >> sb is < 2^17 and freq <= 2^16, so its product is one
        // 24-bit multiply added to the high word -- one v_mad_u64_u32 instead of two plus a 64-bit
        // subtract-with-borrow
        const uint64_t xs = x >> T.scale_bits;
        const uint32_t bias = cf - r.y;
        x = (uint64_t)r.x * (uint32_t)xs + bias + ((uint64_t)__umul24(r.x, (uint32_t)(xs >> 32)) << 32);
        return s;
    } else if constexpr (FMT == FMT_R64S) {
        // rans64.h:118-121 (get) with the cum2sym lookup replaced by a search: t0 = cum[] padded with ~0 to a
        // power of two (2^bucket_shift entries, cum[0] = 0); the symbol is the LAST index whose cumulative
        // frequency is <= cf (symbols of frequency 0 share their successor's value and are never hit).
        // rans64.h:126-142 (advance): freq < 2^31 and x >> scale_bits < 2^(63 - scale_bits): the product is
        // below 2^63, computed in full.
        const uint32_t cf = (uint32_t)x & T.mask;
        uint32_t s = 0;
        for (uint32_t half = 1u << (T.bucket_shift - 1u); half; half >>= 1) {
            const uint32_t c = reinterpret_cast<const uint32_t *>(T.t0)[s + half];
            s += (c <= cf) ? half : 0u;
        }
        const uint2 r = reinterpret_cast<const uint2 *>(T.t1)[s]; // {freq, start}
        x = (uint64_t)r.x * (x >> T.scale_bits) + (cf - r.y);
        return s;
    } else if constexpr (kIsAlias2<FMT>) {
        // main_alias.cpp:252-267 with the tables of FMT_ALIAS2: t1 (own-slot counts) sits at LDS address 0.
        // bucket = xm >> (sb - log2 nsyms) straight from x; "xm < divider[bucket]" is "position in bucket < own count"
        const uint32_t bucket = __builtin_amdgcn_ubfe(x, T.bucket_shift, T.log2n);
        uint32_t own;
        if constexpr (FMT == FMT_ALIAS2)
            own = lds0_u8(bucket);
        else
            own = *reinterpret_cast<RANS_LDS const uint16_t *>((uintptr_t)(bucket * 2u));
        const uint32_t d = (x & T.lmaskv) - own;                       // negative: the bucket's own symbol (half 2b + 1)
        const uint32_t half = __builtin_amdgcn_alignbit(bucket, d, 31); // 2 * bucket + (d >> 31)
        const uint2 e = reinterpret_cast<const uint2 *>(T.t0)[half];   // {sym | (M - freq) << 16, adjust}
        // freq * (x >> sb) + xm - adjust with xm = x - (x >> sb) * M: x - adjust - (M - freq) * (x >> sb); M - freq is a
        // 16-bit value (freq >= 1 for every half a state can select) and x >> sb < 2^23: one 24-bit multiply
        x = x - e.y - __umul24(e.x >> 16, x >> T.sbv);
        return e.x;
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
// Per-chunk models (the reference builds one model per input, main.cpp:139-162): a wave turns the
// 256 normalised frequencies of ITS chunk (u16 each, sum = 1 << scale_bits, scale_bits 8..12) into the tables
// of the byte coder, in its own LDS region.  Lane l owns symbols 4l .. 4l+3.
// ---------------------------------------------------------------------------
constexpr uint32_t kAdaptMaxScaleBits = 12;
constexpr uint32_t kAdaptDecWaveLds = (1u << kAdaptMaxScaleBits) + 256u * 4u; // cum2sym + packed {freq | start << 16} records
constexpr uint32_t kAdaptEncWaveLds = 256u * 16u;                             // EncRec per symbol

// frequencies and exclusive cumulative frequencies of this lane's four symbols
__device__ __forceinline__ void adapt_load_cum(const uint16_t *chunk_freqs, uint32_t lane, uint32_t (&f)[4], uint32_t (&c)[4])
{
    const u32x2 v = *reinterpret_cast<const u32x2 RANS_GLOBAL *>(reinterpret_cast<uint64_t>(chunk_freqs) + 8u * lane);
    f[0] = v.x & 0xffffu;
    f[1] = v.x >> 16;
    f[2] = v.y & 0xffffu;
    f[3] = v.y >> 16;
    const uint32_t own = f[0] + f[1] + f[2] + f[3];
    uint32_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
        incl += lane >= (uint32_t)d ? t : 0u;
    }
    c[0] = incl - own;
    c[1] = c[0] + f[0];
    c[2] = c[1] + f[1];
    c[3] = c[2] + f[2];
}

// exclusive cumulative frequencies of a lane's four symbols from the frequencies themselves (registers)
__device__ __forceinline__ void adapt_cum(const uint32_t (&f)[4], uint32_t lane, uint32_t (&c)[4])
{
    const uint32_t own = f[0] + f[1] + f[2] + f[3];
    uint32_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
        incl += lane >= (uint32_t)d ? t : 0u;
    }
    c[0] = incl - own;
    c[1] = c[0] + f[0];
    c[2] = c[1] + f[1];
    c[3] = c[2] + f[2];
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// SymbolStats::normalize_freqs (main.cpp:75-129) for ONE chunk on ONE wave, lane l owning symbols 4l .. 4l+3 (the width-based
// restatement of model.cpp normalize_freqs):
//   edge[s]  = target * (counts[0] + .. + counts[s]) / total          (64-bit product, truncating division)
//   width[s] = edge[s] - edge[s-1]
//   every symbol that occurs but got width 0, in ascending order, takes one slot from the narrowest symbol wider than
//   1 (lowest index on ties) -- a sequential repair: one wave-wide arg-min per squeezed symbol.
// Returns false (wave-uniform) when the chunk cannot be normalised (more distinct symbols than slots).
__device__ __forceinline__ bool adapt_normalize(const uint32_t (&cnt)[4], uint32_t nsym, uint32_t target, uint32_t lane, uint32_t (&width)[4])
{
    // inclusive running sums in symbol order; the total is the chunk's symbol count
    const uint32_t own = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    uint32_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
        incl += lane >= (uint32_t)d ? t : 0u;
    }
    uint32_t run = incl - own;
    uint32_t edge[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        run += cnt[i];
        edge[i] = (uint32_t)(((uint64_t)target * run) / nsym); // (nsym >= 1: a chunk holds at least one symbol)
    }
    uint32_t prev = (uint32_t)__shfl_up((int)edge[3], 1, 64);
    prev = lane ? prev : 0u;
    width[0] = edge[0] - prev;
    width[1] = edge[1] - edge[0];
    width[2] = edge[2] - edge[1];
    width[3] = edge[3] - edge[2];
    // repair, in ascending symbol order
    for (;;) {
        uint32_t mine = 256u; // this lane's lowest squeezed symbol
#pragma unroll
        for (int i = 3; i >= 0; --i)
            mine = (cnt[i] != 0u && width[i] == 0u) ? 4u * lane + (uint32_t)i : mine;
        const uint32_t s = wave_min_u32(mine);
        if (s >= 256u)
            return true;
        uint32_t key = 0xffffffffu; // (width << 8 | symbol) of this lane's narrowest symbol wider than 1
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            const uint32_t k = (width[i] << 8) | (4u * lane + (uint32_t)i);
            key = (width[i] > 1u && k < key) ? k : key;
        }
        const uint32_t best = wave_min_u32(key);
        if (best == 0xffffffffu) // nobody can give a slot away
            return false;
        const uint32_t victim = best & 0xffu;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            width[i] -= (4u * lane + (uint32_t)i == victim) ? 1u : 0u;
            width[i] = (4u * lane + (uint32_t)i == s) ? 1u : width[i];
        }
    }
}

// decoder tables (main.cpp:143-148 cum2sym, :159-162 RansDecSymbolInit): recs[s] = freq | start << 16, cum2sym[M]
// Returns false (wave-uniform) when the frequencies do not sum to 1 << scale_bits: they come from the caller's
// container, and a table built from them must not be walked (the fill below relies on the sum).
__device__ __forceinline__ bool adapt_build_dec(const uint16_t *chunk_freqs, uint32_t scale_bits, uint32_t lane,
                                                uint8_t *cum2sym, uint32_t *recs)
{
    uint32_t f[4], c[4];
    adapt_load_cum(chunk_freqs, lane, f, c);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)(c[3] + f[3]), 63);
    if (total != (1u << scale_bits))
        return false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        recs[4u * lane + i] = f[i] | (c[i] << 16); // (a frequency of 4096 = M still fits: the sum check above bounds both)
    // (LDS operations of one wave execute in order: the reads below see every lane's records)
    const uint32_t per = (1u << scale_bits) >> 6; // positions per lane: 4 .. 64
    uint32_t pos = lane * per;
    uint32_t s = 0; // the last symbol whose start is <= pos; symbols of frequency 0 are stepped over below
#pragma unroll
    for (uint32_t half = 128; half; half >>= 1)
        s += (recs[s + half] >> 16) <= pos ? half : 0u;
    uint32_t end = (recs[s] >> 16) + (recs[s] & 0xffffu);
    for (uint32_t i = 0; i < per; i += 4) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            while (pos >= end && s < 255u) { // next symbol with a slot at pos (pos < M = the last symbol's end)
                ++s;
                end = (recs[s] >> 16) + (recs[s] & 0xffffu);
            }
            w |= s << (8 * b);
            ++pos;
        }
        *reinterpret_cast<uint32_t *>(cum2sym + pos - 4u) = w;
    }
    return true;
}

// encoder records (RansEncSymbolInit, rans_byte.h:174-243), in the layout enc_update_byte reads:
// {freq | rshift << 24, bias, rcp, 0}; zero records for symbols the chunk does not hold
__device__ __forceinline__ void adapt_build_enc(const uint16_t *chunk_freqs, uint32_t scale_bits, uint32_t lane, uint4 *recs)
{
    uint32_t f[4], c[4];
    adapt_load_cum(chunk_freqs, lane, f, c);
    const uint32_t M = 1u << scale_bits;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 r = {0u, 0u, 0u, 0u};
        if (f[i] == 1u) {
            r = uint4{1u, c[i] + M - 1u, 0xffffffffu, 0u};
        } else if (f[i] >= 2u) {
            const uint32_t sh = 32u - (uint32_t)__builtin_clz(f[i] - 1u); // ceil(log2 freq)
            const uint32_t rcp = (uint32_t)(((1ull << (sh + 31u)) + f[i] - 1u) / f[i]);
            r = uint4{f[i] | ((sh - 1u) << 24), c[i], rcp, 0u};
        }
        recs[4u * lane + i] = r;
    }
}

// the same for the WORD format (FMT_WORDA; scale_bits = 12): the general path's records {freq, bias, m', cmpl | sh << 24}
// with the round-up reciprocal of Granlund & Montgomery for 32-bit dividends, exactly as model.cpp builds them for a
// whole-input model (rans_word_sse41.h:81-93's x / freq and x % freq without a division in the loop)
__device__ __forceinline__ void adapt_build_enc_word(const uint16_t *chunk_freqs, uint32_t lane, uint4 *recs)
{
    uint32_t f[4], c[4];
    adapt_load_cum(chunk_freqs, lane, f, c);
    const uint32_t M = 1u << 12;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 r = {0u, 0u, 0u, 0u};
        if (f[i] == 1u) {
            r = uint4{1u, c[i] + M - 1u, 0xffffffffu, M - 1u};
        } else if (f[i] >= 2u) {
            const uint32_t l = 32u - (uint32_t)__builtin_clz(f[i] - 1u); // ceil(log2 freq)
            const unsigned long long mprime = ((1ull << 32) * ((1ull << l) - f[i])) / f[i] + 1ull;
            r = uint4{f[i], c[i], (uint32_t)mprime, (M - f[i]) | ((l - 1u) << 24)};
        }
        recs[4u * lane + i] = r;
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

// Word-format encoder update for a renormalised state y (any 32-bit value: y < 2^20 * freq reaches 2^32
// for freq > 2048, beyond the 31-bit range of the Alverson reciprocal): round-up method of Granlund &
// Montgomery, t = mulhi(y, m'); q = (t + ((y - t) >> 1)) >> sh, exact for freq >= 2, q = y - 1 for
// freq == 1 (m' = 2^32 - 1, sh = 0); y + bias + q * (4096 - freq) == (y / freq << 12) + y % freq + start
// (rans_word_sse41.h:92).  rec = {freq, bias, m', cmpl | sh << 24} (model.cpp).
__device__ __forceinline__ uint32_t enc_update_word(uint32_t y, const uint4 &rec)
{
    const uint32_t t = __umulhi(y, rec.z);
    const uint32_t q = (t + ((y - t) >> 1)) >> (rec.w >> 24);
    return y + rec.y + __umul24(q, rec.w);
}

// Byte-format encoder update for a renormalised state y < 2^31, the reference's form
// (RansEncPutSymbol, rans_byte.h:258-280): q = mulhi(y, rcp) >> rshift is floor(y / freq) exactly for
// freq >= 2 (Alverson, rans_byte.h:201-243) and y - 1 for freq == 1; y + bias + q * (M - freq) then
// equals (floor(y / freq) << scale_bits) + y % freq + start.  rec = {freq | rshift << 24, bias, rcp, -}.
// q < 2^(31 - scale_bits) <= 2^23 and M - freq <= 2^16: one 24-bit multiply.
__device__ __forceinline__ uint32_t enc_update_byte(uint32_t y, const uint4 &rec, uint32_t scale_bits)
{
    const uint32_t q = __umulhi(y, rec.z) >> (rec.x >> 24);
    const uint32_t cmpl = (1u << scale_bits) - (rec.x & 0xffffffu);
    return y + rec.y + __umul24(q, cmpl);
}

// rans64 encoder update C(s, y) for a renormalised state y < 2^63, in the form the reference uses
// (Rans64EncPutSymbol, rans64.h:262-278): q = mulhi64(y, rcp) >> rshift is floor(y / freq) exactly for
// freq >= 2 (Alverson reciprocal, rans64.h:167-247) and y - 1 for freq == 1 (rcp = 2^64 - 1), and
//   y + bias + q * (M - freq)   ==   (floor(y / freq) << scale_bits) + y % freq + start
// with bias = start (freq >= 2) or start + M - 1 (freq == 1): no remainder, no correction step, no
// selects.  rec = {freq | rshift << 24, bias, rcp lo, rcp hi} (model.cpp).  q < 2^49 and
// M - freq < 2^16: the high word of q needs only a 24-bit multiply.
// The same for any scale_bits up to 31 (FMT_R64S): rec = {freq, bias, rcp lo, rcp hi}; the shift is
// ceil(log2 freq) - 1 as in rans64.h:207-240, recomputed from freq (there is no room for it in the record), and
// q * (M - freq) is a full 64 x 32 multiply (M - freq reaches 2^31, q 2^62 / freq).
__device__ __forceinline__ uint64_t enc_update_r64s(uint64_t y, const uint4 &rec, uint32_t scale_bits)
{
    const uint64_t rcp = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
    const uint32_t freq = rec.x;
    const uint32_t rshift = freq >= 2u ? 31u - (uint32_t)__builtin_clz(freq - 1u) : 0u; // ceil(log2 freq) - 1
    const uint64_t q = __umul64hi(y, rcp) >> rshift;
    const uint32_t cmpl = (uint32_t)((1ull << scale_bits) - freq);
    return y + rec.y + q * cmpl;
}

__device__ __forceinline__ uint64_t enc_update_r64(uint64_t y, const uint4 &rec, uint32_t scale_bits)
{
    const uint64_t rcp = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
    const uint64_t q = __umul64hi(y, rcp) >> (rec.x >> 24);
    const uint32_t cmpl = (1u << scale_bits) - (rec.x & 0xffffffu);
    return y + rec.y + (uint64_t)(uint32_t)q * cmpl + ((uint64_t)__umul24((uint32_t)(q >> 32), cmpl) << 32);
}

// Mailbox between the coding waves of a block and its copier wave(s) (fused placement, encode_wave.hip k_encode and
// lanes.hip): the coder pushes {unit + 1, bytes} when its unit (chunk / batch of 64 chunks) is in the scratch slots,
// a copier takes the entries in order.  Lives in the block's LDS (EncParams::mailbox_off), zeroed at kernel start.
struct EncMailbox {
    uint32_t tail;     // entries handed out to encoders
    uint32_t claim;    // entries handed out to copiers
    uint32_t finished; // encoder waves that have left their loop
    uint32_t pad;
    uint2 entries[64]; // {unit + 1, stream bytes}; x == 0: empty
};
static_assert(sizeof(EncMailbox) == kEncMailboxBytes, "mailbox layout");

// Every wait of the placement protocols gives up after kWaitSeconds of WALL time (the 100 MHz clock, read every 256th
// poll) and sets a bit of EncParams::flags -- 8: mailbox full, 16: a copier / scanner waiting for its coders, 32:
// look-back, 64: a coder waiting for its batch's place, 128: a coder waiting for its next unit, 256: a coder waiting for
// its scratch slot -- which the host turns into an error: a protocol bug must show up as a failed call, not as a GPU
// that hangs for ever.  Half a minute, not microseconds: a wait is as long as the coding of the largest chunk somebody
// else is still busy with (a 2^31-symbol chunk of a *_host call keeps one wave busy for seconds).  (Rounds 1-2 counted
// polls, 2^28 of them: between 25 seconds and nine minutes depending on what a poll costs -- a protocol bug in round 3
// held a GPU box for a quarter of an hour.)
// The limit travels with the launch (EncParams::wait_ticks): api.cpp adds to the half minute what ONE wave may legitimately
// need for the largest chunk of the call -- a 2^31-symbol chunk of a 1-way stream keeps a lane busy for minutes, and the
// waits for it are as long.
constexpr unsigned long long kWaitTicks = 30ull * 100000000ull;
constexpr uint32_t kProtocolErrorBits = ~7u; // EncParams::flags: everything but bad symbol / no space / LDS layout
struct SpinWatch {
    unsigned long long t0 = 0;
    unsigned long long limit;
    uint32_t polls = 0;
    __device__ __forceinline__ explicit SpinWatch(unsigned long long limit_ticks) : limit(limit_ticks ? limit_ticks : kWaitTicks) {}
    // true when the wait should be abandoned: it has lasted kWaitTicks, or some other wait of this launch has given up
    // already (its flag is set: the launch has failed, and nobody should sit through the timeout a second time --
    // a failed look-back would otherwise cost every chunk behind it another half minute)
    __device__ __forceinline__ bool expired(const uint32_t *flags)
    {
        if ((++polls & 255u) != 0u)
            return false;
        if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kProtocolErrorBits)
            return true;
        const unsigned long long t = wall_clock64();
        if (t0 == 0) {
            t0 = t;
            return false;
        }
        return t - t0 > limit;
    }
};

// Decoupled look-back over the status words of units [0, unit): the sum of the predecessors' values, i.e. where `unit`
// starts.  64 predecessors per trip (lane j reads status[unit - 1 - j]) down to the first published inclusive PREFIX (unit
// 0's virtual predecessor is one), waiting for predecessors that have not published their AGGREGATE yet.  Whole wave;
// returns false when the wait was abandoned (`timeout_bit` of *flags set, or somebody else's protocol error seen).
__device__ __forceinline__ bool status_lookback(const unsigned long long *status, uint64_t unit, uint32_t lane, uint32_t *flags,
                                                unsigned long long wait_ticks, uint32_t timeout_bit, unsigned long long &base_out)
{
    unsigned long long base = 0;
    SpinWatch watch(wait_ticks);
    for (uint64_t j = unit;;) { // status[j-1], status[j-2], ... are still to be added
        unsigned long long st = kStPrefix; // virtual predecessor of unit 0: an inclusive prefix of 0
        if (lane < j)
            st = __hip_atomic_load(status + (j - 1 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t ready = __builtin_amdgcn_ballot_w64((st >> 62) != 0);
        const uint64_t pref = __builtin_amdgcn_ballot_w64((st >> 62) == 2);
        const uint32_t first_pref = pref ? (uint32_t)__builtin_ctzll(pref) : 64u;
        const uint64_t need = first_pref >= 63u ? ~0ull : ((2ull << first_pref) - 1ull); // lanes 0 .. first_pref
        if ((ready & need) != need) { // a predecessor in that range has not published yet
            if (watch.expired(flags)) { // (a protocol error must not hang the GPU)
                if (lane == 0)
                    atomicOr(flags, timeout_bit);
                base_out = base;
                return false;
            }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        unsigned long long v = lane <= first_pref ? (st & kStValue) : 0ull;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
            const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
            v += (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
        base += uniform64(v);
        if (first_pref < 64u)
            break;
        j -= 64;
    }
    base_out = base;
    return true;
}

__device__ __forceinline__ void mailbox_push(EncMailbox *mb, uint32_t unit, uint32_t len, uint32_t *flags, unsigned long long wait_ticks)
{
    const uint32_t i = atomicAdd(&mb->tail, 1u) & 63u;
    volatile uint2 *e = &mb->entries[i];
    SpinWatch watch(wait_ticks);
    while (e->x != 0u) { // (64 entries for at most 15 encoders: the copier would have to be 4 units per encoder behind)
        if (watch.expired(flags)) {
            atomicOr(flags, 8u);
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    *reinterpret_cast<volatile unsigned long long *>(e) = (unsigned long long)(unit + 1u) | ((unsigned long long)len << 32);
}

// Copier side (whole wave): the next entry in order, {ex = unit + 1, ey = bytes}; false when every one of the block's
// `producers` coding waves has left its loop and nothing is left to take.
__device__ __forceinline__ bool mailbox_pop(EncMailbox *mb, uint32_t lane, uint32_t producers, uint32_t &ex, uint32_t &ey,
                                            uint32_t *flags, unsigned long long wait_ticks)
{
    uint32_t h = 0;
    if (lane == 0)
        h = atomicAdd(&mb->claim, 1u);
    h = uniform(h);
    volatile uint2 *e = &mb->entries[h & 63u];
    for (SpinWatch watch(wait_ticks);;) {
        const unsigned long long ev = *reinterpret_cast<volatile unsigned long long *>(e);
        ex = uniform((uint32_t)ev);
        ey = uniform((uint32_t)(ev >> 32));
        if (ex != 0u)
            break;
        // nothing there: done when every encoder has left and fewer than h + 1 entries were ever pushed
        const uint32_t fin = uniform(*reinterpret_cast<volatile uint32_t *>(&mb->finished));
        const uint32_t tail = uniform(*reinterpret_cast<volatile uint32_t *>(&mb->tail));
        if (fin == producers && (int32_t)(tail - h) <= 0)
            return false;
        __builtin_amdgcn_s_sleep(8);
        if (watch.expired(flags)) { // (an idle copier polls for as long as its block codes: the clock restarts with every entry)
            atomicOr(flags, lane == 0 ? 16u : 0u);
            return false;
        }
    }
    // (every lane writes the same zero: a trailing `if (lane == 0)` invites the compiler to let the other lanes run ahead
    //  into code whose readfirstlane / ballot assumes the whole wave -- see lanes.hip, lanes_scan_batch)
    *reinterpret_cast<volatile unsigned long long *>(e) = 0ull;
    return true;
}

} // namespace

} // namespace rans_amd
