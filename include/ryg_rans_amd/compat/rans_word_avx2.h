/*
 * rans_word_avx2.h -- 8-lane AVX2 decoder for the word-aligned rANS format (an EXTENSION: the
 * reference ships a 4-lane SSE4.1 decoder only and merely mentions an unpublished AVX2 one,
 * README:120-122).  Decodes the same 8-way interleaved streams as two RansSimdDec do
 * (main_simd.cpp:313-332), with one vector: lane i of the vector is coder i, all eight share one
 * stream and renormalise in lane order.  Needs -mavx2; C++14.
 *
 * Written as the candidate "stronger CPU baseline" of SURVEY.md 8(f).  Measured (Xeon @ 2.6 GHz, 16 MiB
 * of Zipf bytes, tests/compat_avx2_driver.cpp): 7.8 clocks/symbol on 8-way streams against 4.5 for the
 * two interleaved SSE4.1 vectors (one vector is one dependency chain: gather + multiply latency is
 * exposed), 6.2 with 2 or 4 vectors on 16-/32-way streams (two vpgatherdd per 8 symbols cost more than
 * the eight scalar loads + pinsrd they replace).  So the SSE4.1 decoder stays the CPU baseline of
 * bench.py; this header is kept, tested (tests/test_compat_headers.py), for hosts with fast gathers.
 * The GPU path does not depend on it.
 *
 *   RansAvx2DecInit(&r, &ptr);                       // 8 states, 32 bytes
 *   for (i = 0; i + 8 <= n; i += 8) {
 *       uint64_t s = RansAvx2DecSym(&r, &tab);       // 8 symbols, lane 0 in the low byte
 *       memcpy(out + i, &s, 8);
 *       RansAvx2DecRenorm(&r, &ptr);                 // reads 16 bytes at ptr whatever the mask is
 *   }
 */
#ifndef RYG_RANS_AMD_COMPAT_RANS_WORD_AVX2_H
#define RYG_RANS_AMD_COMPAT_RANS_WORD_AVX2_H

#include "rans_word_compat.h"

#if !defined(__AVX2__)
#error "rans_word_avx2.h needs -mavx2"
#endif
#include <immintrin.h>

typedef union {
    __m256i simd;
    uint32_t lane[8];
} RansAvx2Dec;

static inline void RansAvx2DecInit(RansAvx2Dec *r, uint16_t **pptr)
{
    r->simd = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(*pptr));
    *pptr += 16; /* eight states of two words each */
}

/* Eight symbols, lane 0 in the low byte.  Both tables are gathered: the slot records as dwords, the
 * symbols as the TOP byte of the dword that ends at slot2sym[slot] (the three bytes below it are the
 * end of the slots array or earlier symbols -- always inside the struct). */
static inline uint64_t RansAvx2DecSym(RansAvx2Dec *r, RansWordTables const *tab)
{
    const __m256i x = r->simd;
    const __m256i slot = _mm256_and_si256(x, _mm256_set1_epi32(RANS_WORD_M - 1));
    const __m256i fb = _mm256_i32gather_epi32(reinterpret_cast<const int *>(tab->slots), slot, 4);
    const __m256i sy = _mm256_i32gather_epi32(reinterpret_cast<const int *>(tab->slot2sym - 3), slot, 1);
    const __m256i freq = _mm256_and_si256(fb, _mm256_set1_epi32(0xffff));
    const __m256i bias = _mm256_srli_epi32(fb, 16);
    /* freq < 2^12 and x >> 12 < 2^20: the low 32 bits of the product are exact */
    r->simd = _mm256_add_epi32(_mm256_mullo_epi32(_mm256_srli_epi32(x, RANS_WORD_SCALE_BITS), freq), bias);
    /* byte 3 of every dword -> 8 consecutive bytes */
    const __m256i pick = _mm256_setr_epi8(3, 7, 11, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          3, 7, 11, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i b = _mm256_shuffle_epi8(sy, pick);
    const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(b);
    const uint32_t hi = (uint32_t)_mm256_extract_epi32(b, 4);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

/* control words of vpermd for every 8-bit "needs a word" mask: lane i with its bit set takes stream word
 * number popcount(mask & ((1 << i) - 1)); the other lanes' entries are don't-care (blended away) */
struct RansAvx2RenormTable {
    uint32_t ctl[256][8];
    uint8_t words[256];
    constexpr RansAvx2RenormTable() : ctl(), words()
    {
        for (int mask = 0; mask < 256; mask++) {
            int next = 0;
            for (int lane = 0; lane < 8; lane++) {
                ctl[mask][lane] = (uint32_t)next;
                if (mask & (1 << lane))
                    next++;
            }
            words[mask] = (uint8_t)next;
        }
    }
};

/* NOTE: reads 16 bytes at *pptr whatever the mask is; keep 16 bytes of padding behind the stream. */
static inline void RansAvx2DecRenorm(RansAvx2Dec *r, uint16_t **pptr)
{
    static constexpr RansAvx2RenormTable tbl = RansAvx2RenormTable();
    const __m256i x = r->simd;
    /* x < 2^16  <=>  high half is zero */
    const __m256i low = _mm256_cmpeq_epi32(_mm256_srli_epi32(x, 16), _mm256_setzero_si256());
    const int mask = _mm256_movemask_ps(_mm256_castsi256_ps(low));
    const __m256i next = _mm256_cvtepu16_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i *>(*pptr)));
    const __m256i ctl = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(tbl.ctl[mask]));
    const __m256i words = _mm256_permutevar8x32_epi32(next, ctl);
    const __m256i refilled = _mm256_or_si256(_mm256_slli_epi32(x, 16), words);
    r->simd = _mm256_blendv_epi8(x, refilled, low);
    *pptr += tbl.words[mask];
}

#endif /* RYG_RANS_AMD_COMPAT_RANS_WORD_AVX2_H */
