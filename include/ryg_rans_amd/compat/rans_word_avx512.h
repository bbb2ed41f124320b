/*
 * rans_word_avx512.h -- 16-lane AVX-512 decoder for the word-aligned rANS format (an EXTENSION: the reference ships
 * a 4-lane SSE4.1 decoder and mentions an unpublished AVX2 one, README:120-122; SURVEY.md 8(f)4 asks for a stronger
 * CPU baseline).  One vector is 16 coders of one N-way interleaved stream, lane i = coder i, renormalising in lane
 * order like RansSimdDecRenorm (rans_word_sse41.h:182-227) -- but where SSE4.1 needs a movemask, two 16-entry pshufb
 * tables and a blend, AVX-512 has the operation itself:
 *
 *     mask  = x < 2^16                       (vpcmpud -> k register)
 *     words = expand(mask, next 16 words)    (vpexpandd: the i-th set lane takes the i-th word)
 *     x     = mask ? x << 16 | words : x     (masked vpord)
 *     ptr  += popcount(mask)
 *
 * and the D step needs ONE gather per 16 symbols from a 4-byte packed slot record {freq:12 | bias:12 | sym:8}
 * (RansWordTables512; the reference keeps {u16 freq, u16 bias} and a separate slot2sym byte, rans_word_sse41.h:50-61:
 * two lookups).  freq <= 4095 always: a one-symbol model is outside the word format.
 *
 * Two vectors decode a 32-way stream (RansAvx512DecSym / Renorm on each in turn, like the two RansSimdDec of
 * main_simd.cpp:313-325 decode an 8-way one).  Needs -mavx512f; C++11.  Reads 32 bytes at *pptr whatever the mask is:
 * keep 32 bytes of padding behind the stream (the SSE4.1 decoder needs 8, rans_word_sse41.h:218-220).
 */
#ifndef RYG_RANS_AMD_COMPAT_RANS_WORD_AVX512_H
#define RYG_RANS_AMD_COMPAT_RANS_WORD_AVX512_H

#include "rans_word_compat.h"

#if !defined(__AVX512F__)
#error "rans_word_avx512.h needs -mavx512f"
#endif
#include <immintrin.h>

struct RansWordTables512 {
    uint32_t slots[RANS_WORD_M]; /* freq | bias << 12 | sym << 24 */
};

static inline void RansWordTables512Init(RansWordTables512 *t, RansWordTables const *tab)
{
    for (uint32_t s = 0; s < RANS_WORD_M; s++)
        t->slots[s] = (uint32_t)tab->slots[s].freq | ((uint32_t)tab->slots[s].bias << 12) | ((uint32_t)tab->slot2sym[s] << 24);
}

typedef union {
    __m512i simd;
    uint32_t lane[16];
} RansAvx512Dec;

static inline void RansAvx512DecInit(RansAvx512Dec *r, uint16_t **pptr)
{
    r->simd = _mm512_loadu_si512(reinterpret_cast<const void *>(*pptr));
    *pptr += 32; /* sixteen states of two words each */
}

/* Sixteen symbols, lane 0 in byte 0. */
static inline __m128i RansAvx512DecSym(RansAvx512Dec *r, RansWordTables512 const *tab)
{
    const __m512i x = r->simd;
    const __m512i slot = _mm512_and_si512(x, _mm512_set1_epi32(RANS_WORD_M - 1));
    const __m512i e = _mm512_i32gather_epi32(slot, reinterpret_cast<const void *>(tab->slots), 4);
    const __m512i freq = _mm512_and_si512(e, _mm512_set1_epi32(0xfff));
    const __m512i bias = _mm512_and_si512(_mm512_srli_epi32(e, 12), _mm512_set1_epi32(0xfff));
    /* freq < 2^12 and x >> 12 < 2^20: the low 32 bits of the product are exact */
    r->simd = _mm512_add_epi32(_mm512_mullo_epi32(_mm512_srli_epi32(x, RANS_WORD_SCALE_BITS), freq), bias);
    return _mm512_cvtepi32_epi8(_mm512_srli_epi32(e, 24));
}

static inline void RansAvx512DecRenorm(RansAvx512Dec *r, uint16_t **pptr)
{
    const __m512i x = r->simd;
    const __mmask16 need = _mm512_cmplt_epu32_mask(x, _mm512_set1_epi32((int)RANS_WORD_L));
    const __m512i next = _mm512_cvtepu16_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(*pptr)));
    const __m512i words = _mm512_maskz_expand_epi32(need, next);
    r->simd = _mm512_mask_or_epi32(x, need, _mm512_slli_epi32(x, 16), words);
    *pptr += _mm_popcnt_u32((unsigned)need);
}

#endif /* RYG_RANS_AMD_COMPAT_RANS_WORD_AVX512_H */
