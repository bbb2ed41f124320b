/*
 * rans_word_compat.h -- host-side, source-compatible re-provision of the word-aligned rANS
 * coder and its 4-lane SSE4.1 decoder (API of rygorous/ryg_rans rans_word_sse41.h:35-227),
 * written from the format definition.  C++ only (anonymous struct in
 * a union, as the original API exposes one); the SIMD part needs -msse4.1.
 *
 * Format: state x in [2^16, 2^32), 12-bit probabilities (M = 4096), 256 symbols, 16-bit
 * renormalisation units, at most one per symbol.  The decoder is table driven: one record
 * per cumulative slot {freq, slot - start} plus the symbol owning the slot, so
 *     x = freq * (x >> 12) + bias;   if (x < 2^16) x = x << 16 | next word.
 * This is also the table the GPU kernel stages in LDS (ryg_rans_amd/csrc/model.h WordSlot).
 */
#ifndef RYG_RANS_AMD_COMPAT_RANS_WORD_H
#define RYG_RANS_AMD_COMPAT_RANS_WORD_H

#include <stdint.h>
#include <string.h>

#if defined(__SSE4_1__)
#include <smmintrin.h>
#define RANS_WORD_COMPAT_SIMD 1
#endif

#define RANS_WORD_L (1u << 16)
#define RANS_WORD_SCALE_BITS 12
#define RANS_WORD_M (1u << RANS_WORD_SCALE_BITS)
#define RANS_WORD_NSYMS 256

typedef uint32_t RansWordEnc;
typedef uint32_t RansWordDec;

union RansWordSlot {
    uint32_t u32;
    struct {
        uint16_t freq;
        uint16_t bias;
    };
};

struct RansWordTables {
    RansWordSlot slots[RANS_WORD_M];
    uint8_t slot2sym[RANS_WORD_M];
};

/* every slot of [start, start+freq) learns its symbol, the symbol's width and its own rank */
static inline void RansWordTablesInitSymbol(RansWordTables *tab, uint8_t sym, uint32_t start, uint32_t freq)
{
    for (uint32_t rank = 0; rank < freq; rank++) {
        RansWordSlot &s = tab->slots[start + rank];
        s.freq = (uint16_t)freq;
        s.bias = (uint16_t)rank;
        tab->slot2sym[start + rank] = sym;
    }
}

/* ---- scalar coder ------------------------------------------------------------- */

static inline RansWordEnc RansWordEncInit() { return RANS_WORD_L; }

static inline void RansWordEncPut(RansWordEnc *r, uint16_t **pptr, uint32_t start, uint32_t freq)
{
    uint32_t x = *r;
    const uint32_t x_max = ((RANS_WORD_L >> RANS_WORD_SCALE_BITS) << 16) * freq; /* 32-bit product on purpose */
    if (x >= x_max) {
        uint16_t *p = *pptr - 1;
        *p = (uint16_t)x;
        *pptr = p;
        x >>= 16;
    }
    const uint32_t q = x / freq;
    *r = (q << RANS_WORD_SCALE_BITS) + (x - q * freq) + start;
}

static inline void RansWordEncFlush(RansWordEnc *r, uint16_t **pptr)
{
    uint16_t *p = *pptr - 2;
    p[0] = (uint16_t)(*r);
    p[1] = (uint16_t)(*r >> 16);
    *pptr = p;
}

static inline void RansWordDecInit(RansWordDec *r, uint16_t **pptr)
{
    const uint16_t *p = *pptr;
    *r = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
    *pptr += 2;
}

static inline uint8_t RansWordDecSym(RansWordDec *r, RansWordTables const *tab)
{
    const uint32_t x = *r;
    const uint32_t slot = x % RANS_WORD_M;
    const RansWordSlot rec = tab->slots[slot];
    *r = rec.freq * (x >> RANS_WORD_SCALE_BITS) + rec.bias;
    return tab->slot2sym[slot];
}

static inline void RansWordDecRenorm(RansWordDec *r, uint16_t **pptr)
{
    if (*r < RANS_WORD_L) {
        *r = (*r << 16) | **pptr;
        *pptr += 1;
    }
}

/* ---- 4-lane SSE4.1 decoder --------------------------------------------------------
 * Lane i of the vector is coder i; all four share one stream and renormalise in lane
 * order, exactly like four scalar decoders called in turn. */
#ifdef RANS_WORD_COMPAT_SIMD

typedef union {
    __m128i simd;
    uint32_t lane[4];
} RansSimdDec;

static inline void RansSimdDecInit(RansSimdDec *r, uint16_t **pptr)
{
    r->simd = _mm_loadu_si128(reinterpret_cast<const __m128i *>(*pptr));
    *pptr += 8; /* four states of two words each */
}

/* returns the four symbols, lane 0 in the low byte.  The four table records are gathered
 * with pextrd/pinsrd so nothing bounces through memory (a store of four scalars followed
 * by a vector load would miss store forwarding). */
static inline uint32_t RansSimdDecSym(RansSimdDec *r, RansWordTables const *tab)
{
    const __m128i x = r->simd;
    const __m128i slots = _mm_and_si128(x, _mm_set1_epi32(RANS_WORD_M - 1));
    const uint32_t i0 = (uint32_t)_mm_cvtsi128_si32(slots);
    const uint32_t i1 = (uint32_t)_mm_extract_epi32(slots, 1);
    const uint32_t i2 = (uint32_t)_mm_extract_epi32(slots, 2);
    const uint32_t i3 = (uint32_t)_mm_extract_epi32(slots, 3);
    __m128i fb = _mm_cvtsi32_si128((int)tab->slots[i0].u32);
    fb = _mm_insert_epi32(fb, (int)tab->slots[i1].u32, 1);
    fb = _mm_insert_epi32(fb, (int)tab->slots[i2].u32, 2);
    fb = _mm_insert_epi32(fb, (int)tab->slots[i3].u32, 3);
    const uint32_t syms = (uint32_t)tab->slot2sym[i0] | ((uint32_t)tab->slot2sym[i1] << 8) |
                          ((uint32_t)tab->slot2sym[i2] << 16) | ((uint32_t)tab->slot2sym[i3] << 24);
    const __m128i freq = _mm_and_si128(fb, _mm_set1_epi32(0xffff));
    const __m128i bias = _mm_srli_epi32(fb, 16);
    /* freq < 2^12 and x >> 12 < 2^20: the low 32 bits of the product are exact */
    r->simd = _mm_add_epi32(_mm_mullo_epi32(_mm_srli_epi32(x, RANS_WORD_SCALE_BITS), freq), bias);
    return syms;
}

/* NOTE: like the original API this reads 8 bytes at *pptr whatever the mask is; keep 8 bytes
 * of padding behind the stream (the GPU path has no such requirement). */
static inline void RansSimdDecRenorm(RansSimdDec *r, uint16_t **pptr)
{
    /* pshufb controls: the lanes set in the mask receive consecutive stream words, in lane
     * order, in their low half; -1 writes zero.  Literal so no initialisation guard runs. */
    static const int8_t ctl_tab[16][16] __attribute__((aligned(16))) = {
        {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, /* lanes 0000 */
        { 0,  1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, /* lanes 0001 */
        {-1, -1, -1, -1,  0,  1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, /* lanes 0010 */
        { 0,  1, -1, -1,  2,  3, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, /* lanes 0011 */
        {-1, -1, -1, -1, -1, -1, -1, -1,  0,  1, -1, -1, -1, -1, -1, -1}, /* lanes 0100 */
        { 0,  1, -1, -1, -1, -1, -1, -1,  2,  3, -1, -1, -1, -1, -1, -1}, /* lanes 0101 */
        {-1, -1, -1, -1,  0,  1, -1, -1,  2,  3, -1, -1, -1, -1, -1, -1}, /* lanes 0110 */
        { 0,  1, -1, -1,  2,  3, -1, -1,  4,  5, -1, -1, -1, -1, -1, -1}, /* lanes 0111 */
        {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,  0,  1, -1, -1}, /* lanes 1000 */
        { 0,  1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,  2,  3, -1, -1}, /* lanes 1001 */
        {-1, -1, -1, -1,  0,  1, -1, -1, -1, -1, -1, -1,  2,  3, -1, -1}, /* lanes 1010 */
        { 0,  1, -1, -1,  2,  3, -1, -1, -1, -1, -1, -1,  4,  5, -1, -1}, /* lanes 1011 */
        {-1, -1, -1, -1, -1, -1, -1, -1,  0,  1, -1, -1,  2,  3, -1, -1}, /* lanes 1100 */
        { 0,  1, -1, -1, -1, -1, -1, -1,  2,  3, -1, -1,  4,  5, -1, -1}, /* lanes 1101 */
        {-1, -1, -1, -1,  0,  1, -1, -1,  2,  3, -1, -1,  4,  5, -1, -1}, /* lanes 1110 */
        { 0,  1, -1, -1,  2,  3, -1, -1,  4,  5, -1, -1,  6,  7, -1, -1}, /* lanes 1111 */
    };
    /* companion control applied to the states: a refilled lane moves its low half up */
    static const int8_t shift_tab[16][16] __attribute__((aligned(16))) = {
        { 0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15}, /* lanes 0000 */
        {-1, -1,  0,  1,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15}, /* lanes 0001 */
        { 0,  1,  2,  3, -1, -1,  4,  5,  8,  9, 10, 11, 12, 13, 14, 15}, /* lanes 0010 */
        {-1, -1,  0,  1, -1, -1,  4,  5,  8,  9, 10, 11, 12, 13, 14, 15}, /* lanes 0011 */
        { 0,  1,  2,  3,  4,  5,  6,  7, -1, -1,  8,  9, 12, 13, 14, 15}, /* lanes 0100 */
        {-1, -1,  0,  1,  4,  5,  6,  7, -1, -1,  8,  9, 12, 13, 14, 15}, /* lanes 0101 */
        { 0,  1,  2,  3, -1, -1,  4,  5, -1, -1,  8,  9, 12, 13, 14, 15}, /* lanes 0110 */
        {-1, -1,  0,  1, -1, -1,  4,  5, -1, -1,  8,  9, 12, 13, 14, 15}, /* lanes 0111 */
        { 0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, -1, -1, 12, 13}, /* lanes 1000 */
        {-1, -1,  0,  1,  4,  5,  6,  7,  8,  9, 10, 11, -1, -1, 12, 13}, /* lanes 1001 */
        { 0,  1,  2,  3, -1, -1,  4,  5,  8,  9, 10, 11, -1, -1, 12, 13}, /* lanes 1010 */
        {-1, -1,  0,  1, -1, -1,  4,  5,  8,  9, 10, 11, -1, -1, 12, 13}, /* lanes 1011 */
        { 0,  1,  2,  3,  4,  5,  6,  7, -1, -1,  8,  9, -1, -1, 12, 13}, /* lanes 1100 */
        {-1, -1,  0,  1,  4,  5,  6,  7, -1, -1,  8,  9, -1, -1, 12, 13}, /* lanes 1101 */
        { 0,  1,  2,  3, -1, -1,  4,  5, -1, -1,  8,  9, -1, -1, 12, 13}, /* lanes 1110 */
        {-1, -1,  0,  1, -1, -1,  4,  5, -1, -1,  8,  9, -1, -1, 12, 13}, /* lanes 1111 */
    };
    static const uint8_t words_tab[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
    const __m128i x = r->simd;
    /* x < 2^16  <=>  high half is zero */
    const __m128i low = _mm_cmpeq_epi32(_mm_srli_epi32(x, 16), _mm_setzero_si128());
    const int mask = _mm_movemask_ps(_mm_castsi128_ps(low));
    const __m128i next = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(*pptr));
    const __m128i ctl = _mm_load_si128(reinterpret_cast<const __m128i *>(ctl_tab[mask]));
    const __m128i shf = _mm_load_si128(reinterpret_cast<const __m128i *>(shift_tab[mask]));
    r->simd = _mm_or_si128(_mm_shuffle_epi8(x, shf), _mm_shuffle_epi8(next, ctl));
    *pptr += words_tab[mask];
}

#endif /* RANS_WORD_COMPAT_SIMD */

#endif /* RYG_RANS_AMD_COMPAT_RANS_WORD_H */
