/*
 * rans64_compat.h -- host-side, source-compatible re-provision of the 64-bit-state rANS
 * primitives (API of rygorous/ryg_rans rans64.h:59-316), written from the format definition
 * (SURVEY.md appendix A).  State x in [2^31, 2^63), 32-bit renormalisation units (one at
 * most per symbol), scale_bits <= 31.  Needs a 64x64->128 multiply (__int128 or _umul128).
 */
#ifndef RYG_RANS_AMD_COMPAT_RANS64_H
#define RYG_RANS_AMD_COMPAT_RANS64_H

#include <stdint.h>

#ifdef assert
#define Rans64Assert assert
#else
#define Rans64Assert(x)
#endif

#if defined(_MSC_VER)
#include <intrin.h>
static inline uint64_t Rans64MulHi(uint64_t a, uint64_t b) { return __umulh(a, b); }
#elif defined(__SIZEOF_INT128__)
static inline uint64_t Rans64MulHi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
#else
#error rans64 needs a 64x64 -> high 64 multiply
#endif

#define RANS64_L (1ull << 31)

typedef uint64_t Rans64State;

typedef struct {
    uint64_t rcp_freq;
    uint32_t freq;
    uint32_t bias;
    uint32_t cmpl_freq;
    uint32_t rcp_shift;
} Rans64EncSymbol;

typedef struct {
    uint32_t start;
    uint32_t freq;
} Rans64DecSymbol;

static inline uint64_t rans_compat_r64_xmax(uint32_t freq, uint32_t scale_bits)
{
    return ((RANS64_L >> scale_bits) << 32) * freq;
}

/* emit at most one 32-bit unit (the interval is wide enough that one always suffices) */
static inline uint64_t rans_compat_r64_shift_out(uint64_t x, uint32_t **pptr, uint64_t x_max)
{
    if (x >= x_max) {
        uint32_t *p = *pptr - 1;
        *p = (uint32_t)x;
        *pptr = p;
        x >>= 32;
        Rans64Assert(x < x_max);
    }
    return x;
}

static inline void Rans64EncInit(Rans64State *r) { *r = RANS64_L; }

static inline void Rans64EncPut(Rans64State *r, uint32_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    uint64_t x, q;
    Rans64Assert(freq != 0);
    x = rans_compat_r64_shift_out(*r, pptr, rans_compat_r64_xmax(freq, scale_bits));
    q = x / freq;
    *r = (q << scale_bits) + (x - q * freq) + start;
}

static inline void Rans64EncFlush(Rans64State *r, uint32_t **pptr)
{
    uint32_t *p = *pptr - 2;
    p[0] = (uint32_t)(*r);
    p[1] = (uint32_t)(*r >> 32);
    *pptr = p;
}

static inline void Rans64DecInit(Rans64State *r, uint32_t **pptr)
{
    uint32_t *p = *pptr;
    *r = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    *pptr = p + 2;
}

static inline uint32_t Rans64DecGet(Rans64State *r, uint32_t scale_bits)
{
    return (uint32_t)*r & ((1u << scale_bits) - 1);
}

static inline void Rans64DecAdvanceStep(Rans64State *r, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    uint64_t x = *r;
    *r = freq * (x >> scale_bits) + (x & (((uint64_t)1 << scale_bits) - 1)) - start;
}

static inline void Rans64DecRenorm(Rans64State *r, uint32_t **pptr)
{
    uint64_t x = *r;
    if (x < RANS64_L) {
        x = (x << 32) | **pptr;
        *pptr += 1;
        Rans64Assert(x >= RANS64_L);
    }
    *r = x;
}

static inline void Rans64DecAdvance(Rans64State *r, uint32_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    Rans64DecAdvanceStep(r, start, freq, scale_bits);
    Rans64DecRenorm(r, pptr);
}

/* rcp = ceil(2^(shift+63) / freq) with shift = ceil(log2 freq): exact quotients for x < 2^63 */
static inline void Rans64EncSymbolInit(Rans64EncSymbol *s, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    const uint32_t M = 1u << scale_bits;
    Rans64Assert(scale_bits <= 31);
    Rans64Assert(start <= M);
    Rans64Assert(freq <= M - start);
    s->freq = freq;
    s->cmpl_freq = M - freq;
    if (freq >= 2) {
        uint32_t sh = 0;
        uint64_t hi_q, hi_r, lo_q;
        while ((1u << sh) < freq)
            sh++;
        /* long division of (2^(sh+63) + freq - 1) by freq in two 64-bit steps */
        hi_q = ((uint64_t)1 << (sh + 31)) / freq;
        hi_r = ((uint64_t)1 << (sh + 31)) % freq;
        lo_q = ((hi_r << 32) + (freq - 1)) / freq;
        s->rcp_freq = (hi_q << 32) + lo_q;
        s->rcp_shift = sh - 1;
        s->bias = start;
    } else {
        s->rcp_freq = ~(uint64_t)0;
        s->rcp_shift = 0;
        s->bias = start + M - 1;
    }
}

static inline void Rans64DecSymbolInit(Rans64DecSymbol *s, uint32_t start, uint32_t freq)
{
    Rans64Assert(start <= (1u << 31));
    Rans64Assert(freq <= (1u << 31) - start);
    s->start = start;
    s->freq = freq;
}

static inline void Rans64EncPutSymbol(Rans64State *r, uint32_t **pptr, Rans64EncSymbol const *sym, uint32_t scale_bits)
{
    uint64_t x, q;
    Rans64Assert(sym->freq != 0);
    x = rans_compat_r64_shift_out(*r, pptr, rans_compat_r64_xmax(sym->freq, scale_bits));
    q = Rans64MulHi(x, sym->rcp_freq) >> sym->rcp_shift;
    *r = x + sym->bias + q * sym->cmpl_freq;
}

static inline void Rans64DecAdvanceSymbol(Rans64State *r, uint32_t **pptr, Rans64DecSymbol const *sym, uint32_t scale_bits)
{
    Rans64DecAdvance(r, pptr, sym->start, sym->freq, scale_bits);
}

static inline void Rans64DecAdvanceSymbolStep(Rans64State *r, Rans64DecSymbol const *sym, uint32_t scale_bits)
{
    Rans64DecAdvanceStep(r, sym->start, sym->freq, scale_bits);
}

#endif /* RYG_RANS_AMD_COMPAT_RANS64_H */
