/*
 * rans_byte_compat.h -- host-side, source-compatible re-provision of the byte-aligned rANS
 * primitives (API of rygorous/ryg_rans rans_byte.h:50-318), written from the format
 * definition, for callers that keep using the per-symbol API on the
 * CPU next to the bulk GPU ABI (include/ryg_rans_amd.h).  C89-compatible apart from stdint.
 *
 * Model: state x in [2^23, 2^31), probabilities scaled to M = 1 << scale_bits (<= 16).
 *   encode  (reverse order, stream grows DOWN from the end of the buffer):
 *           shift bytes out until x < ((2^23 >> scale_bits) << 8) * freq,
 *           then x = (x / freq) * M + x % freq + start
 *   decode  x = freq * (x / M) + x % M - start, then shift bytes in while x < 2^23
 * Several coders may share one stream provided the decoder mirrors the encoder's order.
 */
#ifndef RYG_RANS_AMD_COMPAT_RANS_BYTE_H
#define RYG_RANS_AMD_COMPAT_RANS_BYTE_H

#include <stdint.h>

#ifdef assert
#define RansAssert assert
#else
#define RansAssert(x)
#endif

#define RANS_BYTE_L (1u << 23)

typedef uint32_t RansState;

typedef struct {
    uint32_t x_max;     /* renormalise while x >= x_max */
    uint32_t rcp_freq;  /* fixed-point 1/freq */
    uint32_t bias;
    uint16_t cmpl_freq; /* M - freq */
    uint16_t rcp_shift;
} RansEncSymbol;

typedef struct {
    uint16_t start;
    uint16_t freq;
} RansDecSymbol;

/* ---- shared pieces --------------------------------------------------------- */

static inline uint32_t rans_compat_byte_xmax(uint32_t freq, uint32_t scale_bits)
{
    return ((RANS_BYTE_L >> scale_bits) << 8) * freq;
}

static inline RansState rans_compat_byte_shift_out(RansState x, uint8_t **pptr, uint32_t x_max)
{
    uint8_t *p = *pptr;
    while (x >= x_max) {
        *--p = (uint8_t)x;
        x >>= 8;
    }
    *pptr = p;
    return x;
}

static inline RansState rans_compat_byte_shift_in(RansState x, uint8_t **pptr)
{
    uint8_t *p = *pptr;
    while (x < RANS_BYTE_L)
        x = (x << 8) | *p++;
    *pptr = p;
    return x;
}

/* ---- encoder ---------------------------------------------------------------- */

static inline void RansEncInit(RansState *r) { *r = RANS_BYTE_L; }

static inline RansState RansEncRenorm(RansState x, uint8_t **pptr, uint32_t freq, uint32_t scale_bits)
{
    return rans_compat_byte_shift_out(x, pptr, rans_compat_byte_xmax(freq, scale_bits));
}

static inline void RansEncPut(RansState *r, uint8_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    RansState x = RansEncRenorm(*r, pptr, freq, scale_bits);
    uint32_t q = x / freq;
    *r = (q << scale_bits) + (x - q * freq) + start;
}

static inline void RansEncFlush(RansState *r, uint8_t **pptr)
{
    uint8_t *p = *pptr - 4;
    uint32_t x = *r;
    int i;
    for (i = 0; i < 4; i++)
        p[i] = (uint8_t)(x >> (8 * i));
    *pptr = p;
}

/* Reciprocal form: q = floor(x / freq) from a multiply-high, exact for x < 2^31.
 * With shift = ceil(log2 freq) and rcp = ceil(2^(shift+31) / freq) the identity
 * x_new = x + bias + q * (M - freq) reproduces RansEncPut; freq == 1 uses rcp = 2^32 - 1
 * (q = x - 1) and folds the missing M - 1 into bias. */
static inline void RansEncSymbolInit(RansEncSymbol *s, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    const uint32_t M = 1u << scale_bits;
    RansAssert(scale_bits <= 16);
    RansAssert(start <= M);
    RansAssert(freq <= M - start);
    s->x_max = rans_compat_byte_xmax(freq, scale_bits);
    s->cmpl_freq = (uint16_t)(M - freq);
    if (freq >= 2) {
        uint32_t sh = 0;
        while ((1u << sh) < freq)
            sh++;
        s->rcp_freq = (uint32_t)((((uint64_t)1 << (sh + 31)) + freq - 1) / freq);
        s->rcp_shift = (uint16_t)(sh - 1);
        s->bias = start;
    } else {
        s->rcp_freq = 0xffffffffu;
        s->rcp_shift = 0;
        s->bias = start + M - 1;
    }
}

static inline void RansEncPutSymbol(RansState *r, uint8_t **pptr, RansEncSymbol const *sym)
{
    RansState x;
    uint32_t q;
    RansAssert(sym->x_max != 0);
    x = rans_compat_byte_shift_out(*r, pptr, sym->x_max);
    q = (uint32_t)(((uint64_t)x * sym->rcp_freq) >> 32) >> sym->rcp_shift;
    *r = x + sym->bias + q * sym->cmpl_freq;
}

/* ---- decoder ---------------------------------------------------------------- */

static inline void RansDecInit(RansState *r, uint8_t **pptr)
{
    uint8_t *p = *pptr;
    *r = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    *pptr = p + 4;
}

static inline uint32_t RansDecGet(RansState *r, uint32_t scale_bits) { return *r & ((1u << scale_bits) - 1); }

static inline void RansDecAdvanceStep(RansState *r, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    RansState x = *r;
    *r = freq * (x >> scale_bits) + (x & ((1u << scale_bits) - 1)) - start;
}

static inline void RansDecRenorm(RansState *r, uint8_t **pptr) { *r = rans_compat_byte_shift_in(*r, pptr); }

static inline void RansDecAdvance(RansState *r, uint8_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits)
{
    RansDecAdvanceStep(r, start, freq, scale_bits);
    RansDecRenorm(r, pptr);
}

static inline void RansDecSymbolInit(RansDecSymbol *s, uint32_t start, uint32_t freq)
{
    RansAssert(start <= (1 << 16));
    RansAssert(freq <= (1 << 16) - start);
    s->start = (uint16_t)start;
    s->freq = (uint16_t)freq;
}

static inline void RansDecAdvanceSymbol(RansState *r, uint8_t **pptr, RansDecSymbol const *sym, uint32_t scale_bits)
{
    RansDecAdvance(r, pptr, sym->start, sym->freq, scale_bits);
}

static inline void RansDecAdvanceSymbolStep(RansState *r, RansDecSymbol const *sym, uint32_t scale_bits)
{
    RansDecAdvanceStep(r, sym->start, sym->freq, scale_bits);
}

#endif /* RYG_RANS_AMD_COMPAT_RANS_BYTE_H */
