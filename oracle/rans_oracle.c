/*
 * rans_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see rans_oracle.h).
 *
 * Every routine names the reference lines it restates.  Nothing here is shared
 * with the product library.  The code is written for clarity and for arbitrary
 * lane counts N; the reference itself only ships N = 1, 2 (all formats) and
 * N = 8 (word format).  For those N the streams produced here are byte-identical
 * to the reference's (pinned by tests/test_oracle_golden.py).
 */
#include "rans_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */

static inline uint32_t sym_at(const void *syms, size_t i, int sym_bytes)
{
    return sym_bytes == 1 ? ((const uint8_t *)syms)[i] : ((const uint16_t *)syms)[i];
}

static inline void sym_put(void *out, size_t i, int sym_bytes, uint32_t s)
{
    if (sym_bytes == 1)
        ((uint8_t *)out)[i] = (uint8_t)s;
    else
        ((uint16_t *)out)[i] = (uint16_t)s;
}

static uint32_t ilog2_exact(uint32_t v)
{
    uint32_t l = 0;
    while ((1u << l) < v)
        l++;
    return l;
}

/* ------------------------------------------------------------------ model */

/* main.cpp:59-66 (count_freqs) */
void orc_count_freqs(const void *syms, size_t n, int sym_bytes, uint32_t nsyms, uint32_t *freqs)
{
    memset(freqs, 0, sizeof(uint32_t) * nsyms);
    for (size_t i = 0; i < n; i++) {
        uint32_t s = sym_at(syms, i, sym_bytes);
        if (s < nsyms)
            freqs[s]++;
    }
}

/* main.cpp:68-129 (calc_cum_freqs + normalize_freqs), main_alias.cpp:83-144 for
 * the NSYMS-generic form.  Steps: prefix-sum the counts; rescale the cumulative
 * table to target_total with 64-bit intermediates; any present symbol squeezed
 * to width 0 steals one slot from the narrowest symbol that still has width > 1
 * (first such symbol in index order), shifting the boundaries in between. */
int orc_normalize_freqs(uint32_t *freqs, uint32_t *cum, uint32_t nsyms, uint32_t target_total)
{
    if (target_total < nsyms)
        return ORC_E_ARG;

    cum[0] = 0;
    for (uint32_t s = 0; s < nsyms; s++)
        cum[s + 1] = cum[s] + freqs[s];
    uint32_t total = cum[nsyms];
    if (total == 0)
        return ORC_E_ARG;

    for (uint32_t s = 1; s <= nsyms; s++)
        cum[s] = (uint32_t)(((uint64_t)target_total * cum[s]) / total);

    for (uint32_t s = 0; s < nsyms; s++) {
        if (freqs[s] == 0 || cum[s + 1] != cum[s])
            continue;
        /* victim = smallest width > 1, lowest index wins ties */
        uint32_t best_w = 0xffffffffu;
        int64_t victim = -1;
        for (uint32_t j = 0; j < nsyms; j++) {
            uint32_t w = cum[j + 1] - cum[j];
            if (w > 1 && w < best_w) {
                best_w = w;
                victim = j;
            }
        }
        if (victim < 0)
            return ORC_E_ARG;
        if ((uint32_t)victim < s) {
            for (uint32_t j = (uint32_t)victim + 1; j <= s; j++)
                cum[j]--;
        } else {
            for (uint32_t j = s + 1; j <= (uint32_t)victim; j++)
                cum[j]++;
        }
    }

    if (cum[0] != 0 || cum[nsyms] != target_total)
        return ORC_E_ARG;
    for (uint32_t s = 0; s < nsyms; s++) {
        uint32_t w = cum[s + 1] - cum[s];
        if ((freqs[s] == 0) != (w == 0))
            return ORC_E_ARG;
        freqs[s] = w;
    }
    return ORC_OK;
}

/* main_alias.cpp:147-237 (make_alias_table), NSYMS-generic.  Two passes:
 * (1) a Vose-style sweep pairs each under-full bucket ("small") with a donor
 * ("large") and records how many of the bucket's tgt slots stay with its own
 * symbol; (2) buckets are laid out in index order, handing every symbol's slots
 * out in increasing order and recording, per bucket half, the symbol, its
 * frequency and the offset that maps a cumulative value back to "k-th slot of
 * the symbol"; alias_remap is the encoder-side inverse of that map. */
static int build_alias(orc_model *m)
{
    const uint32_t ns = m->nsyms;
    const uint32_t M = 1u << m->scale_bits;
    if (ns == 0 || (ns & (ns - 1)) || M % ns != 0 || M < ns)
        return ORC_E_ARG;
    const uint32_t tgt = M / ns;

    m->divider = (uint32_t *)malloc(sizeof(uint32_t) * ns);
    m->slot_adjust = (uint32_t *)malloc(sizeof(uint32_t) * 2 * ns);
    m->slot_freqs = (uint32_t *)malloc(sizeof(uint32_t) * 2 * ns);
    m->sym_id = (uint32_t *)malloc(sizeof(uint32_t) * 2 * ns);
    m->alias_remap = (uint32_t *)malloc(sizeof(uint32_t) * M);
    uint32_t *left = (uint32_t *)malloc(sizeof(uint32_t) * ns);
    uint32_t *given = (uint32_t *)calloc(ns, sizeof(uint32_t));
    if (!m->divider || !m->slot_adjust || !m->slot_freqs || !m->sym_id || !m->alias_remap || !left || !given) {
        free(left);
        free(given);
        return ORC_E_ARG;
    }

    for (uint32_t i = 0; i < ns; i++) {
        left[i] = m->freqs[i];
        m->divider[i] = tgt;
        m->sym_id[2 * i] = i;
        m->sym_id[2 * i + 1] = i;
    }

    /* pass 1: main_alias.cpp:166-204 */
    uint32_t large = 0, small = 0;
    while (large < ns && left[large] < tgt)
        large++;
    while (small < ns && left[small] >= tgt)
        small++;
    uint32_t small_next = small + 1;

    while (large < ns && small < ns) {
        m->sym_id[2 * small] = large;
        m->divider[small] = left[small];
        left[large] -= tgt - m->divider[small];

        if (left[large] >= tgt || small_next <= large) {
            small = small_next;
            while (small < ns && left[small] >= tgt)
                small++;
            small_next = small + 1;
        } else {
            small = large; /* the donor just became under-full and lies behind us */
        }
        while (large < ns && left[large] < tgt)
            large++;
    }

    /* pass 2: main_alias.cpp:206-236 */
    for (uint32_t i = 0; i < ns; i++) {
        uint32_t j = m->sym_id[2 * i];
        uint32_t own_h = m->divider[i];
        uint32_t donor_h = tgt - own_h;
        uint32_t own_base = given[i];
        uint32_t donor_base = given[j];
        uint32_t own_c = m->cum[i] + own_base;
        uint32_t donor_c = m->cum[j] + donor_base;

        m->divider[i] = i * tgt + own_h;
        m->slot_freqs[2 * i + 1] = m->freqs[i];
        m->slot_freqs[2 * i] = m->freqs[j];
        m->slot_adjust[2 * i + 1] = i * tgt - own_base;
        m->slot_adjust[2 * i] = i * tgt - (donor_base - own_h);
        for (uint32_t k = 0; k < own_h; k++)
            m->alias_remap[own_c + k] = i * tgt + k;
        for (uint32_t k = 0; k < donor_h; k++)
            m->alias_remap[donor_c + k] = i * tgt + own_h + k;

        given[i] += own_h;
        given[j] += donor_h;
    }

    int ok = 1;
    for (uint32_t i = 0; i < ns; i++)
        if (given[i] != m->freqs[i])
            ok = 0;
    free(left);
    free(given);
    return ok ? ORC_OK : ORC_E_ARG;
}

orc_model *orc_model_create(const uint32_t *norm_freqs, uint32_t nsyms, uint32_t scale_bits, int with_alias)
{
    if (!norm_freqs || nsyms == 0 || scale_bits == 0 || scale_bits > 31)
        return NULL;
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    if (!m)
        return NULL;
    m->nsyms = nsyms;
    m->log2nsyms = ilog2_exact(nsyms);
    m->scale_bits = scale_bits;
    m->freqs = (uint32_t *)malloc(sizeof(uint32_t) * nsyms);
    m->cum = (uint32_t *)malloc(sizeof(uint32_t) * (nsyms + 1));
    m->cum2sym = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)1 << scale_bits));
    if (!m->freqs || !m->cum || !m->cum2sym) {
        orc_model_destroy(m);
        return NULL;
    }
    uint64_t run = 0;
    for (uint32_t s = 0; s < nsyms; s++) {
        m->freqs[s] = norm_freqs[s];
        m->cum[s] = (uint32_t)run;
        run += norm_freqs[s];
    }
    m->cum[nsyms] = (uint32_t)run;
    if (run != ((uint64_t)1 << scale_bits)) {
        orc_model_destroy(m);
        return NULL;
    }
    /* main.cpp:143-148 (cum2sym) */
    for (uint32_t s = 0; s < nsyms; s++)
        for (uint32_t c = m->cum[s]; c < m->cum[s + 1]; c++)
            m->cum2sym[c] = s;
    if (with_alias && build_alias(m) != ORC_OK) {
        orc_model_destroy(m);
        return NULL;
    }
    return m;
}

void orc_model_destroy(orc_model *m)
{
    if (!m)
        return;
    free(m->freqs);
    free(m->cum);
    free(m->cum2sym);
    free(m->divider);
    free(m->slot_adjust);
    free(m->slot_freqs);
    free(m->sym_id);
    free(m->alias_remap);
    free(m);
}

/* --------------------------------------------------------------- formats */

#define BYTE_L (1u << 23)   /* rans_byte.h:50 */
#define WORD_L (1u << 16)   /* rans_word_sse41.h:35 */
#define WORD_SB 12u         /* rans_word_sse41.h:37 */
#define R64_L (1ull << 31)  /* rans64.h:59 */

static size_t state_bytes(int fmt)
{
    return fmt == ORC_FMT_R64 ? 8 : 4;
}

size_t orc_stream_bound(int fmt, size_t n, uint32_t n_ways)
{
    /* per symbol at most 2 bytes (byte/alias, scale_bits<=16), 2 (word), 4 (r64) */
    size_t per = fmt == ORC_FMT_R64 ? 4 : 2;
    return n * per + (size_t)n_ways * state_bytes(fmt) + 16;
}

static int check_args(int fmt, const orc_model *m, int sym_bytes, uint32_t n_ways)
{
    if (!m || n_ways == 0 || (sym_bytes != 1 && sym_bytes != 2))
        return ORC_E_ARG;
    if (sym_bytes == 1 && m->nsyms > 256)
        return ORC_E_ARG;
    switch (fmt) {
    case ORC_FMT_BYTE:
        return m->scale_bits <= 16 ? ORC_OK : ORC_E_ARG; /* rans_byte.h:176 */
    case ORC_FMT_ALIAS:
        return (m->scale_bits <= 16 && m->divider) ? ORC_OK : ORC_E_ARG;
    case ORC_FMT_WORD:
        return m->scale_bits == WORD_SB ? ORC_OK : ORC_E_ARG;
    case ORC_FMT_R64:
        return m->scale_bits <= 31 ? ORC_OK : ORC_E_ARG; /* rans64.h:169 */
    default:
        return ORC_E_ARG;
    }
}

/* ----------------------------------------------------------------- encode
 *
 * Layout rule shared by all formats (main.cpp:226-246, main_simd.cpp:287-300):
 * symbol i belongs to lane i mod N; symbols are visited last-to-first, each
 * lane first pushing renormalisation units *down* from the write cursor, then
 * applying C(s,x); finally lanes N-1 .. 0 push their states, so lane 0's state
 * is the first thing in the stream. */

int orc_encode(int fmt, const orc_model *m, const void *syms, size_t n, int sym_bytes,
               uint32_t n_ways, uint8_t *buf, size_t cap, size_t *out_len)
{
    int rc = check_args(fmt, m, sym_bytes, n_ways);
    if (rc)
        return rc;
    const uint32_t N = n_ways;
    const uint32_t sb = m->scale_bits;
    uint8_t *wp = buf + cap; /* write cursor, moves down */
    uint64_t *st = (uint64_t *)malloc(sizeof(uint64_t) * N);
    if (!st)
        return ORC_E_ARG;
    const uint64_t L0 = fmt == ORC_FMT_R64 ? R64_L : (fmt == ORC_FMT_WORD ? WORD_L : BYTE_L);
    for (uint32_t l = 0; l < N; l++)
        st[l] = L0;

    rc = ORC_OK;
    uint32_t lane = n ? (uint32_t)((n - 1) % N) : 0;
    for (size_t i = n; i-- > 0;) {
        uint32_t s = sym_at(syms, i, sym_bytes);
        if (s >= m->nsyms || m->freqs[s] == 0) {
            rc = ORC_E_ARG;
            break;
        }
        const uint32_t freq = m->freqs[s];
        const uint32_t start = m->cum[s];

        if (fmt == ORC_FMT_BYTE || fmt == ORC_FMT_ALIAS) {
            /* rans_byte.h:62-74 renorm, :83-90 put; alias put main_alias.cpp:241-250 */
            uint32_t x = (uint32_t)st[lane];
            uint32_t x_max = ((BYTE_L >> sb) << 8) * freq;
            while (x >= x_max) {
                if (wp - buf < 1) { rc = ORC_E_SPACE; goto done; }
                *--wp = (uint8_t)(x & 0xff);
                x >>= 8;
            }
            if (fmt == ORC_FMT_BYTE)
                x = ((x / freq) << sb) + (x % freq) + start;
            else
                x = ((x / freq) << sb) + m->alias_remap[(x % freq) + start];
            st[lane] = x;
        } else if (fmt == ORC_FMT_WORD) {
            /* rans_word_sse41.h:81-93; the threshold is a 32-bit product */
            uint32_t x = (uint32_t)st[lane];
            uint32_t x_max = ((WORD_L >> WORD_SB) << 16) * freq;
            if (x >= x_max) {
                if (wp - buf < 2) { rc = ORC_E_SPACE; goto done; }
                wp -= 2;
                uint16_t w = (uint16_t)(x & 0xffff);
                memcpy(wp, &w, 2);
                x >>= 16;
            }
            x = ((x / freq) << WORD_SB) + (x % freq) + start;
            st[lane] = x;
        } else {
            /* rans64.h:77-93 */
            uint64_t x = st[lane];
            uint64_t x_max = ((R64_L >> sb) << 32) * freq;
            if (x >= x_max) {
                if (wp - buf < 4) { rc = ORC_E_SPACE; goto done; }
                wp -= 4;
                uint32_t w = (uint32_t)x;
                memcpy(wp, &w, 4);
                x >>= 32;
            }
            x = ((x / freq) << sb) + (x % freq) + start;
            st[lane] = x;
        }
        lane = lane ? lane - 1 : N - 1;
    }

    /* flush: rans_byte.h:93-105, rans_word_sse41.h:96-106, rans64.h:96-103 */
    for (uint32_t l = N; l-- > 0;) {
        size_t sz = state_bytes(fmt);
        if ((size_t)(wp - buf) < sz) { rc = ORC_E_SPACE; goto done; }
        wp -= sz;
        if (fmt == ORC_FMT_R64) {
            uint64_t x = st[l]; /* lo32 then hi32, little endian */
            memcpy(wp, &x, 8);
        } else {
            uint32_t x = (uint32_t)st[l]; /* 4 LE bytes == lo16,hi16 */
            wp[0] = (uint8_t)(x >> 0);
            wp[1] = (uint8_t)(x >> 8);
            wp[2] = (uint8_t)(x >> 16);
            wp[3] = (uint8_t)(x >> 24);
        }
    }
done:
    free(st);
    if (rc == ORC_OK && out_len)
        *out_len = (size_t)(buf + cap - wp);
    return rc;
}

/* ----------------------------------------------------------------- decode
 *
 * Round structure (main.cpp:259-280, main_simd.cpp:313-332): lanes 0..N-1 load
 * their states in order; a round first lets every lane that still has a symbol
 * decode it and apply D (no stream access), then the same lanes, in ascending
 * order, pull the renormalisation units they need.  The byte formats can need
 * up to two units for a lane even in the last (partial) round, because the
 * encoder may have renormalised its *initial* state (L >= x_max when freq is
 * small): main.cpp:276-280 / main_alias.cpp:400-405 renormalise in the tail. */

int orc_decode(int fmt, const orc_model *m, const uint8_t *stream, size_t len, size_t n,
               int sym_bytes, uint32_t n_ways, void *out)
{
    int rc = check_args(fmt, m, sym_bytes, n_ways);
    if (rc)
        return rc;
    const uint32_t N = n_ways;
    const uint32_t sb = m->scale_bits;
    const uint32_t mask = (uint32_t)(((uint64_t)1 << sb) - 1);
    const uint8_t *rp = stream;
    const uint8_t *end = stream + len;
    const size_t ssz = state_bytes(fmt);
    if (len < (size_t)N * ssz)
        return ORC_E_CORRUPT;

    uint64_t *st = (uint64_t *)malloc(sizeof(uint64_t) * N);
    if (!st)
        return ORC_E_ARG;
    for (uint32_t l = 0; l < N; l++) { /* rans_byte.h:109-122, rans_word_sse41.h:109-120, rans64.h:107-115 */
        if (fmt == ORC_FMT_R64) {
            uint64_t x;
            memcpy(&x, rp, 8);
            st[l] = x;
        } else {
            st[l] = (uint32_t)rp[0] | ((uint32_t)rp[1] << 8) | ((uint32_t)rp[2] << 16) | ((uint32_t)rp[3] << 24);
        }
        rp += ssz;
    }

    rc = ORC_OK;
    for (size_t base = 0; base < n && rc == ORC_OK; base += N) {
        uint32_t cnt = (n - base < N) ? (uint32_t)(n - base) : N;
        /* phase 1: symbol lookup + D */
        for (uint32_t l = 0; l < cnt; l++) {
            uint32_t s;
            if (fmt == ORC_FMT_R64) {
                /* rans64.h:118-121 get, :286-292 step */
                uint64_t x = st[l];
                uint32_t cf = (uint32_t)x & mask;
                s = m->cum2sym[cf];
                st[l] = (uint64_t)m->freqs[s] * (x >> sb) + cf - m->cum[s];
            } else if (fmt == ORC_FMT_ALIAS) {
                /* main_alias.cpp:252-267; 32-bit wrap-around is intended */
                uint32_t x = (uint32_t)st[l];
                uint32_t xm = x & mask;
                uint32_t bucket = xm >> (sb - m->log2nsyms);
                uint32_t half = 2 * bucket + (xm < m->divider[bucket] ? 1u : 0u);
                s = m->sym_id[half];
                st[l] = (uint32_t)(m->slot_freqs[half] * (x >> sb) + xm - m->slot_adjust[half]);
            } else {
                /* rans_byte.h:125-128 get, :291-298 step; rans_word_sse41.h:123-131
                 * (slot table == cum2sym + {freq, slot - start}) */
                uint32_t x = (uint32_t)st[l];
                uint32_t cf = x & mask;
                s = m->cum2sym[cf];
                st[l] = (uint32_t)(m->freqs[s] * (x >> sb) + cf - m->cum[s]);
            }
            sym_put(out, base + l, sym_bytes, s);
        }
        /* phase 2: renormalise in lane order */
        for (uint32_t l = 0; l < cnt; l++) {
            if (fmt == ORC_FMT_R64) { /* rans64.h:305-316 */
                uint64_t x = st[l];
                if (x < R64_L) {
                    if (end - rp < 4) { rc = ORC_E_CORRUPT; break; }
                    uint32_t w;
                    memcpy(&w, rp, 4);
                    rp += 4;
                    x = (x << 32) | w;
                }
                st[l] = x;
            } else if (fmt == ORC_FMT_WORD) { /* rans_word_sse41.h:134-141 */
                uint32_t x = (uint32_t)st[l];
                if (x < WORD_L) {
                    if (end - rp < 2) { rc = ORC_E_CORRUPT; break; }
                    uint16_t w;
                    memcpy(&w, rp, 2);
                    rp += 2;
                    x = (x << 16) | w;
                }
                st[l] = x;
            } else { /* rans_byte.h:307-318 */
                uint32_t x = (uint32_t)st[l];
                while (x < BYTE_L) {
                    if (end - rp < 1) { rc = ORC_E_CORRUPT; break; }
                    x = (x << 8) | *rp++;
                }
                if (x < BYTE_L) { rc = ORC_E_CORRUPT; break; }
                st[l] = x;
            }
        }
    }

    if (rc == ORC_OK) {
        const uint64_t L0 = fmt == ORC_FMT_R64 ? R64_L : (fmt == ORC_FMT_WORD ? WORD_L : BYTE_L);
        for (uint32_t l = 0; l < N; l++)
            if (st[l] != L0)
                rc = ORC_E_CORRUPT;
        if (rp != end)
            rc = ORC_E_CORRUPT;
    }
    free(st);
    return rc;
}

/* --------------------------------------------------------------- chunked */

static size_t align_up(size_t v, size_t a)
{
    return a > 1 ? (v + a - 1) / a * a : v;
}

int orc_encode_chunked(int fmt, const orc_model *m, const void *syms, size_t n, int sym_bytes,
                       uint32_t n_ways, size_t chunk_syms, size_t align,
                       uint8_t *out, size_t cap, uint64_t *offsets, uint32_t *lengths,
                       size_t *out_total)
{
    if (chunk_syms == 0)
        return ORC_E_ARG;
    size_t nchunks = (n + chunk_syms - 1) / chunk_syms;
    size_t bound = orc_stream_bound(fmt, chunk_syms < n ? chunk_syms : n, n_ways);
    uint8_t *tmp = (uint8_t *)malloc(bound);
    if (!tmp)
        return ORC_E_ARG;
    size_t pos = 0;
    int rc = ORC_OK;
    for (size_t c = 0; c < nchunks; c++) {
        size_t first = c * chunk_syms;
        size_t cnt = n - first < chunk_syms ? n - first : chunk_syms;
        size_t len = 0;
        rc = orc_encode(fmt, m, (const uint8_t *)syms + first * sym_bytes, cnt, sym_bytes, n_ways, tmp, bound, &len);
        if (rc)
            break;
        pos = align_up(pos, align);
        if (pos + len > cap) { rc = ORC_E_SPACE; break; }
        memcpy(out + pos, tmp + bound - len, len);
        offsets[c] = pos;
        lengths[c] = (uint32_t)len;
        pos += len;
    }
    offsets[nchunks] = pos;
    if (out_total)
        *out_total = pos;
    free(tmp);
    return rc;
}

/* Compare the chunks [c0, c1) of somebody else's container with this oracle's own streams for the same symbols:
 * returns -1 when every chunk has the oracle's length and bytes, the index of the first chunk that differs
 * otherwise, -2 on a bad argument.  No shared state: ranges may be compared from several threads at once (that is
 * how the full-size GPU containers -- 32768 chunks of a 1 GiB shard -- are checked chunk by chunk in seconds). */
int64_t orc_compare_chunks(int fmt, const orc_model *m, const void *syms, size_t n, int sym_bytes, uint32_t n_ways,
                           size_t chunk_syms, uint64_t c0, uint64_t c1, const uint8_t *container,
                           const uint64_t *offsets, const uint32_t *lengths)
{
    if (chunk_syms == 0 || c0 > c1)
        return -2;
    size_t nchunks = (n + chunk_syms - 1) / chunk_syms;
    if (c1 > nchunks)
        return -2;
    size_t bound = orc_stream_bound(fmt, chunk_syms < n ? chunk_syms : n, n_ways);
    uint8_t *tmp = (uint8_t *)malloc(bound);
    if (!tmp)
        return -2;
    int64_t bad = -1;
    for (uint64_t c = c0; c < c1; c++) {
        size_t first = (size_t)c * chunk_syms;
        size_t cnt = n - first < chunk_syms ? n - first : chunk_syms;
        size_t len = 0;
        int rc = orc_encode(fmt, m, (const uint8_t *)syms + first * sym_bytes, cnt, sym_bytes, n_ways, tmp, bound, &len);
        if (rc || len != lengths[c] || memcmp(tmp + bound - len, container + offsets[c], len) != 0) {
            bad = (int64_t)c;
            break;
        }
    }
    free(tmp);
    return bad;
}

/* one chunk as its own input: main.cpp:139-162 (count_freqs, normalize_freqs, tables), then the N-way encoder loop */
static int adaptive_chunk(int fmt, const uint8_t *part, size_t cnt, uint32_t n_ways, uint32_t scale_bits, uint8_t *tmp,
                          size_t bound, size_t *len, uint32_t *freqs /* [256] */)
{
    uint32_t cum[257];
    orc_count_freqs(part, cnt, 1, 256, freqs);
    int rc = orc_normalize_freqs(freqs, cum, 256, 1u << scale_bits);
    if (rc)
        return rc;
    orc_model *m = orc_model_create(freqs, 256, scale_bits, 0);
    if (!m)
        return ORC_E_ARG;
    rc = orc_encode(fmt, m, part, cnt, 1, n_ways, tmp, bound, len);
    orc_model_destroy(m);
    return rc;
}

int64_t orc_compare_chunks_adaptive(int fmt, const uint8_t *syms, size_t n, uint32_t n_ways, size_t chunk_syms,
                                    uint32_t scale_bits, uint64_t c0, uint64_t c1, const uint8_t *container,
                                    const uint64_t *offsets, const uint32_t *lengths, const uint16_t *rows)
{
    if (chunk_syms == 0 || c0 > c1 || (fmt != ORC_FMT_BYTE && fmt != ORC_FMT_WORD))
        return -2;
    size_t nchunks = (n + chunk_syms - 1) / chunk_syms;
    if (c1 > nchunks)
        return -2;
    size_t bound = orc_stream_bound(fmt, chunk_syms < n ? chunk_syms : n, n_ways);
    uint8_t *tmp = (uint8_t *)malloc(bound);
    if (!tmp)
        return -2;
    int64_t bad = -1;
    for (uint64_t c = c0; c < c1 && bad == -1; c++) {
        size_t first = (size_t)c * chunk_syms;
        size_t cnt = n - first < chunk_syms ? n - first : chunk_syms;
        size_t len = 0;
        uint32_t freqs[256];
        int rc = adaptive_chunk(fmt, syms + first, cnt, n_ways, scale_bits, tmp, bound, &len, freqs);
        if (rc) {
            bad = -2;
            break;
        }
        for (int s = 0; s < 256; s++)
            if (rows[c * 256 + s] != freqs[s])
                bad = (int64_t)(2 * c);
        if (bad == -1 && (len != lengths[c] || memcmp(tmp + bound - len, container + offsets[c], len) != 0))
            bad = (int64_t)(2 * c + 1);
    }
    free(tmp);
    return bad;
}

int orc_encode_chunks_adaptive(int fmt, const uint8_t *syms, size_t n, uint32_t n_ways, size_t chunk_syms,
                               uint32_t scale_bits, uint64_t c0, uint64_t c1, uint8_t *out, size_t slot,
                               uint32_t *lengths, uint16_t *rows)
{
    if (chunk_syms == 0 || c0 > c1 || (fmt != ORC_FMT_BYTE && fmt != ORC_FMT_WORD))
        return ORC_E_ARG;
    size_t nchunks = (n + chunk_syms - 1) / chunk_syms;
    size_t bound = orc_stream_bound(fmt, chunk_syms < n ? chunk_syms : n, n_ways);
    if (c1 > nchunks || slot < bound)
        return ORC_E_ARG;
    for (uint64_t c = c0; c < c1; c++) {
        size_t first = (size_t)c * chunk_syms;
        size_t cnt = n - first < chunk_syms ? n - first : chunk_syms;
        size_t len = 0;
        uint32_t freqs[256];
        int rc = adaptive_chunk(fmt, syms + first, cnt, n_ways, scale_bits, out + (c - c0) * slot, slot, &len, freqs);
        if (rc)
            return rc;
        lengths[c - c0] = (uint32_t)len;
        for (int s = 0; s < 256; s++)
            rows[(c - c0) * 256 + s] = (uint16_t)freqs[s];
    }
    return ORC_OK;
}

int orc_decode_chunked(int fmt, const orc_model *m, const uint8_t *container,
                       const uint64_t *offsets, const uint32_t *lengths, size_t n, int sym_bytes,
                       uint32_t n_ways, size_t chunk_syms, void *out)
{
    if (chunk_syms == 0)
        return ORC_E_ARG;
    size_t nchunks = (n + chunk_syms - 1) / chunk_syms;
    for (size_t c = 0; c < nchunks; c++) {
        size_t first = c * chunk_syms;
        size_t cnt = n - first < chunk_syms ? n - first : chunk_syms;
        int rc = orc_decode(fmt, m, container + offsets[c], lengths[c], cnt, sym_bytes, n_ways,
                            (uint8_t *)out + first * sym_bytes);
        if (rc)
            return rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------- generator */

static inline uint64_t splitmix64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_gen_zipf(void *out, size_t n, int sym_bytes, uint32_t K, double s, uint64_t seed)
{
    double *cdf = (double *)malloc(sizeof(double) * K);
    double run = 0.0;
    for (uint32_t k = 0; k < K; k++) {
        run += 1.0 / pow((double)(k + 1), s);
        cdf[k] = run;
    }
    uint64_t state = seed;
    for (size_t i = 0; i < n; i++) {
        double u = (double)(splitmix64(&state) >> 11) * (1.0 / 9007199254740992.0) * run;
        uint32_t lo = 0, hi = K - 1; /* first k with cdf[k] > u */
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] > u)
                hi = mid;
            else
                lo = mid + 1;
        }
        sym_put(out, i, sym_bytes, lo);
    }
    free(cdf);
}
