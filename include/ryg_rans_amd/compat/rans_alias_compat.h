/*
 * rans_alias_compat.h -- host-side alias-method symbol lookup on top of the byte-aligned coder
 * (include/ryg_rans_amd/compat/rans_byte.h).  rygorous/ryg_rans keeps this part inside a sample program
 * (main_alias.cpp:47-267: SymbolStats::make_alias_table, RansEncPutAlias, RansDecGetAlias, 256 symbols fixed at
 * compile time); here it is a header, for any power-of-two alphabet up to 65536 symbols, written from the format
 * definition so that host code can make and read the streams the bulk GPU ABI
 * (include/ryg_rans_amd.h, RANS_AMD_FMT_ALIAS) makes and reads.  The tables it builds are the ones the library builds
 * (rans_amd_model_table, RANS_AMD_TAB_ALIAS_*) -- tests/test_compat_headers.py compares them entry by entry.
 *
 * The idea: the M = 1 << scale_bits slots of the coding interval are cut into nsyms buckets of M / nsyms slots; bucket b
 * holds slots of at most two symbols, b itself ("own", below the bucket's divider) and one donor.  The decoder finds a
 * slot's symbol with one shift, one compare and one small table read -- O(nsyms) table bytes instead of O(M) -- and the
 * encoder pays with a remap table of M entries that scatters a symbol's slots over the buckets that took them.
 */
#ifndef RYG_RANS_AMD_COMPAT_RANS_ALIAS_H
#define RYG_RANS_AMD_COMPAT_RANS_ALIAS_H

#include <stdint.h>
#include <stdlib.h>

#include "rans_byte.h"

typedef struct {
    uint32_t scale_bits;
    uint32_t log2_nsyms;
    uint32_t nsyms;
    const uint32_t *freqs;     /* caller's, nsyms entries, sum == 1 << scale_bits */
    const uint32_t *cum_freqs; /* caller's, nsyms + 1 entries */
    uint32_t *divider;         /* [nsyms]     first slot of bucket b that belongs to the donor */
    uint32_t *slot_adjust;     /* [2 * nsyms] half 2b + 1 = own symbol, half 2b = donor */
    uint32_t *slot_freqs;      /* [2 * nsyms] */
    uint32_t *sym_id;          /* [2 * nsyms] */
    uint32_t *alias_remap;     /* [1 << scale_bits] encoder side: slot (start + k) of a symbol -> where bucket filling put it */
} RansAliasTable;

static inline void RansAliasTableFree(RansAliasTable *t)
{
    free(t->divider);
    free(t->slot_adjust);
    free(t->slot_freqs);
    free(t->sym_id);
    free(t->alias_remap);
    t->divider = t->slot_adjust = t->slot_freqs = t->sym_id = t->alias_remap = 0;
}

/* Builds the tables.  Returns 0, or -1 when nsyms is not a power of two that divides 1 << scale_bits (scale_bits <= 16:
 * the byte coder's limit), when the frequencies do not sum to 1 << scale_bits, or when memory runs out.  freqs and
 * cum_freqs must stay valid while the table is in use. */
static inline int RansAliasTableInit(RansAliasTable *t, const uint32_t *freqs, const uint32_t *cum_freqs, uint32_t nsyms,
                                     uint32_t scale_bits)
{
    uint32_t M, per_bucket, b, donor, needy, resume, log2n = 0;
    uint32_t *left, *keep, *from, *given;

    t->divider = t->slot_adjust = t->slot_freqs = t->sym_id = t->alias_remap = 0;
    if (scale_bits == 0 || scale_bits > 16 || nsyms == 0 || (nsyms & (nsyms - 1)) != 0)
        return -1;
    M = 1u << scale_bits;
    if (nsyms > M || cum_freqs[nsyms] != M)
        return -1;
    while ((1u << log2n) < nsyms)
        ++log2n;
    per_bucket = M / nsyms;

    t->scale_bits = scale_bits;
    t->log2_nsyms = log2n;
    t->nsyms = nsyms;
    t->freqs = freqs;
    t->cum_freqs = cum_freqs;
    t->divider = (uint32_t *)calloc(nsyms, sizeof(uint32_t));
    t->slot_adjust = (uint32_t *)calloc(2 * (size_t)nsyms, sizeof(uint32_t));
    t->slot_freqs = (uint32_t *)calloc(2 * (size_t)nsyms, sizeof(uint32_t));
    t->sym_id = (uint32_t *)calloc(2 * (size_t)nsyms, sizeof(uint32_t));
    t->alias_remap = (uint32_t *)calloc(M, sizeof(uint32_t));
    left = (uint32_t *)malloc(4 * (size_t)nsyms * sizeof(uint32_t));
    if (!t->divider || !t->slot_adjust || !t->slot_freqs || !t->sym_id || !t->alias_remap || !left) {
        free(left);
        RansAliasTableFree(t);
        return -1;
    }
    keep = left + nsyms;
    from = keep + nsyms;
    given = from + nsyms;

    /* Who fills whom.  `left[s]` = slots of symbol s not yet placed in a bucket.  A bucket whose own symbol has fewer
     * than per_bucket slots left is "needy": it keeps what its symbol has and takes the rest from the first symbol that
     * can still give a whole bucket's worth ("donor").  Giving may turn the donor itself needy; if the sweep has
     * already passed it, it is served next, otherwise the sweep reaches it in its turn. */
    for (b = 0; b < nsyms; ++b) {
        left[b] = freqs[b];
        keep[b] = per_bucket;
        from[b] = b;
        given[b] = 0;
    }
    donor = 0;
    while (donor < nsyms && left[donor] < per_bucket)
        ++donor;
    needy = 0;
    while (needy < nsyms && left[needy] >= per_bucket)
        ++needy;
    resume = needy + 1;
    while (donor < nsyms && needy < nsyms) {
        from[needy] = donor;
        keep[needy] = left[needy];
        left[donor] -= per_bucket - keep[needy];
        if (left[donor] >= per_bucket || resume <= donor) {
            needy = resume;
            while (needy < nsyms && left[needy] >= per_bucket)
                ++needy;
            resume = needy + 1;
        } else {
            needy = donor;
        }
        while (donor < nsyms && left[donor] < per_bucket)
            ++donor;
    }

    /* Hand the slots out bucket by bucket, own symbol first. */
    for (b = 0; b < nsyms; ++b) {
        const uint32_t d = from[b], own_n = keep[b], donor_n = per_bucket - own_n, first = b * per_bucket;
        const uint32_t own_at = given[b], donor_at = given[d];
        uint32_t k;
        t->divider[b] = first + own_n;
        t->sym_id[2 * b + 1] = b;
        t->sym_id[2 * b] = d;
        t->slot_freqs[2 * b + 1] = freqs[b];
        t->slot_freqs[2 * b] = freqs[d];
        t->slot_adjust[2 * b + 1] = first - own_at;             /* (both differences wrap on purpose) */
        t->slot_adjust[2 * b] = first - (donor_at - own_n);
        for (k = 0; k < own_n; ++k)
            t->alias_remap[cum_freqs[b] + own_at + k] = first + k;
        for (k = 0; k < donor_n; ++k)
            t->alias_remap[cum_freqs[d] + donor_at + k] = first + own_n + k;
        given[b] += own_n;
        given[d] += donor_n;
    }
    for (b = 0; b < nsyms; ++b)
        if (given[b] != freqs[b]) {
            free(left);
            RansAliasTableFree(t);
            return -1;
        }
    free(left);
    return 0;
}

/* Encodes symbol s (frequency > 0): the byte coder's renormalisation, then the usual update with the symbol's slot sent
 * through alias_remap.  Symbols go in reverse order, the stream grows downwards (rans_byte.h). */
static inline void RansEncPutAlias(RansState *r, uint8_t **pptr, const RansAliasTable *t, uint32_t s, uint32_t scale_bits)
{
    const uint32_t freq = t->freqs[s];
    RansState x = rans_compat_byte_shift_out(*r, pptr, rans_compat_byte_xmax(freq, scale_bits));
    *r = ((x / freq) << scale_bits) + t->alias_remap[x % freq + t->cum_freqs[s]];
}

/* Decodes one symbol and advances the state; the caller renormalises afterwards (RansDecRenorm), as with
 * RansDecAdvanceSymbolStep. */
static inline uint32_t RansDecGetAlias(RansState *r, const RansAliasTable *t, uint32_t scale_bits)
{
    const RansState x = *r;
    const uint32_t slot = x & ((1u << scale_bits) - 1u);
    const uint32_t bucket = slot >> (scale_bits - t->log2_nsyms);
    const uint32_t half = 2u * bucket + (slot < t->divider[bucket] ? 1u : 0u);
    *r = t->slot_freqs[half] * (x >> scale_bits) + slot - t->slot_adjust[half]; /* (32-bit wrap-around intended) */
    return t->sym_id[half];
}

#endif /* RYG_RANS_AMD_COMPAT_RANS_ALIAS_H */
