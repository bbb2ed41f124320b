// model.cpp -- host model builder (see model.h).  No HIP in this file.
#include "model.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "../../include/ryg_rans_amd.h"

namespace rans_amd {

namespace {

uint32_t ceil_log2(uint32_t v)
{
    uint32_t l = 0;
    while ((1ull << l) < v)
        ++l;
    return l;
}

template <typename T>
void append_bytes(std::vector<uint8_t> &out, const T *p, size_t count)
{
    const uint8_t *b = reinterpret_cast<const uint8_t *>(p);
    out.insert(out.end(), b, b + count * sizeof(T));
}

} // namespace

// count_freqs: main.cpp:59-66.  Symbols outside the alphabet are an error here
// (the reference indexes freqs[] unchecked).
int count_freqs_host(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *freqs)
{
    if (!freqs || nsyms == 0 || (n && !syms) || (sym_bytes != 1 && sym_bytes != 2))
        return RANS_AMD_E_ARG;
    std::fill(freqs, freqs + nsyms, 0u);
    // four partial histograms break the store-to-load dependency on runs
    std::vector<uint32_t> part(4 * (size_t)nsyms, 0u);
    bool bad = false;
    if (sym_bytes == 1) {
        const uint8_t *p = static_cast<const uint8_t *>(syms);
        if (nsyms >= 256) {
            uint64_t i = 0;
            for (; i + 4 <= n; i += 4) {
                part[0 * nsyms + p[i + 0]]++;
                part[1 * nsyms + p[i + 1]]++;
                part[2 * nsyms + p[i + 2]]++;
                part[3 * nsyms + p[i + 3]]++;
            }
            for (; i < n; ++i)
                part[p[i]]++;
        } else {
            for (uint64_t i = 0; i < n; ++i) {
                if (p[i] >= nsyms) { bad = true; break; }
                part[p[i]]++;
            }
        }
    } else {
        const uint16_t *p = static_cast<const uint16_t *>(syms);
        for (uint64_t i = 0; i < n; ++i) {
            if (p[i] >= nsyms) { bad = true; break; }
            part[(i & 3) * nsyms + p[i]]++;
        }
    }
    if (bad)
        return RANS_AMD_E_ARG;
    for (uint32_t s = 0; s < nsyms; ++s)
        freqs[s] = part[s] + part[nsyms + s] + part[2 * (size_t)nsyms + s] + part[3 * (size_t)nsyms + s];
    return RANS_AMD_OK;
}

// normalize_freqs: main.cpp:75-129.  The reference shifts a run of cumulative
// boundaries for every repaired symbol; that changes exactly two WIDTHS (the
// repaired symbol +1, the victim -1), so the repair is done on widths here and
// the cumulative table is rebuilt once at the end.  Same visiting order, same
// victim rule (narrowest width > 1, lowest index on ties), hence same result.
int normalize_freqs(uint32_t *freqs, uint32_t *cum, uint32_t nsyms, uint32_t target_total)
{
    if (!freqs || !cum || nsyms == 0)
        return RANS_AMD_E_ARG;
    if (target_total < nsyms)
        return RANS_AMD_E_MODEL;
    uint64_t total = 0;
    for (uint32_t s = 0; s < nsyms; ++s)
        total += freqs[s];
    if (total == 0 || total > 0xffffffffull)
        return RANS_AMD_E_MODEL;

    std::vector<uint32_t> width(nsyms);
    uint64_t run = 0;
    uint32_t prev_edge = 0;
    for (uint32_t s = 0; s < nsyms; ++s) {
        run += freqs[s];
        uint32_t edge = (uint32_t)(((uint64_t)target_total * (uint32_t)run) / (uint32_t)total);
        width[s] = edge - prev_edge;
        prev_edge = edge;
    }

    for (uint32_t s = 0; s < nsyms; ++s) {
        if (freqs[s] == 0 || width[s] != 0)
            continue;
        uint32_t victim = nsyms, narrowest = 0xffffffffu;
        for (uint32_t j = 0; j < nsyms; ++j)
            if (width[j] > 1 && width[j] < narrowest) {
                narrowest = width[j];
                victim = j;
            }
        if (victim == nsyms)
            return RANS_AMD_E_MODEL;
        width[victim] -= 1;
        width[s] = 1;
    }

    cum[0] = 0;
    for (uint32_t s = 0; s < nsyms; ++s) {
        if ((freqs[s] == 0) != (width[s] == 0))
            return RANS_AMD_E_MODEL;
        freqs[s] = width[s];
        cum[s + 1] = cum[s] + width[s];
    }
    return cum[nsyms] == target_total ? RANS_AMD_OK : RANS_AMD_E_MODEL;
}

int HostModel::build(int fmt, const uint32_t *norm_freqs, uint32_t ns, uint32_t sb)
{
    if (!norm_freqs || ns == 0 || ns > 65536)
        return RANS_AMD_E_ARG;
    switch (fmt) {
    case RANS_AMD_FMT_BYTE:
    case RANS_AMD_FMT_ALIAS:
        if (sb == 0 || sb > 16) // rans_byte.h:176
            return RANS_AMD_E_UNSUPPORTED;
        break;
    case RANS_AMD_FMT_WORD:
        // rans_word_sse41.h:37 fixes 12 bits; :41 fixes 256 symbols -- the stream format does not depend on the
        // alphabet, so up to 4096 symbols (one slot each at least) are taken, with u16 symbols beyond 256
        if (sb != 12 || ns > 4096)
            return RANS_AMD_E_UNSUPPORTED;
        break;
    case RANS_AMD_FMT_R64:
        if (sb == 0 || sb > 31) // rans64.h:169
            return RANS_AMD_E_UNSUPPORTED;
        break;
    default:
        return RANS_AMD_E_ARG;
    }
    format = fmt;
    // rans64 outside 7..16 bits: no cum2sym table (2^scale_bits entries fit no LDS beyond 16 bits, and below 7
    // the table decoder's 24-bit partial products do not hold): the kernels search the cumulative frequencies
    r64_search = fmt == RANS_AMD_FMT_R64 && (sb > 16 || sb < 7);
    nsyms = ns;
    log2nsyms = ceil_log2(ns);
    scale_bits = sb;
    sym_bytes = ns <= 256 ? 1 : 2;
    const uint32_t M = 1u << sb;

    freqs.assign(norm_freqs, norm_freqs + ns);
    cum.assign(ns + 1, 0);
    uint64_t run = 0;
    for (uint32_t s = 0; s < ns; ++s) {
        if (freqs[s] > M) // garbage
            return RANS_AMD_E_MODEL;
        // A one-symbol model (freq == M) is inside the working range of the byte, alias and rans64 coders
        // (rans_byte.h:176-178, rans64.h:169-171: x_max = 2^31 resp. 2^63 is never reached, C and D are the
        // identity, the stream is the N flushed states).  In the word format the 32-bit renormalisation
        // threshold wraps to 0 for freq == 4096 (rans_word_sse41.h:85): rejected.
        if (freqs[s] == M && fmt == RANS_AMD_FMT_WORD)
            return RANS_AMD_E_MODEL;
        run += freqs[s];
        if (run > M)
            return RANS_AMD_E_MODEL;
        cum[s + 1] = (uint32_t)run;
    }
    if (run != M)
        return RANS_AMD_E_MODEL;

    // main.cpp:143-148 (not for the wide rans64 models: up to 2^31 entries nobody reads)
    cum2sym.clear();
    if (sb <= 16) {
        cum2sym.assign(M, 0);
        for (uint32_t s = 0; s < ns; ++s)
            std::fill(cum2sym.begin() + cum[s], cum2sym.begin() + cum[s + 1], (uint16_t)s);
    }
    cum_padded.clear();
    if (r64_search) {
        // cum[0..ns] followed by ~0 up to a power of two: "last index with cum <= cf" by halving steps
        uint32_t p2 = 2;
        while (p2 < ns + 1)
            p2 <<= 1;
        cum_padded.assign(p2, 0xffffffffu);
        std::copy(cum.begin(), cum.end(), cum_padded.begin());
        // entries ns+1.. stay ~0; cum[ns] = M is > every cf as well, so symbol ns is never chosen
    }

    r64_packed.clear();
    if (fmt == RANS_AMD_FMT_R64 && !r64_search && ns <= 256 && sb <= 14) {
        bool small = true;
        for (uint32_t s = 0; s < ns; ++s)
            small = small && freqs[s] <= 4095u;
        if (small) {
            r64_packed.resize(M);
            for (uint32_t slot = 0; slot < M; ++slot) {
                const uint32_t s = cum2sym[slot];
                r64_packed[slot] = freqs[s] | ((slot - cum[s]) << 12) | (s << 24);
            }
        }
    }

    byte_slots.clear();
    if (fmt == RANS_AMD_FMT_BYTE && ns <= 256 && sb <= 13 && sb >= 8) {
        bool fits = true; // (a frequency of M = 2^sb <= 8192 fits the record's 24 bits; kept as a guard)
        for (uint32_t s = 0; s < ns; ++s)
            fits = fits && freqs[s] < (1u << 24);
        if (fits) {
            byte_slots.resize(M);
            for (uint32_t slot = 0; slot < M; ++slot) {
                const uint32_t s = cum2sym[slot];
                byte_slots[slot] = WordSlot{freqs[s] | (s << 24), slot - cum[s]};
            }
        }
    }

    // per-symbol records
    sym_recs.resize(ns);
    enc_recs.resize(ns);
    for (uint32_t s = 0; s < ns; ++s) {
        sym_recs[s] = SymRec{freqs[s], cum[s]};
        uint32_t f = freqs[s];
        if (fmt == RANS_AMD_FMT_R64) {
            // 64-bit Alverson reciprocal (exact quotients for x < 2^63, rans64.h:167-247):
            // q = mulhi64(x, rcp) >> rshift.  Record layout: {freq | rshift << 24, bias, rcp_lo, rcp_hi}.
            // freq == 1: rcp = 2^64 - 1 gives q = x - 1 and bias = start + M - 1 makes
            // x + bias + q * (M - 1) come out as x * M + start (rans64.h:216-221).
            uint64_t rcp64 = ~0ull;
            uint32_t rshift = 0;
            uint32_t bias = cum[s] + M - 1;
            if (f >= 2) {
                uint32_t sh = ceil_log2(f);
                unsigned __int128 num = ((unsigned __int128)1 << (sh + 63)) + (f - 1);
                rcp64 = (uint64_t)(num / f);
                rshift = sh - 1;
                bias = cum[s];
            }
            // (search variant: freq needs all 32 bits, the kernel recomputes rshift from it)
            enc_recs[s] = EncRec{r64_search ? f : (f | (rshift << 24)), bias, (uint32_t)rcp64, (uint32_t)(rcp64 >> 32)};
            continue;
        }
        if (fmt == RANS_AMD_FMT_BYTE) {
            // 32-bit Alverson reciprocal of RansEncSymbolInit (rans_byte.h:201-243), exact for x < 2^31:
            // {freq | rshift << 24, bias, rcp, -}; freq < 2: rcp = ~0, bias = start + M - 1.
            uint32_t rcp = 0xffffffffu, rshift = 0, bias = cum[s] + M - 1;
            if (f >= 2) {
                const uint32_t sh = ceil_log2(f);
                rcp = (uint32_t)(((1ull << (sh + 31)) + f - 1) / f);
                rshift = sh - 1;
                bias = cum[s];
            }
            enc_recs[s] = EncRec{f | (rshift << 24), bias, rcp, 0u};
            continue;
        }
        if (fmt == RANS_AMD_FMT_WORD) {
            // round-up reciprocal for 32-bit dividends (see WordEncRec in model.h; same numbers):
            // {freq, bias, m', cmpl | sh << 24}
            if (f <= 1) {
                enc_recs[s] = EncRec{f, cum[s] + M - 1, 0xffffffffu, M - 1};
            } else {
                const uint32_t l = ceil_log2(f);
                const uint64_t mprime = (((uint64_t)1 << 32) * (((uint64_t)1 << l) - f)) / f + 1;
                enc_recs[s] = EncRec{f, cum[s], (uint32_t)mprime, (M - f) | ((l - 1) << 24)};
            }
            continue;
        }
        // alias: remainder needed for the remap lookup -> floor(2^32 / freq) and one correction step
        uint32_t rcp = f <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / f);
        enc_recs[s] = EncRec{f, cum[s], rcp, cum[s]};
    }

    dense256 = ns == 256;
    for (uint32_t s = 0; s < ns && dense256; ++s)
        dense256 = freqs[s] != 0;

    if (fmt == RANS_AMD_FMT_WORD) {
        // rans_word_sse41.h:64-72, packed for the device
        word_slots.resize(M);
        for (uint32_t slot = 0; slot < M; ++slot) {
            uint32_t s = cum2sym[slot];
            // byte symbols: {freq | sym << 24, bias}; u16 symbols: {freq, bias | sym << 16}
            word_slots[slot] = ns <= 256 ? WordSlot{freqs[s] | (s << 24), slot - cum[s]}
                                         : WordSlot{freqs[s], (slot - cum[s]) | (s << 16)};
        }
        word_enc_recs.clear(); // (the full-wave encoder path and its 256 records are for byte symbols)
        word_small = true;
        for (uint32_t s = 0; s < ns; ++s)
            word_small = word_small && freqs[s] <= 2048u;
        if (ns <= 256)
            word_enc_recs.assign(256, WordEncRec{0u, 0xffffffffu, 0x80000000u, 0u});
        auto rec16 = [M](uint32_t mprime, uint32_t cmpl, uint32_t bias, uint32_t sh) { // (freq = M - cmpl; thresh = (freq << 20) - 1)
            return WordEncRec{mprime, (uint32_t)(((uint64_t)(M - cmpl) << 20) - 1u), cmpl | (sh << 24), bias};
        };
        for (uint32_t s = 0; s < ns && ns <= 256; ++s) {
            const uint32_t f = freqs[s];
            if (f == 0)
                continue;
            if (f == 1) { // q = x - 1 either way: mulhi(x, 2^32 - 1) = x - 1, no shift; bias = start + M - 1 (13 bits)
                word_enc_recs[s] = rec16(0xffffffffu, M - 1, cum[s] + M - 1, 0);
                continue;
            }
            const uint32_t l = ceil_log2(f);
            if (word_small) { // x < 2^31 after renormalisation: Alverson (rans_byte.h:201-243), q = mulhi(x, rcp) >> (l - 1)
                const uint32_t rcp = (uint32_t)((((uint64_t)1 << (l + 31)) + f - 1) / f);
                word_enc_recs[s] = rec16(rcp, M - f, cum[s], l - 1);
            } else {
                const uint64_t mprime = (((uint64_t)1 << 32) * (((uint64_t)1 << l) - f)) / f + 1;
                word_enc_recs[s] = rec16((uint32_t)mprime, M - f, cum[s], l - 1);
            }
        }
    }
    if (fmt == RANS_AMD_FMT_ALIAS) {
        int rc = build_alias();
        if (rc)
            return rc;
    }
    return RANS_AMD_OK;
}

void adapt_rcp_tables(std::vector<uint32_t> &out)
{
    constexpr uint32_t kEntries = 4097;
    out.assign(2 * kEntries, 0u);
    out[1] = out[kEntries + 1] = 0xffffffffu; // freq 1: q = mulhi(x, 2^32 - 1) = x - 1, the bias makes up for it
    for (uint32_t f = 2; f < kEntries; ++f) {
        const uint32_t l = ceil_log2(f);
        out[f] = (uint32_t)((((uint64_t)1 << (l + 31)) + f - 1) / f);
        out[kEntries + f] = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - f)) / f + 1);
    }
}

// make_alias_table: main_alias.cpp:147-237, for any power-of-two alphabet.
//
// Phase A decides, for every bucket b (bucket b nominally belongs to symbol b
// and holds tgt = M / nsyms slots), how many of its slots symbol b keeps
// (`keep[b]`) and which donor symbol fills the rest.  The sweep keeps a cursor
// on the first symbol that still has >= tgt slots to give ("donor") and one on
// the first symbol with < tgt ("needy"); topping up a needy bucket may turn the
// donor itself needy, in which case the sweep steps back to it if it has
// already been passed.  Phase B hands the slots out in bucket order.
int HostModel::build_alias()
{
    const uint32_t ns = nsyms;
    const uint32_t M = 1u << scale_bits;
    if ((ns & (ns - 1)) != 0 || M < ns || M % ns != 0) // main_alias.cpp:151-152
        return RANS_AMD_E_UNSUPPORTED;
    const uint32_t tgt = M / ns;

    std::vector<uint32_t> remaining(freqs);
    std::vector<uint32_t> keep(ns, tgt);
    std::vector<uint32_t> donor_of(ns);
    for (uint32_t b = 0; b < ns; ++b)
        donor_of[b] = b;

    auto skip_to_donor = [&](uint32_t from) {
        while (from < ns && remaining[from] < tgt)
            ++from;
        return from;
    };
    auto skip_to_needy = [&](uint32_t from) {
        while (from < ns && remaining[from] >= tgt)
            ++from;
        return from;
    };

    uint32_t donor = skip_to_donor(0);
    uint32_t needy = skip_to_needy(0);
    uint32_t resume = needy + 1; // where the forward scan for needy buckets continues
    while (donor < ns && needy < ns) {
        donor_of[needy] = donor;
        keep[needy] = remaining[needy];
        remaining[donor] -= tgt - keep[needy];

        if (remaining[donor] >= tgt || resume <= donor) {
            needy = skip_to_needy(resume);
            resume = needy + 1;
        } else {
            needy = donor; // donor became needy and is behind the scan position
        }
        donor = skip_to_donor(donor);
    }

    divider.assign(ns, 0);
    slot_adjust.assign(2 * (size_t)ns, 0);
    slot_freqs.assign(2 * (size_t)ns, 0);
    sym_id.assign(2 * (size_t)ns, 0);
    alias_remap.assign(M, 0);
    std::vector<uint32_t> handed(ns, 0);

    for (uint32_t b = 0; b < ns; ++b) {
        const uint32_t d = donor_of[b];
        const uint32_t own_n = keep[b];
        const uint32_t donor_n = tgt - own_n;
        const uint32_t bucket0 = b * tgt;
        const uint32_t own_first = handed[b];   // index of symbol b's first slot placed here
        const uint32_t donor_first = handed[d]; // (for d == b this is before own slots are added)

        divider[b] = bucket0 + own_n;
        // half 1 = the bucket's own symbol (taken when xm < divider), half 0 = the donor
        sym_id[2 * b + 1] = b;
        sym_id[2 * b + 0] = d;
        slot_freqs[2 * b + 1] = freqs[b];
        slot_freqs[2 * b + 0] = freqs[d];
        slot_adjust[2 * b + 1] = bucket0 - own_first;
        slot_adjust[2 * b + 0] = bucket0 - (donor_first - own_n);

        for (uint32_t k = 0; k < own_n; ++k)
            alias_remap[cum[b] + own_first + k] = bucket0 + k;
        for (uint32_t k = 0; k < donor_n; ++k)
            alias_remap[cum[d] + donor_first + k] = bucket0 + own_n + k;

        handed[b] += own_n;
        handed[d] += donor_n;
    }
    for (uint32_t s = 0; s < ns; ++s)
        if (handed[s] != freqs[s]) // main_alias.cpp:235-236
            return RANS_AMD_E_MODEL;

    alias_halves.resize(2 * (size_t)ns);
    for (size_t h = 0; h < alias_halves.size(); ++h)
        alias_halves[h] = AliasHalf{slot_freqs[h] | (sym_id[h] << 16), slot_adjust[h]};
    // the encoder's LDS tables (kernels.h kKernelFormatAliasLds), where they fit and every frequency and start
    // fits 16 bits (a 65536-wide symbol does not: that model is host-only anyway)
    alias_recs8.clear();
    alias_remap16.clear();
    const size_t nrecs = ns < 256 ? 256 : ns;
    // decoder tables of the two-chunks-per-wave kernel
    alias2_halves.clear();
    alias2_own.clear();
    alias2_wide = tgt > 255;
    bool fits16 = ns >= 2 && tgt <= 0xffffu;
    for (uint32_t f : slot_freqs)
        fits16 = fits16 && f <= 0xffffu;
    if (fits16) {
        alias2_halves.resize(2 * (size_t)ns);
        for (size_t h = 0; h < alias2_halves.size(); ++h) // (M - 0 = 65536 for a half no state can select: any value)
            alias2_halves[h] = AliasHalf{sym_id[h] | (((M - slot_freqs[h]) & 0xffffu) << 16), slot_adjust[h]};
        alias2_own.assign(alias2_wide ? 2 * (size_t)ns : (size_t)ns, 0);
        for (uint32_t b = 0; b < ns; ++b) {
            const uint32_t own = divider[b] - b * tgt;
            if (alias2_wide) {
                alias2_own[2 * (size_t)b] = (uint8_t)own;
                alias2_own[2 * (size_t)b + 1] = (uint8_t)(own >> 8);
            } else {
                alias2_own[b] = (uint8_t)own;
            }
        }
    }
    // (only symbols the encoder can meet constrain the record width: an unused symbol at the end of a 16-bit
    //  alphabet has cum == 65536 and a record nobody reads)
    bool narrow = true;
    for (uint32_t s = 0; s < ns; ++s)
        narrow = narrow && (freqs[s] == 0 || (freqs[s] <= 0xffffu && cum[s] <= 0xffffu));
    if (narrow && nrecs * 8 + ((size_t)2 << scale_bits) <= 160 * 1024) {
        alias_recs8.assign(nrecs, 0);
        for (uint32_t s = 0; s < ns; ++s) {
            const uint32_t f = freqs[s];
            const uint32_t rcp = f <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / f);
            if (f == 0)
                continue; // zero record
            alias_recs8[s] = (uint64_t)(f | (cum[s] << 16)) | ((uint64_t)rcp << 32);
        }
        alias_remap16.resize(alias_remap.size());
        for (size_t i = 0; i < alias_remap.size(); ++i)
            alias_remap16[i] = (uint16_t)alias_remap[i];
    }
    return RANS_AMD_OK;
}

int HostModel::export_table(int which, std::vector<uint8_t> &out) const
{
    out.clear();
    const uint32_t M = 1u << scale_bits;
    const bool alias = format == RANS_AMD_FMT_ALIAS;
    switch (which) {
    case RANS_AMD_TAB_FREQS:
        append_bytes(out, freqs.data(), freqs.size());
        return RANS_AMD_OK;
    case RANS_AMD_TAB_CUM_FREQS:
        append_bytes(out, cum.data(), cum.size());
        return RANS_AMD_OK;
    case RANS_AMD_TAB_CUM2SYM:
        if (cum2sym.empty())
            return RANS_AMD_E_ARG; // wide rans64 model: there is no such table
        if (sym_bytes == 1) {
            out.resize(M);
            for (uint32_t i = 0; i < M; ++i)
                out[i] = (uint8_t)cum2sym[i];
        } else {
            append_bytes(out, cum2sym.data(), cum2sym.size());
        }
        return RANS_AMD_OK;
    case RANS_AMD_TAB_WORD_SLOTS: {
        if (format != RANS_AMD_FMT_WORD)
            return RANS_AMD_E_ARG;
        // RansWordTables memory image: slots[4096] {u16 freq, u16 bias}, then slot2sym[4096]
        std::vector<uint16_t> fb(2 * (size_t)M);
        std::vector<uint8_t> s2s(M);
        for (uint32_t slot = 0; slot < M; ++slot) {
            fb[2 * slot] = (uint16_t)(word_slots[slot].lo & 0xffffu);
            fb[2 * slot + 1] = (uint16_t)word_slots[slot].hi;
            s2s[slot] = (uint8_t)(word_slots[slot].lo >> 24);
        }
        append_bytes(out, fb.data(), fb.size());
        append_bytes(out, s2s.data(), s2s.size());
        return RANS_AMD_OK;
    }
    case RANS_AMD_TAB_ALIAS_DIVIDER:
        if (!alias) return RANS_AMD_E_ARG;
        append_bytes(out, divider.data(), divider.size());
        return RANS_AMD_OK;
    case RANS_AMD_TAB_ALIAS_SLOT_ADJUST:
        if (!alias) return RANS_AMD_E_ARG;
        append_bytes(out, slot_adjust.data(), slot_adjust.size());
        return RANS_AMD_OK;
    case RANS_AMD_TAB_ALIAS_SLOT_FREQS:
        if (!alias) return RANS_AMD_E_ARG;
        append_bytes(out, slot_freqs.data(), slot_freqs.size());
        return RANS_AMD_OK;
    case RANS_AMD_TAB_ALIAS_SYM_ID:
        if (!alias) return RANS_AMD_E_ARG;
        if (sym_bytes == 1) {
            out.resize(sym_id.size());
            for (size_t i = 0; i < sym_id.size(); ++i)
                out[i] = (uint8_t)sym_id[i];
        } else {
            std::vector<uint16_t> t(sym_id.begin(), sym_id.end());
            append_bytes(out, t.data(), t.size());
        }
        return RANS_AMD_OK;
    case RANS_AMD_TAB_ALIAS_REMAP:
        if (!alias) return RANS_AMD_E_ARG;
        append_bytes(out, alias_remap.data(), alias_remap.size());
        return RANS_AMD_OK;
    case RANS_AMD_TAB_ENC_SYMBOLS: {
        // Values per RansEncSymbolInit (rans_byte.h:174-243) / Rans64EncSymbolInit
        // (rans64.h:167-247): exact-division reciprocals after Alverson.
        if (format == RANS_AMD_FMT_R64) {
            std::vector<EncSymbol64> t(nsyms);
            for (uint32_t s = 0; s < nsyms; ++s) {
                uint32_t f = freqs[s];
                EncSymbol64 e;
                e.freq = f;
                e.cmpl_freq = M - f;
                if (f < 2) {
                    e.rcp_freq = ~0ull;
                    e.rcp_shift = 0;
                    e.bias = cum[s] + M - 1;
                } else {
                    uint32_t sh = ceil_log2(f);
                    unsigned __int128 num = ((unsigned __int128)1 << (sh + 63)) + (f - 1);
                    e.rcp_freq = (uint64_t)(num / f);
                    e.rcp_shift = sh - 1;
                    e.bias = cum[s];
                }
                t[s] = e;
            }
            append_bytes(out, t.data(), t.size());
        } else {
            std::vector<EncSymbol32> t(nsyms);
            for (uint32_t s = 0; s < nsyms; ++s) {
                uint32_t f = freqs[s];
                EncSymbol32 e;
                e.x_max = ((0x800000u >> scale_bits) << 8) * f;
                e.cmpl_freq = (uint16_t)(M - f);
                if (f < 2) {
                    e.rcp_freq = ~0u;
                    e.rcp_shift = 0;
                    e.bias = cum[s] + M - 1;
                } else {
                    uint32_t sh = ceil_log2(f);
                    e.rcp_freq = (uint32_t)(((1ull << (sh + 31)) + f - 1) / f);
                    e.rcp_shift = (uint16_t)(sh - 1);
                    e.bias = cum[s];
                }
                t[s] = e;
            }
            append_bytes(out, t.data(), t.size());
        }
        return RANS_AMD_OK;
    }
    case RANS_AMD_TAB_DEC_SYMBOLS:
        if (format == RANS_AMD_FMT_R64) {
            std::vector<DecSymbol64> t(nsyms);
            for (uint32_t s = 0; s < nsyms; ++s)
                t[s] = DecSymbol64{cum[s], freqs[s]};
            append_bytes(out, t.data(), t.size());
        } else {
            std::vector<DecSymbol32> t(nsyms);
            for (uint32_t s = 0; s < nsyms; ++s)
                t[s] = DecSymbol32{(uint16_t)cum[s], (uint16_t)freqs[s]};
            append_bytes(out, t.data(), t.size());
        }
        return RANS_AMD_OK;
    default:
        return RANS_AMD_E_ARG;
    }
}

} // namespace rans_amd
