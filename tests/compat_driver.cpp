// Test driver for include/ryg_rans_amd/compat/*: N-way encode/decode loops written against the
// per-symbol API (the shape of the reference drivers), exported with a C ABI so that
// tests/test_compat_headers.py can compare the streams with the oracle byte for byte.
#include <assert.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "rans_byte.h"
#include "rans64.h"
#include "rans_word_sse41.h"

extern "C" {

// fmt: 0 byte (RansEncPutSymbol), 1 word, 2 rans64 (Rans64EncPutSymbol), 4 byte via RansEncPut (division form)
int compat_encode(int fmt, const uint32_t *freqs, const uint32_t *cum, uint32_t scale_bits, const uint8_t *in, size_t n,
                  uint32_t N, uint8_t *buf, size_t cap, size_t *out_len)
{
    if (fmt == 0 || fmt == 4) {
        std::vector<RansEncSymbol> es(256);
        for (int s = 0; s < 256; s++)
            RansEncSymbolInit(&es[s], cum[s], freqs[s], scale_bits);
        std::vector<RansState> st(N);
        for (auto &x : st) RansEncInit(&x);
        uint8_t *p = buf + cap;
        for (size_t i = n; i-- > 0;) {
            if (fmt == 0) RansEncPutSymbol(&st[i % N], &p, &es[in[i]]);
            else RansEncPut(&st[i % N], &p, cum[in[i]], freqs[in[i]], scale_bits);
        }
        for (uint32_t l = N; l-- > 0;) RansEncFlush(&st[l], &p);
        *out_len = (size_t)(buf + cap - p);
    } else if (fmt == 1) {
        std::vector<RansWordEnc> st(N, RansWordEncInit());
        uint16_t *p = (uint16_t *)(buf + cap);
        for (size_t i = n; i-- > 0;) RansWordEncPut(&st[i % N], &p, cum[in[i]], freqs[in[i]]);
        for (uint32_t l = N; l-- > 0;) RansWordEncFlush(&st[l], &p);
        *out_len = (size_t)(buf + cap - (uint8_t *)p);
    } else if (fmt == 2) {
        std::vector<Rans64EncSymbol> es(256);
        for (int s = 0; s < 256; s++)
            Rans64EncSymbolInit(&es[s], cum[s], freqs[s], scale_bits);
        std::vector<Rans64State> st(N);
        for (auto &x : st) Rans64EncInit(&x);
        uint32_t *p = (uint32_t *)(buf + cap);
        for (size_t i = n; i-- > 0;) Rans64EncPutSymbol(&st[i % N], &p, &es[in[i]], scale_bits);
        for (uint32_t l = N; l-- > 0;) Rans64EncFlush(&st[l], &p);
        *out_len = (size_t)(buf + cap - (uint8_t *)p);
    } else {
        return 1;
    }
    return 0;
}

int compat_decode(int fmt, const uint32_t *freqs, const uint32_t *cum, uint32_t scale_bits, const uint8_t *stream,
                  size_t len, size_t n, uint32_t N, uint8_t *out)
{
    std::vector<uint8_t> c2s((size_t)1 << scale_bits);
    for (int s = 0; s < 256; s++)
        for (uint32_t c = cum[s]; c < cum[s + 1]; c++) c2s[c] = (uint8_t)s;
    if (fmt == 0) {
        std::vector<RansDecSymbol> ds(256);
        for (int s = 0; s < 256; s++) RansDecSymbolInit(&ds[s], cum[s], freqs[s]);
        std::vector<RansState> st(N);
        uint8_t *p = (uint8_t *)stream;
        for (auto &x : st) RansDecInit(&x, &p);
        for (size_t base = 0; base < n; base += N) {
            uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
            for (uint32_t l = 0; l < cnt; l++) {
                uint8_t s = c2s[RansDecGet(&st[l], scale_bits)];
                out[base + l] = s;
                RansDecAdvanceSymbolStep(&st[l], &ds[s], scale_bits);
            }
            for (uint32_t l = 0; l < cnt; l++) RansDecRenorm(&st[l], &p);
        }
        return p == stream + len ? 0 : 3;
    } else if (fmt == 1) {
        RansWordTables *tab = new RansWordTables;
        for (int s = 0; s < 256; s++) RansWordTablesInitSymbol(tab, (uint8_t)s, cum[s], freqs[s]);
        uint16_t *p = (uint16_t *)stream;
        int rc = 0;
        if (N == 8) { // the SSE4.1 path, loop shape of the reference's SIMD driver
            RansSimdDec r0, r1;
            RansSimdDecInit(&r0, &p);
            RansSimdDecInit(&r1, &p);
            size_t i = 0;
            for (; i + 8 <= n; i += 8) {
                uint32_t a = RansSimdDecSym(&r0, tab), b = RansSimdDecSym(&r1, tab);
                memcpy(out + i, &a, 4);
                memcpy(out + i + 4, &b, 4);
                RansSimdDecRenorm(&r0, &p);
                RansSimdDecRenorm(&r1, &p);
            }
            for (; i < n; i++) out[i] = RansWordDecSym(&((i & 4) ? r1 : r0).lane[i & 3], tab);
        } else {
            std::vector<RansWordDec> st(N);
            for (auto &x : st) RansWordDecInit(&x, &p);
            for (size_t base = 0; base < n; base += N) {
                uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
                for (uint32_t l = 0; l < cnt; l++) out[base + l] = RansWordDecSym(&st[l], tab);
                if (cnt == N)
                    for (uint32_t l = 0; l < cnt; l++) RansWordDecRenorm(&st[l], &p);
            }
        }
        rc = (uint8_t *)p == stream + len ? 0 : 3;
        delete tab;
        return rc;
    } else if (fmt == 2) {
        std::vector<Rans64DecSymbol> ds(256);
        for (int s = 0; s < 256; s++) Rans64DecSymbolInit(&ds[s], cum[s], freqs[s]);
        std::vector<Rans64State> st(N);
        uint32_t *p = (uint32_t *)stream;
        for (auto &x : st) Rans64DecInit(&x, &p);
        for (size_t base = 0; base < n; base += N) {
            uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
            for (uint32_t l = 0; l < cnt; l++) {
                uint8_t s = c2s[Rans64DecGet(&st[l], scale_bits)];
                out[base + l] = s;
                Rans64DecAdvanceSymbol(&st[l], &p, &ds[s], scale_bits);
            }
        }
        return (uint8_t *)p == stream + len ? 0 : 3;
    }
    return 1;
}

// struct layouts the reference promises: 16, 4, 24, 8, 4, 20480, 16
int compat_sizeof(int which)
{
    switch (which) {
    case 0: return (int)sizeof(RansEncSymbol);
    case 1: return (int)sizeof(RansDecSymbol);
    case 2: return (int)sizeof(Rans64EncSymbol);
    case 3: return (int)sizeof(Rans64DecSymbol);
    case 4: return (int)sizeof(RansWordSlot);
    case 5: return (int)sizeof(RansWordTables);
    case 6: return (int)sizeof(RansSimdDec);
    default: return -1;
    }
}
}

// ---- rans_alias_compat.h: the alias coder over the byte stream format, u8 or u16 symbols ----
#include "rans_alias_compat.h"

extern "C" {

static uint32_t alias_sym(const void *in, int sym_bytes, size_t i)
{
    return sym_bytes == 1 ? ((const uint8_t *)in)[i] : ((const uint16_t *)in)[i];
}

int compat_alias_encode(const uint32_t *freqs, const uint32_t *cum, uint32_t nsyms, uint32_t scale_bits, const void *in,
                        int sym_bytes, size_t n, uint32_t N, uint8_t *buf, size_t cap, size_t *out_len)
{
    RansAliasTable t;
    if (RansAliasTableInit(&t, freqs, cum, nsyms, scale_bits))
        return 2;
    std::vector<RansState> st(N);
    for (auto &x : st) RansEncInit(&x);
    uint8_t *p = buf + cap;
    for (size_t i = n; i-- > 0;) RansEncPutAlias(&st[i % N], &p, &t, alias_sym(in, sym_bytes, i), scale_bits);
    for (uint32_t l = N; l-- > 0;) RansEncFlush(&st[l], &p);
    *out_len = (size_t)(buf + cap - p);
    RansAliasTableFree(&t);
    return 0;
}

int compat_alias_decode(const uint32_t *freqs, const uint32_t *cum, uint32_t nsyms, uint32_t scale_bits, const uint8_t *stream,
                        size_t len, size_t n, uint32_t N, void *out, int sym_bytes)
{
    RansAliasTable t;
    if (RansAliasTableInit(&t, freqs, cum, nsyms, scale_bits))
        return 2;
    std::vector<RansState> st(N);
    uint8_t *p = (uint8_t *)stream;
    for (auto &x : st) RansDecInit(&x, &p);
    for (size_t base = 0; base < n; base += N) {
        uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
        for (uint32_t l = 0; l < cnt; l++) {
            uint32_t s = RansDecGetAlias(&st[l], &t, scale_bits);
            if (sym_bytes == 1) ((uint8_t *)out)[base + l] = (uint8_t)s;
            else ((uint16_t *)out)[base + l] = (uint16_t)s;
        }
        for (uint32_t l = 0; l < cnt; l++) RansDecRenorm(&st[l], &p);
    }
    RansAliasTableFree(&t);
    return p == stream + len ? 0 : 3;
}

// the tables, for comparison with the library's (rans_amd_model_table): divider[n], slot_adjust[2n], slot_freqs[2n],
// sym_id[2n], alias_remap[M] back to back
int compat_alias_tables(const uint32_t *freqs, const uint32_t *cum, uint32_t nsyms, uint32_t scale_bits, uint32_t *out)
{
    RansAliasTable t;
    if (RansAliasTableInit(&t, freqs, cum, nsyms, scale_bits))
        return 2;
    memcpy(out, t.divider, 4 * (size_t)nsyms); out += nsyms;
    memcpy(out, t.slot_adjust, 8 * (size_t)nsyms); out += 2 * nsyms;
    memcpy(out, t.slot_freqs, 8 * (size_t)nsyms); out += 2 * nsyms;
    memcpy(out, t.sym_id, 8 * (size_t)nsyms); out += 2 * nsyms;
    memcpy(out, t.alias_remap, 4u << scale_bits);
    RansAliasTableFree(&t);
    return 0;
}

}
