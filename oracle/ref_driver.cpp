// ref_driver.cpp -- builds oracle/_ref/libryg_ref.so from the UNMODIFIED reference
// sources where they lie (-I/root/reference).  TEST INFRASTRUCTURE ONLY.
//
// Nothing from /root/reference is copied: this translation unit #includes
// rans_byte.h / rans64.h / rans_word_sse41.h and main_alias.cpp (for the
// SymbolStats model builder, RansEncPutAlias and RansDecGetAlias, which only
// exist inside the reference's mains) and wraps them in a C ABI.  The N-way
// loops below are the reference driver loops (main.cpp:226-280,
// main_simd.cpp:287-332, main_alias.cpp:353-405) with the lane count as a
// run-time value; for N = 1, 2, 8 they issue exactly the same sequence of
// reference calls as the mains do.
//
// Used to (a) validate oracle/rans_oracle.c, (b) generate tests/golden fixtures,
// (c) time the reference CPU path (bench.py cpu_baseline.kind == "reference").

#include <assert.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define main ryg_ref_alias_main_unused
#include "main_alias.cpp" // SymbolStats (256 symbols), RansEncPutAlias, RansDecGetAlias, rans_byte.h, platform.h
#undef main
#include "rans64.h"
#include "rans_word_sse41.h"

enum { FMT_BYTE = 0, FMT_WORD = 1, FMT_R64 = 2, FMT_ALIAS = 3 };

namespace {

// Fill a SymbolStats from already-normalised frequencies.
void stats_from_freqs(SymbolStats &st, const uint32_t *freqs)
{
    for (int i = 0; i < 256; i++)
        st.freqs[i] = freqs[i];
    st.calc_cum_freqs();
}

} // namespace

extern "C" {

// count_freqs + normalize_freqs exactly as every main does (main.cpp:140-141).
int ref_build_model_u8(const uint8_t *in, size_t n, uint32_t target_total, uint32_t *freqs, uint32_t *cum)
{
    SymbolStats st;
    st.count_freqs(in, n);
    st.normalize_freqs(target_total);
    memcpy(freqs, st.freqs, sizeof(st.freqs));
    memcpy(cum, st.cum_freqs, sizeof(st.cum_freqs));
    return 0;
}

// normalize_freqs on caller-provided raw counts.
int ref_normalize_u8(uint32_t *freqs, uint32_t *cum, uint32_t target_total)
{
    SymbolStats st;
    memcpy(st.freqs, freqs, sizeof(st.freqs));
    st.normalize_freqs(target_total);
    memcpy(freqs, st.freqs, sizeof(st.freqs));
    memcpy(cum, st.cum_freqs, sizeof(st.cum_freqs));
    return 0;
}

// make_alias_table (main_alias.cpp:147-237) on normalised freqs.
int ref_alias_tables_u8(const uint32_t *freqs, uint32_t *divider, uint32_t *slot_adjust, uint32_t *slot_freqs,
                        uint8_t *sym_id, uint32_t *alias_remap)
{
    SymbolStats st;
    stats_from_freqs(st, freqs);
    st.make_alias_table();
    memcpy(divider, st.divider, sizeof(st.divider));
    memcpy(slot_adjust, st.slot_adjust, sizeof(st.slot_adjust));
    memcpy(slot_freqs, st.slot_freqs, sizeof(st.slot_freqs));
    memcpy(sym_id, st.sym_id, sizeof(st.sym_id));
    memcpy(alias_remap, st.alias_remap, sizeof(uint32_t) * st.cum_freqs[256]);
    return 0;
}

// RansWordTables image (slots then slot2sym), main_simd.cpp:141-143.
int ref_word_tables_u8(const uint32_t *freqs, uint8_t *image /* 20480 bytes */)
{
    SymbolStats st;
    stats_from_freqs(st, freqs);
    RansWordTables *tab = new RansWordTables;
    memset(tab, 0, sizeof(*tab));
    for (int s = 0; s < 256; s++)
        RansWordTablesInitSymbol(tab, (uint8_t)s, st.cum_freqs[s], st.freqs[s]);
    memcpy(image, tab, sizeof(*tab));
    delete tab;
    return 0;
}

// N-way encode with the reference primitives.  Stream = buf[cap-*out_len .. cap).
int ref_encode_u8(int fmt, const uint32_t *freqs, uint32_t scale_bits, const uint8_t *in, size_t n, uint32_t n_ways,
                  uint8_t *buf, size_t cap, size_t *out_len)
{
    SymbolStats st;
    stats_from_freqs(st, freqs);
    const uint32_t N = n_ways;

    if (fmt == FMT_BYTE) {
        RansEncSymbol esyms[256];
        for (int i = 0; i < 256; i++)
            RansEncSymbolInit(&esyms[i], st.cum_freqs[i], st.freqs[i], scale_bits);
        std::vector<RansState> rans(N);
        for (uint32_t l = 0; l < N; l++)
            RansEncInit(&rans[l]);
        uint8_t *ptr = buf + cap;
        for (size_t i = n; i > 0; i--)
            RansEncPutSymbol(&rans[(i - 1) % N], &ptr, &esyms[in[i - 1]]);
        for (uint32_t l = N; l > 0; l--)
            RansEncFlush(&rans[l - 1], &ptr);
        *out_len = (size_t)(buf + cap - ptr);
    } else if (fmt == FMT_ALIAS) {
        st.make_alias_table();
        std::vector<RansState> rans(N);
        for (uint32_t l = 0; l < N; l++)
            RansEncInit(&rans[l]);
        uint8_t *ptr = buf + cap;
        for (size_t i = n; i > 0; i--)
            RansEncPutAlias(&rans[(i - 1) % N], &ptr, &st, in[i - 1], scale_bits);
        for (uint32_t l = N; l > 0; l--)
            RansEncFlush(&rans[l - 1], &ptr);
        *out_len = (size_t)(buf + cap - ptr);
    } else if (fmt == FMT_WORD) {
        if (scale_bits != RANS_WORD_SCALE_BITS || (cap & 1))
            return 1;
        std::vector<RansWordEnc> rans(N);
        for (uint32_t l = 0; l < N; l++)
            rans[l] = RansWordEncInit();
        uint16_t *ptr = (uint16_t *)(buf + cap);
        for (size_t i = n; i > 0; i--) {
            int s = in[i - 1];
            RansWordEncPut(&rans[(i - 1) % N], &ptr, st.cum_freqs[s], st.freqs[s]);
        }
        for (uint32_t l = N; l > 0; l--)
            RansWordEncFlush(&rans[l - 1], &ptr);
        *out_len = (size_t)(buf + cap - (uint8_t *)ptr);
    } else if (fmt == FMT_R64) {
        if (cap & 3)
            return 1;
        Rans64EncSymbol esyms[256];
        for (int i = 0; i < 256; i++)
            Rans64EncSymbolInit(&esyms[i], st.cum_freqs[i], st.freqs[i], scale_bits);
        std::vector<Rans64State> rans(N);
        for (uint32_t l = 0; l < N; l++)
            Rans64EncInit(&rans[l]);
        uint32_t *ptr = (uint32_t *)(buf + cap);
        for (size_t i = n; i > 0; i--)
            Rans64EncPutSymbol(&rans[(i - 1) % N], &ptr, &esyms[in[i - 1]], scale_bits);
        for (uint32_t l = N; l > 0; l--)
            Rans64EncFlush(&rans[l - 1], &ptr);
        *out_len = (size_t)(buf + cap - (uint8_t *)ptr);
    } else {
        return 1;
    }
    return 0;
}

// N-way decode with the reference primitives (scalar).  The stream must be
// readable for `len` bytes; returns 0 when the cursor ends on stream+len.
int ref_decode_u8(int fmt, const uint32_t *freqs, uint32_t scale_bits, const uint8_t *stream, size_t len, size_t n,
                  uint32_t n_ways, uint8_t *out)
{
    SymbolStats st;
    stats_from_freqs(st, freqs);
    const uint32_t N = n_ways;
    const uint8_t *endp = stream + len;

    if (fmt == FMT_BYTE || fmt == FMT_ALIAS) {
        std::vector<uint8_t> cum2sym((size_t)1 << scale_bits);
        RansDecSymbol dsyms[256];
        for (int s = 0; s < 256; s++) {
            RansDecSymbolInit(&dsyms[s], st.cum_freqs[s], st.freqs[s]);
            for (uint32_t i = st.cum_freqs[s]; i < st.cum_freqs[s + 1]; i++)
                cum2sym[i] = (uint8_t)s;
        }
        if (fmt == FMT_ALIAS)
            st.make_alias_table();
        std::vector<RansState> rans(N);
        uint8_t *ptr = (uint8_t *)stream;
        for (uint32_t l = 0; l < N; l++)
            RansDecInit(&rans[l], &ptr);
        for (size_t base = 0; base < n; base += N) {
            uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
            for (uint32_t l = 0; l < cnt; l++) {
                uint32_t s;
                if (fmt == FMT_BYTE) {
                    s = cum2sym[RansDecGet(&rans[l], scale_bits)];
                    RansDecAdvanceSymbolStep(&rans[l], &dsyms[s], scale_bits);
                } else {
                    s = RansDecGetAlias(&rans[l], &st, scale_bits);
                }
                out[base + l] = (uint8_t)s;
            }
            for (uint32_t l = 0; l < cnt; l++)
                RansDecRenorm(&rans[l], &ptr);
        }
        return ptr == endp ? 0 : 3;
    } else if (fmt == FMT_WORD) {
        RansWordTables *tab = new RansWordTables;
        for (int s = 0; s < 256; s++)
            RansWordTablesInitSymbol(tab, (uint8_t)s, st.cum_freqs[s], st.freqs[s]);
        std::vector<RansWordDec> rans(N);
        uint16_t *ptr = (uint16_t *)stream;
        for (uint32_t l = 0; l < N; l++)
            RansWordDecInit(&rans[l], &ptr);
        for (size_t base = 0; base < n; base += N) {
            uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
            for (uint32_t l = 0; l < cnt; l++)
                out[base + l] = RansWordDecSym(&rans[l], tab);
            if (cnt == N) // main_simd.cpp:328-332: the tail never renormalises
                for (uint32_t l = 0; l < cnt; l++)
                    RansWordDecRenorm(&rans[l], &ptr);
        }
        delete tab;
        return (uint8_t *)ptr == endp ? 0 : 3;
    } else if (fmt == FMT_R64) {
        std::vector<uint8_t> cum2sym((size_t)1 << scale_bits);
        Rans64DecSymbol dsyms[256];
        for (int s = 0; s < 256; s++) {
            Rans64DecSymbolInit(&dsyms[s], st.cum_freqs[s], st.freqs[s]);
            for (uint32_t i = st.cum_freqs[s]; i < st.cum_freqs[s + 1]; i++)
                cum2sym[i] = (uint8_t)s;
        }
        std::vector<Rans64State> rans(N);
        uint32_t *ptr = (uint32_t *)stream;
        for (uint32_t l = 0; l < N; l++)
            Rans64DecInit(&rans[l], &ptr);
        for (size_t base = 0; base < n; base += N) {
            uint32_t cnt = n - base < N ? (uint32_t)(n - base) : N;
            for (uint32_t l = 0; l < cnt; l++) {
                uint32_t s = cum2sym[Rans64DecGet(&rans[l], scale_bits)];
                out[base + l] = (uint8_t)s;
                Rans64DecAdvanceSymbolStep(&rans[l], &dsyms[s], scale_bits);
            }
            for (uint32_t l = 0; l < cnt; l++)
                Rans64DecRenorm(&rans[l], &ptr);
        }
        return (uint8_t *)ptr == endp ? 0 : 3;
    }
    return 1;
}

// The reference's fastest shipped decoder: 8-way word format through the SSE4.1
// routines, loop shape of main_simd.cpp:313-332.  `stream` must have 16 bytes of
// readable padding after its end (RansSimdDecRenorm over-reads,
// rans_word_sse41.h:218-220).
int ref_decode_word_simd8(const uint32_t *freqs, const uint8_t *stream, size_t n, uint8_t *out)
{
    SymbolStats st;
    stats_from_freqs(st, freqs);
    RansWordTables *tab = new RansWordTables;
    for (int s = 0; s < 256; s++)
        RansWordTablesInitSymbol(tab, (uint8_t)s, st.cum_freqs[s], st.freqs[s]);

    RansSimdDec rans0, rans1;
    uint16_t *ptr = (uint16_t *)stream;
    RansSimdDecInit(&rans0, &ptr);
    RansSimdDecInit(&rans1, &ptr);
    for (size_t i = 0; i < (n & ~(size_t)7); i += 8) {
        uint32_t s03 = RansSimdDecSym(&rans0, tab);
        uint32_t s47 = RansSimdDecSym(&rans1, tab);
        memcpy(out + i, &s03, 4);
        memcpy(out + i + 4, &s47, 4);
        RansSimdDecRenorm(&rans0, &ptr);
        RansSimdDecRenorm(&rans1, &ptr);
    }
    for (size_t i = (n & ~(size_t)7); i < n; i++) {
        RansSimdDec *which = (i & 4) != 0 ? &rans1 : &rans0;
        out[i] = RansWordDecSym(&which->lane[i & 3], tab);
    }
    delete tab;
    return 0;
}

// ---- timing helpers for the CPU baseline -----------------------------------
//
// `shards` independent 8-way word streams (stream s at streams + offsets[s],
// n_per symbols each) are decoded `reps` times by `threads` pthreads, shards dealt
// round robin.  Returns wall seconds for the whole batch (clock_gettime MONOTONIC via
// the reference's own timer(), platform.h:47-55).

struct simd8_job {
    const uint32_t *freqs;
    const uint8_t *streams;
    const uint64_t *offsets;
    size_t n_per;
    uint8_t *out;
    uint32_t shards, threads, tid, reps;
};

static void *simd8_worker(void *p)
{
    simd8_job *j = (simd8_job *)p;
    for (uint32_t rep = 0; rep < j->reps; rep++)
        for (uint32_t s = j->tid; s < j->shards; s += j->threads)
            ref_decode_word_simd8(j->freqs, j->streams + j->offsets[s], j->n_per, j->out + (size_t)s * j->n_per);
    return 0;
}

double ref_time_word_simd8(const uint32_t *freqs, const uint8_t *streams, const uint64_t *offsets, uint32_t shards,
                           size_t n_per, uint8_t *out, uint32_t threads, uint32_t reps)
{
    std::vector<pthread_t> th(threads);
    std::vector<simd8_job> jobs(threads);
    double t0 = timer();
    for (uint32_t t = 0; t < threads; t++) {
        jobs[t] = simd8_job{freqs, streams, offsets, n_per, out, shards, threads, t, reps};
        pthread_create(&th[t], 0, simd8_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; t++)
        pthread_join(th[t], 0);
    return timer() - t0;
}

// rdtsc for clocks/symbol, as main.cpp:171.
uint64_t ref_rdtsc(void) { return __rdtsc(); }

} // extern "C"
