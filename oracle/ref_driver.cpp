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
#include <sched.h>
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

// ---- the reference's own 2-way loops, timed the way its mains time them ------------------------------------------
//
// One shard through "interleaved rANS encode" + "interleaved rANS decode" of the format's main: main.cpp:226-246 /
// 259-280 (rans_byte.h), main64.cpp:228-248 / 261-282 (rans64.h), main_alias.cpp:353-373 / 386-405 (alias lookup) --
// two named states, the odd symbol first, the same sequence of reference calls, each pass bracketed by timer() and
// __rdtsc() as main.cpp:222-223,248-249 does.  Tables are built before the clock starts, as in the mains.  With a
// barrier (multi-thread runs) every thread starts each pass at the same moment.

struct ref_loop2_times {
    double enc_s, dec_s;
    uint64_t enc_clocks, dec_clocks, stream_bytes;
    int32_t ok, pad;
};

static void loop2_sync(void *bar)
{
    if (bar)
        pthread_barrier_wait((pthread_barrier_t *)bar);
}

int ref_loop2_u8(int fmt, const uint32_t *freqs, uint32_t prob_bits, const uint8_t *in_bytes, size_t in_size, uint8_t *out_buf,
                 size_t out_max_size, uint8_t *dec_bytes, ref_loop2_times *res, void *bar)
{
    SymbolStats *stp = new SymbolStats;
    SymbolStats &stats = *stp;
    stats_from_freqs(stats, freqs);
    memset(res, 0, sizeof *res);
    memset(dec_bytes, 0xcc, in_size);
    if (fmt == FMT_BYTE) {
        std::vector<uint8_t> cum2sym_v((size_t)1 << prob_bits);
        uint8_t *cum2sym = cum2sym_v.data();
        RansEncSymbol esyms[256];
        RansDecSymbol dsyms[256];
        for (int s = 0; s < 256; s++) {
            RansEncSymbolInit(&esyms[s], stats.cum_freqs[s], stats.freqs[s], prob_bits);
            RansDecSymbolInit(&dsyms[s], stats.cum_freqs[s], stats.freqs[s]);
            for (uint32_t i = stats.cum_freqs[s]; i < stats.cum_freqs[s + 1]; i++)
                cum2sym[i] = (uint8_t)s;
        }
        uint8_t *rans_begin;
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t enc_start_time = __rdtsc();
            RansState rans0, rans1;
            RansEncInit(&rans0);
            RansEncInit(&rans1);
            uint8_t *ptr = out_buf + out_max_size;
            if (in_size & 1) {
                int s = in_bytes[in_size - 1];
                RansEncPutSymbol(&rans0, &ptr, &esyms[s]);
            }
            for (size_t i = (in_size & ~(size_t)1); i > 0; i -= 2) {
                int s1 = in_bytes[i - 1];
                int s0 = in_bytes[i - 2];
                RansEncPutSymbol(&rans1, &ptr, &esyms[s1]);
                RansEncPutSymbol(&rans0, &ptr, &esyms[s0]);
            }
            RansEncFlush(&rans1, &ptr);
            RansEncFlush(&rans0, &ptr);
            rans_begin = ptr;
            res->enc_clocks = __rdtsc() - enc_start_time;
            res->enc_s = timer() - start_time;
        }
        res->stream_bytes = (uint64_t)(out_buf + out_max_size - rans_begin);
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t dec_start_time = __rdtsc();
            RansState rans0, rans1;
            uint8_t *ptr = rans_begin;
            RansDecInit(&rans0, &ptr);
            RansDecInit(&rans1, &ptr);
            for (size_t i = 0; i < (in_size & ~(size_t)1); i += 2) {
                uint32_t s0 = cum2sym[RansDecGet(&rans0, prob_bits)];
                uint32_t s1 = cum2sym[RansDecGet(&rans1, prob_bits)];
                dec_bytes[i + 0] = (uint8_t)s0;
                dec_bytes[i + 1] = (uint8_t)s1;
                RansDecAdvanceSymbolStep(&rans0, &dsyms[s0], prob_bits);
                RansDecAdvanceSymbolStep(&rans1, &dsyms[s1], prob_bits);
                RansDecRenorm(&rans0, &ptr);
                RansDecRenorm(&rans1, &ptr);
            }
            if (in_size & 1) {
                uint32_t s0 = cum2sym[RansDecGet(&rans0, prob_bits)];
                dec_bytes[in_size - 1] = (uint8_t)s0;
                RansDecAdvanceSymbol(&rans0, &ptr, &dsyms[s0], prob_bits);
            }
            res->dec_clocks = __rdtsc() - dec_start_time;
            res->dec_s = timer() - start_time;
        }
    } else if (fmt == FMT_R64) {
        std::vector<uint8_t> cum2sym_v((size_t)1 << prob_bits);
        uint8_t *cum2sym = cum2sym_v.data();
        Rans64EncSymbol esyms[256];
        Rans64DecSymbol dsyms[256];
        for (int s = 0; s < 256; s++) {
            Rans64EncSymbolInit(&esyms[s], stats.cum_freqs[s], stats.freqs[s], prob_bits);
            Rans64DecSymbolInit(&dsyms[s], stats.cum_freqs[s], stats.freqs[s]);
            for (uint32_t i = stats.cum_freqs[s]; i < stats.cum_freqs[s + 1]; i++)
                cum2sym[i] = (uint8_t)s;
        }
        uint32_t *out_end = (uint32_t *)(out_buf + (out_max_size & ~(size_t)3));
        uint32_t *rans_begin;
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t enc_start_time = __rdtsc();
            Rans64State rans0, rans1;
            Rans64EncInit(&rans0);
            Rans64EncInit(&rans1);
            uint32_t *ptr = out_end;
            if (in_size & 1) {
                int s = in_bytes[in_size - 1];
                Rans64EncPutSymbol(&rans0, &ptr, &esyms[s], prob_bits);
            }
            for (size_t i = (in_size & ~(size_t)1); i > 0; i -= 2) {
                int s1 = in_bytes[i - 1];
                int s0 = in_bytes[i - 2];
                Rans64EncPutSymbol(&rans1, &ptr, &esyms[s1], prob_bits);
                Rans64EncPutSymbol(&rans0, &ptr, &esyms[s0], prob_bits);
            }
            Rans64EncFlush(&rans1, &ptr);
            Rans64EncFlush(&rans0, &ptr);
            rans_begin = ptr;
            res->enc_clocks = __rdtsc() - enc_start_time;
            res->enc_s = timer() - start_time;
        }
        res->stream_bytes = (uint64_t)((out_end - rans_begin) * sizeof(uint32_t));
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t dec_start_time = __rdtsc();
            Rans64State rans0, rans1;
            uint32_t *ptr = rans_begin;
            Rans64DecInit(&rans0, &ptr);
            Rans64DecInit(&rans1, &ptr);
            for (size_t i = 0; i < (in_size & ~(size_t)1); i += 2) {
                uint32_t s0 = cum2sym[Rans64DecGet(&rans0, prob_bits)];
                uint32_t s1 = cum2sym[Rans64DecGet(&rans1, prob_bits)];
                dec_bytes[i + 0] = (uint8_t)s0;
                dec_bytes[i + 1] = (uint8_t)s1;
                Rans64DecAdvanceSymbolStep(&rans0, &dsyms[s0], prob_bits);
                Rans64DecAdvanceSymbolStep(&rans1, &dsyms[s1], prob_bits);
                Rans64DecRenorm(&rans0, &ptr);
                Rans64DecRenorm(&rans1, &ptr);
            }
            if (in_size & 1) {
                uint32_t s0 = cum2sym[Rans64DecGet(&rans0, prob_bits)];
                dec_bytes[in_size - 1] = (uint8_t)s0;
                Rans64DecAdvanceSymbol(&rans0, &ptr, &dsyms[s0], prob_bits);
            }
            res->dec_clocks = __rdtsc() - dec_start_time;
            res->dec_s = timer() - start_time;
        }
    } else if (fmt == FMT_ALIAS) {
        stats.make_alias_table();
        uint8_t *rans_begin;
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t enc_start_time = __rdtsc();
            RansState rans0, rans1;
            RansEncInit(&rans0);
            RansEncInit(&rans1);
            uint8_t *ptr = out_buf + out_max_size;
            if (in_size & 1) {
                int s = in_bytes[in_size - 1];
                RansEncPutAlias(&rans0, &ptr, &stats, s, prob_bits);
            }
            for (size_t i = (in_size & ~(size_t)1); i > 0; i -= 2) {
                int s1 = in_bytes[i - 1];
                int s0 = in_bytes[i - 2];
                RansEncPutAlias(&rans1, &ptr, &stats, s1, prob_bits);
                RansEncPutAlias(&rans0, &ptr, &stats, s0, prob_bits);
            }
            RansEncFlush(&rans1, &ptr);
            RansEncFlush(&rans0, &ptr);
            rans_begin = ptr;
            res->enc_clocks = __rdtsc() - enc_start_time;
            res->enc_s = timer() - start_time;
        }
        res->stream_bytes = (uint64_t)(out_buf + out_max_size - rans_begin);
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t dec_start_time = __rdtsc();
            RansState rans0, rans1;
            uint8_t *ptr = rans_begin;
            RansDecInit(&rans0, &ptr);
            RansDecInit(&rans1, &ptr);
            for (size_t i = 0; i < (in_size & ~(size_t)1); i += 2) {
                uint32_t s0 = RansDecGetAlias(&rans0, &stats, prob_bits);
                uint32_t s1 = RansDecGetAlias(&rans1, &stats, prob_bits);
                dec_bytes[i + 0] = (uint8_t)s0;
                dec_bytes[i + 1] = (uint8_t)s1;
                RansDecRenorm(&rans0, &ptr);
                RansDecRenorm(&rans1, &ptr);
            }
            if (in_size & 1) {
                uint32_t s0 = RansDecGetAlias(&rans0, &stats, prob_bits);
                dec_bytes[in_size - 1] = (uint8_t)s0;
                RansDecRenorm(&rans0, &ptr);
            }
            res->dec_clocks = __rdtsc() - dec_start_time;
            res->dec_s = timer() - start_time;
        }
    } else if (fmt == FMT_WORD) {
        // the word format's loop is 8-way: scalar encode main_simd.cpp:287-300, SSE4.1 decode main_simd.cpp:313-332
        // (the stream buffer must leave 16 readable bytes behind its end: RansSimdDecRenorm over-reads,
        //  rans_word_sse41.h:218-220 -- out_max_size is cut short by them here, as main_simd.cpp:146 pads)
        if (prob_bits != RANS_WORD_SCALE_BITS || out_max_size < 64) {
            delete stp;
            return 1;
        }
        RansWordTables *tabp = new RansWordTables;
        RansWordTables &tab = *tabp;
        for (int s = 0; s < 256; s++)
            RansWordTablesInitSymbol(&tab, (uint8_t)s, stats.cum_freqs[s], stats.freqs[s]);
        const size_t out_size = (out_max_size - 16) & ~(size_t)1;
        uint16_t *rans_begin;
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t enc_start_time = __rdtsc();
            RansWordEnc rans[8];
            for (int i = 0; i < 8; i++)
                rans[i] = RansWordEncInit();
            uint16_t *ptr = (uint16_t *)(out_buf + out_size);
            for (size_t i = in_size; i > 0; i--) {
                int s = in_bytes[i - 1];
                RansWordEncPut(&rans[(i - 1) & 7], &ptr, stats.cum_freqs[s], stats.freqs[s]);
            }
            for (int i = 8; i > 0; i--)
                RansWordEncFlush(&rans[i - 1], &ptr);
            rans_begin = ptr;
            res->enc_clocks = __rdtsc() - enc_start_time;
            res->enc_s = timer() - start_time;
        }
        res->stream_bytes = (uint64_t)(out_buf + out_size - (uint8_t *)rans_begin);
        loop2_sync(bar);
        {
            double start_time = timer();
            uint64_t dec_start_time = __rdtsc();
            RansSimdDec rans0, rans1;
            uint16_t *ptr = rans_begin;
            RansSimdDecInit(&rans0, &ptr);
            RansSimdDecInit(&rans1, &ptr);
            for (size_t i = 0; i < (in_size & ~(size_t)7); i += 8) {
                uint32_t s03 = RansSimdDecSym(&rans0, &tab);
                uint32_t s47 = RansSimdDecSym(&rans1, &tab);
                memcpy(dec_bytes + i, &s03, 4);
                memcpy(dec_bytes + i + 4, &s47, 4);
                RansSimdDecRenorm(&rans0, &ptr);
                RansSimdDecRenorm(&rans1, &ptr);
            }
            for (size_t i = (in_size & ~(size_t)7); i < in_size; i++) {
                RansSimdDec *which = (i & 4) != 0 ? &rans1 : &rans0;
                dec_bytes[i] = RansWordDecSym(&which->lane[i & 3], &tab);
            }
            res->dec_clocks = __rdtsc() - dec_start_time;
            res->dec_s = timer() - start_time;
        }
        delete tabp;
    } else {
        delete stp;
        return 1;
    }
    res->ok = memcmp(in_bytes, dec_bytes, in_size) == 0; // "decode ok!" (main.cpp:287-290)
    delete stp;
    return 0;
}

// the same loop over the 4096-symbol alias model (u16 symbols): the sed-edited main_alias.cpp of ref_alias12.o
int ref12_loop2(const uint32_t *freqs, uint32_t prob_bits, const uint16_t *in, size_t n, uint8_t *out_buf, size_t out_max_size,
                uint16_t *dec, ref_loop2_times *res, void *bar);

// `threads` shards of n_per symbols each (shard t = symbols [t * n_per, (t + 1) * n_per) of `in`), one pthread per shard,
// thread t pinned to cpus[t % ncpus] when ncpus > 0; every thread codes its own shard with the reference loop of `which`
// (FMT_BYTE / FMT_R64 / FMT_ALIAS: the 2-way loops; FMT_WORD: the 8-way loop of main_simd.cpp with its SSE4.1 decoder;
// 12 = the 2-way alias loop over the 4096-symbol model and u16 symbols).  times[t] receives
// thread t's record; the passes start together (barrier), so the wall time of a pass is the largest per-thread time.
struct loop2_job {
    int which;
    const uint32_t *freqs;
    uint32_t prob_bits;
    const uint8_t *in;
    size_t n_per;
    ref_loop2_times *res;
    pthread_barrier_t *bar;
    int cpu;
    int rc;
};

static void *loop2_worker(void *p)
{
    loop2_job *j = (loop2_job *)p;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    const size_t sym_bytes = j->which == 12 ? 2 : 1;
    const size_t cap = (j->n_per * (j->which == FMT_R64 ? 4 : 2) + 256 + 15) & ~(size_t)15;
    uint8_t *buf = (uint8_t *)malloc(cap);
    uint8_t *dec = (uint8_t *)malloc(j->n_per * sym_bytes + 16);
    if (!buf || !dec) {
        j->rc = 2;
        pthread_barrier_wait(j->bar);
        pthread_barrier_wait(j->bar);
    } else if (j->which == 12) {
        j->rc = ref12_loop2(j->freqs, j->prob_bits, (const uint16_t *)j->in, j->n_per, buf, cap, (uint16_t *)dec, j->res, j->bar);
    } else {
        j->rc = ref_loop2_u8(j->which, j->freqs, j->prob_bits, j->in, j->n_per, buf, cap, dec, j->res, j->bar);
    }
    free(buf);
    free(dec);
    return 0;
}

int ref_time_loop2_mt(int which, const uint32_t *freqs, uint32_t prob_bits, const void *in, size_t n_per, uint32_t threads,
                      const int *cpus, int ncpus, ref_loop2_times *times)
{
    if (threads == 0 || (which != FMT_BYTE && which != FMT_WORD && which != FMT_R64 && which != FMT_ALIAS && which != 12))
        return 1;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, 0, threads);
    std::vector<pthread_t> th(threads);
    std::vector<loop2_job> jobs(threads);
    const size_t sym_bytes = which == 12 ? 2 : 1;
    for (uint32_t t = 0; t < threads; t++) {
        jobs[t] = loop2_job{which, freqs, prob_bits, (const uint8_t *)in + (size_t)t * n_per * sym_bytes, n_per, &times[t], &bar,
                            ncpus > 0 ? cpus[t % (uint32_t)ncpus] : -1, 0};
        pthread_create(&th[t], 0, loop2_worker, &jobs[t]);
    }
    int rc = 0;
    for (uint32_t t = 0; t < threads; t++) {
        pthread_join(th[t], 0);
        rc |= jobs[t].rc;
    }
    pthread_barrier_destroy(&bar);
    return rc;
}

} // extern "C"
