// compat_avx2_driver.cpp -- test / timing harness for include/ryg_rans_amd/compat/rans_word_avx2.h
// (built with -mavx2 by tests/test_compat_headers.py and tools/cpu_reference_rates.sh).
#include <stdint.h>
#include <string.h>
#include <x86intrin.h>

#include "rans_word_avx2.h"

static RansWordTables *make_tables(const uint32_t *freqs, const uint32_t *cum)
{
    RansWordTables *tab = new RansWordTables;
    memset(tab, 0, sizeof(*tab));
    for (int s = 0; s < 256; s++)
        RansWordTablesInitSymbol(tab, (uint8_t)s, cum[s], freqs[s]);
    return tab;
}

// 8-way word stream (16 readable bytes behind `len`) -> n symbols; 0 when the cursor ends on the stream end
static int decode_avx2(const RansWordTables *tab, const uint8_t *stream, size_t len, size_t n, uint8_t *out)
{
    uint16_t *p = (uint16_t *)stream;
    RansAvx2Dec r;
    RansAvx2DecInit(&r, &p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint64_t s = RansAvx2DecSym(&r, tab);
        memcpy(out + i, &s, 8);
        RansAvx2DecRenorm(&r, &p);
    }
    for (; i < n; i++) // tail: lanes 0 .. n%8-1, scalar, no renormalisation (main_simd.cpp:328-332)
        out[i] = RansWordDecSym(&r.lane[i & 7], tab);
    return (const uint8_t *)p == stream + len ? 0 : 3;
}

// same stream through the 4-lane SSE4.1 routines of rans_word_compat.h (loop shape of main_simd.cpp:313-332)
static int decode_sse41(const RansWordTables *tab, const uint8_t *stream, size_t len, size_t n, uint8_t *out)
{
    uint16_t *p = (uint16_t *)stream;
    RansSimdDec r0, r1;
    RansSimdDecInit(&r0, &p);
    RansSimdDecInit(&r1, &p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint32_t a = RansSimdDecSym(&r0, tab), b = RansSimdDecSym(&r1, tab);
        memcpy(out + i, &a, 4);
        memcpy(out + i + 4, &b, 4);
        RansSimdDecRenorm(&r0, &p);
        RansSimdDecRenorm(&r1, &p);
    }
    for (; i < n; i++)
        out[i] = RansWordDecSym(&((i & 4) ? r1 : r0).lane[i & 3], tab);
    return (const uint8_t *)p == stream + len ? 0 : 3;
}

// (8 V)-way streams with V vectors in flight: V independent dependency chains hide the gather latency
template <int V>
static int decode_avx2_wide(const RansWordTables *tab, const uint8_t *stream, size_t len, size_t n, uint8_t *out)
{
    uint16_t *p = (uint16_t *)stream;
    RansAvx2Dec r[V];
    for (int v = 0; v < V; v++)
        RansAvx2DecInit(&r[v], &p);
    size_t i = 0;
    for (; i + 8 * V <= n; i += 8 * V) {
        for (int v = 0; v < V; v++) {
            const uint64_t s = RansAvx2DecSym(&r[v], tab);
            memcpy(out + i + 8 * v, &s, 8);
        }
        for (int v = 0; v < V; v++)
            RansAvx2DecRenorm(&r[v], &p);
    }
    for (size_t j = 0; i < n; i++, j++)
        out[i] = RansWordDecSym(&r[j >> 3].lane[j & 7], tab);
    return (const uint8_t *)p == stream + len ? 0 : 3;
}

extern "C" {

int compat_decode_avx2(const uint32_t *freqs, const uint32_t *cum, const uint8_t *stream, size_t len, size_t n,
                       uint8_t *out)
{
    RansWordTables *tab = make_tables(freqs, cum);
    const int rc = decode_avx2(tab, stream, len, n, out);
    delete tab;
    return rc;
}

// best-of-`reps` clocks per symbol; which = 0: AVX2 8-lane, 1: SSE4.1 2 x 4-lane (both 8-way streams),
// 2 / 3: AVX2 with 2 / 4 vectors on 16- / 32-way streams
double compat_time_word8(int which, const uint32_t *freqs, const uint32_t *cum, const uint8_t *stream, size_t len,
                         size_t n, uint8_t *out, int reps)
{
    RansWordTables *tab = make_tables(freqs, cum);
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        const uint64_t t0 = __rdtsc();
        const int rc = which == 0   ? decode_avx2(tab, stream, len, n, out)
                       : which == 1 ? decode_sse41(tab, stream, len, n, out)
                       : which == 2 ? decode_avx2_wide<2>(tab, stream, len, n, out)
                                    : decode_avx2_wide<4>(tab, stream, len, n, out);
        const uint64_t t1 = __rdtsc();
        if (rc != 0) {
            best = -1.0;
            break;
        }
        const double c = (double)(t1 - t0) / (double)n;
        if (c < best)
            best = c;
    }
    delete tab;
    return best;
}
}
