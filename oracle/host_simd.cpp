// host_simd.cpp -- CPU-baseline helpers (TEST / BENCH INFRASTRUCTURE ONLY; the product never loads this):
//   * host_decode_word_avx512x2: a 32-way word stream decoded by two 16-lane AVX-512 vectors
//     (include/ryg_rans_amd/compat/rans_word_avx512.h) -- the "stronger CPU baseline" of SURVEY.md 8(f)4;
//   * host_time_threads: any decoder of the shape int f(freqs, stream, n, out) -- the reference's SSE4.1 loop from
//     oracle/_ref included -- over independent shards on PINNED threads: one per physical core first, SMT siblings
//     after (an unpinned sweep on a 256-CPU host fell from 16 GB/s at 32 threads to 12 at 256: the scheduler stacked
//     threads on sibling CPUs; round 2's BENCH line shows it).
// Compiled with -mavx512f; nothing but host_decode_word_avx512x2 executes AVX-512 instructions, and callers ask
// host_has_avx512() first.
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <vector>

#include "../include/ryg_rans_amd/compat/rans_word_avx512.h"

extern "C" {

int host_has_avx512(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") ? 1 : 0;
}

// freqs: 256 normalised 12-bit frequencies.  stream: a 32-way word stream of n symbols (coder i codes symbols
// i, i + 32, ...; main_simd.cpp:287-300 with 8 -> 32), followed by >= 32 bytes of padding.
int host_decode_word_avx512x2(const uint32_t *freqs, const uint8_t *stream, size_t n, uint8_t *out)
{
    RansWordTables *tab = new RansWordTables;
    RansWordTables512 *t5 = new RansWordTables512;
    uint32_t start = 0;
    for (int s = 0; s < 256; s++) {
        RansWordTablesInitSymbol(tab, (uint8_t)s, start, freqs[s]);
        start += freqs[s];
    }
    RansWordTables512Init(t5, tab);
    uint16_t *ptr = (uint16_t *)stream;
    RansAvx512Dec r0, r1;
    RansAvx512DecInit(&r0, &ptr);
    RansAvx512DecInit(&r1, &ptr);
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m128i s0 = RansAvx512DecSym(&r0, t5);
        const __m128i s1 = RansAvx512DecSym(&r1, t5);
        _mm_storeu_si128((__m128i *)(out + i), s0);
        _mm_storeu_si128((__m128i *)(out + i + 16), s1);
        RansAvx512DecRenorm(&r0, &ptr);
        RansAvx512DecRenorm(&r1, &ptr);
    }
    for (; i < n; i++) { // tail: lanes 0 .. (n mod 32) - 1, no renormalisation (main_simd.cpp:328-332)
        RansAvx512Dec *which = (i & 16) ? &r1 : &r0;
        out[i] = RansWordDecSym(&which->lane[i & 15], tab);
    }
    delete t5;
    delete tab;
    return 0;
}

// The CPUs this process may run on, physical cores first: one CPU of every core (ascending), then the second SMT
// sibling of every core, and so on.  Returns how many were written.
int host_cpu_order(int *cpus, int cap)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0)
        return 0;
    struct Cpu { int cpu, first_sibling, rank; };
    std::vector<Cpu> all;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &set))
            continue;
        char path[128];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        int first = c;
        if (FILE *f = fopen(path, "r")) {
            if (fscanf(f, "%d", &first) != 1)
                first = c;
            fclose(f);
        }
        all.push_back(Cpu{c, first, 0});
    }
    // rank of a CPU among the allowed siblings of its core
    for (size_t i = 0; i < all.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (all[j].first_sibling == all[i].first_sibling)
                all[i].rank++;
    std::stable_sort(all.begin(), all.end(), [](const Cpu &a, const Cpu &b) { return a.rank < b.rank; });
    int n = 0;
    for (const Cpu &c : all)
        if (n < cap)
            cpus[n++] = c.cpu;
    return n;
}

typedef int (*host_decode_fn)(const uint32_t *freqs, const uint8_t *stream, size_t n, uint8_t *out);

struct host_job {
    host_decode_fn fn;
    const uint32_t *freqs;
    const uint8_t *streams;
    const uint64_t *offsets;
    size_t n_per;
    uint8_t *out;
    uint32_t shards, threads, tid, reps;
    int cpu; // -1: not pinned
};

static void *host_worker(void *p)
{
    host_job *j = (host_job *)p;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    for (uint32_t rep = 0; rep < j->reps; rep++)
        for (uint32_t s = j->tid; s < j->shards; s += j->threads)
            j->fn(j->freqs, j->streams + j->offsets[s], j->n_per, j->out + (size_t)s * j->n_per);
    return 0;
}

// `shards` independent streams (stream s at streams + offsets[s], n_per symbols each) decoded `reps` times by `threads`
// pthreads, shards dealt round robin, thread t pinned to the t-th CPU of host_cpu_order when pin != 0.  Returns wall
// seconds (CLOCK_MONOTONIC) for the whole batch, thread creation included (reps make that negligible).
double host_time_threads(void *fn, const uint32_t *freqs, const uint8_t *streams, const uint64_t *offsets, uint32_t shards,
                         size_t n_per, uint8_t *out, uint32_t threads, uint32_t reps, int pin)
{
    std::vector<int> order(4096);
    const int ncpu = pin ? host_cpu_order(order.data(), (int)order.size()) : 0;
    std::vector<pthread_t> th(threads);
    std::vector<host_job> jobs(threads);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (uint32_t t = 0; t < threads; t++) {
        jobs[t] = host_job{(host_decode_fn)fn, freqs, streams, offsets, n_per, out, shards, threads, t, reps,
                           ncpu > 0 ? order[t % (uint32_t)ncpu] : -1};
        pthread_create(&th[t], 0, host_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; t++)
        pthread_join(th[t], 0);
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

} // extern "C"
