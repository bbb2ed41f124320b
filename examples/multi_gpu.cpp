// examples/multi_gpu.cpp -- the multi-GPU shape of the decode path for a C/C++ caller: one host thread
// and one rans_amd_ctx per visible device, every device owns an independent shard (no payload crosses xGMI), and the
// ONLY communication is one RCCL all-gather of a 40-byte record per rank -- {elapsed s, symbols, stream bytes, kernel
// ms, ok} -- after which rank 0 prints the per-rank table and the whole-job rate (work = sum over ranks, time = max
// over ranks: weak scaling).  The same protocol as bench.py + ryg_rans_amd/sharding.py, without Python or torch.
// The reference has no multi-device code (its only caller shape is main_simd.cpp:131-349); this is what replaces
// "run the loop of main_simd.cpp:313-332 on every core".
//
//   hipcc -O2 -Iinclude examples/multi_gpu.cpp -Lryg_rans_amd/lib -lryg_rans_amd -lrccl -lpthread \
//         -Wl,-rpath,$PWD/ryg_rans_amd/lib -o build/multi_gpu
//   build/multi_gpu [devices (default: all)] [log2 symbols per device (default 26)] [steps (default 10)]
//   build/multi_gpu --split-one [ranks (default: max(2, devices))] [log2 symbols (default 26)] [steps]
//   build/multi_gpu --share     [ranks] [log2 symbols per rank] [steps]     (independent shards, ranks may share devices)
//
// --split-one: ONE container (made once, on the host side of rank 0's device) is split over the ranks by SURVEY 8(e)'s
// rule -- rank g owns chunks [g C / G, (g + 1) C / G) -- and every rank holds ONLY the bytes rans_amd_container_slice
// assigns it; the pieces side by side must be the input.  Ranks beyond the visible devices share devices (rank g on
// device g mod devices, its own context each): the records then travel through host memory instead of RCCL, which is
// what makes the mode testable on a one-GPU box.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "ryg_rans_amd.h"

namespace {

struct ShardRecord { // (ryg_rans_amd/sharding.py ShardRecord, as five doubles)
    double elapsed_s, symbols, stream_bytes, kernel_ms, ok;
};

// Zipf(256, s = 1) bytes, splitmix64 counter based (element i: state = seed + (i + 1) * golden), inverse
// CDF over the double running sums -- the generator bench.py and the oracle use, so rank r's shard here IS rank r's
// shard there (seed = r + 1).
void gen_zipf(uint8_t *out, size_t n, uint64_t seed)
{
    double cdf[256], run = 0.0;
    for (int k = 0; k < 256; ++k) {
        run += 1.0 / std::pow((double)(k + 1), 1.0);
        cdf[k] = run;
    }
    for (size_t i = 0; i < n; ++i) {
        uint64_t z = seed + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) * run;
        const int k = (int)(std::upper_bound(cdf, cdf + 256, u) - cdf);
        out[i] = (uint8_t)(k > 255 ? 255 : k);
    }
}

// what --split-one shares between the ranks (host memory): the container made once, its index, the input
struct Shared {
    std::vector<uint8_t> syms, container, decoded;
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> lengths;
    uint32_t freqs[256];
    uint64_t n = 0;
};

struct Rank {
    int rank = 0;
    const Shared *shared = nullptr; // --split-one: decode this rank's chunk range of shared->container
    uint8_t *decoded = nullptr;     // ... into this host buffer (the range's place in it)
    int device = 0;
    int world = 1;
    uint64_t n = 0;
    int steps = 10;
    ncclComm_t comm = nullptr;
    std::vector<ShardRecord> gathered; // filled on every rank by the all-gather
    float first_pair_ms = 0.f, chosen_ms = 0.f; // rans_amd_probe_placement: what two plain allocations got / what the chosen pair gets
    int rc = 0;
    char msg[256] = "";
};

#define R_CHECK(call)                                                                                      \
    do {                                                                                                   \
        int rc__ = (call);                                                                                 \
        if (rc__ != RANS_AMD_OK) {                                                                         \
            snprintf(r.msg, sizeof r.msg, "%s -> %s (%s)", #call, rans_amd_status_string(rc__), rans_amd_last_error()); \
            r.rc = 1;                                                                                      \
            goto done;                                                                                     \
        }                                                                                                  \
    } while (0)
#define R_HIP(call)                                                                       \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            snprintf(r.msg, sizeof r.msg, "%s -> %s", #call, hipGetErrorString(e__));    \
            r.rc = 1;                                                                     \
            goto done;                                                                    \
        }                                                                                 \
    } while (0)

void run_rank(Rank &r)
{
    const uint32_t n_ways = 64, chunk = 16384, scale_bits = 12; // (16 Ki-symbol chunks: bench.py's, the measured optimum)
    const uint64_t n = r.n;
    rans_amd_ctx *ctx = nullptr;
    rans_amd_model *model = nullptr;
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_cont = nullptr;
    uint8_t *d_cont2 = nullptr, *d_out2 = nullptr, *d_out3 = nullptr; // placement candidates (the ones not kept are freed at the end)
    uint64_t *d_off = nullptr;
    uint32_t *d_len = nullptr;
    double *d_rec = nullptr, *d_all = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ShardRecord rec = {0, 0, 0, 0, 0};
    std::vector<uint8_t> shard, back;
    {
        // FIRST what the closing collective needs -- the stream and the two record buffers: whatever fails after this
        // point, this rank still joins the all-gather (with ok = 0).  If even these cannot be had, the communicator is
        // aborted, so that the peers' all-gather returns an error instead of waiting for a rank that will never come.
        if (hipSetDevice(r.device) != hipSuccess || // (the device of THIS thread; the library switches to its context's
            hipStreamCreate(&stream) != hipSuccess || //  device inside every call anyway)
            hipMalloc((void **)&d_rec, sizeof(ShardRecord)) != hipSuccess ||
            hipMalloc((void **)&d_all, sizeof(ShardRecord) * (size_t)r.world) != hipSuccess) {
            snprintf(r.msg, sizeof r.msg, "device %d: no stream / record buffers (%s)", r.device, hipGetErrorString(hipGetLastError()));
            r.rc = 1;
            if (r.comm)
                (void)ncclCommAbort(r.comm);
            r.comm = nullptr;
            goto done;
        }
        if (r.shared)
            goto split_one;
        shard.resize(n);
        back.resize(n);
        gen_zipf(shard.data(), n, (uint64_t)r.rank + 1);   // independent shard: seed = rank + 1
        R_CHECK(rans_amd_ctx_create(r.device, &ctx));       // one context per device
        R_HIP(hipEventCreate(&ev0));
        R_HIP(hipEventCreate(&ev1));
        const uint64_t nchunks = rans_amd_num_chunks(n, chunk);
        const uint64_t cap = rans_amd_encode_bound(RANS_AMD_FMT_WORD, n, n_ways, chunk);
        R_HIP(hipMalloc((void **)&d_in, n + 256));
        R_HIP(hipMalloc((void **)&d_out, n + 256));
        R_HIP(hipMalloc((void **)&d_cont, cap + 256));
        R_HIP(hipMalloc((void **)&d_off, 8 * (nchunks + 1)));
        R_HIP(hipMalloc((void **)&d_len, 4 * nchunks));
        R_HIP(hipMemcpy(d_in, shard.data(), n, hipMemcpyHostToDevice));
        uint32_t freqs[256];
        R_CHECK(rans_amd_build_model_o0(ctx, RANS_AMD_FMT_WORD, d_in, n, 1, 256, scale_bits, freqs, &model, stream));
        uint64_t total = 0, bad = 0;
        R_CHECK(rans_amd_encode(ctx, model, d_in, n, n_ways, chunk, d_cont, cap, d_off, d_len, &total, stream));
        {   // Setup: where the two buffers of a streaming decode lie is worth 4-6 % on this part and hipMalloc cannot be
            // steered (include/ryg_rans_amd.h, rans_amd_probe_placement) -- a second copy of the container and two more
            // outputs, the library times the six pairs, the fastest pair is the one the timed loop uses.  The candidates
            // are allocated ~20 GiB APART (spacer allocations nobody touches, freed after the probe): a class of device
            // memory is a window of 20-60 GiB of allocation order, buffers allocated in a row share one
            // (profiles/r05_class_map.md).
            void *spacer[2] = {nullptr, nullptr};
            auto space = [&](int i) {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < ((size_t)96 << 30)) return;
                if (hipMalloc(&spacer[i], (size_t)20 << 30) != hipSuccess) { // (another rank of this device was faster)
                    spacer[i] = nullptr;
                    (void)hipGetLastError();
                }
            };
            space(0);
            R_HIP(hipMalloc((void **)&d_cont2, cap + 256));
            R_HIP(hipMalloc((void **)&d_out2, n + 256));
            space(1);
            R_HIP(hipMalloc((void **)&d_out3, n + 256));
            R_HIP(hipMemcpyAsync(d_cont2, d_cont, total, hipMemcpyDeviceToDevice, stream));
            const void *conts[2] = {d_cont, d_cont2};
            void *outs[3] = {d_out, d_out2, d_out3};
            uint32_t bc = 0, bo = 0;
            float ms[6];
            R_CHECK(rans_amd_probe_placement(ctx, model, conts, 2, total, d_off, d_len, n, n_ways, chunk, outs, 3, 0, 0, &bc, &bo, ms, stream));
            r.first_pair_ms = ms[0];
            r.chosen_ms = ms[bc * 3 + bo];
            std::swap(d_cont, bc ? d_cont2 : d_cont); // (the kept pair in d_cont / d_out; everything is freed at the end)
            std::swap(d_out, bo == 1 ? d_out2 : (bo == 2 ? d_out3 : d_out));
            for (void *sp : spacer)
                if (sp) (void)hipFree(sp);
        }
        // warm-up, then `steps` timed decodes of the whole shard (device resident in, device resident out)
        for (int i = 0; i < 3; ++i)
            R_CHECK(rans_amd_decode(ctx, model, d_cont, total, d_off, d_len, n, n_ways, chunk, d_out, nullptr, stream));
        R_HIP(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        R_HIP(hipEventRecord(ev0, stream));
        for (int i = 0; i < r.steps; ++i)
            R_CHECK(rans_amd_decode(ctx, model, d_cont, total, d_off, d_len, n, n_ways, chunk, d_out, nullptr, stream));
        R_HIP(hipEventRecord(ev1, stream));
        R_HIP(hipStreamSynchronize(stream));
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        float ms = 0;
        R_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        const int drc = rans_amd_decode_errors(ctx, &bad, stream);
        R_HIP(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
        const bool same = drc == RANS_AMD_OK && bad == 0 && memcmp(back.data(), shard.data(), n) == 0;
        rec = ShardRecord{elapsed, (double)n, (double)total, ms / r.steps, same ? 1.0 : 0.0};
        goto done;
    }
split_one: {
        // rank g of G: chunks [g C / G, (g + 1) C / G) of the ONE container; it copies to its device only the bytes
        // rans_amd_container_slice names, with the offsets rebased to them
        const Shared &sh = *r.shared;
        const uint64_t C = sh.lengths.size();
        const uint64_t lo = C * (uint64_t)r.rank / (uint64_t)r.world, hi = C * ((uint64_t)r.rank + 1) / (uint64_t)r.world;
        const uint64_t first = std::min(sh.n, lo * chunk), last = std::min(sh.n, hi * chunk), cnt = last - first;
        uint64_t b = 0, e = 0;
        std::vector<uint64_t> rebased(hi - lo + 1);
        R_CHECK(rans_amd_container_slice(sh.offsets.data(), sh.lengths.data(), C, lo, hi, &b, &e, rebased.data()));
        R_CHECK(rans_amd_ctx_create(r.device, &ctx));
        R_CHECK(rans_amd_model_create(ctx, RANS_AMD_FMT_WORD, sh.freqs, 256, scale_bits, &model));
        R_HIP(hipEventCreate(&ev0));
        R_HIP(hipEventCreate(&ev1));
        R_HIP(hipMalloc((void **)&d_cont, (e - b) + 256));
        R_HIP(hipMalloc((void **)&d_out, cnt + 256));
        R_HIP(hipMalloc((void **)&d_off, 8 * (hi - lo + 1)));
        R_HIP(hipMalloc((void **)&d_len, 4 * (hi - lo) + 4));
        R_HIP(hipMemcpy(d_cont, sh.container.data() + b, e - b, hipMemcpyHostToDevice)); // ONLY this rank's bytes
        R_HIP(hipMemcpy(d_off, rebased.data(), 8 * (hi - lo + 1), hipMemcpyHostToDevice));
        R_HIP(hipMemcpy(d_len, sh.lengths.data() + lo, 4 * (hi - lo), hipMemcpyHostToDevice));
        uint64_t bad = 0;
        for (int i = 0; i < 3; ++i)
            R_CHECK(rans_amd_decode(ctx, model, d_cont, e - b, d_off, d_len, cnt, n_ways, chunk, d_out, nullptr, stream));
        R_HIP(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        R_HIP(hipEventRecord(ev0, stream));
        for (int i = 0; i < r.steps; ++i)
            R_CHECK(rans_amd_decode(ctx, model, d_cont, e - b, d_off, d_len, cnt, n_ways, chunk, d_out, nullptr, stream));
        R_HIP(hipEventRecord(ev1, stream));
        R_HIP(hipStreamSynchronize(stream));
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        float ms = 0;
        R_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        const int drc = rans_amd_decode_errors(ctx, &bad, stream);
        R_HIP(hipMemcpy(r.decoded + first, d_out, cnt, hipMemcpyDeviceToHost));
        const bool same = drc == RANS_AMD_OK && bad == 0 && memcmp(r.decoded + first, sh.syms.data() + first, cnt) == 0;
        uint64_t range_bytes = 0;
        for (uint64_t c = lo; c < hi; ++c)
            range_bytes += sh.lengths[c];
        rec = ShardRecord{elapsed, (double)cnt, (double)range_bytes, ms / r.steps, same ? 1.0 : 0.0};
    }
done:
    // Every rank reaches the collective, whatever happened above (a rank that failed reports ok = 0): an all-gather
    // somebody skips would hang the others.  40 bytes per rank over RCCL; it doubles as the end barrier.
    if (!r.comm && d_rec) { // ranks that share devices (--split-one on fewer GPUs than ranks): records through host memory
        r.gathered.assign(1, rec);
    } else if (d_rec && d_all && stream && r.comm) {
        (void)hipMemcpyAsync(d_rec, &rec, sizeof rec, hipMemcpyHostToDevice, stream);
        const ncclResult_t nr = ncclAllGather(d_rec, d_all, sizeof(ShardRecord) / sizeof(double), ncclDouble, r.comm, stream);
        r.gathered.resize((size_t)r.world);
        if (nr == ncclSuccess && hipMemcpyAsync(r.gathered.data(), d_all, sizeof(ShardRecord) * (size_t)r.world,
                                                hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipStreamSynchronize(stream) == hipSuccess) {
            // gathered[] now holds every rank's record
        } else {
            snprintf(r.msg, sizeof r.msg, "ncclAllGather -> %s", ncclGetErrorString(nr));
            r.rc = 1;
            r.gathered.clear();
        }
    } else if (!r.rc) {
        r.rc = 1;
    }
    if (model)
        rans_amd_model_destroy(model);
    if (ctx)
        rans_amd_ctx_destroy(ctx);
    for (void *p : {(void *)d_in, (void *)d_out, (void *)d_cont, (void *)d_off, (void *)d_len, (void *)d_rec, (void *)d_all,
                    (void *)d_cont2, (void *)d_out2, (void *)d_out3})
        if (p)
            (void)hipFree(p);
    if (ev0)
        (void)hipEventDestroy(ev0);
    if (ev1)
        (void)hipEventDestroy(ev1);
    if (stream)
        (void)hipStreamDestroy(stream);
}

} // namespace

int main(int argc, char **argv)
{
    const int visible = rans_amd_device_count();
    if (visible <= 0) {
        fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 1;
    }
    const bool split_one = argc > 1 && strcmp(argv[1], "--split-one") == 0;
    // --share: more ranks than devices are allowed for independent shards as well (rank g on device g mod devices) -- the
    // dry run of an 8-rank job on a box with fewer GPUs: every rank still owns its own shard, context, stream and probe
    const bool share = argc > 1 && strcmp(argv[1], "--share") == 0;
    if (split_one || share) {
        --argc;
        ++argv;
    }
    int world = argc > 1 ? atoi(argv[1]) : (split_one ? std::max(2, visible) : visible);
    world = world < 1 ? 1 : ((!split_one && !share && world > visible) ? visible : world);
    const int log2n = argc > 2 ? atoi(argv[2]) : 26;
    const int steps = argc > 3 ? atoi(argv[3]) : 10;
    const bool use_rccl = world <= visible; // (more ranks than devices: they share devices and gather through host memory)

    Shared shared;
    if (split_one) { // the ONE container: made once through the bulk ABI on device 0, kept in host memory
        const uint32_t n_ways = 64, chunk = 16384;
        const uint64_t n = 1ull << log2n;
        shared.n = n;
        shared.syms.resize(n);
        shared.decoded.assign(n, 0xcc);
        gen_zipf(shared.syms.data(), n, 1);
        rans_amd_ctx *ctx = nullptr;
        rans_amd_model *model = nullptr;
        const uint64_t C = rans_amd_num_chunks(n, chunk), cap = rans_amd_encode_bound(RANS_AMD_FMT_WORD, n, n_ways, chunk);
        uint8_t *d_in = nullptr, *d_cont = nullptr;
        uint64_t *d_off = nullptr, total = 0;
        uint32_t *d_len = nullptr;
        bool ok = rans_amd_ctx_create(0, &ctx) == RANS_AMD_OK && hipMalloc((void **)&d_in, n + 256) == hipSuccess &&
                  hipMalloc((void **)&d_cont, cap + 256) == hipSuccess && hipMalloc((void **)&d_off, 8 * (C + 1)) == hipSuccess &&
                  hipMalloc((void **)&d_len, 4 * C) == hipSuccess &&
                  hipMemcpy(d_in, shared.syms.data(), n, hipMemcpyHostToDevice) == hipSuccess &&
                  rans_amd_build_model_o0(ctx, RANS_AMD_FMT_WORD, d_in, n, 1, 256, 12, shared.freqs, &model, nullptr) == RANS_AMD_OK &&
                  rans_amd_encode(ctx, model, d_in, n, n_ways, chunk, d_cont, cap, d_off, d_len, &total, nullptr) == RANS_AMD_OK;
        if (ok) {
            shared.container.resize(total + 16);
            shared.offsets.resize(C + 1);
            shared.lengths.resize(C);
            ok = hipMemcpy(shared.container.data(), d_cont, total, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(shared.offsets.data(), d_off, 8 * (C + 1), hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(shared.lengths.data(), d_len, 4 * C, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (model)
            rans_amd_model_destroy(model);
        if (ctx)
            rans_amd_ctx_destroy(ctx);
        for (void *p : {(void *)d_in, (void *)d_cont, (void *)d_off, (void *)d_len})
            if (p)
                (void)hipFree(p);
        if (!ok) {
            fprintf(stderr, "--split-one: could not make the container (%s)\n", rans_amd_last_error());
            return 1;
        }
    }

    std::vector<int> devs((size_t)world);
    for (int i = 0; i < world; ++i)
        devs[(size_t)i] = i;
    std::vector<ncclComm_t> comms((size_t)world, nullptr);
    if (use_rccl) {
        const ncclResult_t nr = ncclCommInitAll(comms.data(), world, devs.data()); // one process, one communicator per device
        if (nr != ncclSuccess) {
            fprintf(stderr, "ncclCommInitAll -> %s\n", ncclGetErrorString(nr));
            return 1;
        }
    }
    std::vector<Rank> ranks((size_t)world);
    std::vector<std::thread> threads;
    for (int i = 0; i < world; ++i) {
        ranks[(size_t)i].rank = i;
        ranks[(size_t)i].device = i % visible;
        ranks[(size_t)i].world = world;
        ranks[(size_t)i].n = 1ull << log2n;
        ranks[(size_t)i].steps = steps;
        ranks[(size_t)i].comm = comms[(size_t)i];
        if (split_one) {
            ranks[(size_t)i].shared = &shared;
            ranks[(size_t)i].decoded = shared.decoded.data();
        }
        threads.emplace_back(run_rank, std::ref(ranks[(size_t)i]));
    }
    for (auto &t : threads)
        t.join();
    for (size_t i = 0; i < comms.size(); ++i)
        if (comms[i] && ranks[i].comm) // (a rank that aborted its communicator has cleared its copy)
            ncclCommDestroy(comms[i]);

    int rc = 0;
    for (const Rank &r : ranks)
        if (r.rc) {
            fprintf(stderr, "rank %d: %s\n", r.device, r.msg);
            rc = 1;
        }
    std::vector<ShardRecord> all = ranks[0].gathered; // what RANK 0 received through RCCL
    if (!use_rccl) {                                  // (ranks sharing devices: every rank left its own record)
        all.clear();
        for (const Rank &r : ranks)
            if (r.gathered.size() == 1)
                all.push_back(r.gathered[0]);
    }
    if (all.size() != (size_t)world)
        return 1;
    const double kPeakGBps = 8000.0; // HBM3E peak of one MI355X
    double max_elapsed = 0, syms = 0, alg = 0, max_kernel_ms = 0;
    bool all_ok = true;
    printf("rank  symbols      stream bytes  kernel ms  decoded GB/s  roofline frac  ok\n");
    for (size_t i = 0; i < all.size(); ++i) {
        const ShardRecord &s = all[i];
        printf("%4zu  %11.0f  %12.0f  %9.4f  %12.1f  %13.4f  %s\n", i, s.symbols, s.stream_bytes, s.kernel_ms,
               s.symbols / s.kernel_ms / 1e6, (s.symbols + s.stream_bytes) / s.kernel_ms / 1e6 / kPeakGBps, s.ok == 1.0 ? "yes" : "NO");
        max_elapsed = std::max(max_elapsed, s.elapsed_s);
        max_kernel_ms = std::max(max_kernel_ms, s.kernel_ms);
        syms += s.symbols;
        alg += s.symbols + s.stream_bytes;
        all_ok = all_ok && s.ok == 1.0;
    }
    if (split_one) // the pieces side by side are the input
        all_ok = all_ok && memcmp(shared.decoded.data(), shared.syms.data(), shared.n) == 0;
    else
        for (const Rank &r : ranks) // rans_amd_probe_placement: two plain allocations against the fastest of 2 x 3 candidates
            printf("rank %d placement: first pair %.4f ms, chosen pair %.4f ms (%+.1f %%)\n", r.rank, r.first_pair_ms, r.chosen_ms,
                   r.first_pair_ms > 0 ? (r.chosen_ms / r.first_pair_ms - 1.0) * 100.0 : 0.0);
    const int gpus_used = std::min(world, visible);
    // frac_job: algorithmic bytes of all ranks over the slowest rank's kernel, against the peak of the GPUs in use
    // (bench.py roofline.frac_job); ranks that share a GPU run one after the other on it, so the figure is low by design there
    printf("{\"n_gpus\": %d, \"ranks\": %d, \"mode\": \"%s\", \"steps\": %d, \"value\": %.2f, \"unit\": \"GB/s\", \"ms_per_step\": %.4f, "
           "\"scaling\": \"%s\", \"frac_job\": %.4f, \"bit_exact_roundtrip\": %s, \"records_gathered_by\": \"%s\"}\n",
           gpus_used, world, split_one ? "one container split by chunk range" : "one shard per rank", steps,
           syms * steps / max_elapsed / 1e9, max_elapsed / steps * 1e3, split_one ? "strong" : "weak",
           alg / (max_kernel_ms * 1e-3) / 1e9 / (gpus_used * kPeakGBps), all_ok ? "true" : "false",
           use_rccl ? "ncclAllGather" : "host memory (ranks share devices)");
    puts(all_ok && !rc ? "decode ok!" : "ERROR: bad decoder!");
    return all_ok && !rc ? 0 : 2;
}
