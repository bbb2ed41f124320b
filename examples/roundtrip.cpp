// examples/roundtrip.cpp -- native C++ host using nothing but the C ABI (include/ryg_rans_amd.h)
// and the HIP runtime: build an order-0 model of a buffer, encode it as chunked N-way
// interleaved rANS on the GPU, decode it back, verify, and report device-side timings.
// This is the shape of code a ryg_rans user writes after deleting the driver loops of
// main_simd.cpp:283-343.
//
//   hipcc -O2 -Iinclude examples/roundtrip.cpp -Lryg_rans_amd/lib -lryg_rans_amd \
//         -Wl,-rpath,$PWD/ryg_rans_amd/lib -o build/roundtrip && build/roundtrip [file] [format] [n_ways]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "ryg_rans_amd.h"

#define CHECK(call)                                                                               \
    do {                                                                                          \
        int rc__ = (call);                                                                        \
        if (rc__ != RANS_AMD_OK) {                                                                \
            fprintf(stderr, "%s -> %s (%s)\n", #call, rans_amd_status_string(rc__), rans_amd_last_error()); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define HIP(call)                                                                 \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e__));       \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv)
{
    // ---- input: a file, or 64 MiB of skewed synthetic bytes
    std::vector<uint8_t> in;
    if (argc > 1 && strcmp(argv[1], "-") != 0) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { perror(argv[1]); return 1; }
        fseek(f, 0, SEEK_END);
        in.resize((size_t)ftell(f));
        fseek(f, 0, SEEK_SET);
        if (fread(in.data(), 1, in.size(), f) != in.size()) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);
    } else {
        in.resize(64u << 20);
        uint64_t s = 1;
        for (auto &b : in) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            uint32_t r = (uint32_t)(s >> 40);
            b = (uint8_t)(__builtin_clz(r | 1u) * 8 + (r & 7)); // geometric-ish: low values dominate
        }
    }
    const char *fmt_name = argc > 2 ? argv[2] : "word";
    const int format = !strcmp(fmt_name, "byte") ? RANS_AMD_FMT_BYTE : !strcmp(fmt_name, "r64") ? RANS_AMD_FMT_R64
                       : !strcmp(fmt_name, "alias") ? RANS_AMD_FMT_ALIAS : RANS_AMD_FMT_WORD;
    const uint32_t scale_bits = format == RANS_AMD_FMT_WORD ? 12 : format == RANS_AMD_FMT_ALIAS ? 16 : 14;
    const uint32_t n_ways = argc > 3 ? (uint32_t)atoi(argv[3]) : 64;
    const uint32_t chunk = 32768;
    const uint64_t n = in.size();

    rans_amd_ctx *ctx = nullptr;
    CHECK(rans_amd_ctx_create(0, &ctx));

    // ---- model: count on the GPU, normalise on the host (bit-identical to SymbolStats)
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_cont = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_len = nullptr;
    HIP(hipMalloc((void **)&d_in, n + 256));
    HIP(hipMalloc((void **)&d_out, n + 256));
    HIP(hipMemcpy(d_in, in.data(), n, hipMemcpyHostToDevice));
    uint32_t freqs[256], cum[257];
    CHECK(rans_amd_count_freqs(ctx, d_in, n, 1, 256, freqs, nullptr));
    CHECK(rans_amd_normalize_freqs(freqs, cum, 256, 1u << scale_bits));
    rans_amd_model *model = nullptr;
    CHECK(rans_amd_model_create(ctx, format, freqs, 256, scale_bits, &model));

    // ---- encode -> container + index, all device resident
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk);
    const uint64_t cap = rans_amd_encode_bound(format, n, n_ways, chunk);
    HIP(hipMalloc((void **)&d_cont, cap + 256));
    HIP(hipMalloc((void **)&d_off, 8 * (nchunks + 1)));
    HIP(hipMalloc((void **)&d_len, 4 * (nchunks ? nchunks : 1)));
    CHECK(rans_amd_set_timing(ctx, 1));
    uint64_t total = 0;
    CHECK(rans_amd_encode(ctx, model, d_in, n, n_ways, chunk, d_cont, cap, d_off, d_len, &total, nullptr));

    // ---- decode + integrity verdict
    uint64_t bad = 0;
    CHECK(rans_amd_decode(ctx, model, d_cont, total, d_off, d_len, n, n_ways, chunk, d_out, &bad, nullptr));
    float dec_ms = 0, enc_ms = 0;
    CHECK(rans_amd_last_kernel_ms(ctx, &dec_ms, &enc_ms));

    std::vector<uint8_t> back(n);
    HIP(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
    const bool same = memcmp(back.data(), in.data(), n) == 0;
    printf("%s format, %u-way, %llu symbols in %llu chunks: %llu bytes (%.4f bytes/symbol)\n", fmt_name, n_ways,
           (unsigned long long)n, (unsigned long long)nchunks, (unsigned long long)total, (double)total / (double)n);
    printf("encode %.3f ms (%.1f GB/s), decode %.3f ms (%.1f GB/s), kernel %s\n", enc_ms, n / enc_ms / 1e6, dec_ms,
           n / dec_ms / 1e6, rans_amd_last_decode_kernel(ctx));

    // ---- the one-trip encoder: slots sized from the model (rans_amd_encode_slots_sized), room for a few overflowed chunks
    //      behind them; every chunk's stream must be byte for byte the one rans_amd_encode placed in the compact container
    bool sized_ok = true;
    if (nchunks) {
        const uint64_t slot = rans_amd_tight_slot_bytes(model, n_ways, chunk);
        const uint64_t cap2 = rans_amd_encode_sized_bound(format, n, n_ways, chunk, slot, nchunks / 64 + 4);
        uint8_t *d_cont2 = nullptr;
        uint64_t *d_off2 = nullptr;
        uint32_t *d_len2 = nullptr;
        HIP(hipMalloc((void **)&d_cont2, cap2 + 256));
        HIP(hipMalloc((void **)&d_off2, 8 * (nchunks + 1)));
        HIP(hipMalloc((void **)&d_len2, 4 * nchunks));
        uint64_t total2 = 0;
        CHECK(rans_amd_encode_slots_sized(ctx, model, d_in, n, n_ways, chunk, slot, d_cont2, cap2, d_off2, d_len2, &total2, nullptr));
        float enc2_ms = 0, dec2_ms = 0;
        CHECK(rans_amd_last_kernel_ms(ctx, &dec2_ms, &enc2_ms));
        HIP(hipMemset(d_out, 0, n));
        CHECK(rans_amd_decode(ctx, model, d_cont2, total2, d_off2, d_len2, n, n_ways, chunk, d_out, &bad, nullptr));
        HIP(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
        sized_ok = memcmp(back.data(), in.data(), n) == 0;
        std::vector<uint8_t> c1(total), c2(total2);
        std::vector<uint64_t> o1(nchunks + 1), o2(nchunks + 1);
        std::vector<uint32_t> l1(nchunks), l2(nchunks);
        HIP(hipMemcpy(c1.data(), d_cont, total, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(c2.data(), d_cont2, total2, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(o1.data(), d_off, 8 * (nchunks + 1), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(o2.data(), d_off2, 8 * (nchunks + 1), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(l1.data(), d_len, 4 * nchunks, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(l2.data(), d_len2, 4 * nchunks, hipMemcpyDeviceToHost));
        for (uint64_t c = 0; c < nchunks && sized_ok; ++c)
            sized_ok = l1[c] == l2[c] && memcmp(&c1[o1[c]], &c2[o2[c]], l1[c]) == 0;
        printf("sized slots (%llu bytes each, worst case %llu): container %llu bytes = %.3f x input, encode %.3f ms, %s\n",
               (unsigned long long)slot, (unsigned long long)rans_amd_slot_bytes(format, n, n_ways, chunk), (unsigned long long)total2,
               (double)total2 / (double)n, enc2_ms, sized_ok ? "every chunk equals the compact container's" : "MISMATCH");
        // ---- "keep it": what the reference does with fwrite(rans_begin, ...) (main.cpp:182-188).  The sized container went to
        //      the host in ONE copy above (c2); rans_amd_container_pack_indexed writes the self-describing file from it chunk by
        //      chunk -- no compaction pass on the device -- and a reader needs nothing but that file.
        if (sized_ok) {
            rans_amd_container_info info;
            memset(&info, 0, sizeof info);
            info.format = (uint32_t)format;
            info.scale_bits = scale_bits;
            info.nsyms = 256;
            info.n_ways = n_ways;
            info.chunk_syms = chunk;
            info.sym_bytes = 1;
            info.n_symbols = n;
            info.n_chunks = nchunks;
            info.payload_bytes = rans_amd_packed_payload_bytes(l2.data(), nchunks);
            std::vector<uint8_t> file((size_t)rans_amd_container_bytes(&info));
            uint64_t wrote = 0;
            CHECK(rans_amd_container_pack_indexed(&info, freqs, o2.data(), l2.data(), c2.data(), total2, file.data(), file.size(), &wrote));
            rans_amd_container_info back_info;
            const uint32_t *f_freqs = nullptr, *f_lens = nullptr;
            const void *f_payload = nullptr;
            CHECK(rans_amd_container_parse(file.data(), wrote, &back_info, &f_freqs, &f_lens, &f_payload));
            std::vector<uint64_t> f_offs(nchunks + 1);
            CHECK(rans_amd_offsets_from_lengths(f_lens, nchunks, f_offs.data()));
            rans_amd_model *m2 = nullptr;
            CHECK(rans_amd_model_create(ctx, (int)back_info.format, f_freqs, back_info.nsyms, back_info.scale_bits, &m2));
            HIP(hipMemcpy(d_cont2, f_payload, back_info.payload_bytes, hipMemcpyHostToDevice));
            HIP(hipMemcpy(d_off2, f_offs.data(), 8 * (nchunks + 1), hipMemcpyHostToDevice));
            HIP(hipMemcpy(d_len2, f_lens, 4 * nchunks, hipMemcpyHostToDevice));
            HIP(hipMemset(d_out, 0, n));
            CHECK(rans_amd_decode(ctx, m2, d_cont2, back_info.payload_bytes, d_off2, d_len2, back_info.n_symbols, back_info.n_ways,
                                  back_info.chunk_syms, d_out, &bad, nullptr));
            HIP(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
            sized_ok = memcmp(back.data(), in.data(), n) == 0;
            printf("file from the sized container (rans_amd_container_pack_indexed): %llu bytes, decodes %s\n", (unsigned long long)wrote,
                   sized_ok ? "to the input" : "WRONG");
            rans_amd_model_destroy(m2);
        }
        for (void *ptr : {(void *)d_cont2, (void *)d_off2, (void *)d_len2})
            (void)hipFree(ptr);
    }
    puts(same && sized_ok ? "decode ok!" : "ERROR: bad decoder!");

    rans_amd_model_destroy(model);
    rans_amd_ctx_destroy(ctx);
    for (void *ptr : {(void *)d_in, (void *)d_out, (void *)d_cont, (void *)d_off, (void *)d_len})
        (void)hipFree(ptr);
    return same && sized_ok ? 0 : 2;
}
