// tools/ubench_mem.hip -- what the memory system of one MI355X gives a streaming kernel: read-only, write-only and
// copy rates over 1 GiB buffers for a few access shapes (bytes per lane per instruction, cache policy bits, grid
// size), 20 launches each after a warm-up.  The headline decoder moves 0.84 GB in and 1.07 GB out per launch: these
// are the ceilings its 0.38-0.41 ms is to be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mem.hip -o build/ubench_mem && timeout 300 build/ubench_mem
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

// MODE: 0 copy, 1 read only (sum kept), 2 write only;  POL: 0 plain, 1 nt loads + nt stores, 2 plain loads + nt stores
template <int MODE, int POL, int W> // W = dwords per lane per access (1, 2 or 4)
__global__ void __launch_bounds__(256) k_stream(const uint32_t *src, uint32_t *dst, uint64_t ndw, uint32_t *sink)
{
    const uint64_t per_block = 256ull * W;
    const uint64_t stride = (uint64_t)gridDim.x * per_block;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * per_block + threadIdx.x * W; i < ndw; i += stride) {
        if constexpr (W == 4) {
            u32x4 v = {1u, 2u, 3u, 4u};
            if constexpr (MODE != 2)
                v = POL == 1 ? __builtin_nontemporal_load((const u32x4 GLOBAL *)(uintptr_t)(src + i))
                             : *(const u32x4 GLOBAL *)(uintptr_t)(src + i);
            if constexpr (MODE != 1) {
                if constexpr (POL != 0)
                    __builtin_nontemporal_store(v, (u32x4 GLOBAL *)(uintptr_t)(dst + i));
                else
                    *(u32x4 GLOBAL *)(uintptr_t)(dst + i) = v;
            } else
                acc += v.x ^ v.y ^ v.z ^ v.w;
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                uint32_t v = 7u;
                if constexpr (MODE != 2)
                    v = POL == 1 ? __builtin_nontemporal_load((const uint32_t GLOBAL *)(uintptr_t)(src + i + k))
                                 : src[i + k];
                if constexpr (MODE != 1) {
                    if constexpr (POL != 0)
                        __builtin_nontemporal_store(v, (uint32_t GLOBAL *)(uintptr_t)(dst + i + k));
                    else
                        dst[i + k] = v;
                } else
                    acc += v;
            }
        }
    }
    if (MODE == 1 && acc == 0x12345678u)
        sink[0] = acc;
}

// the decoder's shape: every wave reads a contiguous 26 KiB "stream" 16 bytes per lane and writes a contiguous 32 KiB
// of "symbols" one dword per lane per instruction (256 B per store instruction), chunks claimed by striding
template <int SW> // store width in dwords per lane
__global__ void __launch_bounds__(1024) k_decoder_shape(const uint32_t *src, uint32_t *dst, uint32_t nchunks, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (uint32_t c = blockIdx.x * 16u + wave; c < nchunks; c += gridDim.x * 16u) {
        const uint32_t *s = src + (uint64_t)c * 6656u; // 26 KiB
        uint32_t *d = dst + (uint64_t)c * 8192u;       // 32 KiB
        for (uint32_t k = 0; k < 26u; ++k) {           // 1 KiB of stream per 1.23 KiB of symbols, roughly
            const u32x4 v = __builtin_nontemporal_load((const u32x4 GLOBAL *)(uintptr_t)(s + k * 256u + lane * 4u));
            acc += v.x ^ v.w;
        }
        if constexpr (SW == 1) {
            for (uint32_t k = 0; k < 128u; ++k)
                __builtin_nontemporal_store(acc + k, (uint32_t GLOBAL *)(uintptr_t)(d + k * 64u + lane));
        } else {
            for (uint32_t k = 0; k < 32u; ++k) {
                const u32x4 v = {acc, acc + k, acc, k};
                __builtin_nontemporal_store(v, (u32x4 GLOBAL *)(uintptr_t)(d + k * 256u + lane * 4u));
            }
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

// one tile per block, no loop (the shape of an elementwise library kernel): block b writes bytes [b, b + 1) * TILE
template <int W, int U, int NT> // W dwords per lane per store, U stores per lane, NT = non-temporal
__global__ void __launch_bounds__(256) k_write_tile(uint32_t *dst, uint32_t val)
{
    uint32_t *d = dst + (uint64_t)blockIdx.x * (256u * W * U) + threadIdx.x * W;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if constexpr (W == 4) {
            const u32x4 v = {val, val, val, val};
            if constexpr (NT)
                __builtin_nontemporal_store(v, (u32x4 GLOBAL *)(uintptr_t)(d + u * 256 * W));
            else
                *(u32x4 GLOBAL *)(uintptr_t)(d + u * 256 * W) = v;
        } else {
            if constexpr (NT)
                __builtin_nontemporal_store(val, (uint32_t GLOBAL *)(uintptr_t)(d + u * 256 * W));
            else
                *(uint32_t GLOBAL *)(uintptr_t)(d + u * 256 * W) = val;
        }
    }
}

template <typename F> static float time_ms(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i)
        launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i)
        launch();
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 20.0f;
}

int main()
{
    const uint64_t bytes = 1ull << 30, ndw = bytes / 4;
    uint32_t *src, *dst, *sink;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess)
        return 1;
    (void)hipMemset(src, 1, bytes);
    (void)hipMemset(dst, 2, bytes);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs; 1 GiB buffers; GB/s counts bytes read + bytes written\n", cus);
#define RUN(LABEL, MODE, POL, W, BLOCKS)                                                                              \
    {                                                                                                                \
        const float ms = time_ms([&] { hipLaunchKernelGGL((k_stream<MODE, POL, W>), dim3(BLOCKS), dim3(256), 0, 0, src, dst, ndw, sink); }); \
        const double gb = (MODE == 0 ? 2.0 : 1.0) * bytes / 1e9;                                                      \
        printf("%-44s blocks %6d  %.4f ms  %.0f GB/s\n", LABEL, (int)(BLOCKS), ms, gb / (ms * 1e-3));                  \
        fflush(stdout);                                                                                              \
    }
    for (int mult : {4, 8, 16, 32}) {
        RUN("copy   16 B/lane plain", 0, 0, 4, cus * mult);
        RUN("copy   16 B/lane nt/nt", 0, 1, 4, cus * mult);
        RUN("copy   16 B/lane plain loads, nt stores", 0, 2, 4, cus * mult);
    }
    RUN("copy    4 B/lane nt/nt", 0, 1, 1, cus * 16);
    RUN("copy    8 B/lane (2 dword ops) nt/nt", 0, 1, 2, cus * 16);
    for (int mult : {8, 16, 32}) {
        RUN("read   16 B/lane plain", 1, 0, 4, cus * mult);
        RUN("read   16 B/lane nt", 1, 1, 4, cus * mult);
        RUN("write  16 B/lane plain", 2, 0, 4, cus * mult);
        RUN("write  16 B/lane nt", 2, 1, 4, cus * mult);
    }
    RUN("write   4 B/lane plain", 2, 0, 1, cus * 16);
    RUN("write   4 B/lane nt", 2, 1, 1, cus * 16);
#define RUNT(LABEL, W, U, NT)                                                                                         \
    {                                                                                                                \
        const uint32_t blocks = (uint32_t)(ndw / (256u * W * U));                                                     \
        const float ms = time_ms([&] { hipLaunchKernelGGL((k_write_tile<W, U, NT>), dim3(blocks), dim3(256), 0, 0, dst, 7u); }); \
        printf("%-44s blocks %7u  %.4f ms  %.0f GB/s\n", LABEL, blocks, ms, bytes / 1e9 / (ms * 1e-3));               \
        fflush(stdout);                                                                                              \
    }
    RUNT("write tile 16 B/lane x1 plain", 4, 1, 0);
    RUNT("write tile 16 B/lane x4 plain", 4, 4, 0);
    RUNT("write tile 16 B/lane x4 nt", 4, 4, 1);
    RUNT("write tile 16 B/lane x16 plain", 4, 16, 0);
    RUNT("write tile  4 B/lane x4 plain", 1, 4, 0);
    RUNT("write tile  4 B/lane x16 plain", 1, 16, 0);
    RUNT("write tile  4 B/lane x16 nt", 1, 16, 1);
    RUNT("write tile  4 B/lane x64 plain", 1, 64, 0);
    {
        const uint32_t nchunks = 32768;
        const float ms1 = time_ms([&] { hipLaunchKernelGGL((k_decoder_shape<1>), dim3(cus * 2), dim3(1024), 0, 0, src, dst, nchunks, sink); });
        const float ms4 = time_ms([&] { hipLaunchKernelGGL((k_decoder_shape<4>), dim3(cus * 2), dim3(1024), 0, 0, src, dst, nchunks, sink); });
        const double gb = nchunks * (26624.0 + 32768.0) / 1e9;
        printf("decoder shape (26 KiB in, 32 KiB out per wave-chunk, 2 blocks of 16 waves per CU): dword stores %.4f ms %.0f GB/s | "
               "16-byte stores %.4f ms %.0f GB/s\n", ms1, gb / (ms1 * 1e-3), ms4, gb / (ms4 * 1e-3));
    }
    return 0;
}
