// tools/ubench3.hip -- third issue-cost census for gfx950: the 64-bit VALU forms the rans64 decode step is made of
// (v_lshrrev_b64, v_mad_u64_u32, v_lshl_add_u64, v_cmp_*_u64, v_cndmask, v_addc) and the LDS pipe under the access
// patterns of the lane-per-stream decoder (random byte / dword / qword reads over 16 KiB, ds_read2_b32 into 136-byte
// rows), 16 waves per CU.  Method as ubench2: 8 and 4 waves per SIMD, marginal cost = (8w - 4w).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o build/ubench3 && timeout 200 build/ubench3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITERS = 2048;
constexpr int UNROLL = 8;

// 64-bit chain in %0 (a VGPR pair), %1 %2 32-bit chains, %3 %4 constant VGPRs, %5 a constant SGPR pair
#define KERNEL64(NAME, ASM)                                                                        \
    __global__ void __launch_bounds__(512) NAME(uint32_t *out, uint32_t seed)                      \
    {                                                                                              \
        uint64_t a[UNROLL];                                                                        \
        uint32_t b[UNROLL], d[UNROLL];                                                             \
        const uint32_t m = seed | 0x00ff00ffu, c = (seed + threadIdx.x) | 1u;                      \
        const uint64_t lim = (uint64_t)__builtin_amdgcn_readfirstlane(seed) << 20;                 \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                                         \
        {                                                                                          \
            a[i] = ((uint64_t)(threadIdx.x * 2654435761u + i) << 17) | 12345u;                     \
            b[i] = threadIdx.x + i;                                                                \
            d[i] = threadIdx.x * 77u + i;                                                          \
        }                                                                                          \
        for (int it = 0; it < ITERS; ++it) {                                                       \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                                     \
                asm volatile(ASM : "+v"(a[i]), "+v"(b[i]), "+v"(d[i]) : "v"(m), "v"(c), "s"(lim) : "vcc", "s10", "s11"); \
        }                                                                                          \
        uint32_t s = 0;                                                                            \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32) ^ b[i] ^ d[i]; \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }

KERNEL64(k_nop_ref, "v_and_b32 %1, %3, %1")
KERNEL64(k_lshr64_i, "v_lshrrev_b64 %0, 14, %0")
KERNEL64(k_lshl64_i, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %3, %0")
KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_cmp64, "v_cmp_gt_u64 vcc, %5, %0")
KERNEL64(k_cmp32, "v_cmp_gt_u32 vcc, %3, %1")
KERNEL64(k_cmp_cnd1, "v_cmp_gt_u32 vcc, %3, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %4, vcc")
KERNEL64(k_cmp_cnd3, "v_cmp_gt_u32 vcc, %3, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc")
KERNEL64(k_cmp_cnd2, "v_cmp_gt_u32 vcc, %3, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc")
KERNEL64(k_cmp_cnd2i, "v_cmp_gt_u32 vcc, %3, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_and_b32 %1, %3, %1\n\tv_cndmask_b32 %2, %2, %4, vcc")
KERNEL64(k_cmp_cnd2s, "v_cmp_gt_u32 s[10:11], %3, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %4, s[10:11]\n\tv_cndmask_b32 %2, %2, %4, s[10:11]")
KERNEL64(k_cmpx_3mov, "v_cmpx_gt_u32 vcc, %3, %1\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %3\n\tv_add_u32 %2, 4, %2\n\ts_mov_b64 exec, -1")
KERNEL64(k_cmpx64_3mov, "v_cmpx_gt_u64 vcc, %5, %0\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %3\n\tv_add_u32 %2, 4, %2\n\ts_mov_b64 exec, -1")
KERNEL64(k_cmp_addc, "v_cmp_gt_u32 vcc, %3, %1\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc")
KERNEL64(k_alignbit, "v_alignbit_b32 %1, %1, %3, 14")
KERNEL64(k_mullo, "v_mul_lo_u32 %1, %1, %3")
KERNEL64(k_mad24, "v_mad_u32_u24 %1, %1, %3, %4")
KERNEL64(k_mov64, "v_mov_b64 %0, %0")
KERNEL64(k_step_r64, // one whole rans64 decode update on made-up data: what the compiler emits today
         "v_lshrrev_b64 %0, 14, %0\n\tv_sub_u32 %1, %1, %3\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0\n\tv_mad_u32_u24 %2, %4, %1, %2")

typedef void (*fn)(uint32_t *, uint32_t);

// ---- LDS pipe: every lane reads a pseudo-random place, addresses computed by a cheap xorshift-free update so
// that the LDS pipe, not the VALU, is the limiter (1 v_mad_u32_u24 + 1 v_and per read).
template <int MODE> __global__ void __launch_bounds__(1024) k_lds(uint32_t *out, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < 40960u; i += blockDim.x) // 160 KiB
        lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t a = threadIdx.x * 2654435761u + seed, acc = 0;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t row = 18432u + wave * 8704u + lane * 136u; // ring rows of the lane decoder
    for (int it = 0; it < ITERS * 4; ++it) {
        a = a * 1103515245u + 12345u;
        const uint32_t r = a >> 8;
        if constexpr (MODE == 0) { // random u8 over 16 KiB
            uint32_t v;
            asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(r & 16383u));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += v;
        } else if constexpr (MODE == 1) { // random b32 over 64 KiB
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(r & 0xfffcu));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += v;
        } else if constexpr (MODE == 2) { // random b64 over 2 KiB (256 records), uniform symbols
            uint64_t v;
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((r & 0x7f8u) + 16384u));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += (uint32_t)v;
        } else if constexpr (MODE == 3) { // ds_read2_b32 at a random dword of the lane's own 128-byte ring row
            uint64_t v;
            asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(row + (r & 0x7cu)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += (uint32_t)v;
        } else if constexpr (MODE == 4) { // ds_read_b32 at a random dword of the lane's own ring row
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(row + (r & 0x7cu)));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += v;
        } else if constexpr (MODE == 5) { // b64 records, Zipf-like: low symbols far more likely (r*r >> ..)
            uint64_t v;
            const uint32_t u = (r & 0xffffu) * (r & 0xffffu) >> 24; // 0..255, quadratic skew
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(u * 8u + 16384u));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += (uint32_t)v;
        } else if constexpr (MODE == 6) { // address arithmetic only (what the loop costs without the LDS op)
            acc += r & 16383u;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + a;
}

static float time_launch(fn k, uint32_t *d_out, int blocks, int threads, size_t lds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d_out, 12345u);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d_out, 12345u);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best)
            best = ms;
    }
    if (hipGetLastError() != hipSuccess)
        return -1.0f;
    return best;
}

int main()
{
    uint32_t *d_out;
    if (hipMalloc(&d_out, 4096 * 1024 * 4) != hipSuccess)
        return 1;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs, clock %d kHz; VALU: ns per asm body per SIMD (8 waves/SIMD), marginal (8w - 4w), cycles at 2.4 GHz\n", cus,
           prop.clockRate);
    struct Case { const char *name; fn k; };
#define C(n) {#n, n}
    Case cases[] = {C(k_nop_ref), C(k_lshr64_i), C(k_lshl64_i), C(k_mad64), C(k_lshladd64), C(k_cmp64), C(k_cmp32),
                    C(k_cmp_cnd1), C(k_cmp_cnd3), C(k_cmp_cnd2), C(k_cmp_cnd2i), C(k_cmp_cnd2s), C(k_cmpx_3mov), C(k_cmpx64_3mov), C(k_cmp_addc), C(k_alignbit), C(k_mullo), C(k_mad24), C(k_mov64),
                    C(k_step_r64)};
    for (auto &c : cases) {
        const float ms8 = time_launch(c.k, d_out, cus * 4, 512, 0);
        const float ms4 = time_launch(c.k, d_out, cus * 2, 512, 0);
        const double bodies8 = (double)ITERS * UNROLL * 8;
        const double ns = ms8 * 1e6 / bodies8;
        const double ns_marg = (ms8 - ms4) * 1e6 / (bodies8 / 2);
        printf("%-14s 8w %7.3f ms 4w %7.3f ms | %.3f ns/body = %.2f cyc@2.4 | marginal %.3f ns = %.2f cyc@2.4\n", c.name, ms8,
               ms4, ns, ns * 2.4, ns_marg, ns_marg * 2.4);
        fflush(stdout);
    }
    printf("LDS pipe, one 1024-thread block per CU (16 waves): ns and cycles@2.4 per wave-level LDS instruction per CU\n");
    struct LCase { const char *name; fn k; };
    LCase lc[] = {{"u8 random 16K", k_lds<0>}, {"b32 random 64K", k_lds<1>}, {"b64 rec uniform", k_lds<2>},
                  {"read2_b32 ring", k_lds<3>}, {"b32 ring", k_lds<4>}, {"b64 rec skewed", k_lds<5>}, {"no LDS op", k_lds<6>}};
    for (auto &c : lc) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(c.k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const float ms = time_launch(c.k, d_out, cus, 1024, 160 * 1024);
        const double ops = (double)ITERS * 4 * 16; // wave-level instructions per CU
        const double ns = ms * 1e6 / ops;
        printf("%-16s %7.3f ms | %.3f ns = %.2f cyc@2.4 per wave instruction per CU\n", c.name, ms, ns, ns * 2.4);
        fflush(stdout);
    }
    return 0;
}
