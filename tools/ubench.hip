// tools/ubench.hip -- VALU issue-cost microbenchmark for the instructions the rANS decode
// round is made of (gfx950).  Each kernel runs 16 independent chains of one instruction
// form at 8 waves/SIMD on every CU; cost = wall time * clock / instructions per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o build/ubench && timeout 100 build/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;

#define VALU_KERNEL(NAME, ASM)                                                                     \
    __global__ void __launch_bounds__(512) NAME(uint32_t *out, uint32_t seed)                      \
    {                                                                                              \
        uint32_t a[UNROLL];                                                                        \
        const uint32_t m = seed | 0x00ff00ffu, c = seed + threadIdx.x;                             \
        const uint32_t sm = __builtin_amdgcn_readfirstlane(seed * 3u + 1u);                        \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) a[i] = threadIdx.x * 2654435761u + i;   \
        for (int it = 0; it < ITERS; ++it) {                                                       \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                                     \
                asm volatile(ASM : "+v"(a[i]) : "v"(m), "v"(c), "s"(sm) : "vcc", "s20", "s21", "s22");                  \
        }                                                                                          \
        uint32_t s = 0;                                                                            \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s ^= a[i];                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }

VALU_KERNEL(k01, "v_and_b32 %0, %1, %0")
VALU_KERNEL(k02, "v_and_b32 %0, 0xfff0fff, %0")
VALU_KERNEL(k03, "v_and_b32 %0, %3, %0")
VALU_KERNEL(k04, "v_lshlrev_b32 %0, 3, %0")
VALU_KERNEL(k05, "v_lshrrev_b32 %0, 12, %0")
VALU_KERNEL(k06, "v_lshl_add_u32 %0, %0, 3, %2")
VALU_KERNEL(k07, "v_lshl_add_u32 %0, %0, 3, %3")
VALU_KERNEL(k08, "v_lshl_add_u32 %0, %0, 3, 0")
VALU_KERNEL(k09, "v_mad_u32_u24 %0, %0, %1, %2")
VALU_KERNEL(k10, "v_mad_u32_u24 %0, %0, %3, %2")
VALU_KERNEL(k11, "v_mul_u32_u24 %0, %1, %0")
VALU_KERNEL(k12, "v_perm_b32 %0, %0, %1, %2")
VALU_KERNEL(k13, "v_perm_b32 %0, %0, %1, %3")
VALU_KERNEL(k14, "v_mbcnt_lo_u32_b32 %0, %3, %0")
VALU_KERNEL(k15, "v_mbcnt_lo_u32_b32 %0, %3, 0")
VALU_KERNEL(k16, "v_bfe_u32 %0, %0, 3, 12")
VALU_KERNEL(k17, "v_cmp_gt_u32 vcc, %3, %0")
VALU_KERNEL(k18, "v_cmp_gt_u32 vcc, 0x10000, %0")
VALU_KERNEL(k19, "v_lshl_or_b32 %0, %0, 16, %1")
VALU_KERNEL(k20, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
VALU_KERNEL(k21, "v_add_u32 %0, %1, %0")
VALU_KERNEL(k22, "v_cndmask_b32 %0, %0, %1, vcc")
VALU_KERNEL(k23, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3")
VALU_KERNEL(k24, "v_and_b32 %0, %1, %0\n\ts_add_u32 s20, s20, 1")
VALU_KERNEL(k25, "v_and_b32 %0, %1, %0\n\ts_add_u32 s20, s20, 1\n\ts_lshl_b32 s21, s20, 1\n\ts_and_b32 s22, s21, 7")
VALU_KERNEL(k26, "v_mul_lo_u32 %0, %0, %1")
VALU_KERNEL(k27, "v_mul_hi_u32 %0, %0, %1")
VALU_KERNEL(k28, "v_fma_f32 %0, %0, %1, %2")
VALU_KERNEL(k29, "v_add3_u32 %0, %0, %1, %2")
VALU_KERNEL(k30, "v_and_or_b32 %0, %0, %1, %2")

// random LDS gathers out of a 32 KiB table
template <int BYTES, int UNIFORM> __global__ void __launch_bounds__(512) k_lds(uint32_t *out, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x)
        tab[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        x[i] = (threadIdx.x * 40503u + i * 977u + seed) * 2246822519u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t slot = UNIFORM ? ((x[i] >> 7) & 0xfc0u) + (threadIdx.x & 63u) : ((x[i] >> 7) & 0xfffu);
            if (BYTES == 8) {
                uint2 e = reinterpret_cast<const uint2 *>(tab)[slot];
                x[i] = x[i] * 5u + e.x + e.y;
            } else {
                uint32_t e = tab[slot];
                x[i] = x[i] * 5u + e;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
}

typedef void (*valu_fn)(uint32_t *, uint32_t);

static float time_kernel(void (*launch)(uint32_t *, int), uint32_t *d_out, int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch(d_out, blocks); // warm-up
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(d_out, blocks);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    uint32_t *d_out;
    if (hipMalloc(&d_out, 2048 * 512 * 4) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs; 8 waves/SIMD; cost in cycles per wave-instruction per SIMD at 2.4 / 2.1 GHz\n", cus);
    struct Case { const char *name; valu_fn k; int extra; };
    Case cases[] = {
        {"v_and_b32 v,v,v (e32)", k01, 0}, {"v_and_b32 v,lit,v", k02, 0}, {"v_and_b32 v,s,v", k03, 0},
        {"v_lshlrev_b32 v,3,v", k04, 0}, {"v_lshrrev_b32 v,12,v", k05, 0},
        {"v_lshl_add_u32 v,v,3,v", k06, 0}, {"v_lshl_add_u32 v,v,3,s", k07, 0}, {"v_lshl_add_u32 v,v,3,0", k08, 0},
        {"v_mad_u32_u24 v,v,v,v", k09, 0}, {"v_mad_u32_u24 v,v,s,v", k10, 0}, {"v_mul_u32_u24 v,v,v (e32)", k11, 0},
        {"v_perm_b32 v,v,v,v", k12, 0}, {"v_perm_b32 v,v,v,s", k13, 0},
        {"v_mbcnt_lo v,s,v", k14, 0}, {"v_mbcnt_lo v,s,0", k15, 0}, {"v_bfe_u32 v,v,3,12", k16, 0},
        {"v_cmp_gt_u32 vcc,s,v", k17, 0}, {"v_cmp_gt_u32 vcc,lit,v", k18, 0}, {"v_lshl_or_b32 v,v,16,v", k19, 0},
        {"v_mov_b32_dpp quad_perm", k20, 0}, {"v_add_u32 v,v,v", k21, 0}, {"v_cndmask_b32 v,v,v,vcc", k22, 0},
        {"v_mov_b32_sdwa byte", k23, 0}, {"v_and + 1 salu", k24, 0}, {"v_and + 3 salu", k25, 0},
        {"v_mul_lo_u32", k26, 0}, {"v_mul_hi_u32", k27, 0}, {"v_fma_f32", k28, 0}, {"v_add3_u32 v,v,v,v", k29, 0},
        {"v_and_or_b32 v,v,v,v", k30, 0},
    };
    static valu_fn cur;
    for (auto &c : cases) {
        cur = c.k;
        auto launch = [](uint32_t *o, int blocks) { hipLaunchKernelGGL(cur, dim3(blocks), dim3(512), 0, 0, o, 12345u); };
        // baseline overhead: the same launch with 2 waves/SIMD is latency-bound per wave, so use
        // two occupancies and take the difference: 8 waves/SIMD (4 blocks/CU) minus 4 waves/SIMD (2 blocks/CU)
        float ms8 = time_kernel(launch, d_out, cus * 4);
        float ms4 = time_kernel(launch, d_out, cus * 2);
        double instr_per_simd_8 = (double)ITERS * UNROLL * 8;
        double ns_per_instr = (ms8 * 1e6) / instr_per_simd_8;
        double ns_delta = ((ms8 - ms4) * 1e6) / (instr_per_simd_8 / 2);
        printf("%-28s 8w: %7.3f ms  4w: %7.3f ms | %.2f cyc@2.4 (%.2f @2.1) | marginal %.2f cyc@2.4\n", c.name, ms8, ms4,
               ns_per_instr * 2.4, ns_per_instr * 2.1, ns_delta * 2.4);
        fflush(stdout);
    }
    {
        struct L { const char *name; void (*k)(uint32_t *, uint32_t); } ls[] = {
            {"ds_read_b32 random", k_lds<4, 0>}, {"ds_read_b32 conflict-free", k_lds<4, 1>},
            {"ds_read_b64 random", k_lds<8, 0>}, {"ds_read_b64 conflict-free", k_lds<8, 1>}};
        for (auto &l : ls) {
            cur = l.k;
            auto launch = [](uint32_t *o, int blocks) { hipLaunchKernelGGL(cur, dim3(blocks), dim3(512), 0, 0, o, 777u); };
            float ms = time_kernel(launch, d_out, cus * 4);
            double gathers_per_cu = (double)ITERS * 4 * 32; // 32 waves per CU
            printf("%-28s 32 waves/CU: %7.3f ms -> %.2f cyc@2.4 per wave-gather per CU (%.2f @2.1)\n", l.name, ms,
                   ms * 1e6 / gathers_per_cu * 2.4, ms * 1e6 / gathers_per_cu * 2.1);
            fflush(stdout);
        }
    }
    return 0;
}
