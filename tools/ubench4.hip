// tools/ubench4.hip -- fourth issue-cost census for gfx950: the SDWA forms and three-operand integer forms the round-4
// encoder sub-steps are made of (v_lshlrev/v_add/v_sub/v_cmpx with SDWA selects, v_mad_i32_i24, v_add3, v_lshl_or, v_lshl_add,
// v_bitop3, v_pk_max_u16, v_mul_hi_u32, v_perm, v_mbcnt).  Method as ubench2/3: 8 and 4 waves per SIMD, marginal = (8w - 4w).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench4.hip -o build/ubench4 && timeout 200 build/ubench4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITERS = 2048;
constexpr int UNROLL = 8;

// %0 %1 32-bit chains, %2 %3 constant VGPRs
#define KERNEL32(NAME, ASM)                                                                        \
    __global__ void __launch_bounds__(512) NAME(uint32_t *out, uint32_t seed)                      \
    {                                                                                              \
        uint32_t b[UNROLL], d[UNROLL];                                                             \
        const uint32_t m = seed | 0x00ff00ffu, c = ((seed + threadIdx.x) & 15u) | 1u;              \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                                         \
        {                                                                                          \
            b[i] = threadIdx.x * 2654435761u + i;                                                  \
            d[i] = threadIdx.x * 77u + i;                                                          \
        }                                                                                          \
        for (int it = 0; it < ITERS; ++it) {                                                       \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                                     \
                asm volatile(ASM : "+v"(b[i]), "+v"(d[i]) : "v"(m), "v"(c) : "vcc");               \
        }                                                                                          \
        uint32_t s = 0;                                                                            \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s ^= b[i] ^ d[i];                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }

KERNEL32(k_and_ref, "v_and_b32 %0, %2, %0")
KERNEL32(k_add, "v_add_u32 %0, %2, %0")
KERNEL32(k_lshrrev, "v_lshrrev_b32 %0, 1, %0")
KERNEL32(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_min, "v_min_u32 %0, %2, %0")
KERNEL32(k_lshl_sdwa, "v_lshlrev_b32_sdwa %0, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
KERNEL32(k_lshl_sdwa_b1, "v_lshlrev_b32_sdwa %0, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL32(k_add_sdwa, "v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL32(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
KERNEL32(k_lshr_sdwa_b3, "v_lshrrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD")
KERNEL32(k_cmp, "v_cmp_gt_u32 vcc, %0, %2\n\tv_add_u32 %0, %3, %0")
KERNEL32(k_cmp_sdwa, "v_cmp_gt_u32_sdwa vcc, %0, %2 src0_sel:WORD_1 src1_sel:WORD_0\n\tv_add_u32 %0, %3, %0")
KERNEL32(k_mad_i24, "v_mad_i32_i24 %0, %0, %3, %2")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %3, %2")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %2, 1")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, %3, %2")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %2")
KERNEL32(k_add_lshl, "v_add_lshl_u32 %0, %0, %2, 1")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %2, %3")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %2, %3 bitop3:0x80")
KERNEL32(k_pk_max, "v_pk_max_u16 %0, %0, %2")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %2")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %2")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %2, %3")
KERNEL32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 1, %0")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 12")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %2, 14")

typedef void (*fn)(uint32_t *, uint32_t);

static float time_launch(fn k, uint32_t *d_out, int blocks, int threads)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best)
            best = ms;
    }
    return hipGetLastError() == hipSuccess ? best : -1.0f;
}

int main()
{
    uint32_t *d_out;
    if (hipMalloc(&d_out, 4096 * 1024 * 4) != hipSuccess)
        return 1;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs; ns per asm body per SIMD at 8 waves/SIMD, and marginal (8w - 4w); relative to v_and_b32\n", cus);
    struct Case { const char *name; fn k; };
#define C(n) {#n, n}
    Case cases[] = {C(k_and_ref), C(k_add), C(k_lshrrev), C(k_lshlrev), C(k_min), C(k_lshl_sdwa), C(k_lshl_sdwa_b1), C(k_add_sdwa),
                    C(k_sub_sdwa), C(k_lshr_sdwa_b3), C(k_cmp), C(k_cmp_sdwa), C(k_mad_i24), C(k_mad_u24), C(k_add3), C(k_lshl_or),
                    C(k_lshl_add), C(k_add_lshl), C(k_and_or), C(k_bitop3), C(k_pk_max), C(k_mul_hi), C(k_mul_lo), C(k_perm),
                    C(k_mbcnt), C(k_ashr), C(k_mov_dpp), C(k_bfe), C(k_alignbit)};
    double ref = 0;
    for (auto &c : cases) {
        const float ms8 = time_launch(c.k, d_out, cus * 4, 512);
        const float ms4 = time_launch(c.k, d_out, cus * 2, 512);
        const double bodies8 = (double)ITERS * UNROLL * 8;
        const double ns = ms8 * 1e6 / bodies8;
        const double ns_marg = (ms8 - ms4) * 1e6 / (bodies8 / 2);
        if (ref == 0)
            ref = ns;
        printf("%-16s 8w %7.3f ms 4w %7.3f ms | %.3f ns/body (x%.2f) | marginal %.3f ns\n", c.name, ms8, ms4, ns, ns / ref, ns_marg);
        fflush(stdout);
    }
    return 0;
}
