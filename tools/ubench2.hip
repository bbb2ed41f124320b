// tools/ubench2.hip -- second instruction-cost census for gfx950 (round 2): which VALU forms issue at the
// fast rate, what SALU instructions cost a wave that is VALU-bound, v_cmpx + exec restore, v_cndmask with a
// properly written vcc.  Same method as ubench.hip: 16 independent chains, 8 and 4 waves per SIMD on every
// CU, cost = time * clock / instructions per SIMD (the marginal figure is the 8-wave minus the 4-wave run).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o build/ubench2 && timeout 200 build/ubench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;

// NV = VALU instructions per asm body (for the per-VALU cost), body may hold SALU work as well
#define KERNEL(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(512) NAME(uint32_t *out, uint32_t seed)                      \
    {                                                                                              \
        uint32_t a[UNROLL];                                                                        \
        const uint32_t m = seed | 0x00ff00ffu, c = seed + threadIdx.x;                             \
        uint32_t sa = __builtin_amdgcn_readfirstlane(seed * 3u + 1u);                              \
        uint32_t sb = __builtin_amdgcn_readfirstlane(seed * 7u + 5u);                              \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) a[i] = threadIdx.x * 2654435761u + i;   \
        for (int it = 0; it < ITERS; ++it) {                                                       \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                                     \
                asm volatile(ASM : "+v"(a[i]), "+s"(sa), "+s"(sb) : "v"(m), "v"(c) : "vcc", "scc"); \
        }                                                                                          \
        uint32_t s = sa ^ sb;                                                                      \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s ^= a[i];                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }
// operands: %0 chain VGPR, %1 %2 SGPR state, %3 %4 constant VGPRs

KERNEL(k_and, "v_and_b32 %0, %3, %0")
KERNEL(k_or, "v_or_b32 %0, %3, %0")
KERNEL(k_xor, "v_xor_b32 %0, %3, %0")
KERNEL(k_add, "v_add_u32 %0, %3, %0")
KERNEL(k_sub, "v_sub_u32 %0, %0, %3")
KERNEL(k_subrev, "v_subrev_u32 %0, %3, %0")
KERNEL(k_max, "v_max_u32 %0, %3, %0")
KERNEL(k_min, "v_min_u32 %0, %3, %0")
KERNEL(k_mov, "v_mov_b32 %0, %3")
KERNEL(k_not, "v_not_b32 %0, %0")
KERNEL(k_lshr_i, "v_lshrrev_b32 %0, 12, %0")
KERNEL(k_lshr_v, "v_lshrrev_b32 %0, %4, %0")
KERNEL(k_lshl_i, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_lshl_v, "v_lshlrev_b32 %0, %4, %0")
KERNEL(k_ashr_i, "v_ashrrev_i32 %0, 3, %0")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %3, %0")
KERNEL(k_cmp, "v_cmp_gt_u32 vcc, %3, %0")
KERNEL(k_cmp_cnd, "v_cmp_gt_u32 vcc, %3, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %4, vcc")
KERNEL(k_cmpx_restore, "v_cmpx_gt_u32 vcc, %3, %0\n\ts_mov_b64 exec, -1")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %3, %0")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %3, %4")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %3")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %3")
KERNEL(k_add3, "v_add3_u32 %0, %0, %3, %4")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %3, %4")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 16, %3")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %3")
KERNEL(k_bfi, "v_bfi_b32 %0, %3, %0, %4")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 12")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %3, 12")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %3")
KERNEL(k_perm, "v_perm_b32 %0, %0, %3, %4")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_fma, "v_fma_f32 %0, %0, %3, %4")
KERNEL(k_fmac, "v_fmac_f32 %0, %3, %4")
KERNEL(k_addf, "v_add_f32 %0, %3, %0")
KERNEL(k_mulf, "v_mul_f32 %0, %3, %0")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %3")
KERNEL(k_pklshr, "v_pk_lshrrev_b16 %0, 4, %0")
KERNEL(k_pkmad, "v_pk_mad_u16 %0, %0, %3, %4")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_and_dpp, "v_and_b32_dpp %0, %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_sdwa_lshl, "v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
// SALU beside VALU
KERNEL(k_and_1s, "v_and_b32 %0, %3, %0\n\ts_add_u32 %1, %1, 1")
KERNEL(k_and_2s, "v_and_b32 %0, %3, %0\n\ts_add_u32 %1, %1, 1\n\ts_lshl_b32 %2, %1, 1")
KERNEL(k_and_4s, "v_and_b32 %0, %3, %0\n\ts_add_u32 %1, %1, 1\n\ts_lshl_b32 %2, %1, 1\n\ts_and_b32 %2, %2, 7\n\ts_add_u32 %1, %1, %2")
KERNEL(k_mad_1s, "v_mad_u32_u24 %0, %0, %3, %4\n\ts_add_u32 %1, %1, 1")
KERNEL(k_mad_2s, "v_mad_u32_u24 %0, %0, %3, %4\n\ts_add_u32 %1, %1, 1\n\ts_lshl_b32 %2, %1, 1")
KERNEL(k_mad_4s, "v_mad_u32_u24 %0, %0, %3, %4\n\ts_add_u32 %1, %1, 1\n\ts_lshl_b32 %2, %1, 1\n\ts_and_b32 %2, %2, 7\n\ts_add_u32 %1, %1, %2")
KERNEL(k_mad_nop, "v_mad_u32_u24 %0, %0, %3, %4\n\ts_nop 0")
KERNEL(k_mad_2nop, "v_mad_u32_u24 %0, %0, %3, %4\n\ts_nop 0\n\ts_nop 0")
KERNEL(k_salu_only, "s_add_u32 %1, %1, 1\n\ts_lshl_b32 %2, %1, 1")
// mixes of fast and slow forms
KERNEL(k_and_mad, "v_and_b32 %0, %3, %0\n\tv_mad_u32_u24 %0, %0, %3, %4")
KERNEL(k_and_and_mad, "v_and_b32 %0, %3, %0\n\tv_add_u32 %0, %4, %0\n\tv_mad_u32_u24 %0, %0, %3, %4")

typedef void (*valu_fn)(uint32_t *, uint32_t);
static valu_fn cur;

static float time_kernel(uint32_t *d_out, int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(cur, dim3(blocks), dim3(512), 0, 0, d_out, 12345u);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(cur, dim3(blocks), dim3(512), 0, 0, d_out, 12345u);
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
    if (hipMalloc(&d_out, 2048 * 512 * 4) != hipSuccess)
        return 1;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs; ns per asm body per SIMD (8 waves/SIMD), its marginal value (8w - 4w), and cycles at 2.4 GHz\n", cus);
    struct Case { const char *name; valu_fn k; };
#define C(n) {#n, n}
    Case cases[] = {C(k_and), C(k_or), C(k_xor), C(k_add), C(k_sub), C(k_subrev), C(k_max), C(k_min), C(k_mov), C(k_not),
                    C(k_lshr_i), C(k_lshr_v), C(k_lshl_i), C(k_lshl_v), C(k_ashr_i), C(k_addco), C(k_cmp), C(k_cmp_cnd),
                    C(k_cmpx_restore), C(k_mul24), C(k_mad24), C(k_mullo), C(k_mulhi), C(k_add3), C(k_andor), C(k_lshlor),
                    C(k_lshladd), C(k_bfi), C(k_bfe), C(k_alignbit), C(k_bcnt), C(k_perm), C(k_mbcnt), C(k_fma), C(k_fmac),
                    C(k_addf), C(k_mulf), C(k_cvt), C(k_pkadd), C(k_pklshr), C(k_pkmad), C(k_dpp), C(k_and_dpp),
                    C(k_sdwa_lshl), C(k_and_1s), C(k_and_2s), C(k_and_4s), C(k_mad_1s), C(k_mad_2s), C(k_mad_4s),
                    C(k_mad_nop), C(k_mad_2nop), C(k_salu_only), C(k_and_mad), C(k_and_and_mad)};
    for (auto &c : cases) {
        cur = c.k;
        const float ms8 = time_kernel(d_out, cus * 4);
        const float ms4 = time_kernel(d_out, cus * 2);
        const double bodies8 = (double)ITERS * UNROLL * 8; // asm bodies per SIMD at 8 waves/SIMD
        const double ns = ms8 * 1e6 / bodies8;
        const double ns_marg = (ms8 - ms4) * 1e6 / (bodies8 / 2);
        printf("%-18s 8w %7.3f ms 4w %7.3f ms | %.3f ns/body = %.2f cyc@2.4 | marginal %.3f ns = %.2f cyc@2.4\n", c.name, ms8,
               ms4, ns, ns * 2.4, ns_marg, ns_marg * 2.4);
        fflush(stdout);
    }
    return 0;
}
