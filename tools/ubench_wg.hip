// tools/ubench_wg.hip -- how many ONE-WAVE workgroups a CU of gfx950 keeps resident, by LDS bytes per workgroup
// (round 6: the fused per-chunk-model encoder runs one wave per workgroup so that its record table sits at LDS address 0).
// Method: G workgroups of 64 threads that each wait T microseconds of wall time; a launch takes ceil(G / resident) * T.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_wg.hip -o build/ubench_wg && timeout 120 build/ubench_wg
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void __launch_bounds__(64) k_wait(uint32_t *out, unsigned long long ticks)
{
    extern __shared__ uint8_t smem[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0)
        smem[0] = 1;
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0)
        out[blockIdx.x] = smem[0];
}

__global__ void __launch_bounds__(256) k_wait256(uint32_t *out, unsigned long long ticks)
{
    extern __shared__ uint8_t smem[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0)
        smem[0] = 1;
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0)
        out[blockIdx.x] = smem[0];
}

// where a wave runs: HW_REG_HW_ID (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) and HW_REG_XCC_ID (3:0)
__global__ void __launch_bounds__(256) k_where(uint32_t *out, unsigned long long ticks)
{
    extern __shared__ uint8_t smem[];
    const unsigned long long t0 = wall_clock64();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0)
        smem[0] = 1;
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (hw & 0xffffu) | ((xcc & 0xfu) << 16) | (smem[0] ? 0u : 1u << 31);
}

static void where(uint32_t *out, int cus, int threads, int per_cu, int lds)
{
    const int waves = threads / 64, g = per_cu * cus;
    hipMemset(out, 0xff, 1 << 22);
    hipLaunchKernelGGL(k_where, dim3(g), dim3(threads), lds, 0, out, 20000ull);
    hipDeviceSynchronize();
    static uint32_t host[1 << 20];
    hipMemcpy(host, out, (size_t)g * waves * 4, hipMemcpyDeviceToHost);
    // waves per SIMD: key = xcc, se, sh, cu, simd
    static int count[1 << 20];
    for (int i = 0; i < (1 << 20); ++i) count[i] = 0;
    for (int i = 0; i < g * waves; ++i) {
        const uint32_t v = host[i];
        const uint32_t key = ((v >> 16) & 0xf) << 12 | ((v >> 13) & 7) << 9 | ((v >> 12) & 1) << 8 | ((v >> 8) & 0xf) << 4 | ((v >> 4) & 3);
        count[key]++;
    }
    int hist[64] = {0}, simds = 0;
    for (int i = 0; i < (1 << 20); ++i)
        if (count[i]) { hist[count[i] < 63 ? count[i] : 63]++; simds++; }
    printf("%4d-thread workgroups, %2d per CU, lds %6d: %d SIMDs saw waves; SIMDs by waves hosted:", threads, per_cu, lds, simds);
    for (int i = 1; i < 64; ++i)
        if (hist[i]) printf("  %d waves x %d", i, hist[i]);
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, 1 << 22);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_wait), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_wait256), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned long long ticks = 20000; // 200 us at 100 MHz
    printf("CUs %d; a launch of (per-CU x CUs) one-wave workgroups, 200 us each: ms (0.2 = all resident at once)\n", cus);
    const int lds_list[] = {0, 4096, 5120, 6144, 6400, 7168, 8192, 10240, 16384};
    for (int lds : lds_list) {
        printf("lds %6d:", lds);
        for (int per : {8, 16, 17, 20, 24, 25, 26, 28, 32, 33, 40}) {
            const int g = per * cus;
            hipLaunchKernelGGL(k_wait, dim3(g), dim3(64), lds, 0, out, ticks);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k_wait, dim3(g), dim3(64), lds, 0, out, ticks);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("  %d:%.2f", per, ms);
        }
        printf("\n");
    }
    // (launches that fit the chip at once: every wave of the launch is resident while it reports)
    where(out, cus, 64, 24, 6144);
    where(out, cus, 64, 20, 7936);
    where(out, cus, 64, 16, 7936);
    where(out, cus, 128, 10, 2 * 7936);
    where(out, cus, 256, 5, 4 * 7936);
    where(out, cus, 256, 6, 4 * 6144);
    where(out, cus, 512, 4, 0);
    printf("256-thread workgroups (4 waves):\n");
    for (int lds : {0, 16384, 20480, 24576}) {
        printf("lds %6d:", lds);
        for (int per : {2, 4, 5, 6, 7, 8, 9, 10}) {
            const int g = per * cus;
            hipLaunchKernelGGL(k_wait256, dim3(g), dim3(256), lds, 0, out, ticks);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k_wait256, dim3(g), dim3(256), lds, 0, out, ticks);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("  %d:%.2f", per, ms);
        }
        printf("\n");
    }
    return 0;
}
