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
