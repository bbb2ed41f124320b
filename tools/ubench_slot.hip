// tools/ubench_slot.hip -- what the LDS charges for the slot-record gather of the word decoder (rans_word_sse41.h:64-72,
// 123-131: slot = x & 4095, one record per slot) under the layouts VERDICT r02 item 4 asks about:
//   b64        4096 x 8-byte records {freq | sym << 24, bias}, ds_read_b64           (what k_decode_word64 does)
//   b32        4096 x 4-byte packed records {sym:8, freq-1:12, bias:12}, ds_read_b32
//   b64 x2     two copies of the 8-byte table, the second 4 / 8 / 132 bytes further along its banks, lanes 32..63 read
//              the second copy ("bank-staggered copies selected by lane & 32")
//   b64 seq    the same instruction with consecutive slots per lane (no conflicts): the floor
// Slots are uniform pseudo-random per lane, which is what x & 4095 of an rANS state is.  16 waves per block, two blocks
// per CU like the decoder; ns and cycles per wave-level LDS instruction per CU.  Run under rocprofv3 --pmc
// SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE for the conflict share of each kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_slot.hip -o build/ubench_slot && build/ubench_slot
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITERS = 8192;

template <int MODE> __global__ void __launch_bounds__(1024) k_slot(uint32_t *out, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < 18432u; i += blockDim.x) // 72 KiB
        lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t a = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u + seed, acc = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t copy_off = MODE == 3 ? 32768u + 4u : MODE == 4 ? 32768u + 8u : MODE == 5 ? 32768u + 132u : 0u;
    const uint32_t base = (lane & 32u) ? copy_off : 0u;
    for (int it = 0; it < ITERS; ++it) {
        a = a * 1103515245u + 12345u;
        const uint32_t slot = (a >> 10) & 4095u;
        if constexpr (MODE == 0 || (MODE >= 3 && MODE <= 5)) { // b64 gather (MODE 3..5: second copy for the upper half wave)
            uint64_t v;
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(slot * 8u + base));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += (uint32_t)v;
        } else if constexpr (MODE == 1) { // b32 gather
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(slot * 4u));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += v;
        } else if constexpr (MODE == 2) { // b64, consecutive slots: conflict free
            uint64_t v;
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((((uint32_t)it * 64u + lane) & 4095u) * 8u));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += (uint32_t)v + slot;
        } else if constexpr (MODE == 6) { // the address arithmetic alone
            acc += slot;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + a;
}

typedef void (*fn)(uint32_t *, uint32_t);

static float time_launch(fn k, uint32_t *d_out, int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 72 * 1024, 0, d_out, 12345u);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 72 * 1024, 0, d_out, 12345u);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
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
    struct Case { const char *name; fn k; };
    Case cases[] = {{"b64 random (now)", k_slot<0>}, {"b32 packed random", k_slot<1>}, {"b64 sequential", k_slot<2>},
                    {"b64 x2 copies +4 B", k_slot<3>}, {"b64 x2 copies +8 B", k_slot<4>}, {"b64 x2 copies +132 B", k_slot<5>},
                    {"no LDS op", k_slot<6>}};
    printf("%d CUs; two 16-wave blocks per CU; per wave-level LDS instruction per CU\n", cus);
    for (auto &c : cases) {
        const float ms = time_launch(c.k, d_out, cus * 2);
        const double ops = (double)ITERS * 32; // wave-level instructions per CU
        const double ns = ms * 1e6 / ops;
        printf("%-22s %7.3f ms | %.3f ns = %.2f cycles at 2.4 GHz\n", c.name, ms, ns, ns * 2.4);
        fflush(stdout);
    }
    return 0;
}
