// launchers.hpp -- entry points of the kernel translation units (internal; the public launchers of
// kernels.h are assembled from these in dispatch.cpp).
#pragma once

#include "kernels.h"

// Launch and report THIS launch's status: hipGetLastError() is per thread and sticky, so an error left
// behind by other code in the process (PyTorch probing a device, say) would otherwise be blamed on us.
#define RANS_LAUNCH(kern, grid, block, lds, stream, ...)                         \
    do {                                                                         \
        (void)hipGetLastError();                                                 \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);         \
    } while (0)

#include <atomic>

namespace rans_amd {

// Raise a kernel's dynamic-LDS limit once per (kernel, device): function attributes are per device, and
// one process may hold contexts on several GPUs.  `done` is a per-kernel bit set indexed by device.
inline hipError_t allow_large_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit)
        return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess)
        done.fetch_or(bit, std::memory_order_release);
    return e;
}

// wave-per-chunk kernels (N-way streams with N = 64 K lanes): decode_wave.hip, encode_wave.hip
hipError_t launch_decode_wave(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **name);
hipError_t launch_encode_wave(int format, const EncParams &p, int num_cus, hipStream_t stream, const char **name);

// two chunks per wave, 64-way, byte-stream formats (decode_dual.hip); format: kKernelFormatAlias2[W] or RANS_AMD_FMT_BYTE
hipError_t launch_decode_dual(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **name);

// eight chunks per wave, the reference's 8-way word layout over u8 symbols (decode_groups.hip): chunks of a multiple of 4
// symbols on a 4-byte aligned output, a partial last octet and a ragged last chunk included
bool decode_word_groups_applicable(const DecParams &p);
hipError_t launch_decode_word_groups(const DecParams &p, int num_cus, hipStream_t stream, const char **name);
// 32 chunks per wave, the byte format's 2-way layout over u8 symbols (cum2sym tables), same file
bool decode_byte_pairs_applicable(const DecParams &p);
hipError_t launch_decode_byte_pairs(const DecParams &p, int num_cus, hipStream_t stream, const char **name);

// the 8-way word layout's encoder, eight chunks per wave (encode_groups.hip): chunks of a multiple of 4 u8 symbols, the
// three-kernel placement, the slot layout or sized slots (no fused placement)
bool encode_word_groups_applicable(const EncParams &p);
hipError_t launch_encode_word_groups(const EncParams &p, int num_cus, hipStream_t stream, const char **name);

// lane-per-stream kernels (N = 1, 2, 4, 8 with at least kLaneKernelMinChunks chunks): lanes.hip
constexpr uint64_t kLaneKernelMinChunks = 64;
inline bool lanes_applicable(uint64_t nchunks, uint32_t n_ways)
{
    return nchunks >= kLaneKernelMinChunks && (n_ways == 1 || n_ways == 2 || n_ways == 4 || n_ways == 8);
}
hipError_t launch_decode_lanes(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **name);
hipError_t launch_encode_lanes(int format, const EncParams &p, int num_cus, hipStream_t stream, const char **name);
// true when launch_encode_lanes would take the staged kernel for this request -- the one that can place its chunks
// itself (EncParams::status); everything but status / offsets / out / out_cap must be filled in
bool encode_lanes_fused(const EncParams &p, int num_cus);
// true when launch_encode_lanes would take a staged kernel for this SIZED-slot request (EncParams::ovf_ctl): lanes.hip
bool encode_lanes_sized_ok(int format, const EncParams &p, int num_cus);

} // namespace rans_amd
