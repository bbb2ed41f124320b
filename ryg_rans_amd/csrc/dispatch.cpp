// dispatch.cpp -- chooses the kernel family for a (format, n_ways, chunk count) request.
#include "launchers.hpp"

#include "../../include/ryg_rans_amd.h"

namespace rans_amd {

bool ways_supported(int format, uint32_t n_ways)
{
    if (format < 0 || format > 3) // (the public formats; the internal search variant is rans64 to the caller)
        return false;
    return n_ways >= 1 && n_ways <= 512;
}

// narrow interleaves with enough chunks to fill wavefronts run one chunk per lane; everything else
// one chunk per wave
hipError_t launch_decode(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **kernel_name)
{
    if (format != kKernelFormatR64Search && format != kKernelFormatWord16 && format != kKernelFormatByteAdaptive &&
        lanes_applicable(p.nchunks, p.n_ways))
        return launch_decode_lanes(format, p, num_cus, stream, kernel_name);
    return launch_decode_wave(format, p, num_cus, stream, kernel_name);
}

hipError_t launch_encode(int format, const EncParams &p, int num_cus, hipStream_t stream)
{
    // (the lane encoders know the public formats: a narrow alias interleave gathers alias_remap from L2)
    if (format == kKernelFormatByteAdaptive) // one model per chunk: the wave encoder builds them, whatever the interleave
        return launch_encode_wave((int)RANS_AMD_FMT_BYTE, p, num_cus, stream);
    if (lanes_applicable(p.nchunks, p.n_ways) && format != kKernelFormatR64Search && format != kKernelFormatWord16)
        return launch_encode_lanes(format == kKernelFormatAliasLds ? (int)RANS_AMD_FMT_ALIAS : format, p, num_cus, stream);
    // (word format over u16 symbols: the wave encoder's general path, whatever the interleave)
    return launch_encode_wave(format == kKernelFormatWord16 ? (int)RANS_AMD_FMT_WORD : format, p, num_cus, stream);
}

} // namespace rans_amd
