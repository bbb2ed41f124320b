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
    if (format == kKernelFormatAlias2 || format == kKernelFormatAlias2W) // (api.cpp has checked the shape)
        return launch_decode_dual(format, p, num_cus, stream, kernel_name);
    // the reference's 8-way word layout, u8 symbols, from eight full chunks on: eight chunks per wave (decode_groups.hip)
    if (format == (int)RANS_AMD_FMT_WORD && decode_word_groups_applicable(p))
        return launch_decode_word_groups(p, num_cus, stream, kernel_name);
    if (format == (int)RANS_AMD_FMT_BYTE && decode_byte_pairs_applicable(p))
        return launch_decode_byte_pairs(p, num_cus, stream, kernel_name);
    if (format != kKernelFormatR64Search && format != kKernelFormatWord16 && format != kKernelFormatByteAdaptive &&
        format != kKernelFormatWordAdaptive &&
        format != kKernelFormatByteFused && lanes_applicable(p.nchunks, p.n_ways))
        return launch_decode_lanes(format, p, num_cus, stream, kernel_name);
    return launch_decode_wave(format, p, num_cus, stream, kernel_name);
}

// Fused placement needs a mailbox for the block's coders and copier (encode_wave.hip launch_encode_t computes the same
// sizes).  It lives behind the tables in LDS where there is room; the alias tables of a 16-bit model over 4096 symbols
// fill the CU's 160 KiB to the last byte (config 4), and that kernel keeps its mailbox in global memory (round 3; a push
// and a pop per chunk, 175 us apart: L2 latency does not matter there).  16-byte records for more than 8159 symbols
// against the 128 KiB the other kernels may use cannot fuse at all (no model of that shape can be created today: its
// decoder tables do not fit either).
int encode_fused_fits(int format, uint32_t nsyms, uint32_t scale_bits)
{
    const size_t nrecs = nsyms < 256 ? 256 : nsyms;
    if (format == kKernelFormatAliasLds)
        return nrecs * 8 + ((size_t)2 << scale_bits) + 16 + kEncFusedLdsBytes <= 160 * 1024 ? 1 : 2;
    if (format == kKernelFormatByteAdaptive || format == kKernelFormatWordAdaptive) // (per-wave tables; never fused, see api.cpp)
        return 0;
    const bool word_recs = format == (int)RANS_AMD_FMT_WORD || format == (int)RANS_AMD_FMT_BYTE;
    size_t tables = nrecs * 16 + (word_recs ? 256 * 16 : 0);
    if (word_recs && nrecs == 256) // the staging windows of the word / byte encoder's waves
        tables += (size_t)(kEncFusedThreads / 64) * kEncStageBytes;
    return ((tables + 15) & ~(size_t)15) + kEncFusedLdsBytes <= 128 * 1024 ? 1 : 0;
}

// true when launch_encode hands this shape to the lane-per-chunk encoders (no fused placement there)
bool encode_uses_lanes(int format, uint64_t nchunks, uint32_t n_ways)
{
    return format != kKernelFormatByteAdaptive && format != kKernelFormatWordAdaptive && lanes_applicable(nchunks, n_ways) &&
           format != kKernelFormatR64Search &&
           format != kKernelFormatWord16;
}

// (the 8-way word layout's group encoder, encode_groups.hip, has no placement of its own: k_layout / k_compact_small behind it)
bool encode_lanes_can_fuse(int format, const EncParams &p, int num_cus)
{
    if (format == (int)RANS_AMD_FMT_WORD && encode_word_groups_applicable(p))
        return false;
    return encode_uses_lanes(format, p.nchunks, p.n_ways) && encode_lanes_fused(p, num_cus);
}

hipError_t launch_encode(int format, const EncParams &p, int num_cus, hipStream_t stream, const char **name)
{
    // (the lane encoders know the public formats: a narrow alias interleave gathers alias_remap from L2)
    if (format == kKernelFormatByteAdaptive) // one model per chunk: the wave encoder builds them, whatever the interleave
        return launch_encode_wave((int)RANS_AMD_FMT_BYTE, p, num_cus, stream, name);
    if (format == kKernelFormatWordAdaptive)
        return launch_encode_wave(kKernelFormatWordAdaptive, p, num_cus, stream, name);
    if (format == (int)RANS_AMD_FMT_WORD && encode_word_groups_applicable(p)) // the reference's 8-way word layout: eight chunks per wave
        return launch_encode_word_groups(p, num_cus, stream, name);
    if (!p.no_lanes && lanes_applicable(p.nchunks, p.n_ways) && format != kKernelFormatR64Search && format != kKernelFormatWord16)
        return launch_encode_lanes(format == kKernelFormatAliasLds ? (int)RANS_AMD_FMT_ALIAS : format, p, num_cus, stream, name);
    // (word format over u16 symbols: the wave encoder's general path, whatever the interleave)
    return launch_encode_wave(format == kKernelFormatWord16 ? (int)RANS_AMD_FMT_WORD : format, p, num_cus, stream, name);
}

} // namespace rans_amd
