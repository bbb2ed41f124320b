// container_kernels.hip -- offset scan, compaction and histogram kernels around the coders.

#include "device_common.hpp"
#include "launchers.hpp"

namespace rans_amd {

namespace {

// ---------------------------------------------------------------------------
// Layout: offsets[c] = sum_{i<c} align16(lengths[i]); offsets[nchunks] = end of
// the last stream.  Every block scans kLayoutChunksPerBlock chunks; with more than one
// block (narrow interleaves produce 10^5..10^6 chunks) k_layout_sums first leaves every
// block's total in block_sums[] and k_layout adds the totals of the blocks before it.
// ---------------------------------------------------------------------------
constexpr int kLayoutPer = 8; // consecutive chunks per thread
constexpr uint32_t kLayoutChunksPerBlock = 1024 * kLayoutPer;

__device__ __forceinline__ uint64_t block_sum_1024(uint64_t v, uint64_t *wave_sum)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    if (lane_id() == 0)
        wave_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t t = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w)
        t += wave_sum[w];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(1024) k_layout_sums(const LayoutParams p)
{
    __shared__ uint64_t wave_sum[16];
    const uint64_t c0 = (uint64_t)blockIdx.x * kLayoutChunksPerBlock + (uint64_t)threadIdx.x * kLayoutPer;
    uint64_t mine = 0;
#pragma unroll
    for (int i = 0; i < kLayoutPer; ++i)
        mine += c0 + i < p.nchunks ? (((uint64_t)p.lengths[c0 + i] + 15u) & ~uint64_t(15)) : 0;
    const uint64_t total = block_sum_1024(mine, wave_sum);
    if (threadIdx.x == 0)
        p.block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_layout(const LayoutParams p)
{
    __shared__ uint64_t wave_sum[16];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    uint64_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += blockDim.x)
        part += p.block_sums[b];
    const uint64_t carry = blockIdx.x ? block_sum_1024(part, wave_sum) : 0;

    const uint64_t c0 = (uint64_t)blockIdx.x * kLayoutChunksPerBlock + (uint64_t)threadIdx.x * kLayoutPer;
    uint64_t sz[kLayoutPer];
    uint64_t mine = 0;
#pragma unroll
    for (int i = 0; i < kLayoutPer; ++i) {
        sz[i] = c0 + i < p.nchunks ? (((uint64_t)p.lengths[c0 + i] + 15u) & ~uint64_t(15)) : 0;
        mine += sz[i];
    }
    // inclusive scan of the per-thread sums inside the wave
    uint64_t v = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, 64);
        if ((int)lane >= d)
            v += o;
    }
    if (lane == 63)
        wave_sum[wave] = v;
    __syncthreads();
    uint64_t before = carry;
    for (uint32_t w = 0; w < wave; ++w)
        before += wave_sum[w];
    uint64_t at = before + v - mine;
#pragma unroll
    for (int i = 0; i < kLayoutPer; ++i) {
        const uint64_t c = c0 + i;
        if (c < p.nchunks) {
            p.offsets[c] = at;
            if (c == p.nchunks - 1) {
                p.offsets[p.nchunks] = at + p.lengths[c];
                if (at + sz[i] > p.out_cap)
                    atomicOr(p.flags, 2u);
            }
        }
        at += sz[i];
    }
    if (p.nchunks == 0 && threadIdx.x == 0)
        p.offsets[0] = 0;
}

// ---------------------------------------------------------------------------
// Compaction: chunk c's stream sits at the END of its scratch slot (arbitrary
// alignment); copy it to out + offsets[c] (16-byte aligned).  One wave per chunk, 16 bytes per
// lane and trip: two aligned 16-byte loads (the second is the next lane's first, an L1 hit), a
// funnel shift by the source misalignment -- wave-uniform, so the dword part of the shift is a
// scalar branch and the byte part four v_alignbyte_b32 -- and one 16-byte store.  The last store of a
// chunk may spill up to 15 bytes into the alignment padding behind it; the scratch buffer carries
// 64 bytes of slack for the loads past the last slot.
// ---------------------------------------------------------------------------
template <int DSH> __device__ __forceinline__ u32x4 funnel16(const u32x4 &a, const u32x4 &b, uint32_t bsh)
{
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32x4 r;
    r.x = __builtin_amdgcn_alignbyte(w[DSH + 1], w[DSH + 0], bsh);
    r.y = __builtin_amdgcn_alignbyte(w[DSH + 2], w[DSH + 1], bsh);
    r.z = __builtin_amdgcn_alignbyte(w[DSH + 3], w[DSH + 2], bsh);
    r.w = __builtin_amdgcn_alignbyte(w[DSH + 4], w[DSH + 3], bsh);
    return r;
}

__global__ void __launch_bounds__(256) k_compact(const CompactParams p)
{
    if (*p.flags & 2u)
        return;
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t chunk_v = (uint64_t)blockIdx.x * waves_per_block + wave; chunk_v < p.nchunks; chunk_v += total_waves) {
        const uint64_t chunk = uniform64(chunk_v);
        const uint32_t len = uniform(p.lengths[chunk]);
        const uint64_t from = p.src_offsets ? uniform64(p.src_offsets[chunk]) : (chunk + 1) * p.slot_bytes - len;
        if (p.src_offsets && (from > p.src_bytes || len > p.src_bytes - from)) { // (wave-uniform) the index points outside the source
            if (lane == 0)
                atomicOr(p.flags, 512u);
            continue;
        }
        const uint64_t sa = reinterpret_cast<uint64_t>(p.scratch) + from;
        gvec_cptr src = reinterpret_cast<gvec_cptr>(sa & ~uint64_t(15));
        u32x4 RANS_GLOBAL *dst = reinterpret_cast<u32x4 RANS_GLOBAL *>(reinterpret_cast<uint64_t>(p.out) + p.offsets[chunk]);
        const uint32_t dsh = uniform((uint32_t)(sa & 15u) >> 2), bsh = uniform((uint32_t)(sa & 3u));
        const uint32_t n16 = (len + 15u) >> 4;
        for (uint32_t i = lane; i < n16; i += 64u) {
            const u32x4 a = __builtin_nontemporal_load(src + i); // (below src_limit: the chunk lies inside the source, checked above)
            u32x4 b = {0u, 0u, 0u, 0u}; // (the granule behind the source's last one is not there to be read)
            if (reinterpret_cast<uint64_t>(src + i + 1) < p.src_limit)
                b = __builtin_nontemporal_load(src + i + 1);
            u32x4 r;
            switch (dsh) { // wave-uniform
            case 0: r = funnel16<0>(a, b, bsh); break;
            case 1: r = funnel16<1>(a, b, bsh); break;
            case 2: r = funnel16<2>(a, b, bsh); break;
            default: r = funnel16<3>(a, b, bsh); break;
            }
            dst[i] = r; // (a non-temporal store here measured 5 % slower on the 1 GiB word encode)
        }
    }
}

// Small chunks (narrow interleaves: a few hundred bytes each): 16 lanes per chunk, four chunks per
// wave trip, and the source read with UNALIGNED 16-byte global loads (gfx950 runs global memory in
// unaligned-access mode; the funnel shift above would need per-lane selects here).  A wave takes 64
// chunks at a time: lane l loads the length and the offset of chunk l once (two coalesced loads per 64
// chunks -- with every group of 16 lanes fetching its own chunk's pair they were a third of the kernel's
// vector-memory instructions, and the address path is what bounds it: TA 84 % busy), the groups pick
// theirs up by ds_bpermute.
// 16 bytes from any address; what lies at or beyond `limit` reads as zero (the last piece of the last chunk of a
// caller's buffer: the encoders' scratch has slack and passes no limit)
__device__ __forceinline__ u32x4 load16_below(uint64_t a, uint64_t limit)
{
    if (a + 16u <= limit || a + 16u < a)
        return *reinterpret_cast<gvec_cptr>(a);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 16; ++b)
        if (a + (uint32_t)b < limit)
            w[b >> 2] |= (uint32_t) * reinterpret_cast<const uint8_t RANS_GLOBAL *>(a + (uint32_t)b) << (8 * (b & 3));
    return u32x4{w[0], w[1], w[2], w[3]};
}

__global__ void __launch_bounds__(256) k_compact_small(const CompactParams p)
{
    if (*p.flags & 2u)
        return;
    const uint32_t lane = lane_id();
    const uint32_t sub = lane & 15u, grp = lane >> 4;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (uint64_t c0 = wave * 64u; c0 < p.nchunks; c0 += waves * 64u) {
        const uint64_t mine = c0 + lane;
        uint32_t len = 0;
        uint64_t off = 0, from = 0;
        if (mine < p.nchunks) {
            len = p.lengths[mine];
            off = p.offsets[mine];
            from = p.src_offsets ? p.src_offsets[mine] : (mine + 1u) * p.slot_bytes - len;
            if (p.src_offsets && (from > p.src_bytes || len > p.src_bytes - from)) { // the index points outside the source:
                atomicOr(p.flags, 512u);                                             // nothing of this chunk is read or written
                len = 0;
            }
        }
        // four chunks per group at a time, two pieces of each in flight (512 bytes of a chunk per pass): a wave's trip is
        // a chain of memory round trips, and with a load -> store pair per piece (the compiler may not move a load above
        // a store to memory it cannot tell apart) the kernel was bound by that chain, not by any unit: 130 us
        for (int k0 = 0; k0 < 16; k0 += 4) {
            uint64_t sa[4], da[4];
            uint32_t n16[4], most = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int src = 4 * (k0 + q) + (int)grp; // this group's chunk
                const uint32_t l = (uint32_t)__shfl((int)len, src, 64);
                const uint64_t o = (uint64_t)(uint32_t)__shfl((int)(uint32_t)off, src, 64) |
                                   ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(off >> 32), src, 64) << 32);
                sa[q] = reinterpret_cast<uint64_t>(p.scratch) + ((uint64_t)(uint32_t)__shfl((int)(uint32_t)from, src, 64) |
                                                                  ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(from >> 32), src, 64) << 32));
                da[q] = reinterpret_cast<uint64_t>(p.out) + o;
                n16[q] = (l + 15u) >> 4; // (0 behind the last chunk)
                most = n16[q] > most ? n16[q] : most;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t m2 = (uint32_t)__shfl_xor((int)most, d, 64);
                most = m2 > most ? m2 : most;
            }
            most = uniform(most);
            for (uint32_t i0 = 0; i0 < most; i0 += 32u) {
                u32x4 v[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t i = i0 + 16u * h + sub;
                        if (i < n16[q])
                            v[q][h] = load16_below(sa[q] + 16ull * i, p.src_limit);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t i = i0 + 16u * h + sub;
                        if (i < n16[q])
                            *reinterpret_cast<u32x4 RANS_GLOBAL *>(da[q] + 16ull * i) = v[q][h];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Histogram (count_freqs, main.cpp:59-66).  1 byte of HBM traffic per symbol, so the LDS
// atomic rate is what has to keep up: a skewed source (Zipf: the top symbol is 16 % of
// the input) sends ~10 lanes of every wave to the same counter, and same-bank atomics
// serialise.  u8 path: kHistCopies copies of the 256 counters, a lane uses copy
// (lane & (kHistCopies - 1)), and the copies start a few banks apart, so the lanes that hit
// one symbol spread over as many banks; counters of symbols >= nsyms are simply counted and
// flagged at the end (no per-symbol range check).  u16 path (alphabets up to 4096): one
// table per block.
// ---------------------------------------------------------------------------
// Round 4: the copies belong to the BLOCK, not to each of its waves (atomics from different waves never meet inside one
// LDS instruction, so private copies per wave bought nothing and cost LDS): 32 copies 2 banks apart, a lane uses copy
// (lane & 31), 33 KiB per block, twice as many blocks -- 1 GiB of Zipf(256) bytes 0.358 -> 0.281 ms per call, uniform bytes
// 0.303 -> 0.283, one repeated byte 0.495 -> 0.228 (tools/history/r4w_call.sh).
constexpr uint32_t kHistCopies = 32, kHistBlocksPerCu = 8;
constexpr uint32_t kHistCopyStride = 256 + 2; // dwords: copy c starts in bank 2 c

__global__ void __launch_bounds__(256) k_histogram_u8(const void *syms, uint64_t n, uint32_t nsyms, uint32_t *hist,
                                                      uint32_t *flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *all = reinterpret_cast<uint32_t *>(smem);
    const uint32_t waves = 1u; // (the copies belong to the block)
    const uint32_t total = waves * kHistCopies * kHistCopyStride;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x)
        all[i] = 0;
    __syncthreads();
    uint32_t *h = all + (threadIdx.x & (kHistCopies - 1)) * kHistCopyStride;

    const uint8_t *p = static_cast<const uint8_t *>(syms);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 16-byte loads from the first aligned address on; the ragged head and tail go bytewise
    const uint64_t head = (16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    const uint64_t nhead = head < n ? head : n;
    const uint64_t nvec = (n - nhead) / 16;
    gvec_cptr pv = reinterpret_cast<gvec_cptr>(reinterpret_cast<uintptr_t>(p + nhead));
    auto count4 = [&](uint32_t w) {
        atomicAdd(&h[w & 0xffu], 1u);
        atomicAdd(&h[(w >> 8) & 0xffu], 1u);
        atomicAdd(&h[(w >> 16) & 0xffu], 1u);
        atomicAdd(&h[w >> 24], 1u);
    };
    uint64_t i = tid;
    if (i < nvec) {
        u32x4 v = __builtin_nontemporal_load(pv + i);
        for (i += stride; i < nvec; i += stride) { // next load in flight while this one is counted
            const u32x4 nv = __builtin_nontemporal_load(pv + i);
            count4(v.x);
            count4(v.y);
            count4(v.z);
            count4(v.w);
            v = nv;
        }
        count4(v.x);
        count4(v.y);
        count4(v.z);
        count4(v.w);
    }
    for (uint64_t j = tid; j < nhead; j += stride)
        atomicAdd(&h[p[j]], 1u);
    for (uint64_t j = nhead + nvec * 16 + tid; j < n; j += stride)
        atomicAdd(&h[p[j]], 1u);
    __syncthreads();

    bool bad = false;
    for (uint32_t b = threadIdx.x; b < 256u; b += blockDim.x) {
        uint32_t sum = 0;
        for (uint32_t c = 0; c < waves * kHistCopies; ++c)
            sum += all[c * kHistCopyStride + b];
        if (sum) {
            if (b < nsyms)
                atomicAdd(&hist[b], sum);
            else
                bad = true;
        }
    }
    if (bad)
        atomicOr(flags, 1u);
}

// (`copies` tables per block, 8 banks apart, a lane uses table (lane & (copies - 1)): as for the byte symbols above, the lanes
//  that meet on a frequent symbol spread over several counters -- 4 copies of a 4096-symbol table are 64 KiB)
__global__ void __launch_bounds__(256) k_histogram_u16(const void *syms, uint64_t n, uint32_t nsyms, uint32_t *hist,
                                                       uint32_t *flags, uint32_t copies)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *all = reinterpret_cast<uint32_t *>(smem);
    const uint32_t cstride = nsyms + 8u;
    for (uint32_t i = threadIdx.x; i < copies * cstride; i += blockDim.x)
        all[i] = 0;
    __syncthreads();
    uint32_t *h = all + (threadIdx.x & (copies - 1u)) * cstride;
    bool bad = false;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint16_t *p = static_cast<const uint16_t *>(syms);
    auto count1 = [&](uint32_t sym) {
        if (sym < nsyms)
            atomicAdd(&h[sym], 1u);
        else
            bad = true;
    };
    // 16-byte loads (8 symbols) from the first aligned address on
    const uint64_t head = ((16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) / 2;
    const uint64_t nhead = head < n ? head : n;
    const uint64_t nvec = (n - nhead) / 8;
    gvec_cptr pv = reinterpret_cast<gvec_cptr>(reinterpret_cast<uintptr_t>(p + nhead));
    for (uint64_t i = tid; i < nvec; i += stride) {
        const u32x4 v = __builtin_nontemporal_load(pv + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            count1(w[a] & 0xffffu);
            count1(w[a] >> 16);
        }
    }
    for (uint64_t j = tid; j < nhead; j += stride)
        count1(p[j]);
    for (uint64_t j = nhead + nvec * 8 + tid; j < n; j += stride)
        count1(p[j]);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nsyms; i += blockDim.x) {
        uint32_t sum = 0;
        for (uint32_t c = 0; c < copies; ++c)
            sum += all[c * cstride + i];
        if (sum)
            atomicAdd(&hist[i], sum);
    }
    if (bad)
        atomicOr(flags, 1u);
}


} // namespace

uint32_t layout_blocks(uint64_t nchunks)
{
    const uint64_t b = (nchunks + kLayoutChunksPerBlock - 1) / kLayoutChunksPerBlock;
    return (uint32_t)(b ? b : 1);
}

hipError_t launch_layout(const LayoutParams &p, hipStream_t stream)
{
    const uint32_t blocks = layout_blocks(p.nchunks);
    if (blocks > 1) {
        if (!p.block_sums)
            return hipErrorInvalidValue;
        RANS_LAUNCH(k_layout_sums, dim3(blocks), dim3(1024), 0, stream, p);
    }
    RANS_LAUNCH(k_layout, dim3(blocks), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

// kernels.h ZeroParams: every region's unaligned head and tail by words, the body in 16-byte stores
__global__ __launch_bounds__(256) void k_zero(const ZeroParams p)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        uint8_t *b = static_cast<uint8_t *>(p.ptr[r]);
        const uint64_t n = p.bytes[r];
        if (n == 0)
            continue;
        uint64_t head = (16u - (reinterpret_cast<uintptr_t>(b) & 15u)) & 15u;
        head = head < n ? head : n;
        const uint64_t body = (n - head) & ~uint64_t(15);
        for (uint64_t i = tid * 4; i < head; i += nth * 4)
            *reinterpret_cast<uint32_t *>(b + i) = 0u;
        for (uint64_t i = tid * 16; i < body; i += nth * 16)
            *reinterpret_cast<uint4 *>(b + head + i) = make_uint4(0u, 0u, 0u, 0u);
        for (uint64_t i = head + body + tid * 4; i < n; i += nth * 4)
            *reinterpret_cast<uint32_t *>(b + i) = 0u;
    }
}

hipError_t launch_zero(const ZeroParams &p, hipStream_t stream)
{
    uint64_t most = 0;
    for (int r = 0; r < 3; ++r) {
        if (p.bytes[r] && ((reinterpret_cast<uintptr_t>(p.ptr[r]) | p.bytes[r]) & 3u))
            return hipErrorInvalidValue;
        most = p.bytes[r] > most ? p.bytes[r] : most;
    }
    if (most == 0)
        return hipSuccess;
    const uint64_t want = (most + 256u * 16u - 1) / (256u * 16u);
    RANS_LAUNCH(k_zero, dim3((uint32_t)(want < 2048 ? want : 2048)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_compact(const CompactParams &p, int num_cus, hipStream_t stream)
{
    const uint64_t cap = (uint64_t)num_cus * 8;
    if (p.slot_bytes <= 8192) { // small chunks: 16 lanes each
        const uint64_t want = (p.nchunks + 255) / 256; // a wave per 64 chunks, four waves per block
        const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
        RANS_LAUNCH(k_compact_small, dim3(grid), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    const uint64_t want = (p.nchunks + 3) / 4;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    RANS_LAUNCH(k_compact, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// One model per CHUNK, built where the symbols are: one wave per chunk of u8 symbols counts them
// (count_freqs, main.cpp:59-66; 256 counters in the wave's LDS) and normalises the counts to 1 << scale_bits exactly
// as SymbolStats::normalize_freqs does (main.cpp:75-129; the width-based restatement of model.cpp normalize_freqs):
//   (device_common.hpp adapt_normalize: edges by 64-bit scaling, then the sequential repair of squeezed symbols)
// Lane l owns symbols 4l .. 4l+3 (the layout adapt_load_cum reads); the result is u16[256] per chunk.  A chunk that
// cannot be normalised (more distinct symbols than slots) sets bit 0 of *flags.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_chunk_models(const uint8_t *syms, uint64_t n, uint32_t chunk_syms, uint64_t nchunks,
                                                      uint32_t scale_bits, uint16_t *chunk_freqs, uint32_t *flags)
{
    // Round 5: four copies of a wave's 256 counters, eight banks apart (a lane counts into copy lane & 3: the ten-odd lanes
    // that meet a Zipf source's top symbol in one instruction spread over four banks), and 16 bytes per lane and load, two
    // loads in flight -- one dword per lane and trip and a single copy made this kernel 0.68 ms per GiB, three times the
    // whole-input histogram's 0.23.
    constexpr uint32_t kCopies = 4, kStride = 256 + 8;
    __shared__ uint32_t hist[4][kCopies * kStride];
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    uint32_t *h = hist[wave];
    uint32_t *hc = h + (lane & (kCopies - 1u)) * kStride;
    const uint32_t target = 1u << scale_bits;
    const uint64_t total_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t c = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave; c < nchunks; c += total_waves) {
        const uint64_t first = c * chunk_syms;
        const uint32_t nsym = (uint32_t)((n - first) < chunk_syms ? (n - first) : chunk_syms);
        const uint8_t *src = syms + first;
        for (uint32_t i = lane; i < kCopies * kStride; i += 64u)
            h[i] = 0u;
        auto count4 = [&](uint32_t v) {
            atomicAdd(&hc[v & 0xffu], 1u);
            atomicAdd(&hc[(v >> 8) & 0xffu], 1u);
            atomicAdd(&hc[(v >> 16) & 0xffu], 1u);
            atomicAdd(&hc[v >> 24], 1u);
        };
        uint32_t done = 0;
        if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) { // 16 bytes per lane, the next trip's load issued before this trip's counting
            const uint32_t body16 = nsym & ~1023u;
            if (body16) {
                u32x4 v = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(reinterpret_cast<uint64_t>(src) + lane * 16u));
                for (uint32_t i = 0; i < body16; i += 1024u) {
                    u32x4 nx = v;
                    if (i + 1024u < body16)
                        nx = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(reinterpret_cast<uint64_t>(src) + i + 1024u + lane * 16u));
                    count4(v.x);
                    count4(v.y);
                    count4(v.z);
                    count4(v.w);
                    v = nx;
                }
            }
            done = body16;
        }
        const bool aligned = (reinterpret_cast<uintptr_t>(src) & 3u) == 0;
        const uint32_t body = aligned ? (nsym & ~3u) : done;
        for (uint32_t i = done + lane * 4u; i < body; i += 256u)
            count4(*reinterpret_cast<const uint32_t *>(src + i));
        for (uint32_t i = body + lane; i < nsym; i += 64u)
            atomicAdd(&hc[src[i]], 1u);
        // (LDS operations of one wave execute in order: the reads below see every lane's increments)
        uint32_t cnt[4], width[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cnt[i] = 0u;
#pragma unroll
            for (uint32_t k = 0; k < kCopies; ++k)
                cnt[i] += h[k * kStride + 4u * lane + i];
        }
        const bool failed = !adapt_normalize(cnt, nsym, target, lane, width);
        if (failed && lane == 0)
            atomicOr(flags, 1u);
        // u16 x 4 per lane = 8 bytes at 8 * lane (adapt_load_cum's layout)
        const u32x2 packed = {width[0] | (width[1] << 16), width[2] | (width[3] << 16)};
        *reinterpret_cast<u32x2 RANS_GLOBAL *>(reinterpret_cast<uint64_t>(chunk_freqs) + c * 512u + 8u * lane) = packed;
    }
}

hipError_t launch_chunk_models(const void *syms, uint64_t n, uint32_t chunk_syms, uint64_t nchunks, uint32_t scale_bits,
                               uint16_t *d_chunk_freqs, uint32_t *d_flags, int num_cus, hipStream_t stream)
{
    const uint64_t want = (nchunks + 3) / 4;
    const uint64_t cap = (uint64_t)num_cus * 8;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    RANS_LAUNCH(k_chunk_models, dim3(grid), dim3(256), 0, stream, static_cast<const uint8_t *>(syms), n, chunk_syms, nchunks,
                scale_bits, d_chunk_freqs, d_flags);
    return hipGetLastError();
}

hipError_t launch_histogram(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *d_hist,
                            uint32_t *d_flags, int num_cus, hipStream_t stream)
{
    const uint32_t grid = (uint32_t)num_cus * (sym_bytes == 1 ? kHistBlocksPerCu : 4);
    if (sym_bytes == 1) {
        const size_t lds = (size_t)kHistCopies * kHistCopyStride * 4;
        RANS_LAUNCH(k_histogram_u8, dim3(grid), dim3(256), lds, stream, syms, n, nsyms, d_hist, d_flags);
    } else {
        uint32_t copies = 8; // a power of two, as many as fit 33 KiB (four blocks per CU stay resident; one copy above 4096 symbols)
        while (copies > 1 && (size_t)copies * (nsyms + 8) * 4 > 33 * 1024)
            copies >>= 1;
        const size_t lds = (size_t)copies * (nsyms + 8) * 4;
        static std::atomic<uint64_t> lds_ok{0};
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_histogram_u16), 160 * 1024, lds_ok); e != hipSuccess)
            return e;
        RANS_LAUNCH(k_histogram_u16, dim3(grid), dim3(256), lds, stream, syms, n, nsyms, d_hist, d_flags, copies);
    }
    return hipGetLastError();
}

} // namespace rans_amd
