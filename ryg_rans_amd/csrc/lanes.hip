// lanes.hip -- lane-per-stream kernels for narrow interleaves (N = 1, 2, 4, 8), both generations.

#include "device_common.hpp"
#include "launchers.hpp"

namespace rans_amd {

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;

// ===========================================================================
// Lane-per-stream kernels for narrow interleaves (N = 1, 2, 4, 8; BASELINE config 2 is
// the reference's 2-way rans64 loop, main64.cpp:224-287).  An N-way stream with N << 64
// cannot feed a wavefront, so here every LANE owns a whole chunk: its N states live in
// registers, it walks its own stream with its own pointer (the renormalisation order
// inside a chunk is the sequential reference order, no cross-lane work at all), and a
// wave decodes 64 chunks at once.  Tables are shared through LDS as before.
// ===========================================================================

// Per-lane stream window for the lane-per-stream decoder: 16 bytes of the lane's own stream
// in registers plus the next 16 prefetched, so that a lane touches global memory once per
// 16 stream bytes (a dword-per-unit walk would issue 4-16x as many scattered accesses).
// The window is consumed from its low end and shifted down after every unit -- positional
// indexing would be a chain of v_cndmask (~22 issue cycles each on gfx950) -- and the refill
// loads straight into `pre` under the lanes' exec mask, so the load issued at one refill is only
// waited for at the next one.
struct LaneWindow {
    u32x4 win, pre;
    uint64_t next;  // global address of the 16 bytes after `pre`
    uint64_t limit; // 16-byte aligned end of the readable container
    uint32_t left;  // bytes left in win (1..16)
    uint32_t used;  // stream bytes consumed so far

    template <int UNIT> __device__ __forceinline__ void shift()
    {
        if constexpr (UNIT == 4) {
            win.x = win.y;
            win.y = win.z;
            win.z = win.w;
        } else {
            win.x = __builtin_amdgcn_alignbit(win.y, win.x, 8 * UNIT);
            win.y = __builtin_amdgcn_alignbit(win.z, win.y, 8 * UNIT);
            win.z = __builtin_amdgcn_alignbit(win.w, win.z, 8 * UNIT);
            win.w >>= 8 * UNIT;
        }
    }
    __device__ __forceinline__ void refill()
    {
        win = pre;
        left = 16;
        if (next < limit) // past the container: `pre` keeps stale bytes, which only a corrupt chunk consumes
            pre = *reinterpret_cast<const u32x4 RANS_GLOBAL *>(next);
        next += 16;
    }
    template <int UNIT> __device__ __forceinline__ void open(uint64_t addr, uint64_t lim)
    {
        limit = lim;
        const uint64_t base = addr & ~uint64_t(15);
        used = 0;
        win = u32x4{0u, 0u, 0u, 0u};
        pre = win;
        if (base < limit)
            win = *reinterpret_cast<const u32x4 RANS_GLOBAL *>(base);
        if (base + 16 < limit)
            pre = *reinterpret_cast<const u32x4 RANS_GLOBAL *>(base + 16);
        next = base + 32;
        left = 16;
        for (uint32_t skip = (uint32_t)(addr & 15u); skip != 0; skip -= UNIT) { // once per chunk
            shift<UNIT>();
            left -= UNIT;
        }
    }
    template <int UNIT> __device__ __forceinline__ uint32_t take()
    {
        uint32_t w = win.x;
        if constexpr (UNIT == 2)
            w &= 0xffffu;
        else if constexpr (UNIT == 1)
            w &= 0xffu;
        shift<UNIT>();
        used += UNIT;
        left -= UNIT;
        if (left == 0)
            refill();
        return w;
    }
};

template <int FMT>
__device__ __forceinline__ void lane_renorm(typename FmtTraits<FMT>::state_t &x, LaneWindow &W, bool active)
{
    if constexpr (FMT == FMT_WORD) {
        if (active && x < (1u << 16)) // rans_word_sse41.h:134-141
            x = (x << 16) | W.take<2>();
    } else if constexpr (FMT == FMT_R64) {
        if (active && x < (1ull << 31)) // rans64.h:305-316
            x = (x << 32) | W.take<4>();
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) // rans_byte.h:307-318, at most two bytes for scale_bits <= 16
            if (active && x < (1u << 23))
                x = (x << 8) | W.take<1>();
    }
}

// ---------------------------------------------------------------------------
// Staged lane-per-stream decoder.  Per-lane 16-byte loads pull a whole memory line for every
// 16 bytes used, and the line is long evicted when the lane comes back for its next 16 bytes
// (measured: >= 4x over-fetch, the kernel sat on the fabric at ~3.3 TB/s); and a lane that
// refills on its own stalls its whole wave.  Here the WAVE refills for all of its 64 chunks at
// once, every 16 symbols: a lane's stream lives in a 128-byte ring in LDS (two 64-byte lines),
// lanes publish which line they need next, and 4 lanes fetch one chunk's line with coalesced
// 16-byte loads (4 load instructions cover 64 chunks x 64 B).  16 symbols consume at most one
// line (rans64: <= 4 B per symbol), so "at least 64 bytes ahead" before every group is all the
// invariant there is.  Ring rows are 136 bytes apart: equal positions of the 64 lanes spread over
// 32 banks.  Positions are 32-bit offsets from the line of the wave's first chunk.
// ---------------------------------------------------------------------------
constexpr uint32_t kLaneLine = 64;
constexpr uint32_t kLaneRingStride = 2 * kLaneLine + 8;
constexpr uint32_t kLaneWaveLds = 64 * kLaneRingStride + 64 * 4; // rings + one request word per lane

template <int FMT> struct LaneRing {
    const uint8_t *row; // this lane's ring in LDS
    uint32_t cur;       // read position (offset from the wave's region base)
    uint32_t used;      // stream bytes consumed

    template <int UNIT> __device__ __forceinline__ uint32_t take()
    {
        const uint8_t *at = row + (cur & (2 * kLaneLine - 1));
        cur += UNIT;
        used += UNIT;
        if constexpr (UNIT == 4)
            return *reinterpret_cast<const uint32_t *>(at);
        else if constexpr (UNIT == 2)
            return *reinterpret_cast<const uint16_t *>(at);
        else
            return *at;
    }
    __device__ __forceinline__ void renorm(typename FmtTraits<FMT>::state_t &x, bool active)
    {
        // the branch bodies hold an LDS read, so they stay exec-masked branches (no v_cndmask)
        if constexpr (FMT == FMT_WORD) {
            if (active && x < (1u << 16)) // rans_word_sse41.h:134-141
                x = (x << 16) | take<2>();
        } else if constexpr (FMT == FMT_R64) {
            if (active && x < (1ull << 31)) // rans64.h:305-316
                x = (x << 32) | take<4>();
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) // rans_byte.h:307-318, at most two bytes for scale_bits <= 16
                if (active && x < (1u << 23))
                    x = (x << 8) | take<1>();
        }
    }
};

// Smallest stream offset among the lanes that hold a chunk (wave-uniform): the base of the wave's 32-bit window
// onto the container.  Offsets need not ascend with the chunk index.  A wave addresses its batch through 32-bit offsets
// from the batch's lowest chunk: chunks that lie 1 GiB or more above it (an overflowed chunk of a large sized-slot
// container, a scattered hand-made index) wait for another trip of the same wave over the same batch, whose window starts
// at the lowest of THEM (`again` below) -- every well-formed index decodes, the usual batch in one trip.
__device__ __forceinline__ uint64_t wave_min_offset(uint64_t off, bool valid)
{
    uint32_t lo = valid ? (uint32_t)off : 0xffffffffu, hi = valid ? (uint32_t)(off >> 32) : 0xffffffffu;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t olo = (uint32_t)__shfl_xor((int)lo, d), ohi = (uint32_t)__shfl_xor((int)hi, d);
        const bool less = ohi < hi || (ohi == hi && olo < lo);
        lo = less ? olo : lo;
        hi = less ? ohi : hi;
    }
    return uniform64(((uint64_t)hi << 32) | lo);
}

// 4 x 4 transpose of 16-byte pieces inside every quad of lanes: lane 4k+m, piece t  <->  lane 4k+t, piece m.  A
// lane that stores its own 64-byte line issues four 16-byte requests, and 64 lanes 64 of them per instruction -- the
// vector-memory address path (TA) was 86 % busy in these kernels (profiles/r02_lanes_counters.md); after the
// transpose store instruction t writes the whole line of the quad's lane t: one 64-byte request per quad.
template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ void quad_transpose(u32x4 &q0, u32x4 &q1, u32x4 &q2, u32x4 &q3, uint32_t lane)
{
    const bool b0 = (lane & 1u) != 0, b1 = (lane & 2u) != 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) { // lane bit 0 <-> piece bit 0
        const uint32_t a = q0[d], b = q1[d], c = q2[d], e = q3[d];
        const uint32_t fa = quad_perm<0xA0>(b), fb = quad_perm<0xF5>(a); // [0,0,2,2]: from lane - 1, [1,1,3,3]: from lane + 1
        const uint32_t fc = quad_perm<0xA0>(e), fe = quad_perm<0xF5>(c);
        q0[d] = b0 ? fa : a;
        q1[d] = b0 ? b : fb;
        q2[d] = b0 ? fc : c;
        q3[d] = b0 ? e : fe;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) { // lane bit 1 <-> piece bit 1
        const uint32_t a = q0[d], b = q1[d], c = q2[d], e = q3[d];
        const uint32_t fa = quad_perm<0x44>(c), fc = quad_perm<0xEE>(a); // [0,1,0,1]: from lane - 2, [2,3,2,3]: from lane + 2
        const uint32_t fb = quad_perm<0x44>(e), fe = quad_perm<0xEE>(b);
        q0[d] = b1 ? fa : a;
        q2[d] = b1 ? c : fc;
        q1[d] = b1 ? fb : b;
        q3[d] = b1 ? e : fe;
    }
}

template <int FMT, int NW>
__global__ void __launch_bounds__(1024) k_decode_lanes_staged(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
        const uint4 *g1 = reinterpret_cast<const uint4 *>(p.table1);
        uint4 *l1 = reinterpret_cast<uint4 *>(smem + t0_bytes);
        for (uint32_t i = threadIdx.x; i < t1_bytes / 16u; i += blockDim.x)
            l1[i] = g1[i];
    }
    __syncthreads();

    DecTables<FMT> T;
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);
    if (!lds_starts_at_zero(smem)) { // cannot happen without static LDS; never decode on a wrong assumption
        if (threadIdx.x == 0)
            atomicAdd(p.err_count, 1ull << 32);
        return;
    }

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u; // keep the wave kernels' counter ring consistent
    if (p.span_reset && blockIdx.x == 0 && threadIdx.x < 2)
        p.span_reset[threadIdx.x] = 0ull;
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    uint8_t *rings = smem + t0_bytes + t1_bytes + wave * kLaneWaveLds;
    uint32_t *req = reinterpret_cast<uint32_t *>(rings + 64u * kLaneRingStride);
    const uint32_t part = lane & 3u, grp = lane >> 2;

    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t glimit = (cbase + p.container_bytes + 15u) & ~uint64_t(15);
    const bool wide_out = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.out) | p.chunk_syms) & 15u) == 0;
    uint32_t nbad = 0;
    const uint64_t nbatches = (p.nchunks + 63u) / 64u;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t again = 0; // lanes of the batch in hand whose chunks lay beyond the last trip's window
    for (uint64_t batch_v = (uint64_t)blockIdx.x * waves_per_block + wave; batch_v < nbatches; batch_v += again ? 0 : total_waves) {
        const uint64_t batch = uniform64(batch_v);
        const uint64_t chunk = batch * 64u + lane;
        bool valid = chunk < p.nchunks && (again == 0 || ((again >> lane) & 1u));
        const uint64_t off = valid ? p.offsets[chunk] : 0;
        const uint32_t len = valid ? p.lengths[chunk] : 0;
        const uint64_t first = chunk * p.chunk_syms;
        if (valid && ((off & (Tr::kUnit - 1u)) != 0 || len < NW * Tr::kStateBytes || off > p.container_bytes || len > p.container_bytes - off)) {
            nbad++; // (a malformed entry is seen in one trip only: it never joins `again`)
            valid = false;
        }
        // region base: the line of the lowest chunk of this trip
        const uint64_t rb = wave_min_offset(off, valid) & ~uint64_t(kLaneLine - 1);
        const bool far = valid && off - rb >= (1u << 30);
        again = __builtin_amdgcn_ballot_w64(far);
        valid = valid && !far;
        const uint32_t nsym = valid ? (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms) : 0u;
        uint8_t RANS_GLOBAL *dst = (uint8_t RANS_GLOBAL *)p.out + first * p.sym_bytes;

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l) { // RansDecInit order: lane 0's state first
            x[l] = Tr::kL;
            if (valid) {
                const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.container + off;
                if constexpr (FMT == FMT_R64) {
                    const u32x2 v = reinterpret_cast<const u32x2 RANS_GLOBAL *>(src)[l];
                    x[l] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                } else {
                    x[l] = reinterpret_cast<const uint32_t RANS_GLOBAL *>(src)[l];
                }
            }
        }
        LaneRing<FMT> W;
        W.row = rings + lane * kLaneRingStride;
        W.cur = (uint32_t)(off - rb) + NW * Tr::kStateBytes;
        W.used = 0;
        uint32_t ld = W.cur & ~(kLaneLine - 1u); // next line this lane has not staged yet

        // the whole wave takes part (also lanes without a chunk): lane -> which line its chunk needs,
        // then lane (4 g + part) moves 16 bytes of chunk (16 j + g)'s line, j = 0..3
        auto refill = [&]() {
            const bool need = valid && (int32_t)(ld - W.cur) < (int32_t)kLaneLine;
            req[lane] = need ? (ld | 1u) : 0u;
            if (need)
                ld += kLaneLine;
            uint32_t r[4];
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = req[16 * j + grp]; // LDS ops of one wave execute in order: sees the writes above
                v[j] = u32x4{0u, 0u, 0u, 0u};
                const uint64_t a = cbase + rb + (r[j] & ~(kLaneLine - 1u)) + part * 16u;
                if ((r[j] & 1u) && a < glimit)
                    v[j] = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(a));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r[j] & 1u) {
                    uint8_t *at = rings + (16u * j + grp) * kLaneRingStride + (r[j] & kLaneLine) + part * 16u;
                    reinterpret_cast<u32x2 *>(at)[0] = u32x2{v[j].x, v[j].y}; // rows are 8-byte aligned
                    reinterpret_cast<u32x2 *>(at)[1] = u32x2{v[j].z, v[j].w};
                }
        };

        // 16 symbols into four dwords: 16/NW rounds of NW steps + renormalisations
        auto decode16 = [&]() -> u32x4 {
            uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int rr = 0; rr < 16 / NW; ++rr) {
#pragma unroll
                for (int l = 0; l < NW; ++l) {
                    uint32_t sy = dec_step<FMT>(T, x[l]);
                    if constexpr (Tr::kSymByte == 3)
                        sy >>= 24;
                    const int pos = rr * NW + l;
                    pk[pos / 4] |= (sy & 0xffu) << (8 * (pos % 4));
                }
#pragma unroll
                for (int l = 0; l < NW; ++l)
                    W.renorm(x[l], true);
            }
            return u32x4{pk[0], pk[1], pk[2], pk[3]};
        };

        // store addresses of the transposed output: instruction t writes the line of the quad's lane t
        uint64_t at[4];
        bool vq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int src = (int)((lane & ~3u) + t);
            const uint64_t d64 = reinterpret_cast<uint64_t>(dst);
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)d64, src), hi = (uint32_t)__shfl((int)(uint32_t)(d64 >> 32), src);
            at[t] = (((uint64_t)hi << 32) | lo) + (lane & 3u) * 16u;
            vq[t] = __shfl((int)valid, src) != 0;
        }
        refill(); // two lines ahead to start with
        // 64 symbols per trip: four groups of 16 (a refill before each), then the lane writes its 64
        // bytes with four back-to-back 16-byte stores.  One 16-byte store per group left every line
        // dirty in L2 for four groups -- long enough to be evicted half-written: 3.2x the bytes on
        // the write side (WRITE_SIZE 0.87 GB for 0.27 GB of symbols).
        for (uint32_t i0 = 0; __builtin_amdgcn_ballot_w64(i0 < nsym) != 0; i0 += 64u) {
            const uint32_t left = i0 < nsym ? nsym - i0 : 0u;
            if (__builtin_amdgcn_ballot_w64(!(left >= 64u && wide_out) && valid) == 0) { // wave-uniform
                u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0;
                refill();
                if (left)
                    q0 = decode16();
                refill();
                if (left)
                    q1 = decode16();
                refill();
                if (left)
                    q2 = decode16();
                refill();
                if (left)
                    q3 = decode16();
                // (every valid lane has its 64 symbols here; lanes without a chunk carry zeros)
                quad_transpose(q0, q1, q2, q3, lane);
                const uint32_t o = i0;
                if (vq[0])
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(at[0] + o) = q0;
                if (vq[1])
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(at[1] + o) = q1;
                if (vq[2])
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(at[2] + o) = q2;
                if (vq[3])
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(at[3] + o) = q3;
                continue;
            }
            // ragged end of a chunk, unaligned or 16-bit output: 16 symbols at a time
            for (uint32_t g = 0; g < 4u; ++g) {
                refill();
                const uint32_t j0 = i0 + 16u * g;
                if (j0 >= nsym)
                    continue;
                const uint32_t cnt = nsym - j0 < 16u ? nsym - j0 : 16u;
                if (cnt == 16u && wide_out) {
                    const u32x4 q = decode16();
                    __builtin_nontemporal_store(q, reinterpret_cast<u32x4 RANS_GLOBAL *>(dst + j0)); // written once, never read back
                } else {
                    for (uint32_t i = 0; i < cnt; i += NW) {
                        const uint32_t c = cnt - i < (uint32_t)NW ? cnt - i : (uint32_t)NW;
#pragma unroll
                        for (int l = 0; l < NW; ++l)
                            if ((uint32_t)l < c) {
                                uint32_t sy = dec_step<FMT>(T, x[l]);
                                if constexpr (Tr::kSymByte == 3)
                                    sy >>= 24;
                                if (p.sym_bytes == 1)
                                    dst[j0 + i + l] = (uint8_t)sy;
                                else
                                    reinterpret_cast<uint16_t RANS_GLOBAL *>(dst)[j0 + i + l] = (uint16_t)sy;
                            }
#pragma unroll
                        for (int l = 0; l < NW; ++l)
                            W.renorm(x[l], (uint32_t)l < c);
                    }
                }
            }
        }
        bool bad = false;
#pragma unroll
        for (int l = 0; l < NW; ++l)
            bad = bad || (x[l] != Tr::kL);
        if (valid && (bad || W.used + NW * Tr::kStateBytes != len))
            nbad++;
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}

// ---------------------------------------------------------------------------
// k_decode_lanes_r64x2: the reference's 2-way rans64 loop (main64.cpp:261-282), one chunk per lane, third
// generation.  What bounds the staged kernel above (rocprofv3 counters, profiles/r02_lanes_counters.md): neither the
// VALU (30 % busy) nor the LDS pipe (42 %) but the vector-memory ADDRESS path -- TA busy 86 % of the kernel: every
// lane's 16-byte store is its own request (64 per instruction, 16.8 M write requests for 256 MiB), and so is every
// 16-byte piece a lane fetches for itself.  And, per wave, four DEPENDENT LDS round trips per pair of symbols.  Here
//  * global accesses are made by QUADS of lanes: the four lanes of a quad fetch the four 16-byte pieces of ONE
//    chunk's 64-byte line (instruction t serves the quad's lane t: address and exec mask by DPP broadcast) and write
//    them straight into that lane's ring row; the 64 symbols a lane has decoded are transposed inside the quad (4 x 4
//    pieces, v_cndmask_b32_dpp) so that store instruction t writes chunk (quad, t)'s whole line: 64-byte requests;
//  * the next 8 stream bytes of a lane sit in two registers (the "window"); a renormalisation takes the first
//    dword under an exec mask (v_cmpx), and only the lanes that consumed something re-read their window
//    (ds_read2_b32, exec-masked) while the NEXT pair's table lookups are in flight: two LDS round trips per pair;
//  * a line is requested as soon as the ring has room for it (<= 64 B staged ahead), ONE GROUP (16 symbols) before
//    it is written to the ring: the pieces stay in registers over the group.  A valid stream consumes <= 40 B per
//    group (5 renormalisations per state: 8 x 16 bits out, 32 bits in each) and the window reads 8 B ahead, so
//    >= 48 B ahead at a group boundary is the invariant; a lane below that (streams of 2^-16 symbols) is served
//    synchronously on a side path;
//  * the whole trip loop is ONE asm statement: states, window, parked pieces, packed symbols and temporaries in
//    fixed registers (64-bit operands need their halves named; nothing in flight crosses a statement the compiler
//    schedules).  The 64 symbols of a trip are stored at the start of the next trip, after its first vmcnt wait, so
//    that the stores have a whole group to complete.
// Requirements (launcher): scale_bits 7..16 (cum2sym), u8 symbols, chunk_syms % 64 == 0, 64-byte aligned output,
// every chunk full (a ragged last chunk goes through the staged kernel in a second launch).
// ---------------------------------------------------------------------------
constexpr uint32_t kR64RingStride = 2 * kLaneLine + 8; // 128-byte ring + mirror of its first dword (+ 4 pad)
constexpr uint32_t kR64WaveLds = 64 * kR64RingStride;

// Registers of the trip loop: v[8:23] packed symbols of the trip (set g = group g = piece g of the lane's line),
// v[24:39] parked pieces (set t = fetch instruction t), v[40:41] state 0, v[42:43] state 1, v[44:45] window,
// v46..v62 temporaries, v[64:67] ring addresses of the parked pieces; s[36:39] renormalisation masks,
// s[40:47] lanes with a parked piece per fetch instruction, s50 "not the first trip".
//
// One pair of symbols, first half: slots, cum2sym, records (rans64.h:286-292 for both states)
#define R64_LOOKUP                                                                                                      \
    "v_and_b32 v46, %[maskv], v40\n\t"                                                                                  \
    "v_and_b32 v47, %[maskv], v42\n\t"                                                                                  \
    "ds_read_u8 v48, v46\n\t"                                                                                           \
    "ds_read_u8 v49, v47\n\t"                                                                                           \
    "v_lshrrev_b64 v[50:51], %[sbv], v[40:41]\n\t"                                                                      \
    "v_lshrrev_b64 v[52:53], %[sbv], v[42:43]\n\t"                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "v_lshl_add_u32 v62, v48, 3, %[t1v]\n\t"                                                                            \
    "v_lshl_add_u32 v63, v49, 3, %[t1v]\n\t"                                                                            \
    "ds_read_b64 v[54:55], v62\n\t"                                                                                     \
    "ds_read_b64 v[56:57], v63\n\t"
// second half: x = freq * (x >> scale_bits) + slot - start, renormalisation (rans64.h:305-316: x < 2^31 -> x = x << 32
// | next dword) from the window under an exec mask, window re-read by the lanes that took from it
#define R64_UPDATE                                                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "v_sub_u32 v58, v46, v55\n\t"                                                                                       \
    "v_mul_u32_u24 v59, v54, v51\n\t"           /* freq * (q >> 32): freq <= 2^16, q >> 32 < 2^24 */                    \
    "v_mad_u64_u32 v[40:41], vcc, v50, v54, v[58:59]\n\t"                                                               \
    "v_cmpx_gt_u64 s[36:37], %[kL], v[40:41]\n\t"                                                                       \
    "v_mov_b32 v41, v40\n\t"                                                                                            \
    "v_mov_b32 v40, v44\n\t"                                                                                            \
    "v_mov_b32 v44, v45\n\t"                                                                                            \
    "v_add_u32 %[cur], 4, %[cur]\n\t"                                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                                            \
    "v_sub_u32 v60, v47, v57\n\t"                                                                                       \
    "v_mul_u32_u24 v61, v56, v53\n\t"                                                                                   \
    "v_mad_u64_u32 v[42:43], vcc, v52, v56, v[60:61]\n\t"                                                               \
    "v_cmpx_gt_u64 s[38:39], %[kL], v[42:43]\n\t"                                                                       \
    "v_mov_b32 v43, v42\n\t"                                                                                            \
    "v_mov_b32 v42, v44\n\t"                                                                                            \
    "v_add_u32 %[cur], 4, %[cur]\n\t"                                                                                   \
    "s_or_b64 exec, s[36:37], s[38:39]\n\t"                                                                             \
    "v_and_b32 v62, 0x7f, %[cur]\n\t"                                                                                   \
    "v_add_u32 v62, %[row], v62\n\t"                                                                                    \
    "ds_read2_b32 v[44:45], v62 offset1:1\n\t"                                                                          \
    "s_mov_b64 exec, -1\n\t"
// four symbols into one dword of the trip's output
#define R64_QUAD(PK)                                                                                                    \
    R64_LOOKUP "v_lshl_or_b32 " PK ", v49, 8, v48\n\t" R64_UPDATE                                                       \
    R64_LOOKUP "v_lshl_or_b32 " PK ", v48, 16, " PK "\n\tv_lshl_or_b32 " PK ", v49, 24, " PK "\n\t" R64_UPDATE
#define R64_GROUP(P0, P1, P2, P3) R64_QUAD(P0) R64_QUAD(P1) R64_QUAD(P2) R64_QUAD(P3)
// Round 3, the same pair of symbols with ONE gather each: a 4-byte slot record {freq:12 | slot - start:12 | sym:8} per
// cumulative slot (the layout rans_word_sse41.h:64-72 uses for the word format, applied to rans64's tables; it holds every
// model whose largest frequency is below 4096 -- at 14 bits every model whose most probable symbol stays below 25 %), at LDS
// address 4 * slot.  One LDS round trip per pair of symbols instead of two dependent ones, one gather instead of two
// (the LDS pipe was 74 % busy, half of it bank conflicts of the two gathers); one more VALU instruction per symbol.
#define R64P_LOOKUP                                                                                                     \
    "v_and_b32 v46, %[maskv], v40\n\t"                                                                                  \
    "v_and_b32 v47, %[maskv], v42\n\t"                                                                                  \
    "v_lshlrev_b32 v46, 2, v46\n\t"                                                                                     \
    "v_lshlrev_b32 v47, 2, v47\n\t"                                                                                     \
    "ds_read_b32 v54, v46\n\t"                                                                                          \
    "ds_read_b32 v56, v47\n\t"                                                                                          \
    "v_lshrrev_b64 v[50:51], %[sbv], v[40:41]\n\t"                                                                      \
    "v_lshrrev_b64 v[52:53], %[sbv], v[42:43]\n\t"
// PACK: the instruction(s) that put the two symbols (top bytes of v54, v56) into the trip's output dword
#define R64P_UPDATE(PACK)                                                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    PACK                                                                                                                \
    "v_and_b32 v48, %[m12v], v54\n\t"                                                                                   \
    "v_bfe_u32 v58, v54, 12, 12\n\t"                                                                                    \
    "v_mul_u32_u24 v59, v48, v51\n\t"                                                                                   \
    "v_mad_u64_u32 v[40:41], vcc, v50, v48, v[58:59]\n\t"                                                               \
    "v_cmpx_gt_u64 s[36:37], %[kL], v[40:41]\n\t"                                                                       \
    "v_mov_b32 v41, v40\n\t"                                                                                            \
    "v_mov_b32 v40, v44\n\t"                                                                                            \
    "v_mov_b32 v44, v45\n\t"                                                                                            \
    "v_add_u32 %[cur], 4, %[cur]\n\t"                                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                                            \
    "v_and_b32 v49, %[m12v], v56\n\t"                                                                                   \
    "v_bfe_u32 v60, v56, 12, 12\n\t"                                                                                    \
    "v_mul_u32_u24 v61, v49, v53\n\t"                                                                                   \
    "v_mad_u64_u32 v[42:43], vcc, v52, v49, v[60:61]\n\t"                                                               \
    "v_cmpx_gt_u64 s[38:39], %[kL], v[42:43]\n\t"                                                                       \
    "v_mov_b32 v43, v42\n\t"                                                                                            \
    "v_mov_b32 v42, v44\n\t"                                                                                            \
    "v_add_u32 %[cur], 4, %[cur]\n\t"                                                                                   \
    "s_or_b64 exec, s[36:37], s[38:39]\n\t"                                                                             \
    "v_and_b32 v62, 0x7f, %[cur]\n\t"                                                                                   \
    "v_add_u32 v62, %[row], v62\n\t"                                                                                    \
    "ds_read2_b32 v[44:45], v62 offset1:1\n\t"                                                                          \
    "s_mov_b64 exec, -1\n\t"
#define R64P_QUAD(PK)                                                                                                   \
    R64P_LOOKUP R64P_UPDATE("v_perm_b32 " PK ", v56, v54, %[selA]\n\t")                                                 \
    R64P_LOOKUP R64P_UPDATE("v_perm_b32 v63, v56, v54, %[selB]\n\tv_or_b32 " PK ", " PK ", v63\n\t")
#define R64P_GROUP(P0, P1, P2, P3) R64P_QUAD(P0) R64P_QUAD(P1) R64P_QUAD(P2) R64P_QUAD(P3)
// parked pieces -> ring.  Lane 4k+m holds piece m of the line of lane 4k+t (fetch instruction t): it goes to that
// lane's row, rq0 + 136 t + slot (rq0 = own row - 136 m + 16 m); piece 0 of slot 0 also refreshes the mirror dword.
#define R64_COMMIT                                                                                                      \
    "s_waitcnt vmcnt(0)\n\t" \
    "s_mov_b64 exec, s[40:41]\n\t" \
    "ds_write2_b64 v64, v[24:25], v[26:27] offset0:0 offset1:1\n\t" \
    "s_and_b64 exec, exec, %[m0]\n\t" \
    "v_cmpx_eq_u32 vcc, %[rq0], v64\n\t" \
    "ds_write_b32 %[rq0], v24 offset:128\n\t" \
    "s_mov_b64 exec, s[42:43]\n\t" \
    "ds_write2_b64 v65, v[28:29], v[30:31] offset0:17 offset1:18\n\t" \
    "s_and_b64 exec, exec, %[m0]\n\t" \
    "v_cmpx_eq_u32 vcc, %[rq0], v65\n\t" \
    "ds_write_b32 %[rq0], v28 offset:264\n\t" \
    "s_mov_b64 exec, s[44:45]\n\t" \
    "ds_write2_b64 v66, v[32:33], v[34:35] offset0:34 offset1:35\n\t" \
    "s_and_b64 exec, exec, %[m0]\n\t" \
    "v_cmpx_eq_u32 vcc, %[rq0], v66\n\t" \
    "ds_write_b32 %[rq0], v32 offset:400\n\t" \
    "s_mov_b64 exec, s[46:47]\n\t" \
    "ds_write2_b64 v67, v[36:37], v[38:39] offset0:51 offset1:52\n\t" \
    "s_and_b64 exec, exec, %[m0]\n\t" \
    "v_cmpx_eq_u32 vcc, %[rq0], v67\n\t" \
    "ds_write_b32 %[rq0], v36 offset:536\n\t" \
    "s_mov_b64 exec, -1\n\t"
// bytes staged ahead in v46 -> lanes with room for their next line (<= 64 ahead) ask for it: ld advances, the quad
// fetches (a = ld of the quad's lane t + 16 m by DPP; the sign bit marks "nothing asked")
#define R64_FETCH                                                                                                       \
    "v_bfrev_b32 v63, 1\n\t" \
    "v_cmp_ge_i32 vcc, 64, v46\n\t" \
    "v_cndmask_b32 v47, v63, %[ld], vcc\n\t" \
    "s_mov_b64 exec, vcc\n\t" \
    "v_add_u32 %[ld], 64, %[ld]\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    "s_nop 4\n\t" \
    "v_add_u32_dpp v64, v47, %[l16] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp v65, v47, %[l16] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp v66, v47, %[l16] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp v67, v47, %[l16] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmpx_le_i32 s[40:41], 0, v64\n\t" \
    "buffer_load_dwordx4 v[24:27], v64, %[rsrc], 0 offen nt\n\t" \
    "v_and_b32 v64, 64, v64\n\t" \
    "v_add_u32 v64, %[rq0], v64\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    "v_cmpx_le_i32 s[42:43], 0, v65\n\t" \
    "buffer_load_dwordx4 v[28:31], v65, %[rsrc], 0 offen nt\n\t" \
    "v_and_b32 v65, 64, v65\n\t" \
    "v_add_u32 v65, %[rq0], v65\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    "v_cmpx_le_i32 s[44:45], 0, v66\n\t" \
    "buffer_load_dwordx4 v[32:35], v66, %[rsrc], 0 offen nt\n\t" \
    "v_and_b32 v66, 64, v66\n\t" \
    "v_add_u32 v66, %[rq0], v66\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    "v_cmpx_le_i32 s[46:47], 0, v67\n\t" \
    "buffer_load_dwordx4 v[36:39], v67, %[rsrc], 0 offen nt\n\t" \
    "v_and_b32 v67, 64, v67\n\t" \
    "v_add_u32 v67, %[rq0], v67\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    "s_nop 4\n\t"
// 4 x 4 transpose of 16-byte pieces inside every quad: lane 4k+m, set t  <->  lane 4k+t, set m
#define R64_TRANSPOSE                                                                                                   \
    "v_mov_b32 v46, v8\n\t" \
    "v_mov_b32 v47, v9\n\t" \
    "v_mov_b32 v48, v10\n\t" \
    "v_mov_b32 v49, v11\n\t" \
    "v_mov_b32 v50, v16\n\t" \
    "v_mov_b32 v51, v17\n\t" \
    "v_mov_b32 v52, v18\n\t" \
    "v_mov_b32 v53, v19\n\t" \
    "s_mov_b32 vcc_lo, 0x55555555\n\t" \
    "s_mov_b32 vcc_hi, 0x55555555\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_dpp v8, v12, v8, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v9, v13, v9, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v10, v14, v10, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v11, v15, v11, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v16, v20, v16, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v17, v21, v17, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v18, v22, v18, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v19, v23, v19, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "s_mov_b32 vcc_lo, 0xaaaaaaaa\n\t" \
    "s_mov_b32 vcc_hi, 0xaaaaaaaa\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_dpp v12, v46, v12, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v13, v47, v13, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v14, v48, v14, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v15, v49, v15, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v20, v50, v20, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v21, v51, v21, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v22, v52, v22, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v23, v53, v23, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32 v46, v8\n\t" \
    "v_mov_b32 v47, v9\n\t" \
    "v_mov_b32 v48, v10\n\t" \
    "v_mov_b32 v49, v11\n\t" \
    "v_mov_b32 v50, v12\n\t" \
    "v_mov_b32 v51, v13\n\t" \
    "v_mov_b32 v52, v14\n\t" \
    "v_mov_b32 v53, v15\n\t" \
    "s_mov_b32 vcc_lo, 0x33333333\n\t" \
    "s_mov_b32 vcc_hi, 0x33333333\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_dpp v8, v16, v8, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v9, v17, v9, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v10, v18, v10, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v11, v19, v11, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v12, v20, v12, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v13, v21, v13, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v14, v22, v14, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v15, v23, v15, vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "s_mov_b32 vcc_lo, 0xcccccccc\n\t" \
    "s_mov_b32 vcc_hi, 0xcccccccc\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_dpp v16, v46, v16, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v17, v47, v17, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v18, v48, v18, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v19, v49, v19, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v20, v50, v20, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v21, v51, v21, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v22, v52, v22, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_dpp v23, v53, v23, vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
// store instruction t: the line of chunk (quad, t), 16 bytes per lane of the quad
#define R64_STORES                                                                                                      \
    "s_mov_b64 exec, %[vq0]\n\t" \
    "global_store_dwordx4 %[at0], v[8:11], off\n\t" \
    "s_mov_b64 exec, %[vq1]\n\t" \
    "global_store_dwordx4 %[at1], v[12:15], off\n\t" \
    "s_mov_b64 exec, %[vq2]\n\t" \
    "global_store_dwordx4 %[at2], v[16:19], off\n\t" \
    "s_mov_b64 exec, %[vq3]\n\t" \
    "global_store_dwordx4 %[at3], v[20:23], off\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    "v_lshl_add_u64 %[at0], %[at0], 0, 64\n\t" \
    "v_lshl_add_u64 %[at1], %[at1], 0, 64\n\t" \
    "v_lshl_add_u64 %[at2], %[at2], 0, 64\n\t" \
    "v_lshl_add_u64 %[at3], %[at3], 0, 64\n\t"
// Paired stores (round 5): a trip leaves 64 bytes per chunk -- HALF a 128-byte line of L2 -- and the other half followed a
// whole trip later, by when the line could have left L2 half written (WRITE_SIZE 1.34 x the symbols).
// Now the transposed pieces of every other trip wait in v68..v83 and the pair goes out together: instruction t writes
// bytes [0, 64) and [64, 128) of chunk (quad, t)'s line back to back.
#define R64_HOLD                                                                                                        \
    "v_mov_b32 v68, v8\n\tv_mov_b32 v69, v9\n\tv_mov_b32 v70, v10\n\tv_mov_b32 v71, v11\n\t"                             \
    "v_mov_b32 v72, v12\n\tv_mov_b32 v73, v13\n\tv_mov_b32 v74, v14\n\tv_mov_b32 v75, v15\n\t"                           \
    "v_mov_b32 v76, v16\n\tv_mov_b32 v77, v17\n\tv_mov_b32 v78, v18\n\tv_mov_b32 v79, v19\n\t"                           \
    "v_mov_b32 v80, v20\n\tv_mov_b32 v81, v21\n\tv_mov_b32 v82, v22\n\tv_mov_b32 v83, v23\n\t"
#define R64_STORES2                                                                                                     \
    "s_mov_b64 exec, %[vq0]\n\t"                                                                                        \
    "global_store_dwordx4 %[at0], v[68:71], off\n\t"                                                                    \
    "global_store_dwordx4 %[at0], v[8:11], off offset:64\n\t"                                                           \
    "s_mov_b64 exec, %[vq1]\n\t"                                                                                        \
    "global_store_dwordx4 %[at1], v[72:75], off\n\t"                                                                    \
    "global_store_dwordx4 %[at1], v[12:15], off offset:64\n\t"                                                          \
    "s_mov_b64 exec, %[vq2]\n\t"                                                                                        \
    "global_store_dwordx4 %[at2], v[76:79], off\n\t"                                                                    \
    "global_store_dwordx4 %[at2], v[16:19], off offset:64\n\t"                                                          \
    "s_mov_b64 exec, %[vq3]\n\t"                                                                                        \
    "global_store_dwordx4 %[at3], v[80:83], off\n\t"                                                                    \
    "global_store_dwordx4 %[at3], v[20:23], off offset:64\n\t"                                                          \
    "s_mov_b64 exec, -1\n\t"                                                                                            \
    "v_lshl_add_u64 %[at0], %[at0], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at1], %[at1], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at2], %[at2], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at3], %[at3], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at0], %[at0], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at1], %[at1], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at2], %[at2], 0, 64\n\t"                                                                          \
    "v_lshl_add_u64 %[at3], %[at3], 0, 64\n\t"
// s50 = trips decoded and not stored yet (0, 1: in v8..v23, 2: the older one transposed in v68..v83).  At the top of a trip:
#define R64_TRIP_TOP                                                                                                    \
    "s_cmp_eq_u32 s50, 0\n\t"                                                                                           \
    "s_cbranch_scc1 .Lr64first_%=\n\t" R64_TRANSPOSE                                                                    \
    "s_cmp_eq_u32 s50, 1\n\t"                                                                                           \
    "s_cbranch_scc1 .Lr64hold_%=\n\t" R64_STORES2                                                                       \
    "s_mov_b32 s50, 0\n\t"                                                                                              \
    "s_branch .Lr64first_%=\n\t"                                                                                        \
    ".Lr64hold_%=:\n\t" R64_HOLD ".Lr64first_%=:\n\t"
#define R64_TRIP_END                                                                                                    \
    "s_add_u32 s50, s50, 1\n\t"
#define R64_LAST                                                                                                        \
    R64_TRANSPOSE                                                                                                       \
    "s_cmp_eq_u32 s50, 2\n\t"                                                                                           \
    "s_cbranch_scc1 .Lr64pair_%=\n\t" R64_STORES                                                                        \
    "s_branch .Lr64done_%=\n\t"                                                                                         \
    ".Lr64pair_%=:\n\t" R64_STORES2 ".Lr64done_%=:\n\t"
#define R64_HOLD_CLOBBERS , "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"
#define R64_CLOBBERS                                                                                                    \
    "vcc", "scc", "memory", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21",   \
        "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", \
        "v39", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", \
        "v62", "v63", "v64", "v65", "v66", "v67", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", \
        "s47", "s50"

template <bool PACKED> __global__ void __launch_bounds__(1024) k_decode_lanes_r64x2(const DecParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
        const uint4 *g1 = reinterpret_cast<const uint4 *>(p.table1);
        uint4 *l1 = reinterpret_cast<uint4 *>(smem + t0_bytes);
        for (uint32_t i = threadIdx.x; i < t1_bytes / 16u; i += blockDim.x)
            l1[i] = g1[i];
    }
    __syncthreads();
    if (!lds_starts_at_zero(smem)) { // the asm addresses LDS by raw offsets
        if (threadIdx.x == 0)
            atomicAdd(p.err_count, 1ull << 32);
        return;
    }
    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u;
    if (p.span_reset && blockIdx.x == 0 && threadIdx.x < 2)
        p.span_reset[threadIdx.x] = 0ull;

    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t row = t0_bytes + t1_bytes + wave * kR64WaveLds + lane * kR64RingStride; // LDS byte offset
    const uint32_t m = lane & 3u;
    const uint32_t rq0 = row - m * kR64RingStride + m * 16u; // piece m in the row of the quad's lane 0
    const uint32_t l16 = m * 16u;
    const uint8_t *rowp = smem + row;
    uint32_t maskv = (1u << p.scale_bits) - 1u, sbv = p.scale_bits, t1v = t0_bytes;
    asm volatile("v_mov_b32 %0, %0" : "+v"(maskv)); // VGPR copies: a VALU op with an SGPR operand issues slower
    asm volatile("v_mov_b32 %0, %0" : "+v"(sbv));
    asm volatile("v_mov_b32 %0, %0" : "+v"(t1v));
    uint32_t m12v = 0xfffu;
    asm volatile("v_mov_b32 %0, %0" : "+v"(m12v));
    (void)m12v;
    const uint64_t kL = 1ull << 31;
    const uint64_t m0 = 0x1111111111111111ull; // lane 0 of every quad

    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t cbytes16 = (p.container_bytes + 15u) & ~uint64_t(15);
    uint32_t nbad = 0;
    const uint64_t nbatches = (p.nchunks + 63u) / 64u;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t again = 0; // lanes of the batch in hand whose chunks lay beyond the last trip's window (wave_min_offset)
    for (uint64_t batch_v = (uint64_t)blockIdx.x * waves_per_block + wave; batch_v < nbatches; batch_v += again ? 0 : total_waves) {
        const uint64_t batch = uniform64(batch_v);
        const uint64_t chunk = batch * 64u + lane;
        bool valid = chunk < p.nchunks && (again == 0 || ((again >> lane) & 1u));
        const uint64_t off = valid ? p.offsets[chunk] : 0;
        const uint32_t len = valid ? p.lengths[chunk] : 0;
        if (valid && ((off & 3u) != 0 || len < 16u || off > p.container_bytes || len > p.container_bytes - off)) {
            nbad++;
            valid = false;
        }
        const uint64_t rb = wave_min_offset(off, valid) & ~uint64_t(kLaneLine - 1);
        const bool far = valid && off - rb >= (1u << 30);
        again = __builtin_amdgcn_ballot_w64(far);
        valid = valid && !far;
        // the batch's window onto the container: offsets from rb, reads past the last 16-byte granule return 0
        const uint64_t span = cbytes16 - (rb < cbytes16 ? rb : cbytes16);
        const uint32_t nrec = uniform((uint32_t)(span < 0x7ffffff0ull ? span : 0x7ffffff0ull));
        const uint64_t wbase = uniform64(cbase + rb);
        const u32x4 rsrc4 = {uniform((uint32_t)wbase), uniform((uint32_t)(wbase >> 32)) & 0xffffu, nrec, 0x00020000u};
        // store addresses: instruction t writes the line of the quad's lane t, this lane its piece m
        const uint64_t dst = reinterpret_cast<uint64_t>(p.out) + chunk * p.chunk_syms;
        uint64_t at[4], vq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int src = (int)((lane & ~3u) + t);
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)dst, src), hi = (uint32_t)__shfl((int)(uint32_t)(dst >> 32), src);
            at[t] = (((uint64_t)hi << 32) | lo) + l16;
            vq[t] = __builtin_amdgcn_ballot_w64(__shfl((int)valid, src) != 0);
        }

        uint32_t cur = valid ? (uint32_t)(off - rb) : 0u; // position in the batch's window (ring index = cur & 127)
        const uint32_t cur0 = cur;
        uint32_t ld = valid ? (cur & ~(kLaneLine - 1u)) : 0x40000000u; // next line not yet requested (never, without a chunk)
        // opening: the first two lines (the states are the first 16 bytes of the stream, rans64.h:251-262)
        asm volatile("v_sub_u32 v46, %[ld], %[cur]\n\t" R64_FETCH R64_COMMIT "v_sub_u32 v46, %[ld], %[cur]\n\t" R64_FETCH R64_COMMIT
                     "s_waitcnt lgkmcnt(0)"
                     : [ld] "+v"(ld)
                     : [cur] "v"(cur), [l16] "v"(l16), [rq0] "v"(rq0), [m0] "s"(m0), [rsrc] "s"(rsrc4)
                     : R64_CLOBBERS);
        uint64_t xA, xB, win;
        {
            const u32x2 a = *reinterpret_cast<const u32x2 *>(rowp + (cur & 127u));
            const u32x2 b = *reinterpret_cast<const u32x2 *>(rowp + ((cur + 8u) & 127u));
            xA = (uint64_t)a.x | ((uint64_t)a.y << 32);
            xB = (uint64_t)b.x | ((uint64_t)b.y << 32);
            cur += 16u;
            const uint32_t w0 = *reinterpret_cast<const uint32_t *>(rowp + (cur & 127u));
            const uint32_t w1 = *reinterpret_cast<const uint32_t *>(rowp + ((cur + 4u) & 127u));
            win = (uint64_t)w0 | ((uint64_t)w1 << 32);
        }
        uint32_t trips = uniform(p.chunk_syms >> 6);
        if constexpr (PACKED) {
        asm volatile("s_mov_b64 s[40:41], 0\n\t"
                     "s_mov_b64 s[42:43], 0\n\t"
                     "s_mov_b64 s[44:45], 0\n\t"
                     "s_mov_b64 s[46:47], 0\n\t"
                     "s_mov_b32 s50, 0\n\t"
                     ".Lr64trip_%=:\n\t"
                     R64_COMMIT
                     R64_TRIP_TOP
#define R64_BOUNDARY(N)                                                                                                 \
    "v_sub_u32 v46, %[ld], %[cur]\n\t"                                                                                  \
    "v_cmp_gt_i32 vcc, 48, v46\n\t"                                                                                     \
    "s_cbranch_vccz .Lr64ok" N "_%=\n\t"                                                                                \
    /* side path (some lane is about to starve): everybody with room asks now, and we wait */                          \
    R64_FETCH R64_COMMIT                                                             \
    "v_sub_u32 v46, %[ld], %[cur]\n\t"                                                                                  \
    ".Lr64ok" N "_%=:\n\t" R64_FETCH
                     R64_BOUNDARY("0") R64P_GROUP("v8", "v9", "v10", "v11")
                     R64_COMMIT R64_BOUNDARY("1") R64P_GROUP("v12", "v13", "v14", "v15")
                     R64_COMMIT R64_BOUNDARY("2") R64P_GROUP("v16", "v17", "v18", "v19")
                     R64_COMMIT R64_BOUNDARY("3") R64P_GROUP("v20", "v21", "v22", "v23")
                     R64_TRIP_END
                     "s_sub_u32 %[trips], %[trips], 1\n\t"
                     "s_cmp_lg_u32 %[trips], 0\n\t"
                     "s_cbranch_scc1 .Lr64trip_%=\n\t"
                     R64_LAST
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+{v[40:41]}"(xA), "+{v[42:43]}"(xB), "+{v[44:45]}"(win), [cur] "+v"(cur), [ld] "+v"(ld), [at0] "+v"(at[0]),
                       [at1] "+v"(at[1]), [at2] "+v"(at[2]), [at3] "+v"(at[3]), [trips] "+s"(trips)
                     : [maskv] "v"(maskv), [sbv] "v"(sbv), [m12v] "v"(m12v), [selA] "s"(0x0c0c0703u), [selB] "s"(0x07030c0cu), [row] "v"(row), [l16] "v"(l16), [rq0] "v"(rq0),
                       [kL] "s"(kL), [m0] "s"(m0), [vq0] "s"(vq[0]), [vq1] "s"(vq[1]), [vq2] "s"(vq[2]), [vq3] "s"(vq[3]),
                       [rsrc] "s"(rsrc4)
                     : R64_CLOBBERS R64_HOLD_CLOBBERS);
        } else {
        asm volatile("s_mov_b64 s[40:41], 0\n\t"
                     "s_mov_b64 s[42:43], 0\n\t"
                     "s_mov_b64 s[44:45], 0\n\t"
                     "s_mov_b64 s[46:47], 0\n\t"
                     "s_mov_b32 s50, 0\n\t"
                     ".Lr64trip_%=:\n\t"
                     R64_COMMIT
                     R64_TRIP_TOP
#define R64_BOUNDARY(N)                                                                                                 \
    "v_sub_u32 v46, %[ld], %[cur]\n\t"                                                                                  \
    "v_cmp_gt_i32 vcc, 48, v46\n\t"                                                                                     \
    "s_cbranch_vccz .Lr64ok" N "_%=\n\t"                                                                                \
    /* side path (some lane is about to starve): everybody with room asks now, and we wait */                          \
    R64_FETCH R64_COMMIT                                                             \
    "v_sub_u32 v46, %[ld], %[cur]\n\t"                                                                                  \
    ".Lr64ok" N "_%=:\n\t" R64_FETCH
                     R64_BOUNDARY("0") R64_GROUP("v8", "v9", "v10", "v11")
                     R64_COMMIT R64_BOUNDARY("1") R64_GROUP("v12", "v13", "v14", "v15")
                     R64_COMMIT R64_BOUNDARY("2") R64_GROUP("v16", "v17", "v18", "v19")
                     R64_COMMIT R64_BOUNDARY("3") R64_GROUP("v20", "v21", "v22", "v23")
                     R64_TRIP_END
                     "s_sub_u32 %[trips], %[trips], 1\n\t"
                     "s_cmp_lg_u32 %[trips], 0\n\t"
                     "s_cbranch_scc1 .Lr64trip_%=\n\t"
                     R64_LAST
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+{v[40:41]}"(xA), "+{v[42:43]}"(xB), "+{v[44:45]}"(win), [cur] "+v"(cur), [ld] "+v"(ld), [at0] "+v"(at[0]),
                       [at1] "+v"(at[1]), [at2] "+v"(at[2]), [at3] "+v"(at[3]), [trips] "+s"(trips)
                     : [maskv] "v"(maskv), [sbv] "v"(sbv), [t1v] "v"(t1v), [row] "v"(row), [l16] "v"(l16), [rq0] "v"(rq0),
                       [kL] "s"(kL), [m0] "s"(m0), [vq0] "s"(vq[0]), [vq1] "s"(vq[1]), [vq2] "s"(vq[2]), [vq3] "s"(vq[3]),
                       [rsrc] "s"(rsrc4)
                     : R64_CLOBBERS R64_HOLD_CLOBBERS);
        }
        // RansDec end state: both states back at L, every byte of the chunk consumed (main64.cpp has no check; ours)
        if (valid && (xA != kL || xB != kL || cur - cur0 != len))
            nbad++;
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}
#undef R64_BOUNDARY

template <int FMT, int NW>
__global__ void __launch_bounds__(256) k_decode_lanes(const DecParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u;
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u;
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
        const uint4 *g1 = reinterpret_cast<const uint4 *>(p.table1);
        uint4 *l1 = reinterpret_cast<uint4 *>(smem + t0_bytes);
        for (uint32_t i = threadIdx.x; i < t1_bytes / 16u; i += blockDim.x)
            l1[i] = g1[i];
    }
    __syncthreads();

    DecTables<FMT> T;
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);
    if (!lds_starts_at_zero(smem)) { // cannot happen without static LDS; never decode on a wrong assumption
        if (threadIdx.x == 0)
            atomicAdd(p.err_count, 1ull << 32);
        return;
    }

    if (p.work_counter_reset && blockIdx.x == 0 && threadIdx.x < kWorkPools)
        p.work_counter_reset[threadIdx.x * kWorkPoolStride] = 0u; // keep the wave kernels' counter ring consistent
    if (p.span_reset && blockIdx.x == 0 && threadIdx.x < 2)
        p.span_reset[threadIdx.x] = 0ull;
    const uint8_t RANS_GLOBAL *cbase = (const uint8_t RANS_GLOBAL *)p.container;
    const uint64_t glimit = (reinterpret_cast<uint64_t>(p.container) + p.container_bytes + 15u) & ~uint64_t(15);
    const bool wide_out = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.out) | p.chunk_syms) & 15u) == 0;
    uint32_t nbad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; chunk < p.nchunks; chunk += stride) {
        const uint64_t off = p.offsets[chunk];
        const uint32_t len = p.lengths[chunk];
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        if ((off & (Tr::kUnit - 1u)) != 0 || len < NW * Tr::kStateBytes || off > p.container_bytes || len > p.container_bytes - off) {
            nbad++;
            continue;
        }
        const uint8_t RANS_GLOBAL *src = cbase + off;
        uint8_t RANS_GLOBAL *dst = (uint8_t RANS_GLOBAL *)p.out + first * p.sym_bytes;

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l) { // RansDecInit order: lane 0's state first
            if constexpr (FMT == FMT_R64) {
                const u32x2 v = reinterpret_cast<const u32x2 RANS_GLOBAL *>(src)[l];
                x[l] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            } else {
                x[l] = reinterpret_cast<const uint32_t RANS_GLOBAL *>(src)[l];
            }
        }
        LaneWindow W;
        W.open<(int)Tr::kUnit>(reinterpret_cast<uint64_t>(p.container) + off + NW * Tr::kStateBytes, glimit);
        bool bad = false;

        const uint32_t rounds = nsym / NW;
        uint32_t i = 0; // symbol index inside the chunk
        if (wide_out) {
            // 16 symbols per 16-byte store: 16/NW rounds per group
            constexpr int kRoundsPerGroup = 16 / NW;
            const uint32_t groups = rounds / kRoundsPerGroup;
            for (uint32_t g = 0; g < groups; ++g) {
                u32x4 pack = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int rr = 0; rr < kRoundsPerGroup; ++rr) {
#pragma unroll
                    for (int l = 0; l < NW; ++l) {
                        uint32_t sy = dec_step<FMT>(T, x[l]);
                        if constexpr (Tr::kSymByte == 3)
                            sy >>= 24;
                        const int pos = rr * NW + l;
                        pack[pos / 4] |= (sy & 0xffu) << (8 * (pos % 4));
                    }
#pragma unroll
                    for (int l = 0; l < NW; ++l)
                        lane_renorm<FMT>(x[l], W, true);
                }
                __builtin_nontemporal_store(pack, reinterpret_cast<u32x4 RANS_GLOBAL *>(dst + i));
                i += 16;
            }
        }
        // remaining rounds + tail: element stores
        while (i < nsym) {
            const uint32_t cnt = nsym - i < (uint32_t)NW ? nsym - i : (uint32_t)NW;
#pragma unroll
            for (int l = 0; l < NW; ++l)
                if ((uint32_t)l < cnt) {
                    uint32_t sy = dec_step<FMT>(T, x[l]);
                    if constexpr (Tr::kSymByte == 3)
                        sy >>= 24;
                    if (p.sym_bytes == 1)
                        dst[i + l] = (uint8_t)sy;
                    else
                        reinterpret_cast<uint16_t RANS_GLOBAL *>(dst)[i + l] = (uint16_t)sy;
                }
#pragma unroll
            for (int l = 0; l < NW; ++l)
                lane_renorm<FMT>(x[l], W, (uint32_t)l < cnt);
            i += cnt;
        }
#pragma unroll
        for (int l = 0; l < NW; ++l)
            bad = bad || (x[l] != Tr::kL);
        if (bad || W.used + NW * Tr::kStateBytes != len)
            nbad++;
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}

// one symbol of the sequential reference encoder (RansEncPut / RansWordEncPut / Rans64EncPut /
// RansEncPutAlias) for a lane-private state and write pointer
template <int FMT>
__device__ __forceinline__ void lane_put(typename FmtTraits<FMT>::state_t &x, uint32_t sym, const uint4 *recs,
                                         const EncParams &p, uint8_t RANS_GLOBAL *&wp, bool &bad)
{
    const bool known = sym < p.nsyms;
    const uint4 rec = recs[known ? sym : 0u];
    const uint32_t freq = (FMT == FMT_R64 || FMT == FMT_BYTE) ? (rec.x & 0xffffffu) : rec.x, start = rec.y, rcp = rec.z;
    if (!known || freq == 0) {
        bad = true;
        return;
    }
    if constexpr (FMT == FMT_WORD) {
        uint32_t y = x;
        if (y >= (freq << 20)) {
            wp -= 2;
            *reinterpret_cast<uint16_t RANS_GLOBAL *>(wp) = (uint16_t)y;
            y >>= 16;
        }
        x = enc_update_word(y, rec);
    } else if constexpr (FMT == FMT_R64) {
        uint64_t y = x;
        if (y >= (((uint64_t)freq) << (63u - p.scale_bits))) {
            wp -= 4;
            *reinterpret_cast<uint32_t RANS_GLOBAL *>(wp) = (uint32_t)y;
            y >>= 32;
        }
        x = enc_update_r64(y, rec, p.scale_bits);
    } else {
        uint32_t y = x;
        const uint32_t x_max = freq << (31u - p.scale_bits);
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (y >= x_max) {
                *--wp = (uint8_t)y;
                y >>= 8;
            }
        if constexpr (FMT == FMT_ALIAS) {
            uint32_t q, rem;
            divmod_rcp(y, freq, rcp, q, rem);
            x = (q << p.scale_bits) + p.alias_remap[rem + start];
        } else {
            x = enc_update_byte(y, rec, p.scale_bits);
        }
    }
}

// ---------------------------------------------------------------------------
// Staged lane-per-stream encoder: the mirror image of k_decode_lanes_staged.  Per-lane stores of
// every emitted unit reached HBM as partial lines (measured 6.4x the stream bytes on the write
// side) and per-lane 16-byte symbol loads pulled whole lines (3.6x).  Here
//   * symbols: the wave loads one 64-byte block of each of its 64 chunks with coalesced 16-byte
//     loads (4 lanes per chunk) into per-lane rows in LDS; every lane then walks its row from
//     the top, 16 symbols per ds_read_b128;
//   * stream: units go into a 128-byte ring per lane (two 64-byte lines, written downwards); after
//     every 16 symbols (at most 64 bytes emitted) a line that has filled up is written out by 4
//     lanes with 16-byte stores -- whole 64-byte lines, each written once.
// Slots are whole lines (api.cpp, encode_slot_bytes); what lies below the stream start inside the
// lowest line is never read.
// ---------------------------------------------------------------------------
constexpr uint32_t kEncRowStride = 80; // 64 symbol bytes, rows 16-byte aligned, 20 dwords apart
constexpr uint32_t kEncWaveLds = 64 * kEncRowStride + 64 * kLaneRingStride + 64 * 4;

// slot layout (EncParams::slot_layout): the chunk stays in its slot, its stream is [slot end - len, slot end)
// (sized slots, EncParams::ovf_ctl: a lane whose chunk did not fit its slot -- ovf -- lists it for the redo launch instead;
//  whatever it stored lies inside its own slot and counts for nothing)
__device__ __forceinline__ void lanes_publish_slot(const EncParams &p, uint64_t chunk, uint32_t len, bool ovf = false)
{
    if (ovf) {
        p.ovf_list[atomicAdd(p.ovf_ctl, 1u)] = (uint32_t)chunk;
        return;
    }
    if (p.slot_layout) {
        p.offsets[chunk] = (chunk + 1u) * p.slot_bytes - len;
        if (chunk + 1 == p.nchunks)
            p.offsets[p.nchunks] = p.nchunks * p.slot_bytes;
    }
}

template <int FMT> struct LaneOut {
    uint8_t *row;  // this lane's output ring in LDS
    uint32_t w;    // write offset inside the chunk's slot, moves down
    template <int UNIT> __device__ __forceinline__ void emit(uint32_t v)
    {
        w -= UNIT;
        uint8_t *at = row + (w & (2 * kLaneLine - 1));
        if constexpr (UNIT == 4)
            *reinterpret_cast<uint32_t *>(at) = v;
        else if constexpr (UNIT == 2)
            *reinterpret_cast<uint16_t *>(at) = (uint16_t)v;
        else
            *at = (uint8_t)v;
    }
};

// one symbol of the sequential reference encoder for a lane-private state, emitting into the ring
template <int FMT>
__device__ __forceinline__ void lane_put_staged(typename FmtTraits<FMT>::state_t &x, uint32_t sym, const uint4 *recs,
                                                const EncParams &p, LaneOut<FMT> &O, bool &bad)
{
    const bool known = sym < p.nsyms;
    const uint4 rec = recs[known ? sym : 0u];
    const uint32_t freq = (FMT == FMT_R64 || FMT == FMT_BYTE) ? (rec.x & 0xffffffu) : rec.x, start = rec.y, rcp = rec.z;
    if (!known || freq == 0) {
        bad = true;
        return;
    }
    if constexpr (FMT == FMT_WORD) {
        uint32_t y = x;
        if (y >= (freq << 20)) { // rans_word_sse41.h:85-89
            O.template emit<2>(y);
            y >>= 16;
        }
        x = enc_update_word(y, rec);
    } else if constexpr (FMT == FMT_R64) {
        uint64_t y = x;
        if (y >= (((uint64_t)freq) << (63u - p.scale_bits))) { // rans64.h:83-88
            O.template emit<4>((uint32_t)y);
            y >>= 32;
        }
        x = enc_update_r64(y, rec, p.scale_bits);
    } else {
        uint32_t y = x;
        const uint32_t x_max = freq << (31u - p.scale_bits); // rans_byte.h:64-70
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (y >= x_max) {
                O.template emit<1>(y);
                y >>= 8;
            }
        if constexpr (FMT == FMT_ALIAS) {
            uint32_t q, rem;
            divmod_rcp(y, freq, rcp, q, rem);
            x = (q << p.scale_bits) + p.alias_remap[rem + start];
        } else {
            x = enc_update_byte(y, rec, p.scale_bits);
        }
    }
}

// ---------------------------------------------------------------------------
// Fused placement of the lane encoders (EncParams::status != NULL): no k_layout / k_compact_small afterwards.  The
// container's layout is the oracle's (chunk c starts at the sum of the 16-byte aligned lengths before it), so the
// chunks of a batch can be copied to their place once the total of everything before the batch is known.
//
// The unit of the scan is a ROUND OF A BLOCK: its C coding waves take C consecutive batches (one claim of the block's
// scanner wave on a counter behind the status words), code them, and post their totals in LDS; the scanner adds them
// up, publishes status[unit] = AGGREGATE | total, looks back over the units before it (decoupled look-back,
// device_common.hpp; kScanWords * 64 units per step) until it meets a PREFIX, publishes its own PREFIX and leaves the
// place of every coder's batch in LDS; kLaneCopyWaves copier waves -- no LDS of their own, so they come on top of the
// coding waves the rings allow -- then move the round's batches to the container while the coders are a round ahead.
// 256 units (one per CU) finish together and one look-back step resolves them all.
//
// What this replaced, all of them measured on config 2 (0.36-0.40 ms with the two extra kernels): the coding wave
// placing its own batch at once (0.54: every wave of the first round finishes at the same moment and the prefix travels
// 64 batches per memory round trip); a copier wave per block that scans and copies batch by batch through a mailbox
// (0.375-0.52: 6-8 us per batch where the coders deliver one every 8 us); a scanner wave per block working batch by
// batch while the coders copy (0.40-0.42: with 2816 batches ending together a scanner walks back thousands of status
// words for each of its 11 batches); the round-of-a-block units with the coders copying their previous batch (0.42-0.47:
// the protocol costs 0.01 ms then, the copy 0.13 -- these kernels run 3-4 waves per SIMD, each bound by its own
// dependency chain, and a wave that spends 40 us per batch in memory round trips is not replaced by anybody).  With
// the copier waves the fused launch is as fast as the three kernels (0.34-0.37 ms against 0.35-0.37: 11 coding waves in
// three rounds instead of 16 in two pay for what the copy no longer costs) -- hence still opt-in.
// Units are claimed in ascending order by running blocks and a scanner waits only for smaller units: the smallest
// unfinished unit always belongs to a running block.  The copy is quad-cooperative like every other access of these
// kernels: instruction t moves 64 bytes of the chunk of the quad's lane t (16 bytes per lane, source unaligned),
// kLaneCopyDepth pieces per chunk in flight.
//
// A word on control flow: none of the loops below ends in an `if (lane == 0) { ... }`.  With such a tail the compiler
// let lanes 1..63 run ahead into the next iteration -- whose readfirstlane then read a lane that had not taken part --
// and parked lane 0's store behind the loop, for ever (found with the watchdogs below: flags 0x40, every status word
// still AGGREGATE).  Stores that one lane would do are done by all of them with the same value.
// ---------------------------------------------------------------------------
constexpr uint32_t kLaneCopyWaves = 3; // copier waves per block
constexpr uint32_t kLaneCopiers = 1 + kLaneCopyWaves; // scanner + copier waves (on top of the coding waves, <= 16 in all)
constexpr int kLaneCopyDepth = 4;
constexpr int kScanWords = 4;

struct LaneRounds { // the block's control words in LDS (EncParams::mailbox_off), zero at kernel start; index = round & 1
    // (units: round & 3 -- the scanner names round r + 1's unit while a copier may still be busy with round r - 1)
    uint32_t unit[4];     // unit claimed for the round ...
    uint32_t unit_seq[4]; // ... valid when this is round + 1
    uint32_t posted[2];   // coding waves that have posted their total
    uint32_t base_seq[2]; // bases[] valid when this is round + 1
    uint32_t copied[2];   // copier waves that are through with the round
    uint32_t totals[2][16];
    unsigned long long bases[2][16];
};
static_assert(sizeof(LaneRounds) <= kEncMailboxBytes, "LaneRounds must fit the LDS reserved for the placement");

struct LaneCoder { // a coding wave's view
    uint32_t round; // rounds begun
};

// poll an LDS word until it has the value (whole wave; gives up after kWaitTicks and says so in flags)
__device__ __forceinline__ bool lanes_wait_lds(volatile uint32_t *word, uint32_t value, uint32_t *flags, uint32_t flag_bit, uint32_t lane,
                                               unsigned long long wait_ticks)
{
    for (SpinWatch watch(wait_ticks);;) {
        if (uniform(*word) == value)
            return true;
        if (watch.expired(flags)) {
            atomicOr(flags, lane == 0 ? flag_bit : 0u);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// coding wave, start of a round: its batch, ~0 - 1 when it has none in this (the last) unit, ~0 when the launch is over
__device__ __forceinline__ uint64_t lanes_round_begin(const EncParams &p, LaneRounds *ctl, const LaneCoder &cs, uint32_t wave,
                                                      uint32_t coders, uint32_t lane)
{
    const uint32_t r = cs.round;
    if (!lanes_wait_lds(&ctl->unit_seq[r & 3u], r + 1u, p.flags, 128u, lane, p.wait_ticks))
        return ~0ull;
    const uint64_t first = p.batch_begin + (uint64_t)uniform(*(volatile uint32_t *)&ctl->unit[r & 3u]) * coders;
    if (first >= p.batch_end)
        return ~0ull;
    return first + wave < p.batch_end ? first + wave : ~0ull - 1u;
}

// coding wave: offsets[] of a batch and its copy out of the scratch slots; base = where the batch starts
__device__ __forceinline__ void lanes_copy_batch(const EncParams &p, uint64_t batch, uint32_t len, unsigned long long base, uint32_t lane)
{
    const uint64_t chunk = batch * 64u + lane;
    const bool valid = chunk < p.nchunks;
    const uint32_t alen = (len + 15u) & ~15u;
    uint32_t incl = alen; // inclusive sum over the lanes below
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        incl += lane >= (uint32_t)d ? o : 0u;
    }
    const uint64_t off = base + incl - alen;
    if (valid) {
        p.offsets[chunk] = off;
        if (chunk + 1 == p.nchunks)
            p.offsets[p.nchunks] = off + len;
    }
    if (__builtin_amdgcn_ballot_w64(valid && off + len > p.out_cap) != 0) { // (wave-uniform)
        atomicOr(p.flags, lane == 0 ? 2u : 0u);
        return;
    }
    const uint32_t m = lane & 3u;
    const uint64_t sa = reinterpret_cast<uint64_t>(p.scratch) + (chunk + 1u) * p.slot_bytes - len; // a stream ends at its slot's end
    const uint64_t da = reinterpret_cast<uint64_t>(p.out) + off;
    uint64_t s_t[4], d_t[4];
    uint32_t n16[4], most = 0; // (the last piece of a chunk may read up to 15 bytes of the next slot: the scratch is padded)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int from = (int)((lane & ~3u) + t);
        s_t[t] = (uint64_t)(uint32_t)__shfl((int)(uint32_t)sa, from, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(sa >> 32), from, 64) << 32);
        d_t[t] = (uint64_t)(uint32_t)__shfl((int)(uint32_t)da, from, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(da >> 32), from, 64) << 32);
        n16[t] = (uint32_t)__shfl((int)(valid ? alen >> 4 : 0u), from, 64);
        most = n16[t] > most ? n16[t] : most;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)most, d, 64);
        most = o > most ? o : most;
    }
    most = uniform(most);
    for (uint32_t i0 = 0; i0 < most; i0 += 4u * kLaneCopyDepth) {
        u32x4 v[4][kLaneCopyDepth];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < kLaneCopyDepth; ++j) {
                const uint32_t i = i0 + 4u * j + m;
                if (i < n16[t])
                    v[t][j] = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(s_t[t] + 16ull * i));
            }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < kLaneCopyDepth; ++j) {
                const uint32_t i = i0 + 4u * j + m;
                if (i < n16[t])
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(d_t[t] + 16ull * i) = v[t][j];
            }
    }
}

// coding wave, end of a round: post the total (len: this lane's chunk, 0 without one)
__device__ __forceinline__ void lanes_round_end(const EncParams &p, LaneRounds *ctl, LaneCoder &cs, uint32_t wave, uint32_t lane,
                                                uint64_t batch, uint32_t len)
{
    (void)batch;
    const uint32_t r = cs.round;
    uint32_t sum = (len + 15u) & ~15u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
        sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // lengths[] and every flushed line of the batch have left the wave
    *(volatile uint32_t *)&ctl->totals[r & 1u][wave] = sum; // (every lane, the same value)
    atomicAdd(&ctl->posted[r & 1u], lane == 0 ? 1u : 0u);
    cs.round = r + 1u;
}

// copier wave number k of the block: when the places of a round's batches are known, the batches k, k + kLaneCopyWaves,
// ... of the round go to the container
__device__ __forceinline__ void lanes_copier(const EncParams &p, LaneRounds *ctl, uint32_t k, uint32_t lane, uint32_t coders)
{
    for (uint32_t r = 0;; ++r) {
        if (!lanes_wait_lds(&ctl->unit_seq[r & 3u], r + 1u, p.flags, 128u, lane, p.wait_ticks))
            return;
        const uint64_t first = p.batch_begin + (uint64_t)uniform(*(volatile uint32_t *)&ctl->unit[r & 3u]) * coders;
        if (first >= p.batch_end)
            return;
        if (!lanes_wait_lds(&ctl->base_seq[r & 1u], r + 1u, p.flags, 64u, lane, p.wait_ticks))
            return;
        for (uint32_t w = k; w < coders && first + w < p.batch_end; w += kLaneCopyWaves) {
            const volatile uint32_t *b = reinterpret_cast<const volatile uint32_t *>(&ctl->bases[r & 1u][w]);
            const unsigned long long base = (unsigned long long)uniform(b[0]) | ((unsigned long long)uniform(b[1]) << 32);
            const uint64_t chunk = (first + w) * 64u + lane;
            // (written by a wave of this CU before it posted its total, read through L2)
            const uint32_t len = chunk < p.nchunks ? __hip_atomic_load(p.lengths + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            lanes_copy_batch(p, first + w, len, base, lane);
        }
        atomicAdd(&ctl->copied[r & 1u], lane == 0 ? 1u : 0u);
    }
}

// the scanner wave's life
__device__ __forceinline__ void lanes_scanner(const EncParams &p, LaneRounds *ctl, uint32_t lane, uint32_t coders)
{
    const uint64_t nunits = (p.batch_end - p.batch_begin + coders - 1u) / coders;
    unsigned int *counter = reinterpret_cast<unsigned int *>(p.status + ((p.nchunks + 63u) / 64u) + 8u * p.claim_slot);
    auto claim = [&]() {
        uint32_t got = 0;
        if (lane == 0)
            got = atomicAdd(counter, 1u);
        return uniform(got);
    };
    const uint32_t slot = lane < 15u ? lane : 15u; // (coders <= 15: slot 15 is nobody's)
    uint32_t u = claim();
    *(volatile uint32_t *)&ctl->unit[0] = u;
    *(volatile uint32_t *)&ctl->unit_seq[0] = 1u;
    for (uint32_t r = 0; u < nunits; ++r) {
        const uint32_t un = claim(); // the coders find their next unit as soon as they are through with this one
        *(volatile uint32_t *)&ctl->unit[(r + 1u) & 3u] = un;
        *(volatile uint32_t *)&ctl->unit_seq[(r + 1u) & 3u] = r + 2u;
        if (!lanes_wait_lds(&ctl->posted[r & 1u], coders, p.flags, 16u, lane, p.wait_ticks))
            return;
        const uint32_t mine = *(volatile uint32_t *)&ctl->totals[r & 1u][slot];
        const uint32_t t = lane < coders ? mine : 0u;
        uint32_t incl = t;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
            incl += lane >= (uint32_t)d ? o : 0u;
        }
        const unsigned long long total = (uint32_t)__shfl((int)incl, 15, 64);
        *(volatile uint32_t *)&ctl->posted[r & 1u] = 0u; // (for round r + 2)
        const uint64_t gu = p.unit_base + u;
        __hip_atomic_store(p.status + gu, kStAggregate | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (every lane)
        unsigned long long base = 0;
        SpinWatch watch(p.wait_ticks);
        for (uint64_t j = gu;;) { // status[j-1], status[j-2], ... are still to be added
            unsigned long long st[kScanWords];
            uint64_t ready[kScanWords], pref[kScanWords];
#pragma unroll
            for (int k = 0; k < kScanWords; ++k) {
                const uint64_t back = lane + 64u * k; // distance - 1
                st[k] = kStPrefix; // virtual predecessors of unit 0: an inclusive prefix of 0
                if (back < j)
                    st[k] = __hip_atomic_load(p.status + (j - 1 - back), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < kScanWords; ++k) {
                ready[k] = __builtin_amdgcn_ballot_w64((st[k] >> 62) != 0);
                pref[k] = __builtin_amdgcn_ballot_w64((st[k] >> 62) == 2);
            }
            // the nearest PREFIX: word K, lane `first`; everything nearer must be there (AGGREGATE or PREFIX)
            int K = kScanWords;
            uint32_t first = 64u;
            bool all_ready = true;
#pragma unroll
            for (int k = 0; k < kScanWords; ++k) {
                if (K == kScanWords) {
                    if (pref[k]) {
                        K = k;
                        first = (uint32_t)__builtin_ctzll(pref[k]);
                        const uint64_t need = first >= 63u ? ~0ull : ((2ull << first) - 1ull); // lanes 0 .. first
                        all_ready = all_ready && (ready[k] & need) == need;
                    } else {
                        all_ready = all_ready && ready[k] == ~0ull;
                    }
                }
            }
            if (!all_ready) { // a unit in that range is still being coded
                if (watch.expired(p.flags)) { // (a protocol error must not hang the GPU)
                    atomicOr(p.flags, lane == 0 ? 32u : 0u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            unsigned long long v = 0;
#pragma unroll
            for (int k = 0; k < kScanWords; ++k)
                if (k < K || (k == K && lane <= first))
                    v += st[k] & kStValue;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
                const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
                v += (unsigned long long)lo | ((unsigned long long)hi << 32);
            }
            base += uniform64(v);
            if (K < kScanWords)
                break;
            j -= 64u * kScanWords;
        }
        __hip_atomic_store(p.status + gu, kStPrefix | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (every lane)
        const unsigned long long place = base + incl - t; // of coder `lane`'s batch
        if (r >= 2u) { // the copiers are through with round r - 2, whose places these words still hold
            if (!lanes_wait_lds(&ctl->copied[r & 1u], kLaneCopyWaves, p.flags, 16u, lane, p.wait_ticks))
                return;
            *(volatile uint32_t *)&ctl->copied[r & 1u] = 0u;
        }
        volatile uint32_t *b = reinterpret_cast<volatile uint32_t *>(&ctl->bases[r & 1u][slot]);
        b[0] = (uint32_t)place; // (lanes >= 15 all write slot 15)
        b[1] = (uint32_t)(place >> 32);
        *(volatile uint32_t *)&ctl->base_seq[r & 1u] = r + 1u;
        u = un;
    }
}

template <int FMT, int NW>
__global__ void __launch_bounds__(1024) k_encode_lanes_staged(const EncParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < p.nsyms; i += blockDim.x)
            l[i] = g[i];
        if (p.status) // (a block may be as small as one coding wave + the scanner: 128 threads for 132 words)
            for (uint32_t i = threadIdx.x; i < kEncMailboxBytes / 4u; i += blockDim.x)
                reinterpret_cast<uint32_t *>(smem + p.mailbox_off)[i] = 0u;
    }
    __syncthreads();
    const uint4 *recs = reinterpret_cast<const uint4 *>(smem);
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const bool fused = p.status != nullptr;
    const uint32_t waves_per_block = (blockDim.x >> 6) - (fused ? kLaneCopiers : 0u); // coding waves
    LaneRounds *ctl = reinterpret_cast<LaneRounds *>(smem + p.mailbox_off);
    if (fused && wave >= waves_per_block) { // ---- the scanner wave and the copier waves
        if (wave == waves_per_block)
            lanes_scanner(p, ctl, lane, waves_per_block);
        else
            lanes_copier(p, ctl, wave - waves_per_block - 1u, lane, waves_per_block);
        return;
    }
    uint8_t *rows = smem + p.nsyms * (uint32_t)sizeof(EncRec) + wave * kEncWaveLds;
    uint8_t *rings = rows + 64u * kEncRowStride;
    uint32_t *req = reinterpret_cast<uint32_t *>(rings + 64u * kLaneRingStride);
    const uint32_t part = lane & 3u, grp = lane >> 2;
    const uint32_t slot_lines = (uint32_t)(p.slot_bytes / kLaneLine);

    bool bad = false;
    LaneCoder cs{0};
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t batch_v = p.batch_begin + (uint64_t)blockIdx.x * waves_per_block + wave;; batch_v += total_waves) {
        if (fused) { // the block's scanner hands out the rounds
            batch_v = lanes_round_begin(p, ctl, cs, wave, waves_per_block, lane);
            if (batch_v == ~0ull - 1u) { // nothing for this wave in the last unit
                lanes_round_end(p, ctl, cs, wave, lane, batch_v, 0u);
                continue;
            }
        }
        if (batch_v >= p.batch_end)
            break;
        const uint64_t chunk0 = uniform64(batch_v) * 64u;
        const uint64_t chunk = chunk0 + lane;
        const bool valid = chunk < p.nchunks;
        auto syms_of = [&](uint64_t c) -> uint32_t { // symbols in chunk c (0 past the end)
            if (c >= p.nchunks)
                return 0u;
            const uint64_t first = c * p.chunk_syms;
            return (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        };
        const uint32_t nsym = syms_of(chunk);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + chunk * (uint64_t)p.chunk_syms;
        uint8_t RANS_GLOBAL *slots0 = (uint8_t RANS_GLOBAL *)p.scratch + chunk0 * p.slot_bytes; // wave-uniform

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l)
            x[l] = Tr::kL;
        LaneOut<FMT> O;
        O.row = rings + lane * kLaneRingStride;
        O.w = (uint32_t)p.slot_bytes;
        uint32_t flushed = slot_lines; // lines [flushed, slot_lines) are in memory

        // the whole wave takes part: lanes publish the line they have filled (or, at the end, the lines
        // that hold anything), lane (4 g + part) writes 16 bytes of chunk (16 j + g)'s line
        auto flush = [&](bool final) {
            const bool need = valid && flushed != 0u &&
                              (final ? O.w < flushed * kLaneLine : O.w <= (flushed - 1u) * kLaneLine);
            req[lane] = need ? (((flushed - 1u) << 1) | 1u) : 0u;
            if (need)
                flushed -= 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = 16u * j + grp;
                const uint32_t r = req[q]; // LDS ops of one wave execute in order
                if (r & 1u) {
                    const uint32_t line = r >> 1;
                    const uint8_t *at = rings + q * kLaneRingStride + (line & 1u) * kLaneLine + part * 16u;
                    const u32x2 a = reinterpret_cast<const u32x2 *>(at)[0], b = reinterpret_cast<const u32x2 *>(at)[1];
                    u32x4 RANS_GLOBAL *o = reinterpret_cast<u32x4 RANS_GLOBAL *>(
                        slots0 + (uint64_t)q * p.slot_bytes + (uint64_t)line * kLaneLine + part * 16u);
                    *o = u32x4{a.x, a.y, b.x, b.y};
                }
            }
        };

        // symbol i belongs to state i mod NW; visit i = nsym-1 .. 0 (main.cpp:233-243).  The top
        // nsym % 16 symbols come one by one from memory, the rest through the staged rows.
        const uint32_t nsym16 = nsym & ~15u;
        for (uint32_t i = nsym; i > nsym16; --i) {
            const uint32_t sym = (uint32_t)src[i - 1];
            const uint32_t l = (i - 1) % NW;
#pragma unroll
            for (int ll = 0; ll < NW; ++ll) // static register indexing
                if ((uint32_t)ll == l)
                    lane_put_staged<FMT>(x[ll], sym, recs, p, O, bad);
        }
        flush(false);

        const uint32_t my_blocks = (nsym16 + 63u) >> 6;
        uint32_t max_blocks = my_blocks;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)max_blocks, d, 64);
            max_blocks = o > max_blocks ? o : max_blocks;
        }
        max_blocks = uniform(max_blocks);
        for (uint32_t k = max_blocks; k-- > 0;) {
            // stage block k of every chunk: 16 bytes per lane, 4 instructions for 64 chunks
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = 16u * j + grp;
                const uint32_t qsyms16 = syms_of(chunk0 + q) & ~15u;
                const uint32_t at = 64u * k + 16u * part;
                if (at + 16u <= qsyms16) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(
                        reinterpret_cast<uint64_t>(p.syms) + (chunk0 + q) * (uint64_t)p.chunk_syms + at));
                    *reinterpret_cast<u32x4 *>(rows + q * kEncRowStride + 16u * part) = v;
                }
            }
            const uint8_t *row = rows + lane * kEncRowStride;
#pragma unroll
            for (int g = 3; g >= 0; --g) {
                if (64u * k + 16u * g + 16u <= nsym16) {
                    const u32x4 cur = *reinterpret_cast<const u32x4 *>(row + 16 * g);
#pragma unroll
                    for (int j = 15; j >= 0; --j) {
                        const uint32_t sym = (cur[j >> 2] >> (8 * (j & 3))) & 0xffu;
                        lane_put_staged<FMT>(x[j % NW], sym, recs, p, O, bad);
                    }
                }
                flush(false);
            }
        }
        // flush states NW-1 .. 0 (lane 0's first in memory), then whatever the ring still holds
        if (valid) {
#pragma unroll
            for (int l = NW - 1; l >= 0; --l) {
                if constexpr (FMT == FMT_R64) {
                    O.template emit<4>((uint32_t)(x[l] >> 32));
                    O.template emit<4>((uint32_t)x[l]);
                } else if constexpr (FMT == FMT_WORD) {
                    O.template emit<2>(x[l] >> 16);
                    O.template emit<2>(x[l]);
                } else {
                    O.template emit<1>(x[l] >> 24);
                    O.template emit<1>(x[l] >> 16);
                    O.template emit<1>(x[l] >> 8);
                    O.template emit<1>(x[l]);
                }
            }
            p.lengths[chunk] = (uint32_t)p.slot_bytes - O.w;
            // (sized slots: a write offset that went below 0 has wrapped -- the flushes stopped at line 0, the ring took the rest)
            lanes_publish_slot(p, chunk, (uint32_t)p.slot_bytes - O.w, O.w > (uint32_t)p.slot_bytes);
        }
        flush(true);
        flush(true);
        if (fused)
            lanes_round_end(p, ctl, cs, wave, lane, chunk0 / 64u, valid ? (uint32_t)p.slot_bytes - O.w : 0u);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
}

// ---------------------------------------------------------------------------
// k_encode_lanes_r64x2: the reference's own 2-way rans64 layout (main64.cpp:224-246, config 2) on its own -- the mirror
// image of k_decode_lanes_r64x2.  The staged kernel above spends 38 VALU instructions per symbol (compiler-scheduled
// 64-bit arithmetic full of register-pair moves, a branch around every renormalisation, a wait after every record read);
// here one 16-symbol group is ONE asm statement:
//  * record {rcp lo, rcp hi, bias << 6 | rcp_shift, cmpl} (16 bytes, one ds_read_b128 per symbol; the LDS pipe could
//    not feed two), the records of the next pair of symbols are read while the current pair is worked on;
//  * renormalisation test (rans64.h:83: x >= ((L >> scale_bits) << 32) * freq) on the high dword alone: the low dword of
//    that bound is 0, and x.hi >= (M - cmpl) << k  <=>  x.hi + (cmpl << k) >= 2^31 (k = 31 - scale_bits): one
//    v_lshl_add + v_cmpx; the lanes that emit run under the exec mask (dword into the ring, x >>= 32 as two moves);
//  * q = mulhi64(x, rcp) >> rcp_shift (rans64.h:91, exact): v_mul_hi + 3 x v_mad_u64_u32, the middle sum's carry through
//    vcc (E64_BACK below); x += bias + q * cmpl (rans64.h:92, Rans64EncSymbolInit's identity) as v_lshl_add_u64 +
//    v_mad_u64_u32 + v_mad_u32_u24; 18.5 VALU per symbol (22.5 until late in round 4);
//  * output ring per lane: 32 dwords, dword d of lane l at ring + 256 d + 4 l (every ds_write_b32 conflict-free), the ring
//    8 KiB aligned so that the write position wraps with one v_bfi; a lane's bytes written are never counted per symbol,
//    the flush derives them from the ring position (at most 64 bytes per group);
//  * symbols: four 64-byte lines per quad and block (instruction t = the line of the quad's lane t), transposed in
//    registers, the next block in flight during the current one; flushes by quads as in the staged kernel.
// Requirements (launcher): rans64 with scale_bits 7..16 and no frequency of 2^16, u8 symbols, chunk_syms % 64 == 0,
// 16-byte aligned input, full batches of full chunks (the rest goes through the staged kernel in a second launch, which
// continues the same status array).
// ---------------------------------------------------------------------------
constexpr uint32_t kR64EncRing = 8192;   // per coding wave
constexpr uint32_t kR64EncTable = 8192;  // 256 records of 16 bytes at LDS address 0 (rings behind, 8 KiB aligned)

// state 0 = v[40:41], state 1 = v[42:43]; record sets v[48:51] / v[52:55] (pair in hand) and v[56:59] / v[60:63] (next);
// temporaries v64..v79 with the permanently zero v65, v69, v77 (the two states are worked on one after the other)
#define E64_ZERO                                                                                                        \
    "v_mov_b32 v65, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v77, 0\n\t"
// address of the record of byte J of symbol dword S: sym << 4 -- one SDWA shift of the selected byte (the count in a VGPR:
// SDWA takes no literal)
#define E64_SDWA(D, S, SEL)                                                                                             \
    "v_lshlrev_b32_sdwa " D ", %[k4], " S " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL "\n\t"
#define E64_ADDR3(D, S) E64_SDWA(D, S, "BYTE_3")
#define E64_ADDR2(D, S) E64_SDWA(D, S, "BYTE_2")
#define E64_ADDR1(D, S) E64_SDWA(D, S, "BYTE_1")
#define E64_ADDR0(D, S) E64_SDWA(D, S, "BYTE_0")
// renormalisation of state {XL, XH} with record dword C = cmpl (T: temporary)
#define E64_FRONT(XL, XH, C, T)                                                                                         \
    "v_lshl_add_u32 " T ", " C ", %[kv], " XH "\n\t"                                                                    \
    "v_cmpx_gt_i32 vcc, 0, " T "\n\t"                                                                                   \
    "v_add_u32 %[wk], %[m256], %[wk]\n\t"                                                                               \
    "v_bfi_b32 %[wk], %[m1fff], %[wk], %[ring]\n\t"                                                                     \
    "ds_write_b32 %[wk], " XL "\n\t"                                                                                    \
    "v_mov_b32 " XL ", " XH "\n\t"                                                                                      \
    "v_mov_b32 " XH ", 0\n\t"                                                                                           \
    "s_mov_b64 exec, -1\n\t"
// x = x + bias + (mulhi64(x, rcp) >> rcp_shift) * cmpl; X = "v[a:b]" of {XL, XH}; record R0..R3 = {rcp lo, rcp hi,
// bias << 6 | rcp_shift, cmpl}; TA / BP: pairs whose high register is permanently zero (TAL, BPL their low registers),
// TZ = {TZL, TZH}: the high dword of the middle sum and its carry.  mulhi64 as
//   t  = hi(xl * r0)
//   P1 = xh * r0 + t                 (< 2^63 + 2^32: xh < 2^31)
//   P2 = xl * r1 + P1                (may pass 2^64: the carry comes out of v_mad_u64_u32 in vcc)
//   RR = xh * r1 + {hi(P2), carry}
// -- the second product takes the WHOLE first one as its addend instead of a {low dword, 0} pair (round 4; three moves and
// a 64-bit add less); the shift takes its count from the low six bits of the record's third dword as they are
// (v_lshrrev_b64 looks at no others).  11 VALU (14 before).  Two instructions stand between the v_mad_u64_u32 that
// writes vcc and the v_addc that reads it (gfx940: a VALU write of an SGPR needs two wait states before a VALU reads it).
#define E64_BACK(X, XL, XH, R0, R1, R2, R3, TA, TAL, P1, TZ, TZL, TZH, P2, P2H, RR, RRL, RRH, BP, BPL, ZERO)             \
    "v_mul_hi_u32 " TAL ", " XL ", " R0 "\n\t"                                                                          \
    "v_mad_u64_u32 " P1 ", vcc, " XH ", " R0 ", " TA "\n\t"                                                             \
    "v_mad_u64_u32 " P2 ", vcc, " XL ", " R1 ", " P1 "\n\t"                                                             \
    "v_lshrrev_b32 " BPL ", 6, " R2 "\n\t"                                                                              \
    "v_mov_b32 " TZL ", " P2H "\n\t"                                                                                    \
    "v_addc_co_u32 " TZH ", vcc, 0, " ZERO ", vcc\n\t"                                                                  \
    "v_mad_u64_u32 " RR ", vcc, " XH ", " R1 ", " TZ "\n\t"                                                             \
    "v_lshrrev_b64 " RR ", " R2 ", " RR "\n\t"                                                                          \
    "v_lshl_add_u64 " X ", " X ", 0, " BP "\n\t"                                                                        \
    "v_mad_u64_u32 " X ", vcc, " RRL ", " R3 ", " X "\n\t"                                                              \
    "v_mad_u32_u24 " XH ", " RRH ", " R3 ", " XH "\n\t"
#define E64_BACK_A(R0, R1, R2, R3)                                                                                      \
    E64_BACK("v[40:41]", "v40", "v41", R0, R1, R2, R3, "v[64:65]", "v64", "v[66:67]", "v[68:69]", "v68", "v69",          \
             "v[70:71]", "v71", "v[72:73]", "v72", "v73", "v[76:77]", "v76", "v65")
#define E64_BACK_B(R0, R1, R2, R3)                                                                                      \
    E64_BACK("v[42:43]", "v42", "v43", R0, R1, R2, R3, "v[64:65]", "v64", "v[66:67]", "v[68:69]", "v68", "v69",          \
             "v[70:71]", "v71", "v[72:73]", "v72", "v73", "v[76:77]", "v76", "v65")
// one pair of symbols (the odd one = state 1 first: it is the later symbol) with its records in set 0 (v48..v55: state 1
// in v[48:51], state 0 in v[52:55]) or set 1 (v56..v63); PRE = address + read instructions of the NEXT pair, WAIT = lgkmcnt
#define E64_PAIR0(PRE, WAIT)                                                                                            \
    PRE "s_waitcnt lgkmcnt(" WAIT ")\n\t"                                                                               \
    E64_TRACK0                                                                                                          \
    E64_FRONT("v42", "v43", "v51", "v74") E64_FRONT("v40", "v41", "v55", "v74")                                         \
    E64_BACK_B("v48", "v49", "v50", "v51") E64_BACK_A("v52", "v53", "v54", "v55")
#define E64_PAIR1(PRE, WAIT)                                                                                            \
    PRE "s_waitcnt lgkmcnt(" WAIT ")\n\t"                                                                               \
    E64_TRACK1                                                                                                          \
    E64_FRONT("v42", "v43", "v59", "v74") E64_FRONT("v40", "v41", "v63", "v74")                                         \
    E64_BACK_B("v56", "v57", "v58", "v59") E64_BACK_A("v60", "v61", "v62", "v63")
// reads of a pair into set 0 / set 1: bytes (JB, JA) of symbol dword S
#define E64_READ0(ADDRB, ADDRA, S)                                                                                      \
    ADDRB("v78", S) ADDRA("v79", S) "ds_read_b128 v[48:51], v78\n\tds_read_b128 v[52:55], v79\n\t"
#define E64_READ1(ADDRB, ADDRA, S)                                                                                      \
    ADDRB("v78", S) ADDRA("v79", S) "ds_read_b128 v[56:59], v78\n\tds_read_b128 v[60:63], v79\n\t"
#define E64_CLOBBERS                                                                                                    \
    "vcc", "memory", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62",  \
        "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"

// 16 symbols (dwords s3 = the last four .. s0 = the first four of the group), last symbol first
#define E64_GROUP_ASM                                                                                                   \
    asm volatile(E64_ZERO                                                                                               \
                 E64_READ0(E64_ADDR3, E64_ADDR2, "%[s3]")                                                               \
                 E64_PAIR0(E64_READ1(E64_ADDR1, E64_ADDR0, "%[s3]"), "2")                                               \
                 E64_PAIR1(E64_READ0(E64_ADDR3, E64_ADDR2, "%[s2]"), "2")                                               \
                 E64_PAIR0(E64_READ1(E64_ADDR1, E64_ADDR0, "%[s2]"), "2")                                               \
                 E64_PAIR1(E64_READ0(E64_ADDR3, E64_ADDR2, "%[s1]"), "2")                                               \
                 E64_PAIR0(E64_READ1(E64_ADDR1, E64_ADDR0, "%[s1]"), "2")                                               \
                 E64_PAIR1(E64_READ0(E64_ADDR3, E64_ADDR2, "%[s0]"), "2")                                               \
                 E64_PAIR0(E64_READ1(E64_ADDR1, E64_ADDR0, "%[s0]"), "2")                                               \
                 E64_PAIR1("", "0")                                                                                     \
                 : "+{v[40:41]}"(xA), "+{v[42:43]}"(xB), [wk] "+v"(wk), [worst] "+v"(worst)                             \
                 : [s3] "v"(s3), [s2] "v"(s2), [s1] "v"(s1), [s0] "v"(s0), [k4] "v"(k4), [m256] "v"(m256),              \
                   [m1fff] "v"(m1fff), [ring] "v"(ring), [kv] "v"(kv)                                                   \
                 : E64_CLOBBERS)
// TRACK: the model has byte values without a record (their cmpl = M is the largest value any record holds: v_max3 over
// the pairs' cmpl words finds it); EncParams::dense256 models run without
template <bool TRACK>
__device__ __forceinline__ void r64x2_encode_group(uint64_t &xA, uint64_t &xB, uint32_t &wk, uint32_t &worst, uint32_t s3,
                                                   uint32_t s2, uint32_t s1, uint32_t s0, uint32_t k4, uint32_t m256,
                                                   uint32_t m1fff, uint32_t ring, uint32_t kv)
{
    if constexpr (TRACK) {
#define E64_TRACK0 "v_max3_u32 %[worst], %[worst], v51, v55\n\t"
#define E64_TRACK1 "v_max3_u32 %[worst], %[worst], v59, v63\n\t"
        E64_GROUP_ASM;
#undef E64_TRACK0
#undef E64_TRACK1
    } else {
#define E64_TRACK0 ""
#define E64_TRACK1 ""
        E64_GROUP_ASM;
#undef E64_TRACK0
#undef E64_TRACK1
    }
}

template <bool TRACK>
__global__ void __launch_bounds__(1024) k_encode_lanes_r64x2(const EncParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    {   // EncRec {freq | rcp_shift << 24, bias, rcp lo, rcp hi} (model.h) -> {rcp lo, rcp hi, bias << 6 | rcp_shift, cmpl};
        // a symbol without a frequency: cmpl = M (the largest value any record holds: v_max3 finds it), rcp = 0
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) {
            uint4 r = i < p.nsyms ? g[i] : uint4{0u, 0u, 0u, 0u};
            const uint32_t freq = r.x & 0xffffffu;
            l[i] = freq ? uint4{r.z, r.w, (r.y << 6) | (r.x >> 24), (1u << p.scale_bits) - freq}
                        : uint4{0u, 0u, 0u, 1u << p.scale_bits};
        }
        if (p.status) // (a block may be as small as one coding wave + the scanner: 128 threads for 132 words)
            for (uint32_t i = threadIdx.x; i < kEncMailboxBytes / 4u; i += blockDim.x)
                reinterpret_cast<uint32_t *>(smem + p.mailbox_off)[i] = 0u;
    }
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const bool fused = p.status != nullptr;
    const uint32_t waves_per_block = (blockDim.x >> 6) - (fused ? kLaneCopiers : 0u); // coding waves
    LaneRounds *ctl = reinterpret_cast<LaneRounds *>(smem + p.mailbox_off);
    if (!lds_starts_at_zero(smem)) { // the asm addresses the record table by raw LDS offsets
        if (threadIdx.x == 0)
            atomicOr(p.flags, 4u);
        return;
    }
    if (fused && wave >= waves_per_block) { // ---- the scanner wave and the copier waves
        if (wave == waves_per_block)
            lanes_scanner(p, ctl, lane, waves_per_block);
        else
            lanes_copier(p, ctl, wave - waves_per_block - 1u, lane, waves_per_block);
        return;
    }
    const uint32_t ringbase = kR64EncTable + wave * kR64EncRing; // LDS byte offset, 8 KiB aligned
    const uint32_t *ringp = reinterpret_cast<const uint32_t *>(smem + ringbase);
    uint32_t k4 = 4u, m256 = 0xffffff00u, m1fff = 0x1fffu, ringv = ringbase, kv = 31u - p.scale_bits;
    asm volatile("v_mov_b32 %0, %0" : "+v"(k4)); // VGPR copies: a VALU op with a literal or an SGPR operand issues slower
    asm volatile("v_mov_b32 %0, %0" : "+v"(m256));
    asm volatile("v_mov_b32 %0, %0" : "+v"(m1fff));
    asm volatile("v_mov_b32 %0, %0" : "+v"(ringv));
    asm volatile("v_mov_b32 %0, %0" : "+v"(kv));
    const uint32_t m = lane & 3u, q4 = lane & ~3u;
    const uint32_t slot_lines = (uint32_t)(p.slot_bytes / kLaneLine);
    const uint32_t nblocks = p.chunk_syms >> 6;
    uint32_t worst = 0;
    LaneCoder cs{0};

    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t batch_v = p.batch_begin + (uint64_t)blockIdx.x * waves_per_block + wave;; batch_v += total_waves) {
        if (fused) { // the block's scanner hands out the rounds
            batch_v = lanes_round_begin(p, ctl, cs, wave, waves_per_block, lane);
            if (batch_v == ~0ull - 1u) { // nothing for this wave in the last unit
                lanes_round_end(p, ctl, cs, wave, lane, batch_v, 0u);
                continue;
            }
        }
        if (batch_v >= p.batch_end)
            break;
        const uint64_t batch = uniform64(batch_v);
        const uint64_t chunk0 = batch * 64u;
        const uint64_t slots0 = reinterpret_cast<uint64_t>(p.scratch) + chunk0 * p.slot_bytes; // wave-uniform
        // symbol lines: instruction t = the line of the quad's lane t, this lane its piece m
        const uint64_t src0 = reinterpret_cast<uint64_t>(p.syms) + chunk0 * (uint64_t)p.chunk_syms + 16u * m;
        auto load_block = [&](u32x4 (&q)[4], uint32_t b) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                q[t] = __builtin_nontemporal_load(
                    reinterpret_cast<gvec_cptr>(src0 + (uint64_t)(q4 + t) * p.chunk_syms + 64ull * b));
        };

        uint64_t xA = 1ull << 31, xB = 1ull << 31; // Rans64EncInit
        uint32_t wk = ringbase + ((((uint32_t)p.slot_bytes & 127u) >> 2) << 8) + lane * 4u; // ring position of slot offset w
        uint32_t flushed = slot_lines; // lines [flushed, slot_lines) of this lane's slot are in memory

        // bytes of this lane's stream that are in the ring only: below line `flushed`, at most 127
        auto pending = [&]() { return (((flushed & 1u) << 6) - ((wk >> 6) & 124u)) & 127u; };
        // the whole wave takes part: lanes say which line they have filled (or, at the end, hold anything of), the quad
        // writes the line of its lane t with instruction t
        bool ovf = false; // sized slots: this lane's chunk needs a line below its slot
        auto flush = [&](bool need) {
            if (need && flushed == 0u) {
                ovf = true;
                need = false;
            }
            const int32_t line = need ? (int32_t)(flushed - 1u) : -1;
            if (need)
                flushed -= 1u;
            if (__builtin_amdgcn_ballot_w64(need) == 0)
                return;
            const int32_t lts[4] = {(int32_t)quad_perm<0x00>((uint32_t)line), (int32_t)quad_perm<0x55>((uint32_t)line),
                                    (int32_t)quad_perm<0xAA>((uint32_t)line), (int32_t)quad_perm<0xFF>((uint32_t)line)};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int32_t lt = lts[t]; // the line asked for by the quad's lane t
                if (lt >= 0) {
                    // ring half (lt & 1), dwords 4 m .. 4 m + 3 of lane q4 + t
                    const uint32_t *at = ringp + ((uint32_t)lt & 1u) * 1024u + m * 256u + q4 + t;
                    const u32x4 v = {at[0], at[64], at[128], at[192]};
                    *reinterpret_cast<u32x4 RANS_GLOBAL *>(slots0 + (uint64_t)(q4 + t) * p.slot_bytes + (uint64_t)lt * kLaneLine +
                                                           16u * m) = v;
                }
            }
        };

        u32x4 cur[4], nxt[4];
        load_block(cur, nblocks - 1u);
        quad_transpose(cur[0], cur[1], cur[2], cur[3], lane);
        for (uint32_t b = nblocks; b-- > 0;) {
            if (b > 0)
                load_block(nxt, b - 1u);
#pragma unroll
            for (int g = 3; g >= 0; --g) {
                r64x2_encode_group<TRACK>(xA, xB, wk, worst, cur[g][3], cur[g][2], cur[g][1], cur[g][0], k4, m256, m1fff, ringv,
                                          kv);
                flush(pending() >= 64u);
            }
            if (b > 0) {
                quad_transpose(nxt[0], nxt[1], nxt[2], nxt[3], lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    cur[t] = nxt[t];
            }
        }
        // flush: state 1 first, state 0 ends up first in memory (main64.cpp:244-245; Rans64EncFlush: two dwords, low first)
        {
            auto emit = [&](uint32_t v) {
                wk = ringbase + ((wk - 256u) & 0x1fffu);
                *reinterpret_cast<uint32_t *>(smem + wk) = v;
            };
            emit((uint32_t)(xB >> 32));
            emit((uint32_t)xB);
            emit((uint32_t)(xA >> 32));
            emit((uint32_t)xA);
        }
        const uint32_t in_ring = pending(); // <= 63 + 16
        const uint32_t len = (uint32_t)p.slot_bytes - (flushed * kLaneLine - in_ring);
        flush(in_ring > 0u); // the line(s) that hold anything; what lies below the stream start in the lowest one is never read
        flush(in_ring > 64u);
        p.lengths[chunk0 + lane] = len;
        lanes_publish_slot(p, chunk0 + lane, len, ovf);
        if (fused)
            lanes_round_end(p, ctl, cs, wave, lane, batch, len);
    }
    if (__builtin_amdgcn_ballot_w64(worst >= (1u << p.scale_bits)) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
}

// Lane-per-stream encoder, second generation: symbols arrive as 16-byte per-lane loads
// (one scattered access per 16 symbols instead of per symbol), one group prefetched.
template <int FMT, int NW>
__global__ void __launch_bounds__(256) k_encode_lanes16(const EncParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < p.nsyms; i += blockDim.x)
            l[i] = g[i];
    }
    __syncthreads();
    const uint4 *recs = reinterpret_cast<const uint4 *>(smem);
    const bool wide_in = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 15u) == 0;

    bool bad = false;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; chunk < p.nchunks; chunk += stride) {
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + first * p.sym_bytes;
        uint8_t RANS_GLOBAL *slot = (uint8_t RANS_GLOBAL *)p.scratch + chunk * p.slot_bytes;
        uint8_t RANS_GLOBAL *wp = slot + p.slot_bytes;

        state_t x[NW];
#pragma unroll
        for (int l = 0; l < NW; ++l)
            x[l] = Tr::kL;

        // symbol i belongs to state i mod NW; visit i = nsym-1 .. 0 (main.cpp:233-243).
        // [0, fast_end) is walked in 16-symbol groups; the ragged top part one by one.
        const uint32_t fast_end = wide_in ? (nsym & ~15u) : 0u;
        for (uint32_t i = nsym; i > fast_end; --i) {
            const uint32_t sym = p.sym_bytes == 1 ? (uint32_t)src[i - 1]
                                                  : (uint32_t) reinterpret_cast<const uint16_t RANS_GLOBAL *>(src)[i - 1];
            const uint32_t l = (i - 1) % NW;
#pragma unroll
            for (int ll = 0; ll < NW; ++ll) // static register indexing
                if ((uint32_t)ll == l)
                    lane_put<FMT>(x[ll], sym, recs, p, wp, bad);
        }
        if (fast_end) {
            const u32x4 RANS_GLOBAL *g16 = reinterpret_cast<const u32x4 RANS_GLOBAL *>(src);
            uint32_t g = fast_end >> 4;
            u32x4 cur = g16[g - 1], nxt = cur;
            while (g-- > 0) {
                if (g > 0)
                    nxt = g16[g - 1];
#pragma unroll
                for (int j = 15; j >= 0; --j) {
                    const uint32_t sym = (cur[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    lane_put<FMT>(x[j % NW], sym, recs, p, wp, bad);
                }
                cur = nxt;
            }
        }
        // flush states NW-1 .. 0 (lane 0's first in memory)
#pragma unroll
        for (int l = NW - 1; l >= 0; --l) {
            wp -= Tr::kStateBytes;
            if constexpr (FMT == FMT_R64) {
                reinterpret_cast<uint32_t RANS_GLOBAL *>(wp)[0] = (uint32_t)x[l];
                reinterpret_cast<uint32_t RANS_GLOBAL *>(wp)[1] = (uint32_t)(x[l] >> 32);
            } else if constexpr (FMT == FMT_WORD) {
                reinterpret_cast<uint16_t RANS_GLOBAL *>(wp)[0] = (uint16_t)x[l];
                reinterpret_cast<uint16_t RANS_GLOBAL *>(wp)[1] = (uint16_t)(x[l] >> 16);
            } else {
                wp[0] = (uint8_t)x[l];
                wp[1] = (uint8_t)(x[l] >> 8);
                wp[2] = (uint8_t)(x[l] >> 16);
                wp[3] = (uint8_t)(x[l] >> 24);
            }
        }
        p.lengths[chunk] = (uint32_t)((slot + p.slot_bytes) - wp);
        lanes_publish_slot(p, chunk, (uint32_t)((slot + p.slot_bytes) - wp));
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane_id() == 0)
        atomicOr(p.flags, 1u);
}

template <int FMT, int NW>
hipError_t launch_decode_lanes_t(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const uint32_t t0 = (p.table0_bytes + 15u) & ~15u, t1 = (p.table1_bytes + 15u) & ~15u;
    // staged kernel: the tables are shared by the block, every wave adds kLaneWaveLds of rings, so one
    // large block per CU keeps the most waves resident (rans64, 14 bits: 15 waves; 4-wave blocks: 12)
    const size_t table_lds = (size_t)t0 + t1;
    uint32_t sw = table_lds + kLaneWaveLds <= 160 * 1024 ? (uint32_t)((160 * 1024 - table_lds) / kLaneWaveLds) : 0;
    sw = sw > 16 ? 16 : sw;
    // 64 chunks of one wave must lie within 2^30 bytes (32-bit ring positions): any sane chunk size
    // NAMED FALLBACK: chunks of 512 Ki symbols and more, or tables that leave no room for a wave's rings, go to the first
    // generation's per-lane register window (k_decode_lanes: >= 4 x over-fetch, one chunk per lane all the same -- a wave per
    // chunk would leave 62 of 64 lanes idle on a 2-way stream); tests/test_gpu_parity.py test_lane_decoder_fallback_*
    const bool staged = sw >= 1 && (uint64_t)p.chunk_syms * 8u < (1u << 22);
    {   // One batch (64 chunks) is a long latency-bound job, so a last round with a few waves per CU
        // costs as much as a full one: take the fewest rounds the LDS allows and split the batches
        // evenly over them (16 batches per CU: 16 waves x 1 round, or 8 x 2 -- never 14 + 2).
        const uint64_t batches = (p.nchunks + 63) / 64;
        const uint64_t per_cu = (batches + (uint64_t)num_cus - 1) / (uint64_t)num_cus;
        if (sw >= 1) {
            const uint64_t rounds = (per_cu + sw - 1) / sw;
            const uint64_t even = rounds ? (per_cu + rounds - 1) / rounds : 1;
            sw = (uint32_t)(even ? even : 1);
        }
    }
    if constexpr (FMT == FMT_R64 && NW == 2) {
        // third generation for the reference's own 2-way rans64 layout (config 2): full 64-symbol trips only; a ragged
        // last chunk is decoded by the staged kernel in a second launch
        const bool aligned = p.sym_bytes == 1 && ((reinterpret_cast<uintptr_t>(p.out) | p.chunk_syms) & 63u) == 0 &&
                             (reinterpret_cast<uintptr_t>(p.container) & 15u) == 0;
        if (staged && aligned && p.nchunks >= 64 && !p.trace) {
            const uint64_t full = p.n / p.chunk_syms; // chunks with chunk_syms symbols
            // packed slot records (one gather per symbol) where the model has them and at least 8 waves' rings fit beside
            // the table
            const size_t packed_lds = (p.packed_bytes + 15u) & ~(size_t)15;
            const bool packed = p.packed && packed_lds + 8u * kR64WaveLds <= 160 * 1024;
            const size_t table_lds3 = packed ? packed_lds : table_lds;
            uint32_t sw3 = (uint32_t)((160 * 1024 - table_lds3) / kR64WaveLds);
            sw3 = sw3 > 16 ? 16 : sw3;
            const uint64_t batches = (full + 63) / 64;
            const uint64_t per_cu = (batches + (uint64_t)num_cus - 1) / (uint64_t)num_cus;
            const uint64_t rounds = (per_cu + sw3 - 1) / sw3;
            const uint64_t even = rounds ? (per_cu + rounds - 1) / rounds : 1;
            sw3 = (uint32_t)(even ? even : 1);
            auto kern3 = packed ? k_decode_lanes_r64x2<true> : k_decode_lanes_r64x2<false>;
            static std::atomic<uint64_t> lds_ok3[2] = {{0}, {0}};
            if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern3), 160 * 1024, lds_ok3[packed]); e != hipSuccess)
                return e;
            DecParams q = p;
            q.nchunks = full;
            q.n = full * p.chunk_syms;
            if (packed) { // the kernel stages "table 0" and "table 1": the packed records and nothing
                q.table0 = p.packed;
                q.table0_bytes = p.packed_bytes;
                q.table1_bytes = 0;
            }
            const uint64_t want_blocks = (batches + sw3 - 1) / sw3;
            const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)num_cus ? want_blocks : (uint64_t)num_cus);
            if (name)
                *name = packed ? "k_decode_lanes_r64x2<packed slots>" : "k_decode_lanes_r64x2";
            RANS_LAUNCH(kern3, dim3(grid), dim3(64 * sw3), table_lds3 + (size_t)sw3 * kR64WaveLds, stream, q);
            if (hipError_t e = hipGetLastError(); e != hipSuccess)
                return e;
            if (full == p.nchunks)
                return hipSuccess;
            DecParams r = p; // the ragged last chunk
            r.offsets = p.offsets + full;
            r.lengths = p.lengths + full;
            r.out = static_cast<uint8_t *>(p.out) + full * p.chunk_syms;
            r.n = p.n - full * p.chunk_syms;
            r.nchunks = 1;
            r.work_counter_reset = nullptr;
            r.span_reset = nullptr;
            auto tail = k_decode_lanes_staged<FMT, NW>;
            static std::atomic<uint64_t> lds_ok4{0};
            if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(tail), 160 * 1024, lds_ok4); e != hipSuccess)
                return e;
            RANS_LAUNCH(tail, dim3(1), dim3(64), table_lds + kLaneWaveLds, stream, r);
            return hipGetLastError();
        }
    }
    const size_t lds = staged ? table_lds + (size_t)sw * kLaneWaveLds : table_lds;
    auto kern = staged ? k_decode_lanes_staged<FMT, NW> : k_decode_lanes<FMT, NW>;
    static std::atomic<uint64_t> lds_ok[2] = {{0}, {0}}; // per kernel generation, one bit per device
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 160 * 1024, lds_ok[staged]); e != hipSuccess)
        return e;
    if (staged) {
        const uint64_t batches = (p.nchunks + 63) / 64;
        const uint64_t want_blocks = (batches + sw - 1) / sw;
        const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)num_cus ? want_blocks : (uint64_t)num_cus);
        if (name)
            *name = "k_decode_lanes_staged";
        RANS_LAUNCH(kern, dim3(grid), dim3(64 * sw), lds, stream, p);
        return hipGetLastError();
    }
    const uint64_t want = (p.nchunks + 255) / 256;
    // 256-thread blocks: up to 8 per CU (32 waves) when the tables leave room in LDS; the kernel
    // is latency-bound (per-lane scattered loads), so residency matters more than anything else
    const size_t lds_room = lds ? (160 * 1024) / lds : 8;
    const uint64_t cap = (uint64_t)num_cus * (lds_room >= 8 ? 8 : (lds_room >= 1 ? lds_room : 1));
    const uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (name)
        *name = "k_decode_lanes";
    RANS_LAUNCH(kern, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

// The staged lane encoder (coalesced symbol loads, whole-line stream stores, fused placement) takes u8 symbols in
// 16-byte aligned chunks with slots made of whole lines, from six batches per CU on.  Everything else -- u16 symbols, chunk
// sizes that are not multiples of 16, unaligned buffers, a handful of batches -- is the NAMED FALLBACK's: the first generation's
// per-lane encoder k_encode_lanes16 (one chunk per lane, unit-by-unit stores).  Returns the waves per block the LDS allows, 0 = no.
static uint32_t encode_lanes_staged_waves(const EncParams &p, int num_cus, uint64_t min_batches_per_cu = 6)
{
    const size_t table_lds = (size_t)p.nsyms * sizeof(EncRec);
    // (room for the scanner wave and the control words of the fused placement, whether or not this launch uses them)
    const size_t fixed_lds = table_lds + 16 + kEncMailboxBytes;
    uint32_t sw = fixed_lds + kEncWaveLds <= 160 * 1024 ? (uint32_t)((160 * 1024 - fixed_lds) / kEncWaveLds) : 0;
    sw = sw > 16 - kLaneCopiers ? 16 - kLaneCopiers : sw;
    const bool staged = sw >= 1 && p.sym_bytes == 1 && (p.slot_bytes % kLaneLine) == 0 &&
                        ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 15u) == 0 &&
                        (reinterpret_cast<uintptr_t>(p.scratch) & 15u) == 0 &&
                        // fewer, longer batches: the per-lane kernel's many small blocks hide latency better
                        (p.nchunks + 63) / 64 >= (uint64_t)num_cus * min_batches_per_cu;
    return staged ? sw : 0u;
}

template <int FMT, int NW> hipError_t launch_encode_lanes_t(const EncParams &p_in, int num_cus, hipStream_t stream, const char **name)
{
    EncParams p = p_in;
    const size_t table_lds = (size_t)p.nsyms * sizeof(EncRec);
    if (table_lds > 128 * 1024)
        return hipErrorInvalidValue;
    // (the dedicated 2-way rans64 encoder is worth it from one batch per CU on: 4096-symbol chunks, 1024 batches of
    //  config 2's 256 MiB, 0.78 ms with the per-lane kernel; with the fused placement the predicate must stay the one
    //  encode_lanes_fused() gave api.cpp)
    const bool r64x2_shape = FMT == FMT_R64 && NW == 2 && !p.status && (p.chunk_syms & 63u) == 0 && p.scale_bits >= 7 &&
                             p.scale_bits <= 16 && p.nsyms <= 256 && (p.n / p.chunk_syms) / 64 >= (uint64_t)num_cus;
    uint32_t sw = encode_lanes_staged_waves(p, num_cus, r64x2_shape ? 1 : 6);
    const bool staged = sw >= 1;
    if (!staged && p.status)
        return hipErrorInvalidValue; // (api.cpp asks encode_lanes_fused() before it sets up the fused placement)
    if (staged) {
        // same split as the staged decoder: fewest rounds, batches spread evenly over them
        uint64_t batches = (p.nchunks + 63) / 64;
        p.batch_begin = 0;
        p.batch_end = batches;
        p.claim_slot = 0;
        p.unit_base = 0;
        if constexpr (FMT == FMT_R64 && NW == 2) {
            // the reference's 2-way rans64 layout (config 2) on its own kernel: whole batches of full chunks; what is
            // left (fewer than 64 chunks, the last one perhaps ragged) goes through the staged kernel below
            const uint64_t full_batches = (p.n / p.chunk_syms) / 64;
            if ((p.chunk_syms & 63u) == 0 && p.scale_bits >= 7 && p.scale_bits <= 16 &&
                p.nsyms <= 256 && full_batches >= (uint64_t)num_cus) {
                // 8 KiB of table + 8 KiB of ring per coding wave (+ the scanner wave and its LDS words when it places the chunks)
                uint32_t sw3 = p.status ? 16 - kLaneCopiers : 16;
                const uint64_t per_cu = (full_batches + (uint64_t)num_cus - 1) / (uint64_t)num_cus;
                const uint64_t rounds = (per_cu + sw3 - 1) / sw3;
                const uint64_t even = (per_cu + rounds - 1) / rounds;
                sw3 = (uint32_t)(even ? even : 1);
                // (a model in which every byte value has a frequency: the variant without the search for record-less symbols)
                auto kern3 = p.dense256 ? k_encode_lanes_r64x2<false> : k_encode_lanes_r64x2<true>;
                static std::atomic<uint64_t> lds_ok3[2] = {{0}, {0}};
                if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern3), 160 * 1024, lds_ok3[p.dense256 ? 0 : 1]);
                    e != hipSuccess)
                    return e;
                EncParams q = p;
                q.batch_end = full_batches;
                size_t lds3 = kR64EncTable + (size_t)sw3 * kR64EncRing;
                uint32_t waves3 = sw3;
                if (q.status) {
                    q.mailbox_off = (uint32_t)lds3;
                    lds3 += kEncMailboxBytes;
                    waves3 += kLaneCopiers;
                }
                const uint64_t want3 = (full_batches + sw3 - 1) / sw3;
                const uint32_t grid3 = (uint32_t)(want3 < (uint64_t)num_cus ? want3 : (uint64_t)num_cus);
                if (name)
                    *name = "k_encode_lanes_r64x2";
                RANS_LAUNCH(kern3, dim3(grid3), dim3(64 * waves3), lds3, stream, q);
                if (hipError_t e = hipGetLastError(); e != hipSuccess)
                    return e;
                if (full_batches == batches)
                    return hipSuccess;
                p.batch_begin = full_batches; // the tail: its own claim counter, its units behind the ones of this launch
                p.claim_slot = 1;
                p.unit_base = (full_batches + sw3 - 1) / sw3;
                batches -= full_batches;
            }
        }
        const uint64_t per_cu = (batches + (uint64_t)num_cus - 1) / (uint64_t)num_cus;
        const uint64_t rounds = (per_cu + sw - 1) / sw;
        const uint64_t even = rounds ? (per_cu + rounds - 1) / rounds : 1;
        sw = (uint32_t)(even ? even : 1);
        auto kern = k_encode_lanes_staged<FMT, NW>;
        static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 160 * 1024, lds_ok); e != hipSuccess)
            return e;
        const uint64_t want_blocks = (batches + sw - 1) / sw;
        const uint32_t grid = (uint32_t)(want_blocks < (uint64_t)num_cus ? want_blocks : (uint64_t)num_cus);
        size_t lds = table_lds + (size_t)sw * kEncWaveLds;
        uint32_t waves = sw;
        if (p.status) { // fused placement: the block's scanner wave and its control words
            lds = (lds + 15) & ~(size_t)15;
            p.mailbox_off = (uint32_t)lds;
            lds += kEncMailboxBytes;
            waves += kLaneCopiers;
        }
        if (name && p.batch_begin == 0)
            *name = "k_encode_lanes_staged";
        RANS_LAUNCH(kern, dim3(grid), dim3(64 * waves), lds, stream, p);
        return hipGetLastError();
    }
    const size_t lds = table_lds;
    auto kern = k_encode_lanes16<FMT, NW>;
    static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 128 * 1024, lds_ok); e != hipSuccess)
        return e;
    const uint64_t want = (p.nchunks + 255) / 256;
    const uint64_t cap = (uint64_t)num_cus * 8;
    const uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (name)
        *name = "k_encode_lanes16";
    RANS_LAUNCH(kern, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int FMT> hipError_t launch_decode_lanes_f(const DecParams &p, int num_cus, hipStream_t s, const char **name)
{
    switch (p.n_ways) {
    case 1: return launch_decode_lanes_t<FMT, 1>(p, num_cus, s, name);
    case 2: return launch_decode_lanes_t<FMT, 2>(p, num_cus, s, name);
    case 4: return launch_decode_lanes_t<FMT, 4>(p, num_cus, s, name);
    case 8: return launch_decode_lanes_t<FMT, 8>(p, num_cus, s, name);
    default: return hipErrorInvalidValue;
    }
}

template <int FMT> hipError_t launch_encode_lanes_f(const EncParams &p, int num_cus, hipStream_t s, const char **name)
{
    switch (p.n_ways) {
    case 1: return launch_encode_lanes_t<FMT, 1>(p, num_cus, s, name);
    case 2: return launch_encode_lanes_t<FMT, 2>(p, num_cus, s, name);
    case 4: return launch_encode_lanes_t<FMT, 4>(p, num_cus, s, name);
    case 8: return launch_encode_lanes_t<FMT, 8>(p, num_cus, s, name);
    default: return hipErrorInvalidValue;
    }
}

} // namespace

hipError_t launch_decode_lanes(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    switch (format) {
    case FMT_WORD: return launch_decode_lanes_f<FMT_WORD>(p, num_cus, stream, name);
    case FMT_BYTE: return launch_decode_lanes_f<FMT_BYTE>(p, num_cus, stream, name);
    case FMT_R64: return launch_decode_lanes_f<FMT_R64>(p, num_cus, stream, name);
    case FMT_ALIAS: return launch_decode_lanes_f<FMT_ALIAS>(p, num_cus, stream, name);
    default: return hipErrorInvalidValue;
    }
}

bool encode_lanes_fused(const EncParams &p, int num_cus) { return encode_lanes_staged_waves(p, num_cus) >= 1; }

// Sized slots: would launch_encode_lanes_t take one of its STAGED kernels (the 2-way rans64 one included) for this
// request?  Those flush whole lines and stop at their slot's first one; the per-lane kernel stores unit by unit and cannot
// tell that a chunk does not fit.  (The same predicate as the launcher's, r64x2 shape and all.)
bool encode_lanes_sized_ok(int format, const EncParams &p, int num_cus)
{
    if (format == FMT_WORD && encode_word_groups_applicable(p)) // (its block stores stop at the slot's first byte: encode_groups.hip)
        return true;
    const bool r64x2_shape = format == FMT_R64 && p.n_ways == 2 && !p.status && (p.chunk_syms & 63u) == 0 && p.scale_bits >= 7 &&
                             p.scale_bits <= 16 && p.nsyms <= 256 && (p.n / p.chunk_syms) / 64 >= (uint64_t)num_cus;
    return encode_lanes_staged_waves(p, num_cus, r64x2_shape ? 1 : 6) >= 1;
}

hipError_t launch_encode_lanes(int format, const EncParams &p, int num_cus, hipStream_t stream, const char **name)
{
    switch (format) {
    case FMT_WORD: return launch_encode_lanes_f<FMT_WORD>(p, num_cus, stream, name);
    case FMT_BYTE: return launch_encode_lanes_f<FMT_BYTE>(p, num_cus, stream, name);
    case FMT_R64: return launch_encode_lanes_f<FMT_R64>(p, num_cus, stream, name);
    case FMT_ALIAS: return launch_encode_lanes_f<FMT_ALIAS>(p, num_cus, stream, name);
    default: return hipErrorInvalidValue;
    }
}

} // namespace rans_amd
