// decode_common.hpp -- pieces shared by the wave-per-chunk decoders (decode_wave.hip, decode_dual.hip): the
// per-wave stream window in LDS, the renormalisation sub-steps, symbol packing, launch span record.
// Included inside each .hip file after device_common.hpp; everything lives in an anonymous namespace.
#pragma once

#include "device_common.hpp"
#include "launchers.hpp"

namespace rans_amd {

namespace {

// ---------------------------------------------------------------------------
// Stream window: a 2 KiB ring per wave in LDS, filled in aligned 1 KiB blocks
// (one buffer_load_dwordx4 per lane), the next block prefetched in registers.
//
//   cur   raw LDS byte address of the read cursor (wave-uniform, an SGPR)
//   mark  the cursor value at which the next refill is due: ring + 1024 while the cursor
//         walks the first half, ring + 2048 while it walks the second
//
// The ring holds the block under the cursor and the one after it.  When the cursor enters
// the second half, the first half is dead: the prefetched block goes there (and its first
// kRingMirror bytes are mirrored behind the ring's end, so reads never wrap), the next
// fetch is issued.  When the cursor runs past the ring's end (into the mirror) it is wrapped
// by -2048 and the second half is refilled.  A checkpoint is therefore one scalar compare +
// branch; between two checkpoints the decoder consumes at most kMaxAdvance bytes and reads
// at most one more sub-step beyond that.
//
// Fetches go through a buffer descriptor of the chunk's own stream (base = chunk start,
// num_records = its 16-byte aligned length): the hardware returns zeros beyond the end, so
// nothing outside the chunk's granules is ever read and the refill needs no address compare.
// ---------------------------------------------------------------------------
constexpr uint32_t kMaxAdvance = 512;
static_assert(kRingMirror >= kMaxAdvance + 256, "mirror must cover one checkpoint interval plus one sub-step");
constexpr uint32_t kRsrcFlags = 0x00020000u; // raw buffer, 32-bit data format (gfx9 family)
constexpr int kAuxNt = 2;                    // 2 = non-temporal: every stream byte is read exactly once

typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct StreamWindow {
    uint8_t *ring;      // LDS, wave-private
    uint32_t ring_addr; // the same as a raw LDS byte address
    uint32_t cur, mark;
    uint32_t lapped;    // stream bytes that lie before ring offset 0 of the current lap
    uint32_t gnext;     // byte offset (within the descriptor, i.e. from the first stream block) of the next fetch
    rsrc_t rsrc;
    u32x4 pre;          // prefetched block (16 B per lane)

    __device__ __forceinline__ u32x4 fetch(uint32_t lane)
    {
        // the whole offset travels in voffset: the range check then is simply offset >= num_records
        // (with an soffset part it is offset >= num_records - soffset, which wraps once soffset is beyond the end)
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gnext + lane * 16u, 0, kAuxNt);
        gnext += kRingBlock;
        return v;
    }
    __device__ __forceinline__ void put(uint32_t lane, uint32_t at, const u32x4 &v)
    {
        *reinterpret_cast<u32x4 *>(ring + at + lane * 16u) = v;
        if (at == 0 && lane < kRingMirror / 16u)
            *reinterpret_cast<u32x4 *>(ring + kRingBytes + lane * 16u) = v;
    }
    // The descriptor of a chunk's stream starts at the 16-byte granule of its first renormalisation unit (behind
    // the initial states): block k of the stream is at buffer offset 1024 k, so the blocks of the opening are
    // reached with immediate offsets from one per-lane VGPR.
    // chunk_base: 16-byte aligned global address of the chunk; first: offset of the first unit in it;
    // fetchable: 16-byte aligned number of bytes of the chunk that may be read.
    static __device__ __forceinline__ rsrc_t stream_rsrc(uint64_t chunk_base, uint32_t first, uint32_t fetchable)
    {
        const uint32_t skip = first & ~15u;
        // fetchable >= skip: callers have checked len >= the initial states.  (Plain subtraction on purpose: a
        // saturating one is matched to v_sub_u32 ... clamp, the descriptor lands in VGPRs and every fetch
        // becomes a waterfall loop.)
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(chunk_base + skip), 0, fetchable - skip,
                                                 kRsrcFlags);
    }
    __device__ __forceinline__ void open(uint8_t *lds, uint64_t chunk_base, uint32_t first, uint32_t fetchable,
                                         uint32_t lane)
    {
        const rsrc_t r = stream_rsrc(chunk_base, first, fetchable);
        u32x4 b0, b1;
        prefetch(r, lane, b0, b1);
        install(lds, r, first, lane, b0, b1);
    }
    // open() in two halves, so that a chunk's first two blocks can be requested while the previous chunk is
    // still being decoded: prefetch() only issues loads, install() takes over the ring and requests block 2.
    static __device__ __forceinline__ void prefetch(rsrc_t r, uint32_t lane, u32x4 &b0, u32x4 &b1)
    {
        b0 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u, 0, kAuxNt);
        b1 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u + kRingBlock, 0, kAuxNt);
    }
    __device__ __forceinline__ void install(uint8_t *lds, rsrc_t r, uint32_t first, uint32_t lane, const u32x4 &b0,
                                            const u32x4 &b1)
    {
        ring = lds;
        ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
        rsrc = r;
        lapped = first & ~15u;
        gnext = 2u * kRingBlock;
        cur = ring_addr + (first & 15u);
        mark = ring_addr + kRingBlock;
        pre = fetch(lane);
        put(lane, 0, b0);
        put(lane, kRingBlock, b1);
    }
    // kMarker (the hand-scheduled decoders, whose symbol stores sit inside asm statements the compiler cannot see):
    // the compiler would wait for the prefetched block with s_waitcnt vmcnt(0) -- it knows of no younger VMEM
    // operation -- and so for every symbol store in flight as well, the newest one included; under a saturated
    // memory system that is a stall per refill.  A 4-byte marker store the compiler DOES see, issued right behind
    // each prefetch, makes it count vmcnt(1): the prefetch and everything older must have landed, the newest
    // operation (in reality: the newest symbol store) may still be on its way.  Only lane 0's marker is in range
    // of its descriptor (4 records): the hardware drops the other 63, no branch.
    rsrc_t marker_rsrc;
    u32x4 rsrc4; // the stream descriptor once more, as the four words an asm operand can take (RANS_TOUCH_AHEAD)
    template <bool kMarker = false> __device__ __forceinline__ void checkpoint(uint32_t lane)
    {
        if (cur >= mark) { // wave-uniform: a scalar compare + branch
            if (mark == ring_addr + kRingBlock) {
                put(lane, 0, pre);
                mark = ring_addr + kRingBytes;
            } else {
                cur -= kRingBytes;
                lapped += kRingBytes;
                put(lane, kRingBlock, pre);
                mark = ring_addr + kRingBlock;
            }
            pre = fetch(lane);
            if constexpr (kMarker) {
                __builtin_amdgcn_raw_buffer_store_b32(gnext, marker_rsrc, lane * 4u, 0, 0);
            }
        }
    }
    // ring offset of the read cursor (no wrap between checkpoints)
    __device__ __forceinline__ uint32_t cursor() const { return cur - ring_addr; }
    __device__ __forceinline__ void consume(uint32_t bytes) { cur += bytes; }
    // offset within the chunk of the next unread byte
    __device__ __forceinline__ uint32_t position() const { return lapped + (cur - ring_addr); }
};

// ---------------------------------------------------------------------------
// Renormalisation of one sub-step (64 lanes, ascending lane order == ascending
// stream address).  `active` masks lanes that have no symbol in this round.
// Returns the bytes consumed (wave-uniform).
// ---------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ uint32_t dec_renorm(const StreamWindow &W, typename FmtTraits<FMT>::state_t &x,
                                               bool active)
{
    if constexpr (kIsWord<FMT>) {
        // rans_word_sse41.h:134-141 / :182-227
        const bool need = active && x < (1u << 16);
        const uint64_t m = __builtin_amdgcn_ballot_w64(need);
        const uint32_t at = W.cursor() + 2u * rank_below(m);
        const uint32_t w = *reinterpret_cast<const uint16_t *>(W.ring + at);
        x = need ? ((x << 16) | w) : x;
        return 2u * (uint32_t)__builtin_popcountll(m);
    } else if constexpr (kIsR64<FMT>) {
        // rans64.h:305-316
        const bool need = active && x < (1ull << 31);
        const uint64_t m = __builtin_amdgcn_ballot_w64(need);
        const uint32_t at = W.cursor() + 4u * rank_below(m);
        const uint32_t w = *reinterpret_cast<const uint32_t *>(W.ring + at);
        x = need ? ((x << 32) | w) : x;
        return 4u * (uint32_t)__builtin_popcountll(m);
    } else {
        // rans_byte.h:307-318.  With scale_bits <= 16 a lane needs 0, 1 or 2
        // bytes: x >= 2^7 after D, and a second byte is needed iff x < 2^15.
        // The first byte read is the more significant one.
        const bool n1 = active && x < (1u << 23);
        const bool n2 = active && x < (1u << 15);
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(n1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(n2);
        const uint32_t at = W.cursor() + rank_below(m1) + rank_below(m2);
        const uint32_t b0 = W.ring[at];
        const uint32_t b1 = W.ring[at + 1];
        const uint32_t x1 = (x << 8) | b0;
        const uint32_t x2 = (x1 << 8) | b1;
        x = n2 ? x2 : (n1 ? x1 : x);
        return (uint32_t)__builtin_popcountll(m1) + (uint32_t)__builtin_popcountll(m2);
    }
}

// Hand-written renormalisation sub-step of the word format for a FULL wave (all 64
// lanes hold a state and are active): rans_word_sse41.h:134-141 for 64 lanes at once.
//   v_cmpx      lanes with x < 2^16 stay enabled; vcc = the same mask
//   s_bcnt1     words the wave consumes in this sub-step
//   v_mbcnt x2  rank of the lane among the enabled ones = its word index in the stream
//   ds_read_u16 only the enabled lanes read; v_perm merges (x << 16) | word
//   s_lshl1_add the window cursor moves on by 2 bytes per word (cur is updated in place)
// 5 VALU + 1 LDS + 5 SALU, no branch, no v_cndmask.  exec is restored to all ones,
// which is what it was (the caller runs this only in wave-uniform full-wave code).
__device__ __forceinline__ void renorm_word_full(uint32_t &x, uint32_t &cur, uint32_t k65536)
{
    uint32_t t, w, cnt;
    // gfx940+ hazard: a VALU write of an SGPR/VCC needs 2 wait states before a VALU reads it
    // as an operand (LLVM GCNHazardRecognizer, VALUWriteSGPRVALURead); hipcc does not pad
    // inside an asm statement: s_bcnt1 is one of the two, s_nop 0 the other.
    asm volatile("v_cmpx_gt_u32_e32 vcc, %[lim], %[x]\n\t"
                 "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                 "s_nop 0\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "v_lshl_add_u32 %[t], %[t], 1, %[cur]\n\t"
                 "ds_read_u16 %[w], %[t]\n\t"
                 "s_lshl1_add_u32 %[cur], %[cnt], %[cur]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_perm_b32 %[x], %[x], %[w], %[sel]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [t] "=&v"(t), [w] "=&v"(w), [cnt] "=&s"(cnt), [cur] "+s"(cur)
                 : [lim] "v"(k65536), [sel] "s"(0x05040100u)
                 : "vcc", "scc", "memory");
}

// Same for the byte formats (rans_byte.h:307-318): a lane needs 0, 1 or 2 bytes
// (x < 2^23, x < 2^15); its offset in the stream is the sum of both masks' ranks; the
// first byte is the more significant one.  9 VALU + 2 LDS, no branch; cur moves on by the
// bytes consumed.
__device__ __forceinline__ void renorm_byte_full(uint32_t &x, uint32_t &cur, uint32_t k2p23, uint32_t k2p15)
{
    uint32_t t, b0, b1, c1, c2;
    // s[52:53] = lanes that take at least one byte, vcc = lanes that take two (a subset).  Lane i's bytes are
    // consecutive (rans_byte.h:307-318 reads them in one loop), at cur + (bytes taken by the lanes below it); both
    // bytes are read under the first mask (the second read of a one-byte lane is dropped by the exec mask of its
    // v_lshl_or), which saves two exec switches: 7 SALU instead of 11 per renormalisation.
    asm volatile("v_cmp_gt_u32_e64 s[52:53], %[l23], %[x]\n\t"
                 "v_cmp_gt_u32_e32 vcc, %[l15], %[x]\n\t"
                 "s_bcnt1_i32_b64 %[c1], s[52:53]\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], s52, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], s53, %[t]\n\t"
                 "s_bcnt1_i32_b64 %[c2], vcc\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, %[t]\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "v_add_u32_e32 %[t], %[cur], %[t]\n\t"
                 "s_add_i32 %[cur], %[cur], %[c1]\n\t"
                 "s_mov_b64 exec, s[52:53]\n\t"
                 "ds_read_u8 %[b0], %[t]\n\t"
                 "ds_read_u8 %[b1], %[t] offset:1\n\t"
                 "s_add_i32 %[cur], %[cur], %[c2]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_lshl_or_b32 %[x], %[x], 8, %[b0]\n\t"
                 "s_mov_b64 exec, vcc\n\t"
                 "v_lshl_or_b32 %[x], %[x], 8, %[b1]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [t] "=&v"(t), [b0] "=&v"(b0), [b1] "=&v"(b1), [c1] "=&s"(c1), [c2] "=&s"(c2), [cur] "+s"(cur)
                 : [l23] "v"(k2p23), [l15] "v"(k2p15)
                 : "vcc", "scc", "memory", "s52", "s53");
}

// Cache policy of the symbol stores: non-temporal, system scope -- decoded symbols are written once and not
// read back by this kernel, so they should stream through instead of sitting dirty in L2 / Infinity Cache until the
// next launch has to push them out.  Measured on the headline workload (1 GiB, sustained launches): plain stores
// 0.438 ms, nt 0.412, sc1 0.422, nt sc1 0.409 (the policy of the stream LOADS makes no difference).
#define RANS_STORE_MODS " nt sc1"
constexpr int kAuxStore = 2 | 16; // the same for stores issued through builtins: nt | sc1

// Descriptor of this wave's marker word (StreamWindow::checkpoint<true>): 4 records, i.e. lane 0 only.
__device__ __forceinline__ rsrc_t marker_rsrc(const DecParams &p, uint32_t wave_in_grid)
{
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(uniform64(reinterpret_cast<uint64_t>(p.wave_scratch) + (uint64_t)uniform(wave_in_grid) * 64u)),
        0, p.wave_scratch ? 4u : 0u, kRsrcFlags);
}

// First wave start .. last wave end of a launch (rans_amd_launch_spans): the block's waves leave their end
// times in LDS (the tables there are dead by now), one thread folds them into the launch's record with two
// atomics.  One pair per BLOCK: 8192 waves hammering two words cost ~80 us per launch (an L2 atomic unit
// retires ~90 same-address atomics per microsecond).
__device__ __forceinline__ void record_span(const DecParams &p, unsigned long long t_start, uint8_t *smem)
{
    if (!p.span)
        return;
    unsigned long long *ends = reinterpret_cast<unsigned long long *>(smem);
    __syncthreads(); // every wave of the block is done with the tables
    if ((threadIdx.x & 63u) == 0)
        ends[threadIdx.x >> 6] = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long last = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w)
            last = ends[w] > last ? ends[w] : last;
        atomicMax(p.span, ~t_start);
        atomicMax(p.span + 1, last);
    }
}

// byte `kSymByte` of `raw` goes to byte J of acc, the other bytes of acc stay
template <int SYMBYTE, int J> __device__ __forceinline__ uint32_t acc_symbol(uint32_t raw, uint32_t acc)
{
    if constexpr (J == 0) {
        return raw; // fixed up by J == 1
    } else if constexpr (J == 1) {
        // byte0 <- acc[SYMBYTE] (round 0's symbol), byte1 <- raw[SYMBYTE]
        constexpr uint32_t sel = (uint32_t)SYMBYTE | ((4u + SYMBYTE) << 8) | 0x03020000u;
        return __builtin_amdgcn_perm(raw, acc, sel);
    } else {
        constexpr uint32_t ident = 0x03020100u;
        constexpr uint32_t sel = (ident & ~(0xffu << (8 * J))) | ((4u + SYMBYTE) << (8 * J));
        return __builtin_amdgcn_perm(raw, acc, sel);
    }
}

} // namespace

} // namespace rans_amd
