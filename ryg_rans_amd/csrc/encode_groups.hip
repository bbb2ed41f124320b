// encode_groups.hip -- the mirror image of decode_groups.hip's k_decode_word_groups: the reference's 8-way word layout
// (rans_word_sse41.h:57-100, main_simd.cpp:287-300: eight states, one pointer that moves DOWN), EIGHT chunks per wave, lane
// 8 g + i holds state i of chunk g of the wave's octet.
//
// The lane-per-chunk encoder (lanes.hip k_encode_lanes_staged) walks a whole chunk per lane: 10 KiB of staging rows and
// rings per wave, 11..13 waves per CU, 1.2 ms per GiB.  Here a round codes one symbol per lane:
//   * symbols: sixteen rounds are one 128-byte line of the chunk, 16 bytes per lane, taken apart into per-state bytes by
//     the inverse of the decoder's output exchange (v_cndmask_b32_dpp between the halves of the group, then the quad
//     transposes, which are their own inverse); the next line is in flight while this one is coded, rounds run from the
//     chunk's last to its first (main_simd.cpp:287-300);
//   * "who emits" (x > thresh, rans_word_sse41.h:85) is a ballot; t = its bits of MY group, in place; a lane's rank among
//     its group's emitters is v_mbcnt over t, the group's running word count c (a VGPR) moves on by v_bcnt(t); the
//     lane's word lands 2 (c' - rank) bytes below the end of the stream (ascending lane = ascending address inside a round,
//     later rounds below earlier ones: rans_word_sse41.h:93-99);
//   * a group's words go to a 256-byte ring in LDS; eight rounds emit at most 8 x 8 words = one 128-byte block, so every
//     eight rounds at most one finished block per group leaves, 16 bytes per lane, to its place below the end of the
//     chunk's slot.  The flushed states (RansWordEncFlush, state 0 lowest) follow the same way;
//   * the division x / freq is the wave encoder's (encode_common.hpp: Alverson or round-up reciprocal from the 16-byte
//     WordEncRec, exact), one ds_read_b128 per symbol.
// Slots: worst-case scratch slots (k_layout / k_compact_small behind the launch), the caller's container (slot layout), or
// sized slots -- a store that would start below the slot's first byte is dropped, and a chunk whose stream turns out longer
// than its slot is listed for the redo launch exactly as the lane encoders list theirs (EncParams::ovf_ctl).
// Chunks of a multiple of 4 symbols (launcher: the symbol loads are dword-aligned); what a chunk size off 128 leaves, and the
// input's last octet when its last chunk is a ragged one, go round by round.
//
// No MFMA: integer, table-driven, serial per state.

#include "decode_common.hpp"

namespace rans_amd {

namespace {

constexpr uint32_t kEncGrpBlock = 128;               // bytes of a group's output block: 8 lanes x 16 B
constexpr uint32_t kEncGrpRing = 2 * kEncGrpBlock;   // per group
constexpr uint32_t kEncGrpWaveLds = 8 * kEncGrpRing; // per wave
constexpr uint32_t kEncGrpThreads = 1024;
constexpr uint32_t kEncGrpTable = 256 * 16;          // WordEncRec[256] at LDS address 0

// Eight rounds, odd-numbered round first (accumulator B: rounds 15, 13, 11, 9 or 7, 5, 3, 1 -- bytes 3..0; accumulator A the
// even ones).  Per round 15 VALU (18 with the round-up reciprocal, + 1 where symbols without a record have to be found) + 2 LDS:
//   v_lshlrev (SDWA byte J)    LDS address of the symbol's record (the table starts at LDS address 0)
//   ds_read_b128               {m', thresh, cmpl | sh << 24, bias}
//   v_cmp                      vcc = the lanes that emit a word
//   v_and, v_and_or            t = my group's bits of the ballot (in one half of the wave)
//   v_mbcnt x2, v_bcnt, v_sub  rank among my group's emitters; c' = words so far; rank - c' = the word's place from the end
//   v_lshlrev, v_and_or        -> LDS byte address in the group's ring
//   ds_write_b16, v_lshrrev    under exec = vcc: the low half leaves, x >>= 16
//   v_mul_hi .. v_add          x = x + bias + (x / freq) * cmpl (encode_common.hpp RANS_ENC_WORD_TAIL_*)
// Fixed registers v56..v59 hold the record: the parts of a 128-bit asm operand cannot be named.
#define RANS_GE_HEAD(ACC, SEL, TRACKSTR)                                                                               \
    "v_lshlrev_b32_sdwa %[t0], %[k4], " ACC " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL "\n\t"   \
    "ds_read_b128 v[56:59], %[t0]\n\t"                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    "v_cmp_gt_u32_e32 vcc, %[x], v57\n\t"                                                                              \
    TRACKSTR                                                                                                           \
    "s_nop 0\n\t"                                                                                                      \
    "v_and_b32_e32 %[t0], vcc_lo, %[gmlo]\n\t"                                                                         \
    "v_and_or_b32 %[t1], vcc_hi, %[gmhi], %[t0]\n\t"                                                                   \
    "v_mbcnt_lo_u32_b32 %[t0], %[t0], 0\n\t"                                                                           \
    "v_mbcnt_hi_u32_b32 %[t0], %[t1], %[t0]\n\t"                                                                       \
    "v_bcnt_u32_b32 %[c], %[t1], %[c]\n\t"                                                                             \
    "v_sub_u32_e32 %[t0], %[t0], %[c]\n\t"                                                                             \
    "v_lshlrev_b32_e32 %[t0], 1, %[t0]\n\t"                                                                            \
    "v_and_or_b32 %[t0], %[t0], %[k255], %[ring]\n\t"                                                                  \
    "s_mov_b64 exec, vcc\n\t"                                                                                          \
    "ds_write_b16 %[t0], %[x]\n\t"                                                                                     \
    "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                           \
    "v_mul_hi_u32 %[t1], %[x], v56\n\t"
#define RANS_GE_TAIL_SMALL                                                                                             \
    "v_lshrrev_b32_sdwa %[t1], v58, %[t1] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"        \
    "v_mad_u32_u24 %[t1], %[t1], v58, %[x]\n\t"                                                                        \
    "v_add_u32_e32 %[x], %[t1], v59\n\t"
#define RANS_GE_TAIL_GM                                                                                                \
    "v_sub_u32_e32 %[t0], %[x], %[t1]\n\t"                                                                             \
    "v_lshrrev_b32_e32 %[t0], 1, %[t0]\n\t"                                                                            \
    "v_add_u32_e32 %[t1], %[t1], %[t0]\n\t"                                                                            \
    RANS_GE_TAIL_SMALL
#define RANS_GE_TRACK "v_or_b32_e32 %[worst], %[worst], v58\n\t"
#define RANS_GE_PAIR(SEL, TRACKSTR, TAIL)                                                                              \
    RANS_GE_HEAD("%[pb]", SEL, TRACKSTR) TAIL RANS_GE_HEAD("%[pa]", SEL, TRACKSTR) TAIL
#define RANS_GE_EIGHT(TRACKSTR, TAIL)                                                                                  \
    asm volatile(RANS_GE_PAIR("BYTE_3", TRACKSTR, TAIL) RANS_GE_PAIR("BYTE_2", TRACKSTR, TAIL)                           \
                 RANS_GE_PAIR("BYTE_1", TRACKSTR, TAIL) RANS_GE_PAIR("BYTE_0", TRACKSTR, TAIL)                           \
                 : [x] "+v"(x), [c] "+v"(c), [worst] "+v"(worst), [t0] "=&v"(t0), [t1] "=&v"(t1)                         \
                 : [pa] "v"(acc_a), [pb] "v"(acc_b), [k4] "v"(k4), [gmlo] "v"(gm_lo), [gmhi] "v"(gm_hi), [k255] "v"(k255), \
                   [ring] "v"(ring)                                                                                    \
                 : "vcc", "memory", "v56", "v57", "v58", "v59")
template <bool SMALL, bool TRACK>
__device__ __forceinline__ void encode_octet_8rounds(uint32_t &x, uint32_t &c, uint32_t &worst, uint32_t acc_a, uint32_t acc_b,
                                                     uint32_t k4, uint32_t gm_lo, uint32_t gm_hi, uint32_t k255, uint32_t ring)
{
    uint32_t t0, t1;
    if constexpr (SMALL && TRACK)
        RANS_GE_EIGHT(RANS_GE_TRACK, RANS_GE_TAIL_SMALL);
    else if constexpr (SMALL)
        RANS_GE_EIGHT("s_nop 0\n\t", RANS_GE_TAIL_SMALL);
    else if constexpr (TRACK)
        RANS_GE_EIGHT(RANS_GE_TRACK, RANS_GE_TAIL_GM);
    else
        RANS_GE_EIGHT("s_nop 0\n\t", RANS_GE_TAIL_GM);
}
#undef RANS_GE_EIGHT
#undef RANS_GE_PAIR
#undef RANS_GE_TRACK
#undef RANS_GE_TAIL_GM
#undef RANS_GE_TAIL_SMALL
#undef RANS_GE_HEAD

template <bool SMALL, bool TRACK>
__global__ void __launch_bounds__(kEncGrpThreads, 8) k_encode_word_groups(const EncParams p)
{
    using Tr = FmtTraits<FMT_WORD>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.word_enc_recs);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < kEncGrpTable / 16u; i += blockDim.x)
            l0[i] = g0[i];
    }
    __syncthreads();
    if (!lds_starts_at_zero(smem)) { // cannot happen without static LDS; never code on a wrong assumption
        if (threadIdx.x == 0)
            atomicOr(p.flags, 1u);
        return;
    }
    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t g = lane >> 3, i = lane & 7u;
    const uint32_t ring_c = kEncGrpTable + wave * kEncGrpWaveLds + g * kEncGrpRing; // raw LDS address of the group's ring (256-byte aligned)
    uint32_t ring = ring_c;
    uint32_t gm_lo = g < 4 ? 0xffu << (8u * g) : 0u, gm_hi = g >= 4 ? 0xffu << (8u * (g - 4u)) : 0u;
    uint32_t k255 = 255u, k4 = 4u;
    asm volatile("v_mov_b32 %0, %0" : "+v"(ring)); // opaque: keep them in VGPRs
    asm volatile("v_mov_b32 %0, %0" : "+v"(gm_lo));
    asm volatile("v_mov_b32 %0, %0" : "+v"(gm_hi));
    asm volatile("v_mov_b32 %0, %0" : "+v"(k255));
    asm volatile("v_mov_b32 %0, %0" : "+v"(k4));
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t lines = uniform(p.chunk_syms >> 7); // 16 rounds of 8 symbols each
    const uint64_t octets = (p.nchunks + 7u) >> 3;     // (the last one may hold fewer than eight chunks)
    const uint32_t slot = (uint32_t)p.slot_bytes;
    const bool ragged = p.n % p.chunk_syms != 0; // the input's last chunk is short: its octet takes the round-by-round path

    uint32_t worst = 0;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t octet_v = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    // a claim covers at least 8192 symbols (one atomic unit retires ~90 claims per microsecond)
    const uint32_t per_claim = uniform(p.chunk_syms >= 1024u ? 1u : (1024u + p.chunk_syms - 1u) / p.chunk_syms);
    const uint64_t claims = (octets + per_claim - 1u) / per_claim;
    for (;;) {
        if (p.claims) { // dynamic hand-out (pool = blockIdx % 8 owns the claims c with c % 8 == pool), as in the decoders
            uint32_t got = 0;
            if (lane == 0)
                got = atomicAdd(p.claims + pool * kWorkPoolStride, 1u);
            octet_v = (uint64_t)uniform(got) * npools + pool;
        }
        if (octet_v >= claims)
            break;
        const uint64_t claim = uniform64(octet_v);
        octet_v += total_waves;
        const uint64_t o_end = (claim + 1u) * per_claim < octets ? (claim + 1u) * per_claim : octets;
        for (uint64_t vk = claim * per_claim; vk < o_end; ++vk) {
            // (a ragged last chunk: its octet takes three times an octet's usual time -- the work is dealt back to front then, and
            //  that one starts first; 1 GiB + 8191 symbols in 16 Ki-symbol chunks: 0.76 -> 0.63 ms)
            const uint64_t octet = ragged ? octets - 1u - vk : vk;
            const uint64_t chunk = octet * 8u + g;
            const bool valid = chunk < p.nchunks;
            // symbols through a descriptor of the octet's input (the running offset is an SGPR), stream blocks through one of its slots
            const rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t *>(p.syms) + octet * 8u * p.chunk_syms, 0, 8u * p.chunk_syms, kRsrcFlags);
            const rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.scratch + octet * 8u * p.slot_bytes, 0, 8u * slot, kRsrcFlags);
            const uint32_t in_off = valid ? g * p.chunk_syms + 16u * i : 0x80000000u; // (a chunk that does not exist reads zeros)
            const uint32_t slot_end = (g + 1u) * slot;                                 // offset of the END of my chunk's slot in the octet's
            uint32_t x = Tr::kL, c = 0, fb = 0; // state, words emitted so far (the group's), blocks flushed
            // block fb of the ring -> its place below the slot's end; a piece that would start below the slot's first byte is dropped
            // (sized slots: the chunk then does not fit, its length says so at the end)
            auto flush_block = [&](uint32_t have_bytes) { // have_bytes: the group's stream so far
                const uint32_t below = kEncGrpBlock * (fb + 1u) - 16u * i; // this lane's piece starts `below` bytes under the slot's end
                if (valid && below <= slot && below - 16u < have_bytes) {
                    const u32x4 v = *reinterpret_cast<RANS_LDS const u32x4 *>((uintptr_t)(ring_c + ((fb & 1u) ? 0u : kEncGrpBlock) + 16u * i));
                    __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, slot_end - below, 0, 0);
                }
                fb += 1u;
            };
            auto sixteen = [&](const u32x4 &v) { // one 128-byte line of the chunk, last round first
                uint32_t a0, a1, a2, a3;
                // rows 2 i, 2 i + 1 of the line -> this state's column: the halves of the group swap what the decoder's output
                // exchange gave them (decode_groups.hip), then the quad transposes
                asm volatile("s_nop 1\n\t"
                             "s_mov_b64 vcc, %[lower]\n\t"
                             "v_cndmask_b32_dpp %[a0], %[vy], %[vx], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" // lower ? mine : y of lane - 4
                             "v_cndmask_b32_dpp %[a1], %[vw], %[vz], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "s_mov_b64 vcc, %[upper]\n\t"
                             "v_cndmask_b32_dpp %[a2], %[vx], %[vy], vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" // upper ? mine : x of lane + 4
                             "v_cndmask_b32_dpp %[a3], %[vz], %[vw], vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
                             : [vx] "v"(v.x), [vy] "v"(v.y), [vz] "v"(v.z), [vw] "v"(v.w), [lower] "s"(0x0f0f0f0f0f0f0f0full),
                               [upper] "s"(0xf0f0f0f0f0f0f0f0ull)
                             : "vcc");
                a0 = quad_transpose(a0, sel1, sel2); // byte J of a_k: round 8 (k >> 1) + 2 J + (k & 1)
                a1 = quad_transpose(a1, sel1, sel2);
                a2 = quad_transpose(a2, sel1, sel2);
                a3 = quad_transpose(a3, sel1, sel2);
                // (the block that the LAST eight rounds of the line before may have finished leaves here, behind the wait for this
                //  line's symbols: in front of it the wait -- s_waitcnt vmcnt(0), the compiler cannot count conditional stores -- would
                //  be for a store that has only just been issued)
                if (c >= 64u * (fb + 1u))
                    flush_block(2u * c);
                encode_octet_8rounds<SMALL, TRACK>(x, c, worst, a2, a3, k4, gm_lo, gm_hi, k255, ring); // rounds 15 .. 8
                if (c >= 64u * (fb + 1u))
                    flush_block(2u * c);
                encode_octet_8rounds<SMALL, TRACK>(x, c, worst, a0, a1, k4, gm_lo, gm_hi, k255, ring); // rounds 7 .. 0
            };
            // one round of the compiler-scheduled kind: lanes without a symbol sit it out (main_simd.cpp:287-300 with in_size % 8 !=
            // 0: the partial round belongs to states 0 .. in_size % 8 - 1).  For what a chunk size off 128 leaves, and for the
            // input's last octet when its last chunk is a ragged one.
            auto one_round = [&](uint32_t sym, bool active) {
                const u32x4 rec = *reinterpret_cast<RANS_LDS const u32x4 *>((uintptr_t)(sym << 4));
                const bool emit = active && x > rec.y; // rans_word_sse41.h:85
                const uint64_t m = __builtin_amdgcn_ballot_w64(emit);
                const uint32_t t_lo = (uint32_t)m & gm_lo, t_hi = (uint32_t)(m >> 32) & gm_hi;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(t_hi, __builtin_amdgcn_mbcnt_lo(t_lo, 0u));
                c += (uint32_t)__builtin_popcount(t_lo) + (uint32_t)__builtin_popcount(t_hi);
                if (emit) {
                    *reinterpret_cast<RANS_LDS uint16_t *>((uintptr_t)(ring_c | ((2u * (rank - c)) & 255u))) = (uint16_t)x;
                    x >>= 16;
                }
                if (active) { // encode_common.hpp RANS_ENC_WORD_TAIL_*: x += bias + (x / freq) * cmpl
                    uint32_t q = __umulhi(x, rec.x);
                    if constexpr (!SMALL)
                        q += (x - q) >> 1;
                    q >>= rec.z >> 24;
                    x = x + rec.w + (q & 0xffffffu) * (rec.z & 0xffffffu);
                    worst |= rec.z;
                }
            };
            const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + chunk * p.chunk_syms;
            if (octet + 1u == octets && ragged) { // one octet of the whole input: a byte load per lane and round
                uint32_t nsym = 0;
                if (valid)
                    nsym = chunk + 1u == p.nchunks ? (uint32_t)(p.n - chunk * p.chunk_syms) : p.chunk_syms;
                for (uint32_t r = (p.chunk_syms + 7u) >> 3; r-- > 0;) {
                    const bool active = r * 8u + i < nsym;
                    one_round(active ? (uint32_t)src[r * 8u + i] : 0u, active);
                    if ((r & 7u) == 0 && c >= 64u * (fb + 1u))
                        flush_block(2u * c);
                }
            } else {
                for (uint32_t r = (p.chunk_syms + 7u) >> 3; r-- > 16u * lines;) { // the rounds behind the chunk's last whole line
                    const bool active = valid && r * 8u + i < p.chunk_syms;
                    one_round(active ? (uint32_t)src[r * 8u + i] : 0u, active);
                    if ((r & 7u) == 0 && c >= 64u * (fb + 1u))
                        flush_block(2u * c);
                }
                if (lines) {
                    u32x4 next = __builtin_amdgcn_raw_buffer_load_b128(irsrc, in_off, 128u * (lines - 1u), kAuxNt);
                    for (uint32_t q = lines; q-- > 0;) {
                        const u32x4 cur = next;
                        if (q)
                            next = __builtin_amdgcn_raw_buffer_load_b128(irsrc, in_off, 128u * (q - 1u), kAuxNt);
                        sixteen(cur);
                    }
                }
            }
            if (c >= 64u * (fb + 1u)) // (the last eight rounds' block)
                flush_block(2u * c);
            // the final states, state 0 lowest (RansWordEncFlush: rans_word_sse41.h:104-113 reads them back in that order): lane
            // i's dword ends 32 - 4 i bytes ... starts 2 c + 32 - 4 i bytes below the slot's end
            {
                const uint32_t at = 0u - (2u * c + 32u - 4u * i);
                *reinterpret_cast<RANS_LDS uint16_t *>((uintptr_t)(ring_c | (at & 255u))) = (uint16_t)x;
                *reinterpret_cast<RANS_LDS uint16_t *>((uintptr_t)(ring_c | ((at + 2u) & 255u))) = (uint16_t)(x >> 16);
                c += 16u;
            }
            const uint32_t len = 2u * c;
            if (64u * fb < c)
                flush_block(len);
            if (64u * fb < c)
                flush_block(len);
            if (valid && i == 0) {
                p.lengths[chunk] = len;
                if (len > slot) { // (sized slots only: a worst-case slot holds every stream)
                    if (p.ovf_ctl)
                        p.ovf_list[atomicAdd(p.ovf_ctl, 1u)] = (uint32_t)chunk;
                } else if (p.slot_layout) {
                    p.offsets[chunk] = (chunk + 1u) * p.slot_bytes - len;
                    if (chunk + 1 == p.nchunks)
                        p.offsets[p.nchunks] = p.nchunks * p.slot_bytes;
                }
            }
        }
    }
    if (TRACK && __builtin_amdgcn_ballot_w64((worst >> 31) != 0) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
}

} // namespace

// (api.cpp asks this while it still fills the parameters in: only what it sets first counts -- the shape, the symbol buffer, the
//  slot size, and that neither the fused placement nor a redo is asked for; scratch and container are 16-byte aligned by then)
bool encode_word_groups_applicable(const EncParams &p)
{
    return p.n_ways == 8 && p.sym_bytes == 1 && !p.status && !p.redo && !p.no_lanes && p.nsyms <= 256 && (p.chunk_syms & 3u) == 0 &&
           p.chunk_syms >= 32 && p.chunk_syms <= (1u << 20) && p.nchunks >= 8 && (reinterpret_cast<uintptr_t>(p.syms) & 15u) == 0 &&
           (p.slot_bytes & 15u) == 0 && p.slot_bytes >= 48 && p.slot_bytes < (1ull << 28);
}

hipError_t launch_encode_word_groups(const EncParams &p, int num_cus, hipStream_t stream, const char **name)
{
    if (!p.word_enc_recs || (reinterpret_cast<uintptr_t>(p.scratch) & 15u) != 0)
        return hipErrorInvalidValue;
    const size_t lds = kEncGrpTable + (size_t)(kEncGrpThreads / 64) * kEncGrpWaveLds;
    const uint64_t octets = (p.nchunks + 7u) / 8u;
    const uint64_t want_blocks = (octets + kEncGrpThreads / 64 - 1) / (kEncGrpThreads / 64);
    const uint64_t cap = (uint64_t)num_cus * 2u;
    const uint32_t grid = (uint32_t)(want_blocks < cap ? want_blocks : cap);
    if (name)
        *name = "k_encode_word_groups";
    const bool small = p.word_small != 0, track = p.dense256 == 0;
    if (small && track)
        RANS_LAUNCH((k_encode_word_groups<true, true>), dim3(grid), dim3(kEncGrpThreads), lds, stream, p);
    else if (small)
        RANS_LAUNCH((k_encode_word_groups<true, false>), dim3(grid), dim3(kEncGrpThreads), lds, stream, p);
    else if (track)
        RANS_LAUNCH((k_encode_word_groups<false, true>), dim3(grid), dim3(kEncGrpThreads), lds, stream, p);
    else
        RANS_LAUNCH((k_encode_word_groups<false, false>), dim3(grid), dim3(kEncGrpThreads), lds, stream, p);
    return hipGetLastError();
}

} // namespace rans_amd
