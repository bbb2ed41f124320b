// decode_groups.hip -- decoders for two of the reference's own narrow layouts with 64 / N chunks per wave and ONE STATE PER
// LANE: k_decode_word_groups, the 8-way word layout (rans_word_sse41.h:151-227: two SSE registers of four states, one shared
// cursor) with lane 8 g + i = state i of chunk g of the wave's octet, and further down k_decode_byte_pairs, the 2-way byte
// layout of main.cpp:226-280 with lane 2 g + i.
//
// The lane-per-chunk kernels (lanes.hip) give every lane a whole chunk: 64 private streams per wave, a 136-byte ring
// row per lane in LDS (8.7 KiB per wave beside the 32 KiB slot table: 13..14 waves per CU), and their pace is set by
// how few waves a CU can hold (profiles/r06_small_abs.md).  Here a chunk is decoded the way the wave-per-chunk
// kernels do it (decode_wave.hip) -- one symbol per lane per round, "who renormalises" is a ballot, a lane's place in
// its stream is a population count over the lanes below it -- only that the ballot is cut into eight bytes, one per
// chunk:
//   * mask_g = ballot & (bits of group g), held per lane in a VGPR; v_mbcnt counts ITS bits below the lane, and with
//     the group's cursor as the instruction's addend the result is the lane's word position outright; v_bcnt with the
//     cursor as addend is the group's next cursor (every lane of the group computes the same);
//   * a group's stream goes through a 256-byte ring in LDS (2 KiB per wave: 32 waves per CU beside the table), fetched
//     in aligned 128-byte blocks, 16 bytes per lane, the next block parked in registers one refill ahead.  Eight rounds
//     consume at most 8 x 8 x 2 = 128 bytes per group, so a group needs at most one block per eight rounds: one
//     compare per lane every eight rounds, the refill itself runs under the exec mask of the groups that need it;
//   * sixteen rounds are one 128-byte line of the chunk: four rounds of symbols are transposed inside the quads (as in
//     decode_wave.hip), the halves of the group exchange dwords (v_cndmask_b32_dpp), and every lane stores 16 bytes -- pieces
//     of a line that leave at different times reach memory as partial lines (profiles/r06_word8_groups.md: 2.05 ms with
//     one dword per lane every four rounds, 0.51 with whole lines).
// Chunks of a multiple of 4 symbols -- the output is stored in dwords -- (the launcher hands everything else to the lane
// kernel); a ragged last chunk sends the input's last octet one round at a time; what a chunk size off 128 leaves goes four rounds, then one round
// at a time.  The mirror image, the 8-way encoder: encode_groups.hip.
//
// No MFMA: integer, table-driven, serial per state.

#include "decode_common.hpp"

namespace rans_amd {

namespace {

constexpr uint32_t kGrpBlock = 128;             // bytes a group fetches at a time: 8 lanes x 16 B
constexpr uint32_t kGrpRing = 2 * kGrpBlock;    // per group
constexpr uint32_t kGrpWaveLds = 8 * kGrpRing;  // per wave
constexpr uint32_t kGrpThreads = 1024;
constexpr uint32_t kGrpClaimSyms = 8192;        // symbols one claim of the work counter covers, at least

__device__ __forceinline__ uint32_t lds_u16(uint32_t addr)
{
    return *reinterpret_cast<RANS_LDS const uint16_t *>((uintptr_t)addr);
}

// Eight rounds of the eight chunks as one hand-scheduled sequence: rans_word_sse41.h:123-141 (the D step and the
// renormalisation of RansWordDecSym / RansWordDecRenorm), 14 VALU + 2 LDS per round:
//   v_and, v_lshlrev            slot = x & 4095 -> LDS byte address of the slot record (the table starts at LDS address 0)
//   ds_read_b64                 {freq | sym << 24, bias}
//   v_lshrrev, v_mad_u32_u24    x = freq * (x >> 12) + bias
//   v_cmp                       vcc = the lanes that renormalise (x < 2^16)
//   v_perm                      the symbol joins its accumulator (rounds 2..7; it also is one of the two wait states a
//                               VALU read of a VALU-written vcc needs on gfx940+, s_nop the other)
//   v_and, v_and_or             t_lo = vcc_lo & my group's bits, t = (vcc_hi & my group's bits) | t_lo -- a group lies in
//                               one half of the wave, so t is the group's byte of the ballot in place
//   v_mbcnt_lo(t_lo, cursor), v_mbcnt_hi(t, .)   cursor + renormalising lanes of MY group below me: for lanes 0..31
//                               v_mbcnt_hi adds nothing, for lanes 32..63 t_lo is zero and v_mbcnt_lo adds nothing
//   v_bcnt(t, cursor)           the group's next cursor (words)
//   v_lshlrev, v_and_or         word position -> LDS byte address inside the group's 256-byte ring
//   ds_read_u16, v_perm         under exec = vcc: x = (x << 16) | word
// Rounds alternate between two accumulators (A: rounds 0, 2, 4, 6; B: 1, 3, 5, 7), see the output transposition below.
// Fixed registers v56..v63: the halves of a 64-bit asm operand cannot be named.
#define RANS_G_LOOKUP(PAIR, LO, HI)                        \
    "v_and_b32_e32 v62, %[m12], %[x]\n\t"                  \
    "v_lshlrev_b32_e32 v62, 3, v62\n\t"                    \
    "ds_read_b64 " PAIR ", v62\n\t"                        \
    "v_lshrrev_b32_e32 v63, 12, %[x]\n\t"                  \
    "s_waitcnt lgkmcnt(0)\n\t"                             \
    "v_mad_u32_u24 %[x], " LO ", v63, " HI "\n\t"
#define RANS_G_RENORM(FILL)                                \
    "v_cmp_gt_u32_e32 vcc, %[lim], %[x]\n\t"               \
    FILL                                                   \
    "v_and_b32_e32 v62, vcc_lo, %[gmlo]\n\t"               \
    "v_and_or_b32 v63, vcc_hi, %[gmhi], v62\n\t"           \
    "v_mbcnt_lo_u32_b32 v62, v62, %[cur]\n\t"              \
    "v_mbcnt_hi_u32_b32 v62, v63, v62\n\t"                 \
    "v_bcnt_u32_b32 %[cur], v63, %[cur]\n\t"               \
    "v_lshlrev_b32_e32 v62, 1, v62\n\t"                    \
    "v_and_or_b32 v62, v62, %[k255], %[ring]\n\t"          \
    "s_mov_b64 exec, vcc\n\t"                              \
    "ds_read_u16 v63, v62\n\t"                             \
    "s_waitcnt lgkmcnt(0)\n\t"                             \
    "v_perm_b32 %[x], %[x], v63, %[selm]\n\t"              \
    "s_mov_b64 exec, -1\n\t"
__device__ __forceinline__ void decode_octet_8rounds(uint32_t &x, uint32_t &curw, uint32_t &acc_a, uint32_t &acc_b, uint32_t m12,
                                                     uint32_t lim, uint32_t gm_lo, uint32_t gm_hi, uint32_t k255, uint32_t ring,
                                                     uint32_t sel_a, uint32_t sel_b, uint32_t sel_c)
{
    asm volatile(
        RANS_G_LOOKUP("v[56:57]", "v56", "v57") RANS_G_RENORM("s_nop 1\n\t")
        RANS_G_LOOKUP("v[58:59]", "v58", "v59") RANS_G_RENORM("s_nop 1\n\t")
        RANS_G_LOOKUP("v[60:61]", "v60", "v61") RANS_G_RENORM("v_perm_b32 %[pa], v60, v56, %[selA]\n\ts_nop 0\n\t")
        RANS_G_LOOKUP("v[60:61]", "v60", "v61") RANS_G_RENORM("v_perm_b32 %[pb], v60, v58, %[selA]\n\ts_nop 0\n\t")
        RANS_G_LOOKUP("v[60:61]", "v60", "v61") RANS_G_RENORM("v_perm_b32 %[pa], v60, %[pa], %[selB]\n\ts_nop 0\n\t")
        RANS_G_LOOKUP("v[60:61]", "v60", "v61") RANS_G_RENORM("v_perm_b32 %[pb], v60, %[pb], %[selB]\n\ts_nop 0\n\t")
        RANS_G_LOOKUP("v[60:61]", "v60", "v61") RANS_G_RENORM("v_perm_b32 %[pa], v60, %[pa], %[selC]\n\ts_nop 0\n\t")
        RANS_G_LOOKUP("v[60:61]", "v60", "v61") RANS_G_RENORM("v_perm_b32 %[pb], v60, %[pb], %[selC]\n\ts_nop 0\n\t")
        : [x] "+v"(x), [cur] "+v"(curw), [pa] "=&v"(acc_a), [pb] "=&v"(acc_b)
        : [m12] "v"(m12), [lim] "v"(lim), [gmlo] "v"(gm_lo), [gmhi] "v"(gm_hi), [k255] "v"(k255), [ring] "v"(ring),
          [selA] "v"(sel_a), [selB] "v"(sel_b), [selC] "v"(sel_c), [selm] "s"(0x05040100u)
        : "vcc", "memory", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}
#undef RANS_G_LOOKUP
#undef RANS_G_RENORM

__global__ void __launch_bounds__(kGrpThreads, 8) k_decode_word_groups(const DecParams p)
{
    using Tr = FmtTraits<FMT_WORD>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u; // WordSlot[4096]: 32 KiB, a multiple of the ring size
    {
        const uint4 *g0 = reinterpret_cast<const uint4 *>(p.table0);
        uint4 *l0 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < t0_bytes / 16u; i += blockDim.x)
            l0[i] = g0[i];
    }
    __syncthreads();
    DecTables<FMT_WORD> T;
    T.init(smem, smem + t0_bytes, p.scale_bits, p.log2nsyms);
    if (!lds_starts_at_zero(smem)) { // cannot happen without static LDS; never decode on a wrong assumption
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
    const uint32_t g = lane >> 3, i = lane & 7u;
    // raw LDS byte address of this group's ring (LDS starts at zero; 256-byte aligned: positions wrap with an and-or)
    const uint32_t ring = t0_bytes + wave * kGrpWaveLds + g * kGrpRing;
    // Ring positions carry a per-group bias of 16 g bytes: the groups' cursors move at the same average pace, and eight
    // rings 256 bytes apart would otherwise sit on the same banks.
    const uint32_t bias = 16u * g;
    // this group's bits of the ballot's halves (one of the two is zero)
    uint32_t gm_lo = g < 4 ? 0xffu << (8u * g) : 0u, gm_hi = g >= 4 ? 0xffu << (8u * (g - 4u)) : 0u;
    asm volatile("v_mov_b32 %0, %0" : "+v"(gm_lo)); // opaque: keep them in VGPRs
    asm volatile("v_mov_b32 %0, %0" : "+v"(gm_hi));
    uint32_t k255 = 255u, k65536 = 0x10000u;
    // v_perm selectors of the symbol accumulators (decode_common.hpp acc_symbol for byte 3 of the slot record's first word)
    uint32_t sel_a = 0x03020703u, sel_b = 0x03070100u, sel_c = 0x07020100u;
    asm volatile("v_mov_b32 %0, %0" : "+v"(sel_a));
    asm volatile("v_mov_b32 %0, %0" : "+v"(sel_b));
    asm volatile("v_mov_b32 %0, %0" : "+v"(sel_c));
    asm volatile("v_mov_b32 %0, %0" : "+v"(k255));
    asm volatile("v_mov_b32 %0, %0" : "+v"(k65536));

    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t glo = cbase & ~uint64_t(15), glimit = (cbase + p.container_bytes + 15u) & ~uint64_t(15);
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t groups16 = uniform(p.chunk_syms >> 7);      // 16 rounds of 8 symbols
    const uint32_t rem4 = uniform((p.chunk_syms >> 5) & 3u);   // + up to three times 4 rounds
    const uint32_t left_syms = uniform(p.chunk_syms & 31u);    // + up to 31 symbols
    const uint64_t octets = (p.nchunks + 7u) >> 3; // (the last one may hold fewer than eight chunks)
    const uint32_t per_claim = uniform(p.chunk_syms >= kGrpClaimSyms / 8u ? 1u : (kGrpClaimSyms / 8u + p.chunk_syms - 1u) / p.chunk_syms);
    const uint64_t claims = (octets + per_claim - 1u) / per_claim;

    auto load16 = [&](uint64_t a) -> u32x4 { // 16 bytes of the container, zeros beyond its granules
        u32x4 v = {0u, 0u, 0u, 0u};
        if (a >= glo && a < glimit)
            v = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(a));
        return v;
    };

    uint32_t nbad = 0;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t claim_v = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    // Work is handed out dynamically, as in decode_wave.hip k_decode (pool = blockIdx % 8 owns the claims c with c % 8 == pool).
    // Nothing is claimed or fetched ahead of its octet: the kernel is bound by VALU and LDS issue (eight waves per SIMD), a
    // wave's trips to memory in front of an octet are covered by the other seven, and measured (profiles/r06_word8_groups.md)
    // a claim held while another octet is decoded keeps work from the waves that run dry: + 6 % at 16 octets per wave, + 15 %
    // at one; index entries fetched during the last rounds only add to what the refills' s_waitcnt vmcnt(0) waits for.
    auto claim_take = [&]() -> uint64_t {
        if (p.work_counter) {
            uint32_t got = 0;
            if (lane == 0)
                got = atomicAdd(p.work_counter + pool * kWorkPoolStride, 1u);
            return (uint64_t)uniform(got) * npools + pool;
        }
        const uint64_t c = claim_v;
        claim_v += total_waves;
        return uniform64(c);
    };
    // A ragged last chunk (p.n % p.chunk_syms != 0) is decoded here as well: its octet, the input's last, goes one round at a
    // time with the lanes that have no symbol sitting out -- three times an octet's usual time, so the octets are handed out
    // back to front then and that one starts first.
    const bool ragged = p.n % p.chunk_syms != 0;
    auto octet_of = [&](uint64_t k) -> uint64_t { return ragged ? octets - 1u - k : k; }; // k: the order of the hand-out
    uint64_t vk, o_end;
    bool have;
    {
        const uint64_t c = claim_take();
        have = c < claims;
        vk = c * per_claim;
        o_end = (c + 1u) * per_claim < octets ? (c + 1u) * per_claim : octets;
    }
    uint64_t off = 0;
    uint32_t len = 0;
    if (have && octet_of(vk) * 8u + g < p.nchunks) {
        off = p.offsets[octet_of(vk) * 8u + g];
        len = p.lengths[octet_of(vk) * 8u + g];
    }
    while (have) {
        {
            const uint64_t octet = octet_of(vk);
            const bool exists = octet * 8u + g < p.nchunks;
            const bool valid = exists && (off & 1u) == 0 && len >= 8u * 4u && off <= p.container_bytes && len <= p.container_bytes - off;
            if (exists && !valid && i == 0)
                nbad++;
            const uint64_t src = cbase + (valid ? off : 0u);
            uint32_t x = Tr::kL;
            if (valid) // RansDecInit order: state 0 first (rans_word_sse41.h:104-113)
                x = reinterpret_cast<const uint32_t RANS_GLOBAL *>(src)[i];
            // positions count from the 128-byte line of the chunk's first byte
            const uint64_t abase = src & ~uint64_t(kGrpBlock - 1u);
            const uint32_t start = (uint32_t)(src - abase) + 8u * 4u; // < 160
            uint32_t curw = (start + bias) >> 1; // the cursor in words (biased)
            uint64_t ld = abase + 16u * i;
            const u32x4 b0 = load16(ld), b1 = load16(ld + kGrpBlock);
            u32x4 pend = load16(ld + 2u * kGrpBlock); // block 2; the ring holds blocks nb - 2 and nb - 1, pend is block nb
            ld += 3u * kGrpBlock;
            // 16-byte pieces at ld, ld + 128, ... that still lie inside the container's granules: the refills count them down instead
            // of comparing addresses (a piece beyond the end is not fetched; the ring then keeps what it had -- only a damaged
            // stream reads that far, and the integrity check is what catches those)
            uint32_t left = 0;
            if (ld < glimit) {
                const uint64_t pieces = (glimit - ld + (kGrpBlock - 1u)) / kGrpBlock;
                left = pieces < 0x7fffffffu ? (uint32_t)pieces : 0x7fffffffu;
            }
            *reinterpret_cast<RANS_LDS u32x4 *>((uintptr_t)(ring | ((16u * i + bias) & 255u))) = b0;
            *reinterpret_cast<RANS_LDS u32x4 *>((uintptr_t)(ring | ((kGrpBlock + 16u * i + bias) & 255u))) = b1;
            uint32_t wr = ring | ((16u * i + bias) & 255u); // where block nb goes
            uint32_t thr = (kGrpBlock + bias) >> 1;         // cursor (words) from which block nb has to be in the ring
            auto checkpoint = [&]() {
                if (curw >= thr) { // (the same for the eight lanes of a group)
                    *reinterpret_cast<RANS_LDS u32x4 *>((uintptr_t)wr) = pend;
                    wr ^= kGrpBlock;
                    thr += kGrpBlock / 2u;
                    if (left) {
                        pend = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(ld));
                        left--;
                    }
                    ld += kGrpBlock;
                }
            };
            checkpoint(); // the states may end in block 1

            // symbol stores through a descriptor of the octet's output: the running offset is an SGPR
            const rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(reinterpret_cast<uint64_t>(p.out) + octet * 8u * p.chunk_syms), 0, 8u * p.chunk_syms, kRsrcFlags);
            const uint32_t out_off16 = valid ? g * p.chunk_syms + 16u * i : 0x80000000u; // (an invalid chunk's stores: dropped by the range check)
            uint32_t osoff = 0;
#define RANS_GROUP_ROUND(ACC, J)                                                                        \
    {                                                                                                   \
        ACC = acc_symbol<Tr::kSymByte, J>(dec_step<FMT_WORD>(T, x), ACC);                               \
        const bool need = x < k65536;                                                                   \
        const uint64_t m = __builtin_amdgcn_ballot_w64(need);                                           \
        const uint32_t t_lo = (uint32_t)m & gm_lo, t_hi = (uint32_t)(m >> 32) & gm_hi;                  \
        /* the cursor + this group's renormalising lanes below me (v_mbcnt adds its second operand) */  \
        const uint32_t at = __builtin_amdgcn_mbcnt_hi(t_hi, __builtin_amdgcn_mbcnt_lo(t_lo, curw));     \
        curw += __builtin_popcount(t_lo) + __builtin_popcount(t_hi);                                    \
        if (need)                                                                                       \
            x = (x << 16) | lds_u16(((at << 1) & k255) | ring);                                         \
    }
            // Sixteen rounds = one 128-byte line per group.  Round r's symbols go to byte (r >> 1) & 3 of accumulator
            // (r & 1) + 2 (r >> 3): after the quad transposes lane (m = i & 3, h = i >> 2) holds the dwords (row 2 m, half h),
            // (row 2 m + 1, half h), (row 8 + 2 m, h), (row 9 + 2 m, h) of the line's sixteen 8-byte rows; the halves h = 0
            // keep the first two and take their other halves from lane i + 4, the halves h = 1 the last two from lane i - 4:
            // lane i then holds bytes [16 i, 16 i + 16) of the line, one 16-byte store per lane.
            const bool by_rounds = ragged && octet + 1u == octets; // (wave-uniform)
            for (uint32_t q = 0; q < (by_rounds ? 0u : groups16); ++q) {
                uint32_t a0, a1, a2, a3;
                // (the refill check BEHIND each eight rounds, not in front: the store of the line before then is eight rounds old when a
                //  refill's s_waitcnt vmcnt(0) comes -- in front of the rounds it had just been issued; 0.519 -> 0.503 ms)
                decode_octet_8rounds(x, curw, a0, a1, T.mask12v, k65536, gm_lo, gm_hi, k255, ring, sel_a, sel_b, sel_c);
                checkpoint();
                decode_octet_8rounds(x, curw, a2, a3, T.mask12v, k65536, gm_lo, gm_hi, k255, ring, sel_a, sel_b, sel_c);
                checkpoint();
                a0 = quad_transpose(a0, sel1, sel2);
                a1 = quad_transpose(a1, sel1, sel2);
                a2 = quad_transpose(a2, sel1, sel2);
                a3 = quad_transpose(a3, sel1, sel2);
                // (v_cndmask_b32_dpp by hand: left to the compiler, the select becomes a branch around the DPP move, and a DPP
                // move under an exec mask reads zeros from the lanes the mask has switched off -- the very lanes it is after)
                u32x4 v;
                asm volatile("s_nop 1\n\t"
                             "s_mov_b64 vcc, %[lower]\n\t"
                             "v_cndmask_b32_dpp %[vx], %[a2], %[a0], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" // lower ? a0 : a2 of lane - 4
                             "v_cndmask_b32_dpp %[vz], %[a3], %[a1], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "s_mov_b64 vcc, %[upper]\n\t"
                             "v_cndmask_b32_dpp %[vy], %[a0], %[a2], vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" // upper ? a2 : a0 of lane + 4
                             "v_cndmask_b32_dpp %[vw], %[a1], %[a3], vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : [vx] "=&v"(v.x), [vy] "=&v"(v.y), [vz] "=&v"(v.z), [vw] "=&v"(v.w)
                             : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [lower] "s"(0x0f0f0f0f0f0f0f0full),
                               [upper] "s"(0xf0f0f0f0f0f0f0f0ull)
                             : "vcc");
                // (chunk sizes off 64: a chunk's lines straddle the memory's at odd places, and the pieces of a memory line arrive
                //  sixteen rounds apart -- plain stores let L2 put them together: 1000-symbol chunks 1.10 -> 0.77 ms, 4000: 0.90 ->
                //  0.65; on the 64-byte grid `nt sc1` wins: 960-symbol chunks 0.52 against 0.59)
                if (p.chunk_syms & 63u)
                    __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, out_off16, osoff, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, out_off16, osoff, kAuxStore);
                osoff += 128u;
            }
            for (uint32_t q = 0; q < (by_rounds ? 0u : rem4); ++q) { // what is left of a chunk that is not a multiple of 128 symbols: 4 rounds a time
                uint32_t acc = 0;
                RANS_GROUP_ROUND(acc, 0)
                RANS_GROUP_ROUND(acc, 1)
                RANS_GROUP_ROUND(acc, 2)
                RANS_GROUP_ROUND(acc, 3)
                if (q & 1u)
                    checkpoint();
                const uint32_t v = quad_transpose(acc, sel1, sel2);
                const uint32_t li = lane_id() & 7u; // lane (m, h) of the group holds row m, states 4 h .. 4 h + 3
                __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, out_off16 - 16u * li + (li & 3u) * 8u + (li >> 2) * 4u, osoff, 0);
                osoff += 32u;
            }
#undef RANS_GROUP_ROUND
            // what a chunk size off 32 leaves (at most 31 symbols: three rounds and a partial one, main_simd.cpp:313-332 with
            // in_size % 8 != 0) -- or, in the ragged chunk's octet, every round: lanes without a symbol sit the round out, a byte
            // store per lane
            uint32_t nsym = valid ? p.chunk_syms : 0u, r_first = p.chunk_syms - left_syms;
            if (by_rounds) {
                r_first = 0u;
                if (valid && octet * 8u + g + 1u == p.nchunks)
                    nsym = (uint32_t)(p.n - (octet * 8u + g) * p.chunk_syms);
            }
            for (uint32_t r0 = r_first; r0 < p.chunk_syms; r0 += 8u) {
                if (by_rounds && (r0 & 63u) == 0)
                    checkpoint();
                const bool active = r0 + i < nsym;
                uint32_t raw = 0;
                if (active)
                    raw = dec_step<FMT_WORD>(T, x);
                const bool need = active && x < k65536;
                const uint64_t m = __builtin_amdgcn_ballot_w64(need);
                const uint32_t t_lo = (uint32_t)m & gm_lo, t_hi = (uint32_t)(m >> 32) & gm_hi;
                const uint32_t at = __builtin_amdgcn_mbcnt_hi(t_hi, __builtin_amdgcn_mbcnt_lo(t_lo, curw));
                curw += __builtin_popcount(t_lo) + __builtin_popcount(t_hi);
                if (need)
                    x = (x << 16) | lds_u16(((at << 1) & k255) | ring);
                if (active)
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(raw >> 24), orsrc, out_off16 - 16u * i + r0 + i, 0, 0);
            }
            // integrity: every state back at L, the cursor exactly at the end of the chunk's stream
            const bool bad = valid && (x != Tr::kL || 2u * curw - bias - (start - 8u * 4u) != len);
            const uint64_t bm = __builtin_amdgcn_ballot_w64(bad);
            if (i == 0 && ((bm >> (8u * g)) & 0xffu) != 0)
                nbad++;
            // the octet after this one
            uint64_t n_vk = vk + 1u, n_end = o_end;
            bool n_have = true;
            if (n_vk >= o_end) {
                const uint64_t c = claim_take();
                n_have = c < claims;
                n_vk = c * per_claim;
                n_end = (c + 1u) * per_claim < octets ? (c + 1u) * per_claim : octets;
            }
            uint64_t n_off = 0;
            uint32_t n_len = 0;
            if (n_have && octet_of(n_vk) * 8u + g < p.nchunks) {
                n_off = p.offsets[octet_of(n_vk) * 8u + g];
                n_len = p.lengths[octet_of(n_vk) * 8u + g];
            }
            vk = n_vk;
            o_end = n_end;
            have = n_have;
            off = n_off;
            len = n_len;
        }
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}

// ---------------------------------------------------------------------------
// Byte format, 2-way (main.cpp:226-280: two states, one pointer): THIRTY-TWO chunks per wave, lane 2 g + i holds state i of
// chunk g of the wave's batch.  The same plan as above with the group cut down to a pair:
//   * a state takes 0, 1 or 2 bytes (rans_byte.h:307-318, scale_bits <= 16); state 1's bytes follow state 0's, so the odd
//     lane's position is the cursor + what the even lane takes (its count through a DPP swap of the pair, masked to the even
//     lanes' counts beforehand), and the next cursor is the odd lane's position + count, broadcast to the pair (DPP);
//   * a pair's stream goes through a 64-byte ring in LDS (rows 68 bytes apart: byte 64 mirrors byte 0, so that the second
//     byte of a lane can be read at offset 1 without wrapping), fetched in aligned 32-byte blocks, 16 bytes per lane, the
//     next block parked in registers.  Eight rounds consume at most 8 x 2 x 2 = 32 bytes per pair: the refill check again is
//     one compare per lane every eight rounds;
//   * tables as in the lane kernels: cum2sym (LDS address 0) and {freq, start} records -- two dependent gathers per symbol;
//   * four rounds of a state's symbols are one dword; the pair interleaves them (DPP swap + v_perm), sixteen rounds are 32
//     bytes of the chunk, 16 per lane, and after thirty-two rounds two 16-byte stores per lane leave back to back: 64
//     contiguous bytes per chunk.
// Full chunks of a multiple of 4 symbols (what a size off 64 leaves goes one round at a time).
// ---------------------------------------------------------------------------
constexpr uint32_t kPairBlock = 32;              // bytes a pair fetches at a time: 2 lanes x 16 B
constexpr uint32_t kPairRing = 2 * kPairBlock;   // per pair
constexpr uint32_t kPairRow = kPairRing + 4;     // ring + the mirror byte; 17 dwords: equal positions of the 32 pairs on 32 banks
constexpr uint32_t kPairWaveLds = 32 * kPairRow; // per wave
constexpr int kPairAux = kAuxStore; // nt sc1, as the other decoders' symbol stores (plain stores: 0.94 ms against 0.73)

// Eight rounds (one symbol per lane and round) as one hand-scheduled sequence; rans_byte.h:125-128 (get), :291-298 (advance),
// :307-318 (renormalise).  17 VALU + 4 LDS per round:
//   v_and                       cf = x & mask
//   ds_read_u8                  s = cum2sym[cf]
//   v_lshrrev                   q = x >> scale_bits
//   v_lshl_or                   the symbol joins its accumulator (rounds 1..3 of four; round 0's lands there directly)
//   v_lshl_add, ds_read_b64     {freq, start} of s
//   v_mad_u32_u24, v_sub        x = freq * q + cf - start
//   v_cmp x2                    n1 = x < 2^23 (at least one byte), n2 = x < 2^15 (two)
//   v_addc x2                   t = cursor + n1 + n2: where this lane's bytes END if it is the pair's first
//   v_cndmask_dpp               my position = even lane: the cursor; odd lane: the even lane's t (vcc = the even lanes)
//   v_add_dpp, v_sub            the pair's next cursor = t + the other lane's t - cursor
//   v_and, v_add                position -> LDS address in the pair's ring
//   ds_read_u8 x2, v_lshl_or x2 under exec = n1 / n2: x = (x << 8) | byte
#define RANS_P_RECORD(SREG)                                                                        \
    "v_lshl_add_u32 %[t1], " SREG ", 3, %[rec]\n\t"                                                \
    "ds_read_b64 v[60:61], %[t1]\n\t"                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                     \
    "v_mad_u32_u24 %[x], v60, %[t2], %[t0]\n\t"                                                    \
    "v_sub_u32_e32 %[x], %[x], v61\n\t"
#define RANS_P_ROUND(SREG, FILL)                                                                   \
    "v_and_b32_e32 %[t0], %[maskv], %[x]\n\t"                                                      \
    "ds_read_u8 " SREG ", %[t0]\n\t"                                                               \
    "v_lshrrev_b32_e32 %[t2], %[sbv], %[x]\n\t"                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                     \
    RANS_P_RECORD(SREG)                                                                            \
    "v_cmp_gt_u32_e64 %[n1], %[l23], %[x]\n\t"                                                     \
    "v_cmp_gt_u32_e64 %[n2], %[l15], %[x]\n\t"                                                     \
    FILL                                                                                           \
    "v_addc_co_u32_e64 %[t1], %[dm], %[cur], 0, %[n1]\n\t"                                         \
    "v_addc_co_u32_e64 %[t1], %[dm], %[t1], 0, %[n2]\n\t"                                          \
    "s_nop 1\n\t"                                                                                  \
    "v_cndmask_b32_dpp %[t2], %[t1], %[cur], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp %[t0], %[t1], %[t1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         \
    "v_sub_u32_e32 %[cur], %[t0], %[cur]\n\t"                                                      \
    "v_and_b32_e32 %[t2], 63, %[t2]\n\t"                                                           \
    "v_add_u32_e32 %[t2], %[ring], %[t2]\n\t"                                                      \
    "s_mov_b64 exec, %[n1]\n\t"                                                                    \
    "ds_read_u8 %[t1], %[t2]\n\t"                                                                  \
    "s_mov_b64 exec, %[n2]\n\t"                                                                    \
    "ds_read_u8 %[t0], %[t2] offset:1\n\t"                                                         \
    "s_mov_b64 exec, %[n1]\n\t"                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                     \
    "v_lshl_or_b32 %[x], %[x], 8, %[t1]\n\t"                                                       \
    "s_mov_b64 exec, %[n2]\n\t"                                                                    \
    "v_lshl_or_b32 %[x], %[x], 8, %[t0]\n\t"                                                       \
    "s_mov_b64 exec, -1\n\t"
#define RANS_P_FOUR(ACC)                                                                           \
    RANS_P_ROUND(ACC, "s_nop 0\n\t")                                                               \
    RANS_P_ROUND("%[t3]", "v_lshl_or_b32 " ACC ", %[t3], 8, " ACC "\n\t")                          \
    RANS_P_ROUND("%[t3]", "v_lshl_or_b32 " ACC ", %[t3], 16, " ACC "\n\t")                         \
    RANS_P_ROUND("%[t3]", "v_lshl_or_b32 " ACC ", %[t3], 24, " ACC "\n\t")
__device__ __forceinline__ void decode_pairs_8rounds(uint32_t &x, uint32_t &cur, uint32_t &acc_a, uint32_t &acc_b, uint32_t maskv,
                                                     uint32_t sbv, uint32_t rec, uint32_t l23, uint32_t l15, uint32_t ring)
{
    uint32_t t0, t1, t2, t3;
    uint64_t n1, n2, dm;
    asm volatile("s_mov_b64 vcc, %[evens]\n\t" // the even lanes (the pair's state 0), for the whole sequence
                 RANS_P_FOUR("%[pa]") RANS_P_FOUR("%[pb]")
                 : [x] "+v"(x), [cur] "+v"(cur), [pa] "=&v"(acc_a), [pb] "=&v"(acc_b), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                   [t3] "=&v"(t3), [n1] "=&s"(n1), [n2] "=&s"(n2), [dm] "=&s"(dm)
                 : [maskv] "v"(maskv), [sbv] "v"(sbv), [rec] "v"(rec), [l23] "v"(l23), [l15] "v"(l15), [ring] "v"(ring),
                   [evens] "s"(0x5555555555555555ull)
                 : "vcc", "memory", "v60", "v61");
}
#undef RANS_P_FOUR
#undef RANS_P_RECORD
#undef RANS_P_ROUND

// sixteen rounds of a pair -> 32 bytes of its chunk, 16 per lane: d0..d3 are the four 4-round accumulators of this lane's
// state.  Pair-interleave (the even lane keeps the first dword of every eight bytes, the odd lane the second), then the even
// lane collects bytes [0, 16) and the odd lane [16, 32).
__device__ __forceinline__ u32x4 pair_lines(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t sel_p)
{
    u32x4 v;
    uint32_t e0, e1, e2, e3;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %[e0], %[d0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %[e1], %[d1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %[e2], %[d2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %[e3], %[d3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_perm_b32 %[e0], %[e0], %[d0], %[sel]\n\t" // even: [mine0, theirs0, mine1, theirs1]; odd: [theirs2, mine2, theirs3, mine3]
                 "v_perm_b32 %[e1], %[e1], %[d1], %[sel]\n\t"
                 "v_perm_b32 %[e2], %[e2], %[d2], %[sel]\n\t"
                 "v_perm_b32 %[e3], %[e3], %[d3], %[sel]\n\t"
                 "s_mov_b64 vcc, %[evens]\n\t"
                 "v_cndmask_b32_dpp %[vx], %[e2], %[e0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" // even ? mine : the even lane's
                 "v_cndmask_b32_dpp %[vz], %[e3], %[e1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 vcc, %[odds]\n\t"
                 "v_cndmask_b32_dpp %[vy], %[e0], %[e2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" // odd ? mine : the odd lane's
                 "v_cndmask_b32_dpp %[vw], %[e1], %[e3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : [vx] "=&v"(v.x), [vy] "=&v"(v.y), [vz] "=&v"(v.z), [vw] "=&v"(v.w), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2),
                   [e3] "=&v"(e3)
                 : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [sel] "v"(sel_p), [evens] "s"(0x5555555555555555ull),
                   [odds] "s"(0xaaaaaaaaaaaaaaaaull)
                 : "vcc");
    return v;
}

// Four lanes (two pairs, chunks A and B) hold, per chunk, the eight 16-byte pieces of a 128-byte line: piece 2 r + i in
// register set r of the pair's lane i.  quad_half<HI, R> gathers what lane t of the quad stores with ONE instruction so that the
// quad writes 64 contiguous bytes: piece t of half R (pieces 4 R .. 4 R + 3) of chunk HI -- lane (2 HI + (t & 1))'s register set
// 2 R + (t >> 1).  Two DPP operations per dword.
template <int HI>
__device__ __forceinline__ u32x4 quad_half(const u32x4 &lo, const u32x4 &hi)
{
    u32x4 o;
    uint32_t t0, t1, t2, t3;
    if constexpr (HI) {
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %[t0], %[h0] quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %[t1], %[h1] quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %[t2], %[h2] quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %[t3], %[h3] quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "s_mov_b64 vcc, %[upper]\n\t" // lanes 2, 3 of every quad: the register set 2 R + 1
                     "v_cndmask_b32_dpp %[o0], %[l0], %[t0], vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %[o1], %[l1], %[t1], vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %[o2], %[l2], %[t2], vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %[o3], %[l3], %[t3], vcc quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf"
                     : [o0] "=&v"(o.x), [o1] "=&v"(o.y), [o2] "=&v"(o.z), [o3] "=&v"(o.w), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [t3] "=&v"(t3)
                     : [l0] "v"(lo.x), [l1] "v"(lo.y), [l2] "v"(lo.z), [l3] "v"(lo.w), [h0] "v"(hi.x), [h1] "v"(hi.y), [h2] "v"(hi.z),
                       [h3] "v"(hi.w), [upper] "s"(0xccccccccccccccccull)
                     : "vcc");
    } else {
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %[t0], %[h0] quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %[t1], %[h1] quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %[t2], %[h2] quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %[t3], %[h3] quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_mov_b64 vcc, %[upper]\n\t"
                     "v_cndmask_b32_dpp %[o0], %[l0], %[t0], vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %[o1], %[l1], %[t1], vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %[o2], %[l2], %[t2], vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %[o3], %[l3], %[t3], vcc quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf"
                     : [o0] "=&v"(o.x), [o1] "=&v"(o.y), [o2] "=&v"(o.z), [o3] "=&v"(o.w), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [t3] "=&v"(t3)
                     : [l0] "v"(lo.x), [l1] "v"(lo.y), [l2] "v"(lo.z), [l3] "v"(lo.w), [h0] "v"(hi.x), [h1] "v"(hi.y), [h2] "v"(hi.z),
                       [h3] "v"(hi.w), [upper] "s"(0xccccccccccccccccull)
                     : "vcc");
    }
    return o;
}

__global__ void __launch_bounds__(kGrpThreads, 8) k_decode_byte_pairs(const DecParams p)
{
    using Tr = FmtTraits<FMT_BYTE>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t0_bytes = (p.table0_bytes + 15u) & ~15u; // cum2sym u8[M]
    const uint32_t t1_bytes = (p.table1_bytes + 15u) & ~15u; // {freq, start}[nsyms]
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
    if (!lds_starts_at_zero(smem)) { // cannot happen without static LDS; never decode on a wrong assumption
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
    const uint32_t g = lane >> 1, i = lane & 1u;
    const uint32_t ring_c = t0_bytes + t1_bytes + wave * kPairWaveLds + g * kPairRow; // raw LDS byte address of the pair's ring
    // per-lane constants in VGPRs (a VALU operand from an SGPR or a literal issues slower: profiles/r01_ubench.log)
    uint32_t ring = ring_c, maskv = (1u << p.scale_bits) - 1u, sbv = p.scale_bits, rec = t0_bytes;
    uint32_t l23 = 1u << 23, l15 = 1u << 15;
    uint32_t sel_p = i ? 0x03070206u : 0x05010400u; // v_perm(theirs, mine): the pair's bytes in stream order
    asm volatile("v_mov_b32 %0, %0" : "+v"(ring));
    asm volatile("v_mov_b32 %0, %0" : "+v"(maskv));
    asm volatile("v_mov_b32 %0, %0" : "+v"(sbv));
    asm volatile("v_mov_b32 %0, %0" : "+v"(rec));
    asm volatile("v_mov_b32 %0, %0" : "+v"(l23));
    asm volatile("v_mov_b32 %0, %0" : "+v"(l15));

    const uint64_t cbase = reinterpret_cast<uint64_t>(p.container);
    const uint64_t glo = cbase & ~uint64_t(15), glimit = (cbase + p.container_bytes + 15u) & ~uint64_t(15);
    const uint32_t groups64 = uniform(p.chunk_syms >> 7);    // 64 rounds of 2 symbols
    const uint32_t rem32 = uniform((p.chunk_syms >> 6) & 1u); // + 32 rounds
    const uint32_t left_syms = uniform(p.chunk_syms & 63u);   // + up to 31 rounds, one at a time
    const uint64_t batches = (p.nchunks + 31u) >> 5;      // (the last one may hold fewer than 32 chunks)
    const uint32_t per_claim = uniform(p.chunk_syms >= kGrpClaimSyms / 32u ? 1u : (kGrpClaimSyms / 32u + p.chunk_syms - 1u) / p.chunk_syms);
    const uint64_t claims = (batches + per_claim - 1u) / per_claim;

    auto load16 = [&](uint64_t a) -> u32x4 { // 16 bytes of the container, zeros beyond its granules
        u32x4 v = {0u, 0u, 0u, 0u};
        if (a >= glo && a < glimit)
            v = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(a));
        return v;
    };

    uint32_t nbad = 0;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    uint64_t claim_v = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    auto claim_take = [&]() -> uint64_t { // (as in k_decode_word_groups)
        if (p.work_counter) {
            uint32_t got = 0;
            if (lane == 0)
                got = atomicAdd(p.work_counter + pool * kWorkPoolStride, 1u);
            return (uint64_t)uniform(got) * npools + pool;
        }
        const uint64_t c = claim_v;
        claim_v += total_waves;
        return uniform64(c);
    };
    const bool ragged = p.n % p.chunk_syms != 0; // (as in k_decode_word_groups)
    for (;;) {
        const uint64_t claim = claim_take();
        if (claim >= claims)
            break;
        const uint64_t b_end = (claim + 1u) * per_claim < batches ? (claim + 1u) * per_claim : batches;
        for (uint64_t vb = claim * per_claim; vb < b_end; ++vb) {
            const uint64_t batch = ragged ? batches - 1u - vb : vb; // (a ragged last chunk: its batch goes round by round -- first)
            const uint64_t chunk = batch * 32u + g;
            const bool exists = chunk < p.nchunks;
            const uint64_t off = exists ? p.offsets[chunk] : 0u;
            const uint32_t len = exists ? p.lengths[chunk] : 0u;
            const bool valid = exists && len >= 2u * 4u && off <= p.container_bytes && len <= p.container_bytes - off;
            if (exists && !valid && i == 0)
                nbad++;
            const uint64_t src = cbase + (valid ? off : 0u);
            uint32_t x = Tr::kL;
            if (valid) // RansDecInit order: state 0 first (main.cpp:261-262)
                x = reinterpret_cast<const uint32_t RANS_GLOBAL *>(src)[i];
            // positions count from the 32-byte block of the chunk's first byte
            const uint64_t abase = src & ~uint64_t(kPairBlock - 1u);
            const uint32_t start = (uint32_t)(src - abase) + 2u * 4u; // < 40
            uint32_t cur = start;
            uint64_t ld = abase + 16u * i;
            const u32x4 b0 = load16(ld), b1 = load16(ld + kPairBlock);
            u32x4 pend = load16(ld + 2u * kPairBlock); // block 2; the ring holds blocks nb - 2 and nb - 1, pend is block nb
            ld += 3u * kPairBlock;
            uint32_t left = 0; // pieces at ld, ld + 32, ... inside the container's granules (as in k_decode_word_groups)
            if (ld < glimit) {
                const uint64_t pieces = (glimit - ld + (kPairBlock - 1u)) / kPairBlock;
                left = pieces < 0x7fffffffu ? (uint32_t)pieces : 0x7fffffffu;
            }
            auto put = [&](uint32_t at, const u32x4 &v) { // (rows are 4-byte aligned: four dword writes)
                RANS_LDS uint32_t *d = reinterpret_cast<RANS_LDS uint32_t *>((uintptr_t)(ring_c + at + 16u * i));
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
                if (at == 0 && i == 0)
                    *reinterpret_cast<RANS_LDS uint8_t *>((uintptr_t)(ring_c + kPairRing)) = (uint8_t)v.x;
            };
            put(0, b0);
            put(kPairBlock, b1);
            uint32_t thr = kPairBlock; // cursor from which block nb has to be in the ring; block nb goes to ring offset ~thr & 32
            auto checkpoint = [&]() {
                if (cur >= thr) { // (the same for both lanes of a pair)
                    put(~thr & kPairBlock, pend);
                    thr += kPairBlock;
                    if (left) {
                        pend = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(ld));
                        left--;
                    }
                    ld += kPairBlock;
                }
            };
            checkpoint(); // the states may end in block 1

            const rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(reinterpret_cast<uint64_t>(p.out) + batch * 32u * p.chunk_syms), 0, 32u * p.chunk_syms, kRsrcFlags);
            const uint32_t out_off16 = valid ? g * p.chunk_syms + 16u * i : 0x80000000u; // (an invalid chunk's stores: dropped by the range check)
            // the quad's two chunks A = g & ~1 and B = A + 1, as lane t = lane & 3 of the quad addresses them: 16 t bytes in
            const bool valid_a = __shfl((int)valid, (int)(lane & ~3u)) != 0, valid_b = __shfl((int)valid, (int)(lane | 2u)) != 0;
            const uint32_t off_a = valid_a ? (g & ~1u) * p.chunk_syms + 16u * (lane & 3u) : 0x80000000u;
            const uint32_t off_b = valid_b ? (g | 1u) * p.chunk_syms + 16u * (lane & 3u) : 0x80000000u;
            uint32_t osoff = 0;
            auto sixteen = [&]() -> u32x4 { // 16 rounds -> this lane's 16 of the pair's 32 bytes
                uint32_t a0, a1, a2, a3;
                checkpoint(); // (in front of the rounds here: behind them this kernel is 1.5 % slower, k_decode_word_groups 3 % faster)
                decode_pairs_8rounds(x, cur, a0, a1, maskv, sbv, rec, l23, l15, ring);
                checkpoint();
                decode_pairs_8rounds(x, cur, a2, a3, maskv, sbv, rec, l23, l15, ring);
                return pair_lines(a0, a1, a2, a3, sel_p);
            };
            // 64 rounds = one 128-byte line of the chunk: four 16-byte stores per lane leave back to back (a pair writes 32
            // contiguous bytes per instruction; pieces of a line that leave 16 or 32 rounds apart reach memory as partial
            // lines -- measured 1.18 ms against 0.70 without stores)
            const bool by_rounds = ragged && batch + 1u == batches; // (wave-uniform)
            for (uint32_t q = 0; q < (by_rounds ? 0u : groups64); ++q) {
                const u32x4 v0 = sixteen();
                const u32x4 v1 = sixteen();
                const u32x4 v2 = sixteen();
                const u32x4 v3 = sixteen();
                // a quad writes 64 contiguous bytes per instruction, a chunk's line with two instructions back to back
                // (64-byte pieces off the 64-byte grid: plain stores, as in k_decode_word_groups -- 1000-symbol chunks 1.67 -> 1.42 ms;
                //  on the grid `nt sc1` wins: 960-symbol chunks 0.79 against 1.15)
                if (p.chunk_syms & 63u) {
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<0>(v0, v1), orsrc, off_a, osoff, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<0>(v2, v3), orsrc, off_a, osoff + 64u, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<1>(v0, v1), orsrc, off_b, osoff, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<1>(v2, v3), orsrc, off_b, osoff + 64u, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<0>(v0, v1), orsrc, off_a, osoff, kPairAux);
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<0>(v2, v3), orsrc, off_a, osoff + 64u, kPairAux);
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<1>(v0, v1), orsrc, off_b, osoff, kPairAux);
                    __builtin_amdgcn_raw_buffer_store_b128(quad_half<1>(v2, v3), orsrc, off_b, osoff + 64u, kPairAux);
                }
                osoff += 128u;
            }
            if (rem32 && !by_rounds) { // a chunk of an odd multiple of 64 symbols: its last half line
                const u32x4 v0 = sixteen();
                const u32x4 v1 = sixteen();
                __builtin_amdgcn_raw_buffer_store_b128(v0, orsrc, out_off16, osoff, 0);
                __builtin_amdgcn_raw_buffer_store_b128(v1, orsrc, out_off16, osoff + 32u, 0);
            }
            // what a chunk size off 64 leaves (at most 31 rounds) -- or, in the ragged chunk's batch, every round: compiler-scheduled,
            // one round at a time, lanes without a symbol sit the round out, a byte store per lane
            uint32_t nsym = valid ? p.chunk_syms : 0u, s_first = p.chunk_syms - left_syms;
            if (by_rounds) {
                s_first = 0u;
                if (valid && chunk + 1u == p.nchunks)
                    nsym = (uint32_t)(p.n - chunk * p.chunk_syms);
            }
            for (uint32_t s0 = s_first; s0 < p.chunk_syms; s0 += 2u) {
                if (((s0 - s_first) & 15u) == 0)
                    checkpoint();
                const bool active = s0 + i < nsym;
                uint32_t sy = 0, n = 0;
                if (active) {
                    const uint32_t cf = x & maskv; // rans_byte.h:125-128 (get), :291-298 (advance)
                    sy = *reinterpret_cast<RANS_LDS const uint8_t *>((uintptr_t)cf);
                    const u32x2 fr = *reinterpret_cast<RANS_LDS const u32x2 *>((uintptr_t)(rec + 8u * sy));
                    x = (fr.x & 0xffffffu) * ((x >> sbv) & 0xffffffu) + cf - fr.y;
                    n = (uint32_t)(x < l23) + (uint32_t)(x < l15); // rans_byte.h:307-318: state 0's bytes first
                }
                const uint32_t n_other = (uint32_t)__shfl_xor((int)n, 1, 64);
                const uint32_t pos = cur + (i ? n_other : 0u);
                cur += n + n_other;
                if (n >= 1u)
                    x = (x << 8) | *reinterpret_cast<RANS_LDS const uint8_t *>((uintptr_t)(ring_c + (pos & 63u)));
                if (n >= 2u)
                    x = (x << 8) | *reinterpret_cast<RANS_LDS const uint8_t *>((uintptr_t)(ring_c + ((pos + 1u) & 63u)));
                if (active)
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)sy, orsrc, out_off16 - 16u * i + s0 + i, 0, 0);
            }
            // integrity: both states back at L, the cursor exactly at the end of the chunk's stream
            const bool bad = valid && (x != Tr::kL || cur - (start - 2u * 4u) != len);
            const uint64_t bm = __builtin_amdgcn_ballot_w64(bad);
            if (i == 0 && ((bm >> (2u * g)) & 3u) != 0)
                nbad++;
        }
    }
    if (nbad)
        atomicAdd(p.err_count, (unsigned long long)nbad);
}

} // namespace

// (a ragged last chunk included: its octet goes one round at a time and is handed out first)
bool decode_word_groups_applicable(const DecParams &p)
{
    return p.n_ways == 8 && p.sym_bytes == 1 && p.scale_bits == 12 && (p.chunk_syms & 3u) == 0 && p.chunk_syms >= 32 && p.chunk_syms <= (1u << 20) &&
           (reinterpret_cast<uintptr_t>(p.out) & 3u) == 0 && p.n / p.chunk_syms >= 8 && !p.trace &&
           ((p.table0_bytes + 15u) & ~15u) % kGrpRing == 0;
}

hipError_t launch_decode_word_groups(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const size_t table_lds = (p.table0_bytes + 15u) & ~15u;
    const size_t lds = table_lds + (size_t)(kGrpThreads / 64) * kGrpWaveLds;
    auto kern = k_decode_word_groups;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 160 * 1024, lds_ok); e != hipSuccess)
        return e;
    const uint32_t per_cu = 2u * lds <= 160u * 1024u ? 2u : 1u;
    const uint64_t want_blocks = ((p.nchunks + 7u) / 8u + kGrpThreads / 64 - 1) / (kGrpThreads / 64);
    const uint64_t cap = (uint64_t)num_cus * per_cu;
    const uint32_t grid = (uint32_t)(want_blocks < cap ? want_blocks : cap);
    if (name)
        *name = "k_decode_word_groups";
    RANS_LAUNCH(kern, dim3(grid), dim3(kGrpThreads), lds, stream, p);
    return hipGetLastError();
}

// The same for the byte format's 2-way layout (cum2sym tables, scale_bits 8..16, u8 symbols).
bool decode_byte_pairs_applicable(const DecParams &p)
{
    const size_t tables = (size_t)((p.table0_bytes + 15u) & ~15u) + ((p.table1_bytes + 15u) & ~15u);
    return p.n_ways == 2 && p.sym_bytes == 1 && p.scale_bits >= 8 && p.scale_bits <= 16 && (p.chunk_syms & 3u) == 0 &&
           p.chunk_syms >= 64 && p.chunk_syms <= (1u << 20) && (reinterpret_cast<uintptr_t>(p.out) & 3u) == 0 && p.n / p.chunk_syms >= 32 && !p.trace &&
           tables + (size_t)(kGrpThreads / 64) * kPairWaveLds <= 160u * 1024u;
}

hipError_t launch_decode_byte_pairs(const DecParams &p, int num_cus, hipStream_t stream, const char **name)
{
    const size_t tables = (size_t)((p.table0_bytes + 15u) & ~15u) + ((p.table1_bytes + 15u) & ~15u);
    const size_t lds = tables + (size_t)(kGrpThreads / 64) * kPairWaveLds;
    auto kern = k_decode_byte_pairs;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), 160 * 1024, lds_ok); e != hipSuccess)
        return e;
    const uint32_t per_cu = 2u * lds <= 160u * 1024u ? 2u : 1u;
    const uint64_t want_blocks = ((p.nchunks + 31u) / 32u + kGrpThreads / 64 - 1) / (kGrpThreads / 64);
    const uint64_t cap = (uint64_t)num_cus * per_cu;
    const uint32_t grid = (uint32_t)(want_blocks < cap ? want_blocks : cap);
    if (name)
        *name = "k_decode_byte_pairs";
    RANS_LAUNCH(kern, dim3(grid), dim3(kGrpThreads), lds, stream, p);
    return hipGetLastError();
}

} // namespace rans_amd
