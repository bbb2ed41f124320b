// encode_common.hpp -- the encoders' sub-steps: one symbol for each of a wave's 64 lanes, compiler-scheduled (any format, any
// lane mask) and hand-written (word and byte format, full waves, units staged in an LDS window or stored directly).  Shared by
// encode_wave.hip (one model per launch) and encode_adaptive.hip (one model per chunk).  Included inside an anonymous namespace
// of namespace rans_amd, after device_common.hpp.
#pragma once

template <int FMT> struct EncTables {
    const uint4 *recs; // LDS: EncRec {freq, start, rcp, remap}  (FMT_ALIAS_LDS: uint2 {freq | start << 16, rcp})
    const uint16_t *remap16;     // LDS (FMT_ALIAS_LDS)
    uint32_t scale_bits;
    uint32_t nsyms;
    uint32_t swap_sel; // v_perm selector of enc_renorm_byte_full (kept in a VGPR)
    uint32_t split_sel; // ... and of enc_renorm_byte_full_staged
};

// Renormalisation of the byte-stream formats for a FULL wave (rans_byte.h:62-74: zero, one or two bytes leave the
// state while x >= x_max), hand-written -- see enc_byte_full below, whose first half this is: two compares give the
// one-byte and the two-byte mask, four v_mbcnt the lane's place, the two-byte lanes store the swapped low half with one
// global_store_short, the one-byte lanes one global_store_byte, each under its own exec mask.  12 VALU where the
// compiler's version (three byte stores with 64-bit address arithmetic each, the byte count by sign tricks) has ~25.
// A lane that must not emit passes x_max = 0xffffffff.  s[34:35] holds the two-byte mask.
// (with both stores dropped the 4096-symbol alias coder still takes 0.477 of its 0.536 ms: the address unit's share is 11 %,
//  profiles/r05_c4_encoder_no_stores.log)
#define RANS_RENORM_STORE_SHORT "global_store_short %[r], %[t], %[base]\n\t"
#define RANS_RENORM_STORE_BYTE "global_store_byte %[r], %[x], %[base]\n\t"
__device__ __forceinline__ void enc_renorm_byte_full(uint32_t &x, uint32_t x_max, uint32_t &wp, const uint8_t RANS_GLOBAL *slot,
                                                     uint32_t swap_sel)
{
    uint32_t t, r, c1, c2;
    uint32_t wps = uniform(wp); // (an "s" operand fed from a loop-carried value wants the readfirstlane spelled out)
    const uint8_t RANS_GLOBAL *base = reinterpret_cast<const uint8_t RANS_GLOBAL *>(uniform64(reinterpret_cast<uint64_t>(slot)));
    asm volatile("v_cmp_ge_u32_e32 vcc, %[x], %[xm]\n\t"
                 "v_lshrrev_b32_e32 %[t], 8, %[x]\n\t"
                 "v_cmp_ge_u32_e64 s[34:35], %[t], %[xm]\n\t"
                 "s_bcnt1_i32_b64 %[c1], vcc\n\t"
                 "s_bcnt1_i32_b64 %[c2], s[34:35]\n\t"
                 "s_add_u32 %[c1], %[c1], %[c2]\n\t"
                 "s_sub_u32 %[wp], %[wp], %[c1]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], s34, %[r]\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], s35, %[r]\n\t"
                 "v_add_u32_e32 %[r], %[wp], %[r]\n\t"
                 "v_perm_b32 %[t], %[x], %[x], %[sel]\n\t"
                 "s_mov_b64 exec, s[34:35]\n\t"
                 RANS_RENORM_STORE_SHORT
                 "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                 "s_andn2_b64 exec, vcc, s[34:35]\n\t"
                 RANS_RENORM_STORE_BYTE
                 "v_lshrrev_b32_e32 %[x], 8, %[x]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [wp] "+s"(wps), [t] "=&v"(t), [r] "=&v"(r), [c1] "=&s"(c1), [c2] "=&s"(c2)
                 : [xm] "v"(x_max), [base] "s"(base), [sel] "v"(swap_sel)
                 : "vcc", "scc", "memory", "s34", "s35");
    wp = wps;
}

// The same with the bytes written into the wave's LDS staging window (`wp` is its LDS write pointer, see
// enc_word_full_staged / enc_byte_full_staged): split_sel puts byte 1 of x into byte 0 and byte 0 into byte 2.
__device__ __forceinline__ void enc_renorm_byte_full_staged(uint32_t &x, uint32_t x_max, uint32_t &wp, uint32_t split_sel)
{
    uint32_t t, r, c1, c2;
    uint32_t wps = uniform(wp); // (an "s" operand fed from a loop-carried value wants the readfirstlane spelled out)
    asm volatile("v_cmp_ge_u32_e32 vcc, %[x], %[xm]\n\t"
                 "v_lshrrev_b32_e32 %[t], 8, %[x]\n\t"
                 "v_cmp_ge_u32_e64 s[34:35], %[t], %[xm]\n\t"
                 "s_bcnt1_i32_b64 %[c1], vcc\n\t"
                 "s_bcnt1_i32_b64 %[c2], s[34:35]\n\t"
                 "s_add_u32 %[c1], %[c1], %[c2]\n\t"
                 "s_sub_u32 %[wp], %[wp], %[c1]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], s34, %[r]\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], s35, %[r]\n\t"
                 "v_add_u32_e32 %[r], %[wp], %[r]\n\t"
                 "v_perm_b32 %[t], %[x], %[x], %[sel]\n\t"
                 "s_mov_b64 exec, s[34:35]\n\t"
                 "ds_write_b8 %[r], %[t]\n\t"
                 "ds_write_b8_d16_hi %[r], %[t] offset:1\n\t"
                 "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                 "s_andn2_b64 exec, vcc, s[34:35]\n\t"
                 "ds_write_b8 %[r], %[x]\n\t"
                 "v_lshrrev_b32_e32 %[x], 8, %[x]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [wp] "+s"(wps), [t] "=&v"(t), [r] "=&v"(r), [c1] "=&s"(c1), [c2] "=&s"(c2)
                 : [xm] "v"(x_max), [sel] "v"(split_sel)
                 : "vcc", "scc", "memory", "s34", "s35");
    wp = wps;
}

// One encoder sub-step for 64 lanes.  `wp` = write cursor (byte offset inside the
// slot, moves down, wave-uniform).
// PADDED: the record table holds 256 entries (zero records behind nsyms) and `sym` is a byte, so it
// indexes the table as it is -- no range select (a v_cndmask costs ~22 issue cycles on gfx950).
// FULL: all 64 lanes hold a symbol (the byte-stream formats then renormalise with enc_renorm_byte_full).
// dead: 0, or ~0 (wave-uniform) when nothing may leave the states any more -- a coder of sized slots (k_encode MODE 3) whose
// chunk no longer fits keeps running through its loop, storing nothing, and abandons the chunk at the end.
template <int FMT, bool PADDED = false, bool FULL = false, bool STAGED = false> // (STAGED: FULL, and wp is an LDS pointer)
__device__ __forceinline__ void enc_substep(const EncTables<FMT> &T, typename FmtTraits<FMT>::state_t &x,
                                            uint32_t sym, bool active, uint8_t RANS_GLOBAL *slot, uint32_t &wp,
                                            bool &bad, uint32_t dead = 0u)
{
    if constexpr (FMT == FMT_ALIAS_LDS && FULL) {
        // Full waves of the alias coder with its tables in LDS (main_alias.cpp:241-250), without a single select (a
        // v_cndmask behind a VALU compare costs several ordinary instructions on this part, and the general form below
        // has four per symbol):
        //  * the alphabet is a power of two: a symbol beyond it wraps into the table and the chunk is flagged;
        //  * a symbol without slots (freq 0) gets x_max = 0xffffffff from a saturating add -- it never emits --
        //    and is flagged; its lane carries garbage from then on, the call fails with RANS_AMD_E_MODEL;
        //  * the quotient estimate mulhi(y, floor(2^32 / freq)) is exact or one too small: the correction is a sign mask.
        // Whatever a lane holds, it emits at most two bytes per round: the slot (2 bytes per symbol) cannot overflow, and
        // the staged form flushes its window every eight rounds for this format (1024 bytes at most).
        const uint32_t idx = PADDED ? sym : (sym & (T.nsyms - 1u));
        const uint2 r8 = reinterpret_cast<const uint2 *>(T.recs)[idx];
        const uint32_t freq = r8.x & 0xffffu, start = r8.x >> 16, rcp = r8.y;
        bad = bad || (!PADDED && sym >= T.nsyms) || freq == 0u;
        const uint32_t k = 31u - T.scale_bits;
        // (`dead`, sized slots: all ones once the slot has no room for a round -- OR-ed into the wave-uniform addend, the
        //  saturating add then yields 0xffffffff for every lane: nothing leaves, and no instruction per symbol is spent on it)
        uint32_t x_max;
        asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(x_max) : "v"((freq - 1u) << k), "v"((1u << k) | dead));
        uint32_t y = x;
        if constexpr (STAGED)
            enc_renorm_byte_full_staged(y, x_max, wp, T.split_sel);
        else
            enc_renorm_byte_full(y, x_max, wp, slot, T.swap_sel);
        const uint32_t q0 = __umulhi(y, rcp);
        const uint32_t d = y - __umul24(q0, freq) - freq;          // rem0 - freq: negative iff the estimate was exact
        const uint32_t m = (uint32_t)((int32_t)d >> 31);
        const uint32_t rem = d + (freq & m), q = q0 + 1u + m;
        x = (q << T.scale_bits) + T.remap16[(rem + start) & ((1u << T.scale_bits) - 1u)];
        return;
    }
    const bool in_alphabet = PADDED || sym < T.nsyms;
    uint4 rec;
    if constexpr (FMT == FMT_ALIAS_LDS) { // 8-byte records: {freq | start << 16, rcp}
        const uint2 r8 = reinterpret_cast<const uint2 *>(T.recs)[PADDED ? sym : (in_alphabet ? sym : 0u)];
        rec = uint4{r8.x & 0xffffu, r8.x >> 16, r8.y, 0u};
    } else {
        rec = T.recs[PADDED ? sym : (in_alphabet ? sym : 0u)];
    }
    const uint32_t freq = (FMT == FMT_R64 || FMT == FMT_BYTE) ? (rec.x & 0xffffffu) : rec.x, start = rec.y, rcp = rec.z;
    (void)start;
    (void)rcp;
    if (active && (!in_alphabet || freq == 0)) {
        bad = true;
        active = false;
    }
    if constexpr (!FULL || kIsWord<FMT> || kIsR64<FMT>) // (the byte-stream formats' FULL form takes `dead` in its threshold)
        active = active && dead == 0u;

    if constexpr (kIsWord<FMT>) {
        // rans_word_sse41.h:81-93
        const bool emit = active && x >= (freq << 20);
        const uint64_t m = __builtin_amdgcn_ballot_w64(emit);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        wp -= 2u * cnt;
        if (emit)
            *reinterpret_cast<uint16_t RANS_GLOBAL *>(slot + wp + 2u * rank_below(m)) = (uint16_t)(x & 0xffffu);
        uint32_t y = emit ? (x >> 16) : x;
        const uint32_t xn = enc_update_word(y, rec);
        x = active ? xn : x;
    } else if constexpr (kIsR64<FMT>) {
        // rans64.h:77-93
        const uint64_t x_max = ((uint64_t)freq) << (63u - T.scale_bits); // ((L >> sb) << 32) * freq
        const bool emit = active && x >= x_max;
        const uint64_t m = __builtin_amdgcn_ballot_w64(emit);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        wp -= 4u * cnt;
        if (emit)
            *reinterpret_cast<uint32_t RANS_GLOBAL *>(slot + wp + 4u * rank_below(m)) = (uint32_t)x;
        uint64_t y = emit ? (x >> 32) : x;
        const uint64_t xn = FMT == FMT_R64S ? enc_update_r64s(y, rec, T.scale_bits) : enc_update_r64(y, rec, T.scale_bits);
        x = active ? xn : x;
    } else {
        // rans_byte.h:62-74 (renorm: 0, 1 or 2 bytes for scale_bits <= 16), :83-90 (put),
        // main_alias.cpp:241-250 (alias put).  The low byte is emitted first, i.e.
        // ends up at the higher address.
        const uint32_t x_max = freq << (31u - T.scale_bits);
        uint32_t y;
        if constexpr (FULL) {
            y = x;
            if constexpr (STAGED)
                enc_renorm_byte_full_staged(y, active ? x_max : 0xffffffffu, wp, T.split_sel);
            else
                enc_renorm_byte_full(y, (active ? x_max : 0xffffffffu) | dead, wp, slot, T.swap_sel);
        } else {
        const bool e1 = active && x >= x_max;
        const bool e2 = e1 && (x >> 8) >= x_max;
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(e1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(e2);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m1) + (uint32_t)__builtin_popcountll(m2);
        wp -= cnt;
        const uint32_t at = wp + rank_below(m1) + rank_below(m2);
        if (e2) {
            slot[at] = (uint8_t)(x >> 8);
            slot[at + 1] = (uint8_t)x;
        } else if (e1) {
            slot[at] = (uint8_t)x;
        }
        // bytes emitted = [x >= x_max] + [x >> 8 >= x_max] as sign bits (x, x_max < 2^31), then one shift:
        // no selects.  Inactive or invalid lanes may shift by garbage; their result is discarded below.
        const uint32_t nb = ((x_max - 1u - x) >> 31) + ((x_max - 1u - (x >> 8)) >> 31);
        y = x >> (nb << 3);
        }
        uint32_t xn;
        if constexpr (FMT == FMT_ALIAS_LDS) {
            // main_alias.cpp:249 with alias_remap in LDS (u16: slots are < M <= 2^16); the index of an inactive
            // or invalid lane is garbage, hence the mask -- its result is discarded below
            uint32_t q, rem;
            divmod_rcp(y, freq, rcp, q, rem);
            xn = (q << T.scale_bits) + T.remap16[(rem + start) & ((1u << T.scale_bits) - 1u)];
        } else {
            xn = enc_update_byte(y, rec, T.scale_bits);
        }
        x = active ? xn : x;
    }
}

// Hand-written encoder sub-step of the word format for a FULL wave (64 active lanes, symbols
// already turned into LDS addresses of their WordEncRec): rans_word_sse41.h:81-93 for 64 lanes.
//   v_cmpx_gt   x > (freq << 20) - 1  <=>  x >= freq << 20: the lanes that emit a word, in vcc and exec
//   s_bcnt1 ..  words emitted -> the wave's write offset moves down (these SALU ops are also the
//               wait states a VALU write of vcc needs before v_mbcnt may read it)
//   v_mbcnt x2  rank among the emitting lanes = word index (ascending lane = ascending address)
//   global_store_short + v_lshrrev under the emit mask, then exec back to all ones
//   x / freq    reciprocal from the record (model.h, WordEncRec; Alverson or round-up, see below): exact, so no
//               compare/select; x' = x + bias + q * cmpl in one v_mad_u32_u24 + add
// 10 VALU (13 with the round-up reciprocal), no v_cndmask, no branch.  `wp` is the byte offset of the lowest word written.
// (One state per lane -- every 64-way launch -- runs enc_word_full_staged below instead: the words go to LDS first.)
#define RANS_ENC_STORE "global_store_short %[t], %[x], %[base]\n\t"
// rec = WordEncRec {m', (freq << 20) - 1, cmpl | sh << 24, bias} (model.h; sixteen bytes: one ds_read_b128, nothing to take
// apart).  SMALL: no frequency of the model exceeds 2048, the renormalised state is below 2^31 and q = mulhi(x, m') >> sh
// is exact (Alverson, rans_byte.h:201-243): 10 VALU.  Otherwise the round-up method of Granlund & Montgomery for 32-bit
// dividends, t = mulhi(x, m'); q = (t + ((x - t) >> 1)) >> sh: 13 VALU.  (Round 3 read an 8-byte record and spent four
// instructions unpacking it -- 14 / 17 -- because the LDS pipe was the busiest unit then; with the stream staged in LDS
// windows and, in the slot layout, no copier waves beside the coders, instruction issue is what bounds the loop -- and a
// scalar instruction takes an issue slot of the SIMD like a vector one: 3 SALU per sub-step in the staged form, 5 before.)
//   v_or        bit 31 of cmpl_sh marks a symbol without a record; OR-accumulated, looked at once per chunk
//   v_cmpx_gt   x > (freq << 20) - 1 (rans_word_sse41.h:85): the lanes that emit, in vcc AND exec
//   s_bcnt1, s_sub   the wave's write pointer moves down (also the wait state between the VALU write of vcc and the
//               v_mbcnt that reads vcc_lo as a scalar operand)
//   v_mbcnt x2, one VOP3 add-shift   rank among the emitting lanes -> place of the lane's word
//   store + v_lshrrev under the emit mask
//   v_mul_hi, v_lshrrev (count = byte 3 of cmpl_sh: SDWA), v_mad_u32_u24 (q < 2^20, cmpl in the low 24 bits), v_add bias
// HEAD: `wp` counts bytes (the sub-step that stores to memory itself); HEAD_W: `wp` counts 16-bit WORDS (the staged
// sub-step: an LDS address / 2), which saves the s_lshl of the count.
#define RANS_ENC_WORD_TRACK "v_or_b32_e32 %[worst], %[worst], %[cs]\n\t"
#define RANS_ENC_WORD_HEAD                                       \
    "v_cmpx_gt_u32_e32 vcc, %[x], %[thr]\n\t"                     \
    "s_bcnt1_i32_b64 %[cnt], vcc\n\t"                             \
    "s_lshl_b32 %[cnt], %[cnt], 1\n\t"                            \
    "s_sub_u32 %[wp], %[wp], %[cnt]\n\t"                          \
    "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                      \
    "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"                   \
    "v_lshl_add_u32 %[t], %[t], 1, %[wp]\n\t"
#define RANS_ENC_WORD_HEAD_W                                     \
    "v_cmpx_gt_u32_e32 vcc, %[x], %[thr]\n\t"                     \
    "s_bcnt1_i32_b64 %[cnt], vcc\n\t"                             \
    "s_sub_u32 %[wp], %[wp], %[cnt]\n\t"                          \
    "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"                      \
    "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"                   \
    "v_add_lshl_u32 %[t], %[t], %[wp], 1\n\t"
#define RANS_ENC_WORD_TAIL_SMALL                                  \
    "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"                        \
    "s_mov_b64 exec, -1\n\t"                                      \
    "v_mul_hi_u32 %[q], %[x], %[m]\n\t"                           \
    "v_lshrrev_b32_sdwa %[q], %[cs], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mad_u32_u24 %[q], %[q], %[cs], %[x]\n\t"                   \
    "v_add_u32_e32 %[x], %[q], %[bias]"
#define RANS_ENC_WORD_TAIL_GM                                     \
    "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"                        \
    "s_mov_b64 exec, -1\n\t"                                      \
    "v_mul_hi_u32 %[q], %[x], %[m]\n\t"                           \
    "v_sub_u32_e32 %[t], %[x], %[q]\n\t"                          \
    "v_lshrrev_b32_e32 %[t], 1, %[t]\n\t"                         \
    "v_add_u32_e32 %[q], %[q], %[t]\n\t"                          \
    "v_lshrrev_b32_sdwa %[q], %[cs], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mad_u32_u24 %[q], %[q], %[cs], %[x]\n\t"                   \
    "v_add_u32_e32 %[x], %[q], %[bias]"
// TRACK: OR-accumulate the records' cmpl_sh words (models with symbols that have no record; EncParams::dense256 = none)
#define RANS_ENC_WORD_ASM(TRACKSTR, HEAD, STORE, TAIL, ...)                                                              \
    asm volatile(TRACKSTR HEAD STORE TAIL                                                                                  \
                 : [x] "+v"(x), [wp] "+s"(wp), [worst] "+v"(worst), [t] "=&v"(t), [q] "=&v"(q), [cnt] "=&s"(cnt)           \
                 : [m] "v"(rec.x), [thr] "v"(rec.y), [cs] "v"(rec.z), [bias] "v"(rec.w) __VA_ARGS__                        \
                 : "vcc", "scc", "memory")
template <bool SMALL, bool TRACK>
__device__ __forceinline__ void enc_word_full(uint32_t &x, const u32x4 &rec, uint32_t &wp,
                                              const uint8_t RANS_GLOBAL *slot, uint32_t &worst)
{
    uint32_t t, q, cnt;
#define RANS_COMMA_BASE , [base] "s"(slot)
    if constexpr (SMALL && TRACK)
        RANS_ENC_WORD_ASM(RANS_ENC_WORD_TRACK, RANS_ENC_WORD_HEAD, RANS_ENC_STORE, RANS_ENC_WORD_TAIL_SMALL, RANS_COMMA_BASE);
    else if constexpr (SMALL)
        RANS_ENC_WORD_ASM("", RANS_ENC_WORD_HEAD, RANS_ENC_STORE, RANS_ENC_WORD_TAIL_SMALL, RANS_COMMA_BASE);
    else if constexpr (TRACK)
        RANS_ENC_WORD_ASM(RANS_ENC_WORD_TRACK, RANS_ENC_WORD_HEAD, RANS_ENC_STORE, RANS_ENC_WORD_TAIL_GM, RANS_COMMA_BASE);
    else
        RANS_ENC_WORD_ASM("", RANS_ENC_WORD_HEAD, RANS_ENC_STORE, RANS_ENC_WORD_TAIL_GM, RANS_COMMA_BASE);
#undef RANS_COMMA_BASE
}

// The same with the emitted words staged in LDS: `wp` is an LDS address / 2 here, the write pointer into the wave's 2 KiB
// window (one ds_write_b16 per round instead of one global_store_short: the per-round stores were 4.3e7 write requests
// of 19 bytes on the 1 GiB encode and kept the address unit 87 % busy, profiles/r03_encoder_bound.md).  stage_flush() in
// the kernel moves what sixteen rounds have produced to memory in whole 16-byte pieces and sets the pointer back to the
// top of the window, so it never wraps.
template <bool SMALL, bool TRACK>
__device__ __forceinline__ void enc_word_full_staged(uint32_t &x, const u32x4 &rec, uint32_t &wp, uint32_t &worst)
{
    uint32_t t, q, cnt;
#define RANS_DS_STORE "ds_write_b16 %[t], %[x]\n\t"
    if constexpr (SMALL && TRACK)
        RANS_ENC_WORD_ASM(RANS_ENC_WORD_TRACK, RANS_ENC_WORD_HEAD_W, RANS_DS_STORE, RANS_ENC_WORD_TAIL_SMALL, );
    else if constexpr (SMALL)
        RANS_ENC_WORD_ASM("", RANS_ENC_WORD_HEAD_W, RANS_DS_STORE, RANS_ENC_WORD_TAIL_SMALL, );
    else if constexpr (TRACK)
        RANS_ENC_WORD_ASM(RANS_ENC_WORD_TRACK, RANS_ENC_WORD_HEAD_W, RANS_DS_STORE, RANS_ENC_WORD_TAIL_GM, );
    else
        RANS_ENC_WORD_ASM("", RANS_ENC_WORD_HEAD_W, RANS_DS_STORE, RANS_ENC_WORD_TAIL_GM, );
#undef RANS_DS_STORE
}

// The same for the byte format (rans_byte.h:62-74 renormalisation, :83-90 / :258-280 update) -- the compiler's version
// of this sub-step is 33.7 VALU instructions per round, three byte stores with 64-bit address arithmetic each among
// them.  Record {rcp, cmpl | rshift << 24, bias, x_max} (built in the kernel's prologue from the EncRec table):
//   v_cmp x2    x >= x_max (one byte leaves), (x >> 8) >= x_max (two bytes leave); x_max = freq << (31 - scale_bits)
//   s_bcnt1 x2  bytes emitted -> the wave's write offset moves down
//   v_mbcnt x4  rank among the emitting lanes, both masks: the lane's place (ascending lane = ascending address, the
//               low byte of a lane at the higher address)
//   two stores  the two-byte lanes one global_store_short of the swapped low half (v_perm), the one-byte lanes one
//               global_store_byte -- each under its own exec mask, against the chunk's SGPR base
//   x / freq    Alverson: mulhi(x, rcp) >> rshift, exact (the shift takes its count from the record's top byte: SDWA);
//               x' = x + bias + q * cmpl (q < 2^24, cmpl < 2^24: v_mad_u32_u24)
// 16 VALU + 7 SALU, no v_cndmask, no branch.  s[34:35] holds the two-byte mask.
__device__ __forceinline__ void enc_byte_full(uint32_t &x, const u32x4 &rec, uint32_t &wp, const uint8_t RANS_GLOBAL *slot,
                                              uint32_t &worst, uint32_t swap_sel)
{
    uint32_t t, r, q, c1, c2;
    asm volatile("v_cmp_ge_u32_e32 vcc, %[x], %[xm]\n\t"
                 "v_lshrrev_b32_e32 %[t], 8, %[x]\n\t"
                 "v_or_b32_e32 %[worst], %[worst], %[w]\n\t"
                 "v_cmp_ge_u32_e64 s[34:35], %[t], %[xm]\n\t"
                 "s_bcnt1_i32_b64 %[c1], vcc\n\t"
                 "s_bcnt1_i32_b64 %[c2], s[34:35]\n\t"
                 "s_add_u32 %[c1], %[c1], %[c2]\n\t"
                 "s_sub_u32 %[wp], %[wp], %[c1]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], s34, %[r]\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], s35, %[r]\n\t"
                 "v_add_u32_e32 %[r], %[wp], %[r]\n\t"
                 "v_perm_b32 %[t], %[x], %[x], %[sel]\n\t"
                 "s_mov_b64 exec, s[34:35]\n\t"
                 "global_store_short %[r], %[t], %[base]\n\t"
                 "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                 "s_andn2_b64 exec, vcc, s[34:35]\n\t"
                 "global_store_byte %[r], %[x], %[base]\n\t"
                 "v_lshrrev_b32_e32 %[x], 8, %[x]\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 "v_mul_hi_u32 %[q], %[x], %[rcp]\n\t"
                 "v_lshrrev_b32_sdwa %[q], %[w], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
                 "v_mad_u32_u24 %[q], %[q], %[w], %[x]\n\t"
                 "v_add_u32_e32 %[x], %[q], %[bias]"
                 : [x] "+v"(x), [wp] "+s"(wp), [worst] "+v"(worst), [t] "=&v"(t), [r] "=&v"(r), [q] "=&v"(q),
                   [c1] "=&s"(c1), [c2] "=&s"(c2)
                 : [rcp] "v"(rec.x), [w] "v"(rec.y), [bias] "v"(rec.z), [xm] "v"(rec.w), [base] "s"(slot), [sel] "v"(swap_sel)
                 : "vcc", "scc", "memory", "s34", "s35");
}

// The same with the emitted bytes staged in LDS (`wp` is the LDS write pointer of the wave's window, as in
// enc_word_full_staged), and with the lanes in REVERSE order: lane l codes stream 63 - l (the kernel mirrors the states
// before and after its fast loop and loads the symbols mirrored).  The stream grows downwards and a higher stream's bytes
// lie at the higher addresses (rans_byte.h:62-74 run for lane N-1 first): with the lanes mirrored, the bytes ABOVE a
// lane's are those of the lanes BELOW it -- the exclusive prefix v_mbcnt delivers -- so a lane's low byte goes to
// wp - 1 - Q and, if it emits two, the next one to wp - 2 - Q (Q = bytes of the lanes below), whatever the lane's own
// count: one address register, two ds_write_b8 (the second under the two-byte mask, from the x >> 8 the second compare
// needed anyway).  In ascending-lane order the low byte's place depends on the lane's own count: round 4's first version
// spent a v_perm, a third ds_write and an s_andn2 on that.
//   v_cmp (e64) s[34:35] = (x >> 8) >= x_max: two bytes leave;  v_cmpx vcc = exec = x >= x_max: at least one
//   v_mbcnt x4 from 2, v_sub from wp   r = wp - 2 - Q
//   15 VALU + 6 SALU + 2 LDS writes (16 + 7 + 3 before)
#define RANS_ENC_BYTE_STAGED_A "v_lshrrev_b32_e32 %[t], 8, %[x]\n\t"
#define RANS_ENC_BYTE_STAGED_B                                                                                            \
    "v_cmp_ge_u32_e64 s[34:35], %[t], %[xm]\n\t"                                                                          \
    "v_cmpx_ge_u32_e32 vcc, %[x], %[xm]\n\t"                                                                              \
    "s_bcnt1_i32_b64 %[c1], vcc\n\t"                                                                                      \
    "s_bcnt1_i32_b64 %[c2], s[34:35]\n\t"                                                                                 \
    "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 2\n\t"                                                                              \
    "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"                                                                           \
    "v_mbcnt_lo_u32_b32 %[r], s34, %[r]\n\t"                                                                              \
    "v_mbcnt_hi_u32_b32 %[r], s35, %[r]\n\t"                                                                              \
    "v_sub_u32_e32 %[r], %[wp], %[r]\n\t"                                                                                 \
    "s_sub_u32 %[wp], %[wp], %[c1]\n\t"                                                                                   \
    "s_sub_u32 %[wp], %[wp], %[c2]\n\t"                                                                                   \
    "ds_write_b8 %[r], %[x] offset:1\n\t"                                                                                 \
    "v_mov_b32_e32 %[x], %[t]\n\t"                                                                                        \
    "s_mov_b64 exec, s[34:35]\n\t"                                                                                        \
    "ds_write_b8 %[r], %[t]\n\t"                                                                                          \
    "v_lshrrev_b32_e32 %[x], 8, %[x]\n\t"                                                                                 \
    "s_mov_b64 exec, -1\n\t"                                                                                              \
    "v_mul_hi_u32 %[q], %[x], %[rcp]\n\t"                                                                                 \
    "v_lshrrev_b32_sdwa %[q], %[w], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"          \
    "v_mad_u32_u24 %[q], %[q], %[w], %[x]\n\t"                                                                            \
    "v_add_u32_e32 %[x], %[q], %[bias]"
#define RANS_ENC_BYTE_STAGED_ASM(TRACKSTR)                                                                                \
    asm volatile(RANS_ENC_BYTE_STAGED_A TRACKSTR RANS_ENC_BYTE_STAGED_B                                                   \
                 : [x] "+v"(x), [wp] "+s"(wp), [worst] "+v"(worst), [t] "=&v"(t), [r] "=&v"(r), [q] "=&v"(q),             \
                   [c1] "=&s"(c1), [c2] "=&s"(c2)                                                                         \
                 : [rcp] "v"(rec.x), [w] "v"(rec.y), [bias] "v"(rec.z), [xm] "v"(rec.w)                                   \
                 : "vcc", "scc", "memory", "s34", "s35")
template <bool TRACK> // (TRACK: the model has symbols without a record -- OR-accumulate the records' second words)
__device__ __forceinline__ void enc_byte_full_staged(uint32_t &x, const u32x4 &rec, uint32_t &wp, uint32_t &worst)
{
    uint32_t t, r, q, c1, c2;
    if constexpr (TRACK)
        RANS_ENC_BYTE_STAGED_ASM("v_or_b32_e32 %[worst], %[worst], %[w]\n\t");
    else
        RANS_ENC_BYTE_STAGED_ASM("");
}
