// model.h -- host-side order-0 model and the table images derived from it.
//
// Replaces the table-building half of the reference mains (SymbolStats,
// cum2sym, RansEncSymbolInit/RansDecSymbolInit loops, RansWordTablesInitSymbol,
// make_alias_table); see include/ryg_rans_amd.h for the file:line map.
// Pure host C++ (no HIP), so it is unit-testable without a GPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace rans_amd {

// Reference struct layouts, reproduced for table export (rans_byte.h:159-171,
// rans64.h:152-164, rans_word_sse41.h:50-61).  Layout only; code is ours.
struct EncSymbol32 {
    uint32_t x_max;
    uint32_t rcp_freq;
    uint32_t bias;
    uint16_t cmpl_freq;
    uint16_t rcp_shift;
};
struct DecSymbol32 {
    uint16_t start;
    uint16_t freq;
};
struct EncSymbol64 {
    uint64_t rcp_freq;
    uint32_t freq;
    uint32_t bias;
    uint32_t cmpl_freq;
    uint32_t rcp_shift;
};
struct DecSymbol64 {
    uint32_t start;
    uint32_t freq;
};
static_assert(sizeof(EncSymbol32) == 16 && sizeof(DecSymbol32) == 4, "layout");
static_assert(sizeof(EncSymbol64) == 24 && sizeof(DecSymbol64) == 8, "layout");

// Device-side packed records (what the kernels stage into LDS).
//
// WordSlot: one record per cumulative slot of the 12-bit word model.
//   lo = freq (bits 0..11, bits 12..23 zero) | symbol << 24 ; hi = slot - start
// so the decoder's D step is a single v_mad_u32_u24(lo, x >> 12, hi) and the
// symbol is the top byte of lo.
struct WordSlot {
    uint32_t lo;
    uint32_t hi;
};
// SymRec: per-symbol record for the cum2sym-based decoders (byte / r64).
struct SymRec {
    uint32_t freq;
    uint32_t start;
};
// AliasHalf: one half-bucket of the alias table.  lo = freq | symbol << 16.
struct AliasHalf {
    uint32_t lo;
    uint32_t adjust;
};
// EncRec: per-symbol encoder record of the general (any N, lane-per-stream) paths; the four slots
// depend on the format (model.cpp):
//   alias  {freq, start, floor(2^32 / freq), remap base}   q from mulhi + one correction, remainder kept
//   byte   {freq | rshift << 24, bias, rcp, -}             Alverson, x + bias + q * (M - freq)
//   word   {freq, bias, m', cmpl | sh << 24}               round-up method (see WordEncRec)
//   rans64 {freq | rshift << 24, bias, rcp lo, rcp hi}     64-bit Alverson
struct EncRec {
    uint32_t freq;  // (byte, rans64: | rshift << 24)
    uint32_t start; // alias: start; the other formats: bias
    uint32_t rcp;   // reciprocal (rans64: low half)
    uint32_t remap; // alias: offset of this symbol's run in alias_remap; word: cmpl | sh << 24; rans64: rcp high half
};

// WordEncRec: encoder record of the word format for the full-wave kernel path (always 256 of them;
// symbols outside the model carry cmpl_sh = 0xffffffff, which the kernel tracks with one v_max).
// 12 bytes of payload on a 16-byte stride: the kernel reads them with one ds_read_b96 -- the LDS
// pipe (bank conflicts between different symbols' records) is the encoder's bottleneck.
// x / freq comes from the round-up method of Granlund & Montgomery for 32-bit dividends:
//   t = mulhi(x, mprime); q = (t + ((x - t) >> 1)) >> sh      (exact for every 32-bit x, freq >= 2)
// and the state update is x + bias + q * cmpl with cmpl = 4096 - freq, which equals
// (q << 12) + x % freq + start (rans_word_sse41.h:92).  freq == 1 uses mprime = 2^32 - 1, sh = 0
// (q = x - 1) with bias = start + 4095, cmpl = 4095 -- the same identity rans_byte.h:186-199 uses.
// The renormalisation test x >= freq << 20 (rans_word_sse41.h:85) is the carry of
// x + (cmpl << 20): (4096 - freq) << 20 = 2^32 - (freq << 20).
// Round 3: 8 bytes, read with one ds_read_b64.  The gather of 64 random records is what the LDS charges the encoder for:
// 16-byte records cost 9.7 LDS-array cycles per wave instruction on Zipf(256) symbols (ds_read_b128: four groups of 16
// lanes, a record covers four banks), 8-byte records 4.9 (two groups of 32 lanes, two banks each) -- bank model of
// MI355X_MICROARCH "LDS", and measured: every lane reading record 0 took the 1 GiB encode from 0.80 to 0.51 ms
// (profiles/r03_encoder_bound.md).  The fields are taken apart with VALU instructions, which this kernel has to spare.
//   mprime   reciprocal: word_small (no frequency above 2048: the renormalised state is below 2^31) -> Alverson,
//            ceil(2^(31 + ceil(log2 freq)) / freq), q = mulhi(x, mprime) >> sh exact (rans_byte.h:201-243); otherwise the
//            round-up method above
//   packed   cmpl (bits 0..11) | bias (bits 12..24) | sh (bits 27..31); packed << 20 is cmpl << 20 = 2^32 - (freq << 20),
//            the addend whose carry out of x is the renormalisation test (rans_word_sse41.h:85).  No record: 0xffffffff.
// Round 4: sixteen bytes again, one ds_read_b128, every field where the instruction that uses it wants it.  With the
// slot layout (rans_amd_encode_slots) nothing but the coding loop is left in the kernel and its VALU issue is the bound
// (LDS pipe 30 % busy): the four instructions that took the 8-byte record apart -- a slow-class v_lshlrev and v_bfe among
// them -- are worth more than the LDS cycles of the wider gather (9.7 against 4.9 per wave instruction).
//   mprime   as above
//   thresh   (freq << 20) - 1: x > thresh is the renormalisation test (rans_word_sse41.h:85) as ONE v_cmpx, which leaves
//            the emitting lanes in exec and vcc at once (the carry of x + (cmpl << 20), until late in round 4, needed an
//            s_mov to exec behind it -- a scalar instruction costs the SIMD an issue slot like a vector one); "- 1" because
//            freq == 4096 (a one-symbol model) makes freq << 20 = 2^32: 0xffffffff never emits, as it should
//   cmpl_sh  cmpl (bits 0..11; v_mad_u32_u24 reads the low 24 bits as they are) | sh << 24 (the shift takes its count from
//            byte 3: SDWA).  No record: 0x80000000 -- bit 31 is what the kernel OR-accumulates to find such a symbol --
//            with thresh 0xffffffff (never emits), mprime 0 and bias 0 (the state stays put)
//   bias     start (freq >= 2) or start + 4095 (freq == 1)
struct WordEncRec {
    uint32_t mprime;
    uint32_t thresh;
    uint32_t cmpl_sh;
    uint32_t bias;
};
static_assert(sizeof(WordEncRec) == 16, "one ds_read_b128");

int count_freqs_host(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *freqs);
int normalize_freqs(uint32_t *freqs, uint32_t *cum, uint32_t nsyms, uint32_t target_total);

struct HostModel {
    int format = 0;
    uint32_t nsyms = 0;
    uint32_t log2nsyms = 0;
    uint32_t scale_bits = 0;
    int sym_bytes = 1;

    std::vector<uint32_t> freqs;    // [nsyms]
    std::vector<uint32_t> cum;      // [nsyms+1]
    std::vector<uint16_t> cum2sym;  // [M] (exported as u8 when nsyms <= 256); empty for scale_bits > 16
    bool r64_search = false;        // rans64 with scale_bits 1..6 or 17..31: symbol by search, see build()
    std::vector<uint32_t> cum_padded; // r64_search: cum[] padded with ~0 to a power of two
    // rans64 with a cum2sym table, byte symbols, scale_bits <= 14 and no frequency above 4095: one 4-byte record per slot,
    // freq | (slot - start) << 12 | sym << 24 (k_decode_lanes_r64x2<packed>: one gather per symbol instead of two)
    std::vector<uint32_t> r64_packed;
    // byte format, byte symbols, scale_bits <= 13: one 8-byte record per cumulative slot, {freq | sym << 24, slot - start} --
    // rans_word_sse41.h:64-72's slot table for a caller-chosen scale_bits: the decoder's D step is ONE gather and one
    // v_mad_u32_u24 where cum2sym + {freq, start} are two dependent gathers and a subtract (FMT_BYTEF; 8 << scale_bits bytes)
    std::vector<WordSlot> byte_slots;

    // alias (main_alias.cpp:56-63)
    std::vector<uint32_t> divider, slot_adjust, slot_freqs, sym_id, alias_remap;

    // device-format records
    std::vector<WordSlot> word_slots;   // [4096]           FMT_WORD
    std::vector<SymRec> sym_recs;       // [nsyms]          FMT_BYTE / FMT_R64
    std::vector<AliasHalf> alias_halves; // [2*nsyms]        FMT_ALIAS
    std::vector<EncRec> enc_recs;       // [nsyms]          all formats
    std::vector<WordEncRec> word_enc_recs; // [256]         FMT_WORD
    bool dense256 = false;                 // 256 byte symbols, every one with a frequency: an encoder need not look for symbols without a record
    bool word_small = false;               // FMT_WORD: no frequency above 2048 -> Alverson reciprocals in word_enc_recs
    // FMT_ALIAS, when 2 M + 8 max(nsyms, 256) bytes fit in LDS: the encoder's tables in their LDS form --
    // {freq | start << 16, floor(2^32 / freq)} per symbol (zero records up to 256) and alias_remap as u16
    std::vector<uint64_t> alias_recs8;
    std::vector<uint16_t> alias_remap16;
    // FMT_ALIAS, decoder tables of the two-chunks-per-wave kernel (device_common.hpp FMT_ALIAS2): per half bucket
    // {sym | (M - freq) << 16, adjust}; per bucket the number of slots its own symbol keeps -- divider[b] - b * tgt,
    // main_alias.cpp:209 -- as u8 (tgt <= 255) or, alias2_wide, as u16.  Empty when the model cannot take that
    // form (a single bucket, or a 65536-wide symbol).
    std::vector<AliasHalf> alias2_halves;
    std::vector<uint8_t> alias2_own;
    bool alias2_wide = false;

    // Returns a rans_amd_status.
    int build(int format, const uint32_t *norm_freqs, uint32_t nsyms, uint32_t scale_bits);
    // Table export (reference layouts).
    int export_table(int which, std::vector<uint8_t> &out) const;

  private:
    int build_alias();
};

// Reciprocals by frequency for the fused per-chunk-model encoder (encode_adaptive.hip), frequencies 0 .. 4096: first the
// Alverson reciprocals of RansEncSymbolInit (rans_byte.h:201-243: ceil(2^(31 + ceil(log2 f)) / f); the byte format, and the word
// format while no frequency exceeds 2048), then the round-up reciprocals for 32-bit dividends (the word format otherwise) --
// the numbers HostModel::build puts into its records, tabulated so that a wave needs no division for its chunk's model.
void adapt_rcp_tables(std::vector<uint32_t> &out);

} // namespace rans_amd
