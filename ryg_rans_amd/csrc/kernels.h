// kernels.h -- launch interface between the C-ABI layer (api.cpp) and the HIP
// kernels (decode_wave.hip, encode_wave.hip, lanes.hip, container_kernels.hip).  Internal; not installed.
#pragma once

#include <cstdint>
#include <cstdlib>

#include <hip/hip_runtime.h>

namespace rans_amd {

// Measurement knobs.  The shipped library reads NO environment variable: every A/B switch and every knob that changes
// what a kernel writes (dropped stores, skipped copies -- the experiments DESIGN.md quotes) exists only in the
// -DRANS_AMD_MEASURE build (`make measure` -> libryg_rans_amd_measure.so, loaded through RANS_AMD_LIB by the tools).
// Alternatives that produce the same bytes and that the tests exercise are context options (rans_amd_ctx_set_option).
#ifdef RANS_AMD_MEASURE
constexpr bool kMeasureBuild = true;
inline const char *measure_knob(const char *name) { return getenv(name); }
#else
constexpr bool kMeasureBuild = false;
inline const char *measure_knob(const char *) { return nullptr; }
#endif

// DecParams::variant / EncParams::variant: kernel-family choices of the context (rans_amd_ctx_set_option)
constexpr uint32_t kVarLanesFused = 4u;    // lane-per-chunk encoders place their chunks themselves
constexpr uint32_t kVarNoDual = 8u;        // alias decoders: always one chunk per wave (k_decode)
constexpr uint32_t kVarDualAlways = 16u;   // ... two chunks per wave (k_decode_dual) whenever the tables fit, not only when
                                           //     they leave no room for a second block per CU

// Per-wave LDS stream window (see decode_wave.hip "stream window").
constexpr uint32_t kRingBytes = 2048;   // two 1 KiB blocks
constexpr uint32_t kRingBlock = 1024;   // 64 lanes x 16 B
constexpr uint32_t kRingMirror = 768;   // copy of ring[0..768) after the end: no wrap between checkpoints
constexpr uint32_t kRingStride = kRingBytes + kRingMirror;

constexpr int kDecBlockThreads = 1024; // 16 waves share one table image
constexpr uint32_t kWorkPools = 8;       // chunk hand-out counters per launch (one per XCD)
constexpr uint32_t kWorkPoolStride = 16; // in uint32: every counter on its own 64-byte line
constexpr uint32_t kWorkSlots = 64;      // launches that may reuse the counter ring before wrap
constexpr uint32_t kCaptureSlots = 8;    // counter slots behind the ring for launches captured into a hipGraph (api.cpp)
// one ring slot = kWorkPools counters (a 64-byte line each) + one line for the launch's wave span record
constexpr uint32_t kWorkSlotWords = (kWorkPools + 1) * kWorkPoolStride;
constexpr int kEncBlockThreads = 256;
// Kernel-side format number of rans64 with a binary search over the cumulative frequencies instead of the cum2sym
// table (scale_bits 1..6 and 17..31; device_common.hpp FMT_R64S).  launch_decode / launch_encode take it in place
// of RANS_AMD_FMT_R64; DecParams::table0 is then the cum table padded with ~0 to 2^log2nsyms words.
constexpr int kKernelFormatR64Search = 4;
// Kernel-side format number of the alias ENCODER with alias_remap in LDS (device_common.hpp FMT_ALIAS_LDS);
// EncParams::alias_recs8 / alias_remap16 are then set.
constexpr int kKernelFormatAliasLds = 5;
// Kernel-side format number of the word-format DECODER over more than 256 symbols (device_common.hpp FMT_WORD16):
// DecParams::table0 holds {freq, bias | sym << 16} per slot, symbols are u16.
constexpr int kKernelFormatWord16 = 6;
// Kernel-side format number of the byte-format DECODER with one model per chunk (device_common.hpp FMT_BYTEA):
// DecParams::chunk_freqs holds u16[256] per chunk; no table0/table1.
constexpr int kKernelFormatByteAdaptive = 7;
// Kernel-side format numbers of the two-chunks-per-wave alias DECODER (device_common.hpp FMT_ALIAS2 / FMT_ALIAS2W):
// DecParams::table0 = {sym | (M - freq) << 16, adjust} per half bucket, table1 = own-slot count per bucket (u8 / u16).
constexpr int kKernelFormatAlias2 = 8;
constexpr int kKernelFormatAlias2W = 9;
// Kernel-side format number of the byte-format DECODER with one fused 8-byte record per slot (device_common.hpp FMT_BYTEF):
// DecParams::table0 = {freq | sym << 24, slot - start}[1 << scale_bits], no table1.
constexpr int kKernelFormatByteFused = 11;
// Kernel-side format number of the WORD format with one model per chunk (device_common.hpp FMT_WORDA), decoder and encoder:
// DecParams / EncParams::chunk_freqs holds u16[256] per chunk, scale_bits is 12.
constexpr int kKernelFormatWordAdaptive = 12;
constexpr uint32_t kTraceWords = 5;      // per-wave record of DecParams::trace

struct DecParams {
    const uint8_t *container;
    uint64_t container_bytes;
    const uint64_t *offsets;
    const uint32_t *lengths;
    uint8_t *out;
    uint64_t n;
    uint64_t nchunks;
    uint32_t chunk_syms;
    uint32_t n_ways;
    const void *table0; // word: WordSlot[4096]; byte/r64: cum2sym u8[M]; alias: AliasHalf[2*nsyms]
    const void *table1; // byte/r64: SymRec[nsyms];  alias: divider u32[nsyms]
    uint32_t table0_bytes;
    uint32_t table1_bytes;
    const void *packed;    // rans64, cum2sym decoders: 4-byte slot records {freq:12 | slot - start:12 | sym:8} [M], or NULL
    uint32_t packed_bytes; //   (models whose largest frequency is below 4096; the 2-way lane decoder reads them)
    uint32_t scale_bits;
    uint32_t log2nsyms;
    uint32_t sym_bytes;
    unsigned long long *err_count; // failed chunks (device counter)
    unsigned int *work_counter;    // next chunk to hand out (zero at launch); NULL = static striding
    unsigned int *work_counter_reset; // a counter slot of a LATER launch that this launch zeroes
    const uint16_t *chunk_freqs;      // FMT_BYTEA: normalised frequencies, u16[256] per chunk (else NULL)
    uint8_t *wave_scratch;            // one 64-byte line per resident wave (marker stores of the window refills), or NULL
    uint32_t variant;                 // kVar* bits
    unsigned long long *span;         // this launch's {max of ~(wave start), max of wave end} in 100 MHz ticks, or NULL
    unsigned long long *span_reset;   // the span record of a LATER launch that this launch zeroes
    unsigned long long *trace;        // wave clocks: per wave kTraceWords words {start, end (100 MHz ticks), xcc,
                                      // shader cycles, 64-symbol rounds}; NULL = off (wave-per-chunk kernels only)
};

struct EncParams {
    const uint8_t *syms;
    uint64_t n;
    uint64_t nchunks;
    uint32_t chunk_syms;
    uint32_t n_ways;
    uint8_t *scratch;     // nchunks slots of slot_bytes each; chunk stream ends at the slot end
    uint64_t slot_bytes;  // multiple of 16
    uint32_t *lengths;    // out: stream bytes per chunk
    const void *enc_recs; // EncRec[nsyms]
    const void *word_enc_recs; // WordEncRec[256] (FMT_WORD only, else NULL)
    uint32_t dense256;         // every byte value is a symbol of the model: the full-wave sub-steps skip the search for symbols without a record
    uint32_t word_small;       // FMT_WORD: the records hold Alverson reciprocals (no frequency above 2048, model.h)
    const uint32_t *alias_remap;
    const uint16_t *chunk_freqs;    // byte format with one model per chunk: u16[256] per chunk (else NULL); the waves
                                    // build their chunk's records in LDS themselves
    const void *alias_recs8;        // FMT_ALIAS_LDS: {freq | start << 16, floor(2^32 / freq)} per symbol (>= 256 entries)
    const uint16_t *alias_remap16;  // FMT_ALIAS_LDS: alias_remap as u16[1 << scale_bits]
    uint32_t nsyms;
    uint32_t scale_bits;
    uint32_t sym_bytes;
    uint32_t *flags;      // bit0: symbol with freq 0 / outside alphabet met; fused: bit1 = container does not fit out_cap
    // Fused placement (wave-per-chunk encoders, k_encode<.., FUSED = true>): every block carries copier waves that
    // move the finished streams of ITS encoder waves from their scratch slots to their final place, found by a
    // decoupled look-back over status[] -- offsets[], lengths[] and the container come out of this one kernel, no
    // k_layout / k_compact.  status: one word per chunk (zero at launch), then kWorkPools claim counters on a 64-byte
    // line each.
    unsigned long long *status; // or NULL: the three-kernel path
    uint64_t *offsets;          // fused: out, [nchunks + 1]
    uint8_t *out;               // fused: the container
    uint64_t out_cap;
    uint32_t mailbox_off;       // fused: LDS byte offset of the block's mailbox (set by the launcher)
    uint32_t stage_off;         // alias encoder with its tables in LDS: byte offset of the coding waves' staging windows, 0 =
                                // the tables leave no room for them (set by the launcher; word / byte: fixed, behind their tables)
    uint8_t *mailbox_global;    // fused, tables that fill the CU's LDS to the last byte (the 4096-symbol, 16-bit alias model):
                                // one kEncMailboxStride-byte mailbox per block in global memory (zero at launch), else NULL
    uint32_t ring_slots;        // fused wave encoders: 0 = scratch holds one slot per CHUNK (slot of chunk c at c * slot_bytes);
                                // R > 0 = one ring of R slots per CODING WAVE (wave g's slots at (g * R + j) * slot_bytes)
    // Lane-per-chunk encoders (lanes.hip), fused: a launch codes the batches (64 chunks each) [batch_begin, batch_end)
    // of the nchunks chunks; status holds one word per UNIT -- the C batches the C coding waves of a block take in one
    // round -- numbered from unit_base, then (from word ceil(nchunks / 64) on) claim counters on a 64-byte line each,
    // of which this launch uses number claim_slot.  The r64 2-way kernel and the staged kernel that takes the tail
    // after it share one status array, in stream order.
    uint64_t batch_begin, batch_end, unit_base; // set by the launcher
    uint32_t claim_slot;
    uint32_t variant;           // kVar* bits
    // Slot layout (rans_amd_encode_slots): the chunks STAY where they are coded -- `scratch` is the caller's container, chunk
    // c's stream ends at the end of slot c exactly as the reference's encoder ends at the end of its buffer
    // (rans_byte.h:22-26, main.cpp:176-188), and the coding kernel writes offsets[c] = c * slot_bytes + (slot_bytes - len)
    // beside lengths[c]: every stream crosses HBM once, no look-back, no copier waves, no k_layout / k_compact.
    unsigned long long wait_ticks; // fused placement: how long (100 MHz ticks) a wait of the protocol may last before the launch is
                                   // declared failed (device_common.hpp SpinWatch); 0 = the default half minute
    uint32_t slot_layout;       // 1 = that; `offsets` is then set and `status` is NULL
    unsigned int *claims;       // wave encoders with dynamic chunk hand-out (fused placement, slot layout): kWorkPools claim
                                // counters on a 64-byte line each, zero at launch
    // Sized slots (rans_amd_encode_slots_sized): slot_bytes is whatever the caller chose -- normally a little above the
    // chunk's expected stream, the way the reference sizes its one buffer from the input (main_simd.cpp:145:
    // n + n/8 + 128), not from the worst case.  A chunk whose stream does not fit is ABANDONED by its coder: nothing of it
    // counts, its number is appended to ovf_list (ovf_ctl[0] = how many).  A second launch (redo = 1, wave kernel,
    // slot_bytes = the worst-case slot) codes exactly those chunks again, chunk ovf_list[i] into the i-th slot of the
    // overflow region that starts at byte ovf_base of `scratch`, and writes the final offsets[nchunks].
    unsigned int *ovf_ctl;      // NULL: slots cannot overflow.  [0] overflowed chunks, [kWorkPoolStride] the redo launch's claim counter
    uint32_t *ovf_list;         // nchunks entries
    uint32_t ovf_cap;           // redo: slots the overflow region holds (more overflowed chunks than that: flags bit 1, E_SPACE)
    uint32_t redo;              // 1 = the second launch
    uint64_t ovf_base;          // redo: byte offset of the overflow region (= nchunks * the first launch's slot_bytes)
    uint32_t no_lanes;          // the request goes to the wave encoders whatever its interleave (sized slots the lane encoders cannot take)
};
constexpr uint32_t kEncFusedThreads = 512; // 7 encoder waves + 1 copier wave; 4 blocks per CU
constexpr uint32_t kEncFusedCopiers16 = 2; // copier waves of a 16-wave block
constexpr uint32_t kEncMailboxBytes = 16 + 64 * 8;
// Wave-per-chunk encoders, fused placement: behind the mailbox, one "drained" counter per coding wave of the block (the
// scratch ring protocol, EncParams::ring_slots)
constexpr uint32_t kEncDrainBytes = 64;
constexpr uint32_t kEncFusedLdsBytes = kEncMailboxBytes + kEncDrainBytes;
// word encoder, one state per lane: the emitted words of sixteen rounds are staged in a window of LDS per wave
// (encode_wave.hip, enc_word_full_staged); the windows follow the 8 KiB of record tables
constexpr uint32_t kEncStageBytes = 2048;
constexpr uint32_t kEncMailboxStride = 640; // a block's mailbox in global memory (EncParams::mailbox_global): whole 128-byte lines
// Scratch ring of the fused wave encoders: every coding wave owns kEncRingSlots worst-case slots and codes its chunks
// into them in turn (a slot is reused once the block's copier has moved its previous occupant to the container), so the
// scratch a launch touches is (coding waves) x (slots) x (stream of a chunk) instead of the whole container once more --
// small enough to stay in the 256 MiB Infinity Cache, which is what takes the second trip through HBM out of the encoders.
constexpr uint32_t kEncRingSlots = 2;
constexpr uint32_t kEncRingMaxWavesPerCu = 32;          // resident waves of a CU: upper bound of the coding waves
constexpr uint64_t kEncRingMaxSlotBytes = (1ull << 26) - 64; // mailbox entries of the ring protocol keep 26 bits of length: a
                                                             // stream that fills its slot must still be below 2^26 bytes
static_assert(kEncRingSlots <= 4, "the ring protocol's mailbox entry keeps two bits of slot number");

struct LayoutParams {
    const uint32_t *lengths;
    uint64_t *offsets; // [nchunks+1]
    uint64_t nchunks;
    uint64_t out_cap;
    uint32_t *flags;   // bit1: container does not fit out_cap
    uint64_t *block_sums; // [layout_blocks(nchunks)] scratch, needed when that is > 1
};

struct CompactParams {
    const uint8_t *scratch;
    // where chunk c's stream starts in `scratch`: NULL = at the end of its slot ((c + 1) * slot_bytes - lengths[c], the
    // encoders' scratch); else src_offsets[c] (rans_amd_container_compact: any layout the decoders take), and nothing at
    // or beyond address src_limit (16-byte aligned end of the source buffer) is read
    const uint64_t *src_offsets;
    uint64_t src_limit; // ~0 = no limit (the encoders' scratch carries slack)
    // rans_amd_container_compact: the caller's index is DATA (it may have been parsed from a file) -- a chunk whose
    // (src_offsets[c], lengths[c]) does not lie inside [0, src_bytes) is skipped and bit 9 (512) of *flags is set
    // (RANS_AMD_E_CORRUPT), so nothing outside the source buffer is ever read.  Ignored when src_offsets is NULL.
    uint64_t src_bytes;
    uint64_t slot_bytes;
    const uint32_t *lengths;
    const uint64_t *offsets;
    uint8_t *out;
    uint64_t nchunks;
    uint32_t *flags; // skip everything when bit1 is set; bit 9: an index entry outside the source (see src_bytes)
};

// Workspace words that a launch expects to find zero (status flags, chunk claims, placement status words, mailboxes): up
// to three regions cleared by ONE kernel in front of the consumer.  Not hipMemsetAsync: as a node of a captured graph a
// small memset was seen replayed with another fill value from the second replay on (flags read back as 0x81818181,
// profiles/r04_graph_memset.md) -- a kernel's arguments are captured by value.
struct ZeroParams {
    void *ptr[3];      // 4-byte aligned
    uint64_t bytes[3]; // multiples of 4; 0 = unused
};
hipError_t launch_zero(const ZeroParams &p, hipStream_t stream);

// All launchers return hipSuccess or the launch error; they never synchronise.
hipError_t launch_decode(int format, const DecParams &p, int num_cus, hipStream_t stream, const char **kernel_name);
hipError_t launch_encode(int format, const EncParams &p, int num_cus, hipStream_t stream, const char **kernel_names);
bool decode_dual_fits(uint32_t table0_bytes, uint32_t table1_bytes); // LDS room for two stream windows per wave (decode_dual.hip)
bool encode_uses_lanes(int format, uint64_t nchunks, uint32_t n_ways);
// 0: the fused placement does not fit; 1: mailbox in the block's LDS; 2: the tables fill the LDS, mailbox in global memory
int encode_fused_fits(int format, uint32_t nsyms, uint32_t scale_bits);
bool encode_lanes_can_fuse(int format, const EncParams &p, int num_cus); // lane-per-chunk encoders: see launchers.hpp
hipError_t launch_layout(const LayoutParams &p, hipStream_t stream);
uint32_t layout_blocks(uint64_t nchunks); // blocks (and block_sums entries) launch_layout uses
hipError_t launch_compact(const CompactParams &p, int num_cus, hipStream_t stream);
hipError_t launch_histogram(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *d_hist,
                            uint32_t *d_flags, int num_cus, hipStream_t stream);

// per-chunk models of u8 symbols, built on the device: count + normalise to 1 << scale_bits (main.cpp:59-129 per chunk),
// d_chunk_freqs[nchunks][256] as u16; bit 0 of *d_flags when a chunk cannot be normalised
hipError_t launch_chunk_models(const void *syms, uint64_t n, uint32_t chunk_syms, uint64_t nchunks, uint32_t scale_bits,
                               uint16_t *d_chunk_freqs, uint32_t *d_flags, int num_cus, hipStream_t stream);
// The fused per-chunk-model encoder (encode_adaptive.hip; rans_amd_encode_adaptive_sized): ONE launch -- the wave that codes
// a chunk first counts it, normalises the counts, builds its records and codes it into a piece of the container whose size
// it derives from the chunk's own histogram (pieces in index order, placed by a look-back over their sizes).
struct AdaptEncParams {
    const uint8_t *syms;     // u8 symbols
    uint64_t n;
    uint64_t nchunks;
    uint32_t chunk_syms;
    uint32_t n_ways;
    uint32_t scale_bits;
    uint32_t worst_slot;     // worst-case stream of a full chunk, a whole number of 64-byte lines
    uint8_t *out;            // the container
    uint64_t out_cap;
    uint64_t *offsets;       // out, [nchunks + 1]
    uint32_t *lengths;       // out
    uint16_t *chunk_freqs;   // out: u16[256] per chunk
    uint32_t *flags;         // bit 0: a chunk could not be normalised; bit 1: the container does not fit out_cap
    unsigned int *claims;    // kWorkPools claim counters on a 64-byte line each; zero at launch
    unsigned long long *status; // one look-back word per chunk (AGGREGATE | piece size, then PREFIX | end of the piece); zero at launch
    unsigned long long wait_ticks; // how long a look-back may wait (device_common.hpp SpinWatch); 0 = the default half minute
    const uint32_t *rcp;     // [2][kAdaptRcpEntries]: Alverson reciprocals by frequency, then the round-up ones (model.cpp adapt_rcp_tables)
};
constexpr uint32_t kAdaptRcpEntries = 4097; // frequencies 0 .. 4096 (per-chunk models: scale_bits <= 12)
hipError_t launch_encode_adaptive(int format, const AdaptEncParams &p, int num_cus, hipStream_t stream, const char **name);

// (format, n_ways) combinations with a kernel.
bool ways_supported(int format, uint32_t n_ways);

} // namespace rans_amd
