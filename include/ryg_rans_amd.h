/*
 * ryg_rans_amd.h -- C ABI of the MI355X-native interleaved rANS coder.
 *
 * Drop-in boundary for the encode/decode hot path of rygorous/ryg_rans.  The
 * reference is header-only: its "API" is the inline per-symbol primitives
 * (rans_byte.h, rans64.h, rans_word_sse41.h) plus the driver LOOPS in the mains
 * that define the interleaved stream layout.  A per-symbol inline call cannot
 * cross to a device, so this ABI replaces one reference driver loop per call:
 *
 *   reference loop (what a maintainer deletes)            this ABI
 *   ---------------------------------------------------   --------------------------
 *   SymbolStats::count_freqs       main.cpp:59-66         rans_amd_count_freqs[_host]
 *   SymbolStats::normalize_freqs   main.cpp:75-129        rans_amd_normalize_freqs
 *   cum2sym / RansEncSymbolInit /  main.cpp:143-162       rans_amd_model_create
 *   RansWordTablesInitSymbol       main_simd.cpp:141-143    (tables staged for LDS)
 *   make_alias_table               main_alias.cpp:147-237
 *   N-way encode loop + flush      main.cpp:226-246,      rans_amd_encode[_host]
 *                                  main64.cpp:228-248,
 *                                  main_simd.cpp:287-300,
 *                                  main_alias.cpp:353-373
 *   N-way decode loop              main.cpp:259-280,      rans_amd_decode[_host]
 *                                  main64.cpp:261-282,
 *                                  main_simd.cpp:313-332,
 *                                  main_alias.cpp:386-405
 *
 * Streams are bit-exact with the reference formats: a
 * stream produced by the reference's N-way loop decodes here and vice versa.
 * The per-symbol RansEnc / RansDec / Rans64 / RansWord inline API is re-provided
 * for host code in include/ryg_rans_amd/compat/.
 *
 * One N-way stream is one wavefront's worth of sequential work, so inputs that
 * should fill a GPU are cut into CHUNKS of chunk_syms symbols; every chunk is an
 * independent, self-contained reference-format stream and a small index
 * (offsets/lengths) says where each one lives.  With chunk_syms >= n there is
 * exactly one chunk and the container IS the raw reference stream.
 *
 * Conventions
 *   - every function returns a rans_amd_status (0 = ok); no exceptions cross the ABI;
 *   - pointers named d_* are DEVICE memory on the context's GPU, h_* / unprefixed
 *     are host memory; `stream` is a hipStream_t passed as void* (NULL = default
 *     stream); device entry points are asynchronous on that stream unless stated;
 *   - device buffers must come from hipMalloc-like allocators (base 16-byte
 *     aligned, size padded to 16): kernels fetch the stream in aligned 16-byte
 *     granules and may touch the padding of the last granule, never beyond;
 *   - all streams are little-endian (as the reference on x86, README:12).
 */
#ifndef RYG_RANS_AMD_H
#define RYG_RANS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.6.0: additions only since 0.1.0 -- rans_amd_ctx_set_option, rans_amd_build_flags (0.2.0); rans_amd_encode_status, calls
 * inside a hipGraph capture (0.3.0); rans_amd_encode_slots + rans_amd_slot_bytes / rans_amd_encode_slots_bound,
 * rans_amd_container_compact, rans_amd_container_slice, chunk offsets on any multiple of the format's unit in every decoder
 * (0.4.0); rans_amd_encode_slots_sized + rans_amd_tight_slot_bytes / rans_amd_encode_sized_bound, rans_amd_probe_placement,
 * rans_amd_encode_adaptive_fmt / rans_amd_decode_adaptive_fmt (0.5.0); rans_amd_encode_adaptive_sized,
 * rans_amd_container_pack_indexed[_adaptive] (0.6.0).  A caller built
 * against an older header keeps working, with two behaviour changes it can observe: since 0.4.0
 * rans_amd_container_parse[_adaptive] want a 4-byte aligned `src` (RANS_AMD_E_ARG otherwise; an mmap at an odd offset must
 * be copied first), and since 0.5.0 rans_amd_container_compact checks its SOURCE index against src_bytes
 * (RANS_AMD_E_CORRUPT for an entry outside the source, which earlier versions read). */
#define RANS_AMD_VERSION 600

typedef enum rans_amd_status {
    RANS_AMD_OK = 0,
    RANS_AMD_E_ARG = 1,         /* NULL pointer, zero size, bad enum */
    RANS_AMD_E_MODEL = 2,       /* frequencies do not sum to 1<<scale_bits, freq==M, symbol with freq 0 */
    RANS_AMD_E_SPACE = 3,       /* output buffer too small */
    RANS_AMD_E_CORRUPT = 4,     /* decoder: final state != L, cursor != end of stream, read past end */
    RANS_AMD_E_UNSUPPORTED = 5, /* n_ways / scale_bits / alphabet outside what the kernels implement */
    RANS_AMD_E_HIP = 6,         /* HIP runtime error (see rans_amd_last_error) */
    RANS_AMD_E_NOMEM = 7
} rans_amd_status;

/* Bitstream formats. */
typedef enum rans_amd_format {
    RANS_AMD_FMT_BYTE = 0,  /* rans_byte.h:   u32 state, L=2^23, 8-bit renorm, scale_bits <= 16 */
    RANS_AMD_FMT_WORD = 1,  /* rans_word_sse41.h: u32 state, L=2^16, 16-bit renorm, scale_bits == 12 */
    RANS_AMD_FMT_R64 = 2,   /* rans64.h:      u64 state, L=2^31, 32-bit renorm, scale_bits <= 31 */
    RANS_AMD_FMT_ALIAS = 3  /* rans_byte.h stream, alias-method symbol map (main_alias.cpp:241-267) */
} rans_amd_format;

/* Host images of the model tables, for inspection and parity tests. */
typedef enum rans_amd_table {
    RANS_AMD_TAB_FREQS = 0,        /* u32[nsyms]                                       */
    RANS_AMD_TAB_CUM_FREQS = 1,    /* u32[nsyms+1]                                     */
    RANS_AMD_TAB_CUM2SYM = 2,      /* u8[M] (nsyms<=256) or u16[M]; main.cpp:145-148   */
    RANS_AMD_TAB_WORD_SLOTS = 3,   /* RansWordTables image: {u16 freq,u16 bias}[4096] + u8 slot2sym[4096] */
    RANS_AMD_TAB_ALIAS_DIVIDER = 4,     /* u32[nsyms]   main_alias.cpp:57      */
    RANS_AMD_TAB_ALIAS_SLOT_ADJUST = 5, /* u32[2*nsyms] main_alias.cpp:58      */
    RANS_AMD_TAB_ALIAS_SLOT_FREQS = 6,  /* u32[2*nsyms] main_alias.cpp:59      */
    RANS_AMD_TAB_ALIAS_SYM_ID = 7,      /* u8/u16[2*nsyms] main_alias.cpp:60   */
    RANS_AMD_TAB_ALIAS_REMAP = 8,       /* u32[M]       main_alias.cpp:63      */
    RANS_AMD_TAB_ENC_SYMBOLS = 9,  /* RansEncSymbol[nsyms] (16 B each, rans_byte.h:159-165) or
                                      Rans64EncSymbol[nsyms] (24 B each, rans64.h:152-158) for FMT_R64 */
    RANS_AMD_TAB_DEC_SYMBOLS = 10  /* RansDecSymbol[nsyms] (4 B) or Rans64DecSymbol[nsyms] (8 B) */
} rans_amd_table;

/* A context owns ONE encode workspace, error counter and work-counter ring: host calls on it are
 * serialised by a mutex, and its asynchronous work must stay on ONE stream at a time (synchronise
 * before switching streams).  Concurrent streams want one context each.  Models are immutable, may be
 * used from any stream, and may be destroyed after the context they were created with.
 *
 * hipGraph capture: rans_amd_encode / rans_amd_encode_slots[_sized] / rans_amd_container_compact / rans_amd_decode (and the
 * _adaptive pair) may be called while `stream` is being captured -- kernels and nothing else go into the graph (workspace
 * words are cleared by a kernel of the library, not by memset nodes), and a replay does what the call did, on the same
 * buffers.  Rules: the host result pointer (h_total_bytes / h_bad_chunks) must be NULL (fetch with
 * rans_amd_encode_status / rans_amd_decode_errors after a replay); the same call must have run once outside the capture
 * (workspaces are allocated on first use, and nothing may be allocated inside a capture: RANS_AMD_E_ARG says so); the
 * context must not be trimmed or destroyed while the graph exists; at most 8 captured decode launches of one context
 * run at the same time; rans_amd_set_timing has no effect on captured launches. */
typedef struct rans_amd_ctx rans_amd_ctx;     /* one per (process, GPU) and per concurrent stream */
typedef struct rans_amd_model rans_amd_model; /* immutable after creation */

/* ---- library / context ------------------------------------------------- */

int rans_amd_version(void);
/* Bit 0 (RANS_AMD_BUILD_MEASURE): this is the -DRANS_AMD_MEASURE build -- it reads the RANS_AMD_* experiment knobs from
 * the environment, some of which drop stores or skip copies.  0 for the library that ships: it reads no environment. */
#define RANS_AMD_BUILD_MEASURE 1u
unsigned rans_amd_build_flags(void);
const char *rans_amd_status_string(int status);
/* Text of the most recent failure on this thread ("" if none). */
const char *rans_amd_last_error(void);

/* Number of visible GPUs (hipGetDeviceCount); 0 when there is none. */
int rans_amd_device_count(void);

/* Bind a context to GPU `device`.  Fails with RANS_AMD_E_HIP when no GPU is
 * usable: there is no CPU fallback anywhere in this library. */
int rans_amd_ctx_create(int device, rans_amd_ctx **out_ctx);
int rans_amd_ctx_destroy(rans_amd_ctx *ctx);
/* Drop cached device workspaces (encode scratch). */
int rans_amd_ctx_trim(rans_amd_ctx *ctx);

/* Kernel-family choices of a context.  Every setting produces the same bytes -- the alternatives exist because
 * they were measured against each other and the tests run all of them; the defaults are the fast
 * ones.  The library reads no environment variable: this call is the only way to change what it launches.
 * (The separate -DRANS_AMD_MEASURE build adds a watchdog base and a per-wave trace file through the environment; the
 * experiment knobs of rounds 1-5 -- dropped stores, skipped copies, alternative kernels -- were removed in round 6, their
 * patches and logs live under profiles/.) */
enum rans_amd_option {
    RANS_AMD_OPT_LANE_KERNELS = 0,         /* RETIRED in 0.6.0: only 0 (automatic) is accepted, any other value is
                                              RANS_AMD_E_UNSUPPORTED.  (It pinned a generation of the narrow-interleave
                                              kernels; the first generation is now the fallback for the shapes only it
                                              serves -- huge chunks, chunk sizes off 16, u16 symbols, a handful of chunks.) */
    RANS_AMD_OPT_LANE_FUSED_PLACEMENT = 1, /* 1 = the lane-per-chunk encoders place their chunks themselves (default 0:
                                              layout + compaction kernels behind them) */
    RANS_AMD_OPT_FUSED_PLACEMENT = 2,      /* 0 = wave-per-chunk encoders followed by layout + compaction kernels
                                              (default 1: the coding kernel places its chunks itself) */
    RANS_AMD_OPT_DUAL_DECODE = 3,          /* 64-way alias decoders: 1 = two chunks per wavefront for the models whose tables
                                              leave room for one block per CU only (default), 0 = always one chunk per
                                              wavefront, 2 = two chunks per wavefront whenever the tables fit */
    RANS_AMD_OPT_ENC_SCRATCH_RING = 4      /* 1 = the fused wave encoders code into a ring of scratch slots per coding
                                              wave, reused once drained (half the workspace of a large encode, 2-4 %
                                              slower); default 0: one scratch slot per chunk */
};
int rans_amd_ctx_set_option(rans_amd_ctx *ctx, int option, int value);

/* ---- model building (SymbolStats, main.cpp:49-129) ---------------------- */

/* Histogram of n symbols (sym_bytes 1 or 2) into freqs[nsyms]; symbols >= nsyms
 * are an error.  Host version replaces count_freqs (main.cpp:59-66).  The counters are the reference's 32-bit ones
 * (main.cpp:49-57): n >= 2^32 is RANS_AMD_E_UNSUPPORTED, not a wrapped count -- the coders take any n, the model of such
 * an input comes from a part of it (or from per-chunk models). */
int rans_amd_count_freqs_host(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *freqs);
/* Device version: d_syms on the GPU, result copied to host freqs (synchronises `stream`). */
int rans_amd_count_freqs(rans_amd_ctx *ctx, const void *d_syms, uint64_t n, int sym_bytes, uint32_t nsyms,
                         uint32_t *freqs, void *stream);

/* In place: raw counts -> frequencies summing to target_total, exactly as
 * SymbolStats::normalize_freqs (main.cpp:75-129): 64-bit rescale of the
 * cumulative table, then every present symbol squeezed to 0 steals one slot from
 * the narrowest symbol wider than 1 (lowest index on ties).  cum_freqs gets
 * nsyms+1 entries.  Requires target_total >= nsyms. */
int rans_amd_normalize_freqs(uint32_t *freqs, uint32_t *cum_freqs, uint32_t nsyms, uint32_t target_total);

/* Build every host/device table the given format needs from NORMALISED
 * frequencies (sum == 1<<scale_bits) and upload them.
 *   BYTE : cum2sym[M] + RansDecSymbol/RansEncSymbol per symbol   (scale_bits <= 16)
 *   WORD : RansWordTables slots                                  (scale_bits == 12, nsyms <= 4096; u16 symbols
 *          beyond 256 -- rans_word_sse41.h:41 fixes 256, the stream format does not depend on the alphabet)
 *   R64  : cum2sym[M] + Rans64Dec/EncSymbol                      (scale_bits 7..16: table decoder; 1..6 and
 *          17..31: no 2^scale_bits table, the kernels search the cumulative frequencies -- same stream, slower)
 *   ALIAS: divider/slot_adjust/slot_freqs/sym_id (+alias_remap)  (nsyms a power of two dividing M)
 * A model in which one symbol owns the whole range (freq == M) is accepted by the byte, alias and rans64
 * coders as the reference accepts it (rans_byte.h:176-178, rans64.h:169-171: the state never moves, the
 * stream is the flushed initial states); the word format rejects it with RANS_AMD_E_MODEL (its 32-bit
 * renormalisation threshold wraps to 0), and an alias model holding a 65536-wide symbol
 * (scale_bits 16) exists as a host-only model only (RANS_AMD_E_UNSUPPORTED with a context).
 * ctx may be NULL: the model is then host-only (its tables can be exported with
 * rans_amd_model_table, encode/decode reject it with RANS_AMD_E_ARG). */
int rans_amd_model_create(rans_amd_ctx *ctx, int format, const uint32_t *norm_freqs, uint32_t nsyms,
                          uint32_t scale_bits, rans_amd_model **out_model);
int rans_amd_model_destroy(rans_amd_model *model);
/* One call for what every reference main does before coding (main.cpp:139-162):
 * count_freqs + normalize_freqs(1 << scale_bits) + table build.  syms is host memory when
 * syms_on_device == 0, else device memory (histogram on the GPU; synchronises `stream`).
 * norm_freqs_out (optional, nsyms entries) receives the normalised frequencies. */
int rans_amd_build_model_o0(rans_amd_ctx *ctx, int format, const void *syms, uint64_t n, int syms_on_device,
                            uint32_t nsyms, uint32_t scale_bits, uint32_t *norm_freqs_out,
                            rans_amd_model **out_model, void *stream);
int rans_amd_model_format(const rans_amd_model *model);
uint32_t rans_amd_model_scale_bits(const rans_amd_model *model);
uint32_t rans_amd_model_nsyms(const rans_amd_model *model);
/* Bytes per symbol in uncompressed buffers: 1 when nsyms <= 256, else 2. */
int rans_amd_model_sym_bytes(const rans_amd_model *model);
/* Copy the host image of a table; *size receives its byte size (call with
 * dst == NULL to query).  RANS_AMD_E_ARG if the model has no such table. */
int rans_amd_model_table(const rans_amd_model *model, int which, void *dst, size_t cap, size_t *size);

/* ---- chunk layout -------------------------------------------------------- */

/* ceil(n / chunk_syms); 0 for n == 0. */
uint64_t rans_amd_num_chunks(uint64_t n, uint32_t chunk_syms);
/* Worst-case bytes of ONE chunk stream (states + renorm units), rounded up to 16. */
uint64_t rans_amd_chunk_bound(int format, uint32_t chunk_syms, uint32_t n_ways);
/* Worst-case bytes of the whole container for n symbols. */
uint64_t rans_amd_encode_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms);
/* 1 if (format, n_ways) has a GPU kernel: n_ways in 1..512 (64, 128, 256: fastest paths). */
int rans_amd_ways_supported(int format, uint32_t n_ways);
/* Device scratch rans_amd_encode needs for this shape (one worst-case slot per chunk).  The
 * context allocates it on first use and keeps it (rans_amd_ctx_trim releases it); this query lets
 * a caller budget HBM up front. */
uint64_t rans_amd_encode_workspace_bytes(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms);

/* ---- bulk encode / decode on device-resident data ------------------------- */

/* Encode n symbols at d_syms (u8, or u16 when nsyms > 256) into d_out.
 * Chunk c covers symbols [c*chunk_syms, min(n, (c+1)*chunk_syms)) and becomes an
 * n_ways-interleaved reference-format stream at d_out + d_offsets[c] (16-byte
 * aligned) of d_lengths[c] bytes; d_offsets has nchunks+1 entries, the last one
 * is the container size.  If h_total_bytes != NULL the call synchronises
 * `stream` and stores the container size; it then also reports
 * RANS_AMD_E_SPACE if out_cap was too small and RANS_AMD_E_MODEL if a symbol
 * with frequency 0 was met. */
int rans_amd_encode(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n,
                    uint32_t n_ways, uint32_t chunk_syms, void *d_out, uint64_t out_cap,
                    uint64_t *d_offsets, uint32_t *d_lengths, uint64_t *h_total_bytes, void *stream);

/* The same encoder writing every chunk ONCE, in the reference's own buffer convention (rans_byte.h:22-26 "the encoder is
 * handed the END of the output buffer and moves down"; main.cpp:176-188, main64.cpp:178-190, main_simd.cpp:287-306,
 * main_alias.cpp:303-315: the stream is [ptr after the flush, end of the buffer)):
 *
 *     slot c  =  d_out[c * S, (c + 1) * S),   S = rans_amd_slot_bytes(format, n, n_ways, chunk_syms)
 *     chunk c's stream  =  the last d_lengths[c] bytes of slot c;   d_offsets[c] = (c + 1) * S - d_lengths[c]
 *     d_offsets[n_chunks] = n_chunks * S  (the size of the slot container)
 *
 * Every chunk's bytes are exactly what rans_amd_encode produces for it (the oracle's stream for that chunk); only WHERE
 * they lie differs.  rans_amd_encode's compact layout needs every chunk's length before it can place the next one, so its
 * coding kernels write each stream into scratch and move it once more (a second trip through HBM: 1.9 x the bytes);
 * here nothing moves -- no scratch, no look-back between workgroups, no copier waves -- and the call is faster by what
 * the move cost.  The price is capacity: out_cap must be at least rans_amd_encode_slots_bound() = n_chunks * S (S is the
 * worst case of a chunk: ~2 bytes per symbol, 4 for rans64 -- 2.6 x a typical compact container), checked up front
 * (RANS_AMD_E_SPACE, nothing launched).  Chunk starts are NOT 16-byte aligned: rans_amd_decode takes them as they are
 * (any multiple of the format's unit: 1 byte / 2 / 4).  rans_amd_container_compact turns a slot container into the compact
 * layout (needed before rans_amd_container_pack, or to shrink what is kept). */
uint64_t rans_amd_slot_bytes(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms);
uint64_t rans_amd_encode_slots_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms);
int rans_amd_encode_slots(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n,
                          uint32_t n_ways, uint32_t chunk_syms, void *d_out, uint64_t out_cap,
                          uint64_t *d_offsets, uint32_t *d_lengths, uint64_t *h_total_bytes, void *stream);

/* Slots of the CALLER's size -- one trip through HBM and a container about as large as the compact one.
 *
 * The reference does not hand its encoder the worst case either: main_simd.cpp:145 allocates in_size + in_size/8 + 128
 * bytes for a stream it expects to be smaller than the input, main.cpp:150 a fixed 32 MB.  Here:
 *
 *     slot c  =  d_out[c * slot_bytes, (c + 1) * slot_bytes)        slot_bytes: a multiple of 64, the caller's choice
 *     a chunk whose stream fits:   the last d_lengths[c] bytes of slot c, exactly as in rans_amd_encode_slots
 *     a chunk whose stream does NOT fit:  coded again (a second, normally empty launch) into the OVERFLOW region behind
 *         the slots, d_out[n_chunks * slot_bytes + k * W, + W) for the k-th such chunk (W = rans_amd_slot_bytes(): the
 *         worst case, nothing overflows there), d_offsets[c] pointing into it
 *     d_offsets[n_chunks] = n_chunks * slot_bytes + overflowed chunks * W   (the container's size)
 *
 * Every chunk's bytes are exactly what rans_amd_encode produces for it (the reference's stream for that chunk), wherever
 * they lie; the decoders take the index as it is.
 * rans_amd_tight_slot_bytes() sizes a slot from the MODEL: chunk_syms times the model's expected code length (its entropy
 * under itself) plus 2 %, the N flushed states, four standard deviations of a chunk's code length and 16 bytes of slack,
 * rounded up to whole 64-byte lines (the 64-way byte-symbol coders check their slot exactly where they flush their staging
 * window; they do so for symbol buffers and chunk_syms that are multiples of 4 -- any other shape, and every other coder,
 * checks by what two rounds can emit at most, and gets that margin added here when the model's shape alone decides it; an
 * unaligned buffer of a staged shape overflows a little more often, each overflowed chunk at twice its cost) --
 * input that follows the model overflows about one chunk in 30 000; input that does not (a model built from other data, an
 * incompressible stretch) overflows more often and still encodes correctly, each overflowed chunk at twice its cost.
 * out_cap must hold the slots (RANS_AMD_E_SPACE up front otherwise); whatever it has beyond them is overflow region, and a
 * call whose overflowed chunks do not fit there reports RANS_AMD_E_SPACE (with h_total_bytes, or later through
 * rans_amd_encode_status) -- rans_amd_encode_sized_bound(.., overflow_chunks) is the capacity for that many.  slot_bytes at
 * or above rans_amd_slot_bytes() is rans_amd_encode_slots. */
uint64_t rans_amd_tight_slot_bytes(const rans_amd_model *model, uint32_t n_ways, uint32_t chunk_syms);
uint64_t rans_amd_encode_sized_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms, uint64_t slot_bytes,
                                     uint64_t overflow_chunks);
int rans_amd_encode_slots_sized(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n,
                                uint32_t n_ways, uint32_t chunk_syms, uint64_t slot_bytes, void *d_out, uint64_t out_cap,
                                uint64_t *d_offsets, uint32_t *d_lengths, uint64_t *h_total_bytes, void *stream);

/* Copy the n_chunks streams [d_src + d_src_offsets[c], + d_lengths[c]) -- any layout the decoders accept: a slot
 * container, a chunk range of another container, chunks in any order -- into the compact layout at d_dst:
 * d_dst_offsets[c] = sum_{i<c} align16(d_lengths[i]), d_dst_offsets[n_chunks] = end of the last stream (n_chunks + 1
 * entries).  d_src (src_bytes long) and d_dst (dst_cap) must not overlap.  Asynchronous on `stream` unless h_total_bytes is
 * given (then: synchronises, stores d_dst_offsets[n_chunks], reports RANS_AMD_E_SPACE when dst_cap was too small --
 * nothing is copied in that case; rans_amd_encode_status reports the same later for an asynchronous call).  The source
 * index is data: an entry that does not lie inside [0, src_bytes) is neither read nor written and the call reports
 * RANS_AMD_E_CORRUPT (the other chunks are copied). */
int rans_amd_container_compact(rans_amd_ctx *ctx, const void *d_src, uint64_t src_bytes, const uint64_t *d_src_offsets,
                               const uint32_t *d_lengths, uint64_t n_chunks, void *d_dst, uint64_t dst_cap,
                               uint64_t *d_dst_offsets, uint64_t *h_total_bytes, void *stream);

/* Synchronise `stream` and report how the last rans_amd_encode / rans_amd_encode_slots[_sized] / rans_amd_encode_adaptive[_fmt|_sized]
 * of this context ended when it was called without h_total_bytes (asynchronously, or as a graph node): RANS_AMD_OK,
 * RANS_AMD_E_MODEL (a symbol with frequency 0), RANS_AMD_E_SPACE (out_cap too small) or RANS_AMD_E_HIP -- and, when that is
 * RANS_AMD_OK, what the last asynchronous rans_amd_container_compact had to say (RANS_AMD_E_SPACE, RANS_AMD_E_CORRUPT): it
 * keeps its verdict in a word of its own that the next compaction replaces and that this call and a compaction's own
 * synchronous return read and reset -- an encode queued behind a compaction does not discard it (0.6.0; before, every
 * encode did).  The container size is d_offsets[n_chunks]. */
int rans_amd_encode_status(rans_amd_ctx *ctx, void *stream);

/* Decode a container.  d_out receives n symbols.  Chunk c is the d_lengths[c] bytes at d_container + d_offsets[c];
 * offsets may be any multiple of the format's renormalisation unit (byte / alias: 1, word: 2, rans64: 4) and need not
 * ascend nor lie near each other -- compact containers (rans_amd_encode), slot containers (rans_amd_encode_slots[_sized],
 * overflow region included) and hand-made indexes over reference streams all decode.  Every chunk is checked the way
 * the reference's streams allow (all final states == L, cursor == end of the
 * chunk's stream, no read past it); failures are counted on the device.  If
 * h_bad_chunks != NULL the call synchronises `stream`, stores the number of
 * failed chunks and returns RANS_AMD_E_CORRUPT when it is non-zero.  Otherwise
 * the call is asynchronous; fetch the count later with rans_amd_decode_errors.
 * No kernel ever reads outside the 16-byte granules covering
 * [d_container, d_container + container_bytes). */
int rans_amd_decode(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_container,
                    uint64_t container_bytes, const uint64_t *d_offsets, const uint32_t *d_lengths,
                    uint64_t n, uint32_t n_ways, uint32_t chunk_syms, void *d_out,
                    uint64_t *h_bad_chunks, void *stream);
/* Decoding a chunk RANGE [lo, hi) of a container (shard g of G owns chunks [g C / G, (g + 1) C / G)):
 * chunks are independent and the index is per chunk, so
 *
 *     rans_amd_decode(ctx, model, d_container, container_bytes, d_offsets + lo, d_lengths + lo,
 *                     n_range, n_ways, chunk_syms, (char *)d_out + lo * chunk_syms * sym_bytes, ...)
 *
 * with n_range = min(n, hi * chunk_syms) - lo * chunk_syms decodes exactly those chunks (only the container's LAST
 * chunk may be shorter than chunk_syms, and it may only be the last chunk of a range).  A rank that holds only ITS bytes of
 * the payload asks rans_amd_container_slice (host arrays in, host arrays out) for them: it receives the byte range
 * [*byte_begin, *byte_end) that covers the range's streams (*byte_begin 16-byte aligned) and hi - lo + 1 offsets
 * relative to *byte_begin -- copy that byte range to the device, pass it with the rebased offsets and d_lengths + lo. */
int rans_amd_container_slice(const uint64_t *offsets, const uint32_t *lengths, uint64_t n_chunks, uint64_t lo, uint64_t hi,
                             uint64_t *byte_begin, uint64_t *byte_end, uint64_t *rebased_offsets);

/* Placement probe (setup time, not the data path).  On MI355X the same decode over the same bytes runs 4-6 % faster when
 * its container and its output lie in DIFFERENT classes of device memory than when they share one (profiles/r04_allocation.md:
 * a container x output matrix of allocations shows an XOR pattern; what moves is the memory-side read latency).  Which class
 * a hipMalloc lands in is the driver's choice and no allocation flag steers it -- the library cannot pick for the caller.
 * What a caller that keeps its buffers for a while CAN do is allocate a few candidates and ask which pair is fastest:
 *
 *   d_containers[0 .. n_containers)   device buffers holding THE SAME container_bytes bytes (copies of one container)
 *   d_outs[0 .. n_outs)               candidate output buffers (n symbols each)
 *
 * Every pair is decoded `launches` times (after two warm-up launches), `sweeps` times over, interleaved so that a drift of
 * the clocks favours nobody; times come from HIP events on `stream`.  *best_container / *best_out name the fastest pair;
 * ms_matrix (NULL, or n_containers * n_outs floats, row = container) receives the mean milliseconds of every pair, so the
 * caller sees what the choice was worth (ms_matrix[0] is the pair two plain allocations would have got).  Synchronous;
 * every decode is checked (RANS_AMD_E_CORRUPT if one fails) through the context's failed-chunk counter, which the probe
 * reads and resets: call it with no asynchronous decode of this context pending (rans_amd_decode_errors first), or a
 * failure of that earlier decode is reported here, as the probe's.  launches = 0 -> 6, sweeps = 0 -> 2.  The caller frees
 * the candidates it does not keep. */
int rans_amd_probe_placement(rans_amd_ctx *ctx, const rans_amd_model *model, const void *const *d_containers,
                             uint32_t n_containers, uint64_t container_bytes, const uint64_t *d_offsets,
                             const uint32_t *d_lengths, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                             void *const *d_outs, uint32_t n_outs, uint32_t launches, uint32_t sweeps,
                             uint32_t *best_container, uint32_t *best_out, float *ms_matrix, void *stream);

/* Synchronise `stream` and return (and reset) the failed-chunk count accumulated
 * by asynchronous rans_amd_decode calls on this context. */
int rans_amd_decode_errors(rans_amd_ctx *ctx, uint64_t *h_bad_chunks, void *stream);

/* ---- per-chunk adaptive models ------------------------------------------------
 *
 * The reference builds ONE order-0 model per input (main.cpp:139-162: count_freqs, normalize_freqs,
 * RansEncSymbolInit / RansDecSymbolInit, cum2sym).  Here every CHUNK is such an input: its own histogram, its own
 * normalised frequencies (exactly normalize_freqs of that chunk), its own tables -- built by the wavefront that
 * codes the chunk, in its LDS, from the chunk's 256 frequencies.  Byte format (rans_byte.h; scale_bits 8..12) or -- the
 * _fmt entry points -- word format (rans_word_sse41.h; its own 12 bits: the model main_simd.cpp:138-143 builds, per chunk),
 * 256 symbols; n_ways 1..512.  Chunk c's stream is the reference-format stream of a model built for chunk c
 * alone; the container layout and index are those of rans_amd_encode.  d_chunk_freqs (device, u16[256] per chunk,
 * rans_amd_chunk_freqs_bytes) travels with the container: rans_amd_encode_adaptive fills it, rans_amd_decode_adaptive
 * reads it (a chunk whose frequencies do not sum to 1 << scale_bits is counted as corrupt, never decoded). */
uint64_t rans_amd_chunk_freqs_bytes(uint64_t n, uint32_t chunk_syms);
/* Histograms on the GPU, normalize_freqs on the host (synchronises `stream`), encode + layout + compaction on the GPU. */
int rans_amd_encode_adaptive(rans_amd_ctx *ctx, const void *d_syms, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                             uint32_t scale_bits, void *d_out, uint64_t out_cap, uint64_t *d_offsets,
                             uint32_t *d_lengths, uint16_t *d_chunk_freqs, uint64_t *h_total_bytes, void *stream);
int rans_amd_decode_adaptive(rans_amd_ctx *ctx, const void *d_container, uint64_t container_bytes,
                             const uint64_t *d_offsets, const uint32_t *d_lengths, const uint16_t *d_chunk_freqs,
                             uint64_t n, uint32_t n_ways, uint32_t chunk_syms, uint32_t scale_bits, void *d_out,
                             uint64_t *h_bad_chunks, void *stream);
/* The same with the stream format as an argument: RANS_AMD_FMT_BYTE (= the two above) or RANS_AMD_FMT_WORD (scale_bits must be
 * 12); RANS_AMD_E_UNSUPPORTED for any other format.  (Word format, a chunk that holds one symbol value only: its frequency
 * is 4096 = M, for which rans_word_sse41.h:85's 32-bit renormalisation bound wraps to 0 -- the reference's encoder then emits
 * a word per symbol, and so does this one, bit for bit; such chunks cost 2 bytes per symbol.) */
int rans_amd_encode_adaptive_fmt(rans_amd_ctx *ctx, int format, const void *d_syms, uint64_t n, uint32_t n_ways,
                                 uint32_t chunk_syms, uint32_t scale_bits, void *d_out, uint64_t out_cap, uint64_t *d_offsets,
                                 uint32_t *d_lengths, uint16_t *d_chunk_freqs, uint64_t *h_total_bytes, void *stream);
int rans_amd_decode_adaptive_fmt(rans_amd_ctx *ctx, int format, const void *d_container, uint64_t container_bytes,
                                 const uint64_t *d_offsets, const uint32_t *d_lengths, const uint16_t *d_chunk_freqs,
                                 uint64_t n, uint32_t n_ways, uint32_t chunk_syms, uint32_t scale_bits, void *d_out,
                                 uint64_t *h_bad_chunks, void *stream);

/* The same encode as ONE kernel (0.6.0; the recommended entry point).  The wavefront that codes a chunk first counts it,
 * normalises the counts, builds its records and codes the chunk -- and because it knows the chunk's histogram BEFORE it
 * codes, it knows an upper bound of the chunk's stream (sum count[s] * log2(M / freq[s]) bits, what the floor of the rANS
 * update can add, the flushed states) and takes exactly that much room: no scratch trip, no layout pass, no compaction, no
 * second launch.  Chunk c's piece (a whole number of 64-byte lines) starts where the pieces of the chunks before it end
 * -- index order, the same layout from run to run -- and its stream is the LAST d_lengths[c] bytes of the piece, so the
 * container is about as large as the streams (+ ~1.5 %: the bound's slack).  d_offsets / d_lengths say where every stream
 * starts (any decoder takes such an index; rans_amd_container_compact / rans_amd_container_pack_indexed_adaptive close the
 * gaps); d_offsets[n_chunks] = bytes of `d_out` in use.
 * out_cap: rans_amd_encode_adaptive_sized_bound() can never be exceeded; a smaller buffer works as long as the pieces fit
 * (RANS_AMD_E_SPACE from the call or from rans_amd_encode_status otherwise; chunks whose piece does not fit are not coded,
 * their length is 0).  Every chunk's bytes and its row of d_chunk_freqs equal those of rans_amd_encode_adaptive_fmt.  (The
 * first call of a context uploads a 32 KiB table: make one call outside a hipGraph capture first.) */
int rans_amd_encode_adaptive_sized(rans_amd_ctx *ctx, int format, const void *d_syms, uint64_t n, uint32_t n_ways,
                                   uint32_t chunk_syms, uint32_t scale_bits, void *d_out, uint64_t out_cap,
                                   uint64_t *d_offsets, uint32_t *d_lengths, uint16_t *d_chunk_freqs, uint64_t *h_total_bytes,
                                   void *stream);
uint64_t rans_amd_encode_adaptive_sized_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms);

/* ---- host-buffer convenience: one raw reference-format stream -------------- */

/* Exactly the reference encoder loops: the stream is written BACKWARDS and ends
 * at buf + cap (rans_byte.h:22-26); *out_len receives its length, i.e. the
 * stream is buf[cap - *out_len, cap).  Runs on the GPU (copies in/out). */
int rans_amd_encode_host(rans_amd_ctx *ctx, const rans_amd_model *model, const void *syms, uint64_t n,
                         uint32_t n_ways, uint8_t *buf, uint64_t cap, uint64_t *out_len);
/* Decode one raw n_ways stream of exactly len bytes into n symbols. */
int rans_amd_decode_host(rans_amd_ctx *ctx, const rans_amd_model *model, const uint8_t *stream_bytes,
                         uint64_t len, uint64_t n, uint32_t n_ways, void *out);

/* ---- container file format (host memory) -----------------------------------------
 *
 * The reference keeps n, the tables and the stream start out of band (main.cpp:182,196) and
 * defines no file format.  These helpers serialise what a decoder needs into ONE
 * self-describing buffer:
 *
 *   [ 80-byte header | u32 freqs[nsyms] | u32 lengths[n_chunks] | pad to 16 | payload ]
 *
 * header: magic "RANSAMD1", version, format, scale_bits, nsyms, n_ways, chunk_syms, sym_bytes,
 * n_symbols, n_chunks, payload_bytes, FNV-1a-64 of header+freqs+lengths.  Chunk c starts at
 * payload + sum_{i<c} align16(lengths[i]) (rans_amd_offsets_from_lengths) and is a plain
 * reference-format n_ways stream.  Everything is little endian. */
typedef struct rans_amd_container_info {
    uint32_t format;      /* rans_amd_format */
    uint32_t scale_bits;
    uint32_t nsyms;
    uint32_t n_ways;
    uint32_t chunk_syms;
    uint32_t sym_bytes;   /* 1 or 2 */
    uint64_t n_symbols;
    uint64_t n_chunks;
    uint64_t payload_bytes; /* == offsets[n_chunks] */
} rans_amd_container_info;

/* offsets[c] = sum_{i<c} align16(lengths[i]); offsets[n_chunks] = end of the last stream. */
int rans_amd_offsets_from_lengths(const uint32_t *lengths, uint64_t n_chunks, uint64_t *offsets);
/* Total bytes of the serialised container described by info. */
uint64_t rans_amd_container_bytes(const rans_amd_container_info *info);
/* Serialise.  payload = the container bytes produced by rans_amd_encode (copied to the host). */
int rans_amd_container_pack(const rans_amd_container_info *info, const uint32_t *norm_freqs,
                            const uint32_t *lengths, const void *payload, void *dst, uint64_t cap,
                            uint64_t *out_bytes);
/* The same file from a container in ANY layout (0.6.0) -- what the reference does when it writes [rans_begin, end) of a
 * buffer it coded into from the end (main.cpp:182-188): chunk c is read from payload + offsets[c] (lengths[c] bytes) and
 * written to its place in the file's compact payload; info->payload_bytes is ignored (the file's is
 * rans_amd_packed_payload_bytes(lengths, n_chunks); dst needs rans_amd_container_bytes() of an info with that value).
 * `payload` / `payload_bytes`: the host copy of the device container -- sized slots (rans_amd_encode_slots_sized, overflowed
 * chunks included), worst-case slots, a slice, a compact one: encode -> one D2H copy -> this call, no compaction pass on the
 * device.  RANS_AMD_E_CORRUPT when an index entry does not lie inside [0, payload_bytes) (nothing outside is read). */
uint64_t rans_amd_packed_payload_bytes(const uint32_t *lengths, uint64_t n_chunks);
int rans_amd_container_pack_indexed(const rans_amd_container_info *info, const uint32_t *norm_freqs, const uint64_t *offsets,
                                    const uint32_t *lengths, const void *payload, uint64_t payload_bytes, void *dst,
                                    uint64_t cap, uint64_t *out_bytes);
/* Validate and index a serialised container in place: *freqs, *lengths and *payload point INTO
 * src (which must be 4-byte aligned: RANS_AMD_E_ARG otherwise).  RANS_AMD_E_CORRUPT on a bad
 * magic/version/checksum or inconsistent sizes; whatever is accepted lies inside [src, src + bytes). */
int rans_amd_container_parse(const void *src, uint64_t bytes, rans_amd_container_info *info,
                             const uint32_t **freqs, const uint32_t **lengths, const void **payload);

/* Version 2 of the same wrapper, for rans_amd_encode_adaptive containers: the single frequency table is replaced by
 * u16 chunk_freqs[n_chunks][256] (info->format RANS_AMD_FMT_BYTE with scale_bits 8..12 or RANS_AMD_FMT_WORD with 12 -- the
 * header carries it --, nsyms 256, sym_bytes 1):
 *   [ 80-byte header (version 2) | u16 chunk_freqs[n_chunks][256] | u32 lengths[n_chunks] | pad to 16 | payload ] */
uint64_t rans_amd_container_bytes_adaptive(const rans_amd_container_info *info);
int rans_amd_container_pack_adaptive(const rans_amd_container_info *info, const uint16_t *chunk_freqs,
                                     const uint32_t *lengths, const void *payload, void *dst, uint64_t cap,
                                     uint64_t *out_bytes);
/* ... and from any layout (rans_amd_encode_adaptive_sized's pieces): see rans_amd_container_pack_indexed. */
int rans_amd_container_pack_indexed_adaptive(const rans_amd_container_info *info, const uint16_t *chunk_freqs,
                                             const uint64_t *offsets, const uint32_t *lengths, const void *payload,
                                             uint64_t payload_bytes, void *dst, uint64_t cap, uint64_t *out_bytes);
int rans_amd_container_parse_adaptive(const void *src, uint64_t bytes, rans_amd_container_info *info,
                                      const uint16_t **chunk_freqs, const uint32_t **lengths, const void **payload);

/* ---- measurement helpers ---------------------------------------------------- */

/* Duration in milliseconds of the most recent decode / encode kernel group that
 * was launched with timing enabled (HIP events recorded on the launch stream).
 * Synchronises those events. */
int rans_amd_set_timing(rans_amd_ctx *ctx, int enabled);
/* enabled == 2 additionally makes every wave of the next wave-per-chunk decode launches record its own
 * clocks (the GPU-side analogue of the __rdtsc bracket of main.cpp:171,184-186); those launches synchronise
 * `stream` and are a few percent slower, so measure throughput with enabled <= 1. */
typedef struct rans_amd_wave_clocks {
    uint64_t waves;         /* wavefronts that ran */
    uint64_t rounds;        /* 64-symbol rounds they decoded, all together (full rounds only) */
    uint64_t shader_cycles; /* sum over waves of the shader cycles (s_memtime) between a wave's start and end */
    double sclk_hz;         /* shader clock during the kernel: cycles / 100 MHz ticks of the longest-running wave */
    double kernel_ticks_ms; /* first wave start to last wave end on the constant 100 MHz clock */
} rans_amd_wave_clocks;
/* shader_cycles / rounds = clocks one wave spends per round of 64 symbols (waiting included); divide by the
 * waves resident per SIMD (8) for the issue cycles a SIMD spends per round. */
int rans_amd_last_wave_clocks(rans_amd_ctx *ctx, rans_amd_wave_clocks *out);
/* Always on, no cost worth naming: every decode launch records when its first wavefront started and its last
 * one ended (constant 100 MHz clock).  span_ms[0..count) receives those spans for the last `count` (<= 32)
 * decode launches of this context, oldest first (-1 where there was none); synchronises `stream`.  The
 * difference to the HIP-event time of a launch is what the launch spends outside its wavefronts: dispatch,
 * the wait for the previous kernel of the stream, the cache write-back at its end. */
int rans_amd_launch_spans(rans_amd_ctx *ctx, uint32_t count, double *span_ms, void *stream);
int rans_amd_last_kernel_ms(rans_amd_ctx *ctx, float *decode_ms, float *encode_ms);
/* Name of the dominant device kernel the last decode used (for profile matching). */
const char *rans_amd_last_decode_kernel(rans_amd_ctx *ctx);
/* Name of the coding kernel the last encode used; *fused_placement (may be NULL) = 1 when that kernel also placed the
 * chunks in the container itself, 0 when the offset scan and the compaction ran as kernels of their own behind it, 2 when
 * there was nothing to place (rans_amd_encode_slots: the chunks stay where they were coded). */
const char *rans_amd_last_encode_kernel(rans_amd_ctx *ctx, int *fused_placement);

#ifdef __cplusplus
}
#endif
#endif /* RYG_RANS_AMD_H */
