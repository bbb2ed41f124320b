/*
 * rans_oracle.h -- CPU oracle for the interleaved rANS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithms in
 * rygorous/ryg_rans (rans_byte.h, rans64.h, rans_word_sse41.h, the SymbolStats
 * model builder of main.cpp / main_alias.cpp and the N-way driver loops of the
 * four mains), generalised from the reference's 1/2/8 lanes to any N.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product library (ryg_rans_amd/csrc) never links, calls or falls back to it.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function here against
 * (1) the book1 stream sizes published in the reference README:48,62,82,96,110,
 * (2) SHA-256 of streams/tables produced by the unmodified reference (SURVEY.md
 *     appendix B, tests/golden/book1_golden.json), and (3) oracle/_ref (the
 *     reference headers compiled where they lie) on seeded random inputs.
 */
#ifndef RANS_ORACLE_H
#define RANS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum orc_format {
    ORC_FMT_BYTE  = 0, /* rans_byte.h   : u32 state, L=2^23, 8-bit renorm  */
    ORC_FMT_WORD  = 1, /* rans_word_sse41.h: u32 state, L=2^16, 16-bit renorm, scale_bits 12 */
    ORC_FMT_R64   = 2, /* rans64.h      : u64 state, L=2^31, 32-bit renorm */
    ORC_FMT_ALIAS = 3  /* rans_byte.h stream + alias-table symbol mapping (main_alias.cpp) */
};

enum orc_status {
    ORC_OK = 0,
    ORC_E_ARG = 1,        /* bad argument / model */
    ORC_E_SPACE = 2,      /* output buffer too small */
    ORC_E_CORRUPT = 3     /* final state != L, cursor != end, or read past end */
};

/* Order-0 model: normalised frequencies + everything derived from them. */
typedef struct orc_model {
    uint32_t nsyms;        /* alphabet size (power of two for alias) */
    uint32_t log2nsyms;
    uint32_t scale_bits;   /* sum(freqs) == 1 << scale_bits */
    uint32_t *freqs;       /* [nsyms]   */
    uint32_t *cum;         /* [nsyms+1] */
    uint32_t *cum2sym;     /* [1<<scale_bits] symbol owning each cumulative slot */
    /* alias tables (NULL unless built) */
    uint32_t *divider;     /* [nsyms]   */
    uint32_t *slot_adjust; /* [2*nsyms] */
    uint32_t *slot_freqs;  /* [2*nsyms] */
    uint32_t *sym_id;      /* [2*nsyms] */
    uint32_t *alias_remap; /* [1<<scale_bits] */
} orc_model;

/* histogram; sym_bytes is 1 (uint8 symbols) or 2 (uint16 symbols) */
void orc_count_freqs(const void *syms, size_t n, int sym_bytes, uint32_t nsyms, uint32_t *freqs);

/* Rescale counts in freqs[] (in place) so they sum to target_total, never
 * squeezing a present symbol to 0.  cum[] receives nsyms+1 cumulative values. */
int orc_normalize_freqs(uint32_t *freqs, uint32_t *cum, uint32_t nsyms, uint32_t target_total);

/* Build a model from already-normalised freqs (sum must be 1<<scale_bits).
 * with_alias != 0 also builds the alias tables (nsyms must be a power of two
 * dividing 1<<scale_bits). */
orc_model *orc_model_create(const uint32_t *norm_freqs, uint32_t nsyms, uint32_t scale_bits, int with_alias);
void orc_model_destroy(orc_model *m);

/* Worst-case stream bytes for n symbols on n_ways lanes. */
size_t orc_stream_bound(int fmt, size_t n, uint32_t n_ways);

/* Encode n symbols as ONE n_ways-interleaved stream.  The stream is written
 * backwards and ends exactly at buf+cap; *out_len receives its byte length,
 * i.e. the stream is buf[cap-*out_len .. cap). */
int orc_encode(int fmt, const orc_model *m, const void *syms, size_t n, int sym_bytes,
               uint32_t n_ways, uint8_t *buf, size_t cap, size_t *out_len);

/* Decode one n_ways-interleaved stream of exactly len bytes into n symbols.
 * Never reads outside stream[0..len).  Returns ORC_E_CORRUPT unless every lane
 * ends at L and the cursor lands on stream+len. */
int orc_decode(int fmt, const orc_model *m, const uint8_t *stream, size_t len, size_t n,
               int sym_bytes, uint32_t n_ways, void *out);

/* Chunked helpers: the input is cut into chunks of chunk_syms symbols (last one
 * shorter); chunk c becomes an independent stream placed at offsets[c] (aligned
 * up to `align` bytes) in out; offsets[nchunks] = end of the last stream
 * (unaligned).  lengths[c] = stream byte length. */
int orc_encode_chunked(int fmt, const orc_model *m, const void *syms, size_t n, int sym_bytes,
                       uint32_t n_ways, size_t chunk_syms, size_t align,
                       uint8_t *out, size_t cap, uint64_t *offsets, uint32_t *lengths,
                       size_t *out_total);
/* Chunks [c0, c1) of a container against the oracle's own streams for them: -1 = all equal (length and bytes), else
 * the first differing chunk; -2 = bad argument.  Thread-safe. */
int64_t orc_compare_chunks(int fmt, const orc_model *m, const void *syms, size_t n, int sym_bytes, uint32_t n_ways,
                           size_t chunk_syms, uint64_t c0, uint64_t c1, const uint8_t *container,
                           const uint64_t *offsets, const uint32_t *lengths);
/* Per-chunk models: every chunk is coded as the reference codes an input -- count_freqs, normalize_freqs to
 * 1 << scale_bits, tables, the N-way loop (main.cpp:139-162 / main_simd.cpp:138-143 with the chunk as the input).
 * Chunks [c0, c1) of somebody else's container: rows[c] must be normalize(count(chunk c)) (u16[256] per chunk) and the
 * stream at offsets[c] the oracle's stream of chunk c under that model.  Returns -1 when all are equal, -2 on a bad
 * argument, else 2 * c (row of chunk c differs) or 2 * c + 1 (stream or length of chunk c differs).  Thread-safe. */
int64_t orc_compare_chunks_adaptive(int fmt, const uint8_t *syms, size_t n, uint32_t n_ways, size_t chunk_syms,
                                    uint32_t scale_bits, uint64_t c0, uint64_t c1, const uint8_t *container,
                                    const uint64_t *offsets, const uint32_t *lengths, const uint16_t *rows);
/* The oracle's own container of per-chunk models for chunks [c0, c1): stream of chunk c at out + (c - c0) * slot (slot =
 * orc_stream_bound of a chunk, rounded up to 16) -- lengths[c - c0] bytes ENDING at the slot's end -- and rows[c - c0]. */
int orc_encode_chunks_adaptive(int fmt, const uint8_t *syms, size_t n, uint32_t n_ways, size_t chunk_syms,
                               uint32_t scale_bits, uint64_t c0, uint64_t c1, uint8_t *out, size_t slot,
                               uint32_t *lengths, uint16_t *rows);
int orc_decode_chunked(int fmt, const orc_model *m, const uint8_t *container,
                       const uint64_t *offsets, const uint32_t *lengths, size_t n, int sym_bytes,
                       uint32_t n_ways, size_t chunk_syms, void *out);

/* Synthetic data generator shared by tests and bench:
 * Zipf(K, s) symbols from splitmix64(seed), inverse CDF by binary search. */
void orc_gen_zipf(void *out, size_t n, int sym_bytes, uint32_t K, double s, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
