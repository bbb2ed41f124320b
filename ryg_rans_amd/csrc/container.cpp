// container.cpp -- serialised form of a chunked rANS container (host memory only, no HIP).
// The reference has no file format (main.cpp:182,196 keep n, tables and stream start out of
// band); this is the self-describing wrapper SURVEY.md 8(f) item 2 asks for.  See
// include/ryg_rans_amd.h for the layout.
#include "../../include/ryg_rans_amd.h"

#include <cstdint>
#include <cstring>

namespace {

constexpr char kMagic[8] = {'R', 'A', 'N', 'S', 'A', 'M', 'D', '1'};
constexpr uint32_t kVersion = 1;
constexpr uint64_t kHeaderBytes = 80;

// fixed 80-byte little-endian header
struct Header {
    char magic[8];
    uint32_t version;
    uint32_t format;
    uint32_t scale_bits;
    uint32_t nsyms;
    uint32_t n_ways;
    uint32_t chunk_syms;
    uint32_t sym_bytes;
    uint32_t reserved;
    uint64_t n_symbols;
    uint64_t n_chunks;
    uint64_t payload_bytes;
    uint64_t reserved2;
    uint64_t checksum; // FNV-1a 64 over the header with this field zero, then freqs, then lengths
};
static_assert(sizeof(Header) == kHeaderBytes, "header layout");

uint64_t fnv1a(uint64_t h, const void *data, uint64_t n)
{
    const uint8_t *p = static_cast<const uint8_t *>(data);
    for (uint64_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

uint64_t align16(uint64_t v) { return (v + 15) & ~uint64_t(15); }

bool info_sane(const rans_amd_container_info *i)
{
    if (!i || i->format > RANS_AMD_FMT_ALIAS || i->nsyms == 0 || i->nsyms > 65536 || i->scale_bits == 0 ||
        i->scale_bits > 31 || i->chunk_syms == 0 || i->n_ways == 0 || (i->sym_bytes != 1 && i->sym_bytes != 2))
        return false;
    // no n_symbols + chunk_syms - 1: that wraps for n_symbols near 2^64 and would let a forged header
    // (FNV is no protection) claim a huge symbol count with zero chunks
    const uint64_t want_chunks = i->n_symbols / i->chunk_syms + (i->n_symbols % i->chunk_syms != 0);
    return want_chunks == i->n_chunks && i->n_chunks < (1ull << 40);
}

uint64_t meta_bytes(const rans_amd_container_info *i)
{
    return align16(kHeaderBytes + 4ull * i->nsyms + 4ull * i->n_chunks);
}

} // namespace

extern "C" {

int rans_amd_offsets_from_lengths(const uint32_t *lengths, uint64_t n_chunks, uint64_t *offsets)
{
    if (!offsets || (n_chunks && !lengths))
        return RANS_AMD_E_ARG;
    uint64_t at = 0;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        offsets[c] = at;
        if (c + 1 == n_chunks) {
            offsets[n_chunks] = at + lengths[c];
            return RANS_AMD_OK;
        }
        at += align16(lengths[c]);
    }
    offsets[0] = 0;
    return RANS_AMD_OK;
}

int rans_amd_container_slice(const uint64_t *offsets, const uint32_t *lengths, uint64_t n_chunks, uint64_t lo, uint64_t hi,
                             uint64_t *byte_begin, uint64_t *byte_end, uint64_t *rebased_offsets)
{
    if (!offsets || !lengths || !byte_begin || !byte_end || !rebased_offsets || lo > hi || hi > n_chunks)
        return RANS_AMD_E_ARG;
    if (lo == hi) { // an empty range: an empty container
        *byte_begin = *byte_end = lo < n_chunks ? (offsets[lo] & ~uint64_t(15)) : 0;
        rebased_offsets[0] = 0;
        return RANS_AMD_OK;
    }
    uint64_t b = ~uint64_t(0), e = 0;
    for (uint64_t c = lo; c < hi; ++c) { // (offsets need not ascend: the range's extent is the hull of its streams)
        if (offsets[c] > ~uint64_t(0) - lengths[c])
            return RANS_AMD_E_CORRUPT;
        b = offsets[c] < b ? offsets[c] : b;
        e = offsets[c] + lengths[c] > e ? offsets[c] + lengths[c] : e;
    }
    b &= ~uint64_t(15); // whole 16-byte granules: the slice starts as aligned as the container does
    for (uint64_t c = lo; c < hi; ++c)
        rebased_offsets[c - lo] = offsets[c] - b;
    rebased_offsets[hi - lo] = e - b;
    *byte_begin = b;
    *byte_end = e;
    return RANS_AMD_OK;
}

uint64_t rans_amd_container_bytes(const rans_amd_container_info *info)
{
    if (!info_sane(info))
        return 0;
    return meta_bytes(info) + info->payload_bytes;
}

int rans_amd_container_pack(const rans_amd_container_info *info, const uint32_t *norm_freqs, const uint32_t *lengths,
                            const void *payload, void *dst, uint64_t cap, uint64_t *out_bytes)
{
    if (!info_sane(info) || !norm_freqs || !dst || (info->n_chunks && (!lengths || !payload)))
        return RANS_AMD_E_ARG;
    // the index must describe exactly payload_bytes
    uint64_t at = 0;
    for (uint64_t c = 0; c < info->n_chunks; ++c)
        at = (c + 1 == info->n_chunks) ? at + lengths[c] : at + align16(lengths[c]);
    if (at != info->payload_bytes)
        return RANS_AMD_E_ARG;
    uint64_t sum = 0;
    for (uint32_t s = 0; s < info->nsyms; ++s)
        sum += norm_freqs[s];
    if (sum != (1ull << info->scale_bits))
        return RANS_AMD_E_MODEL;
    const uint64_t total = meta_bytes(info) + info->payload_bytes;
    if (cap < total)
        return RANS_AMD_E_SPACE;

    uint8_t *out = static_cast<uint8_t *>(dst);
    memset(out, 0, (size_t)meta_bytes(info));
    Header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.format = info->format;
    h.scale_bits = info->scale_bits;
    h.nsyms = info->nsyms;
    h.n_ways = info->n_ways;
    h.chunk_syms = info->chunk_syms;
    h.sym_bytes = info->sym_bytes;
    h.n_symbols = info->n_symbols;
    h.n_chunks = info->n_chunks;
    h.payload_bytes = info->payload_bytes;
    uint64_t ck = fnv1a(0xcbf29ce484222325ull, &h, sizeof(h));
    ck = fnv1a(ck, norm_freqs, 4ull * info->nsyms);
    ck = fnv1a(ck, lengths, 4ull * info->n_chunks);
    h.checksum = ck;
    memcpy(out, &h, sizeof(h));
    memcpy(out + kHeaderBytes, norm_freqs, 4ull * info->nsyms);
    if (info->n_chunks)
        memcpy(out + kHeaderBytes + 4ull * info->nsyms, lengths, 4ull * info->n_chunks);
    if (info->payload_bytes)
        memcpy(out + meta_bytes(info), payload, (size_t)info->payload_bytes);
    if (out_bytes)
        *out_bytes = total;
    return RANS_AMD_OK;
}

uint64_t rans_amd_packed_payload_bytes(const uint32_t *lengths, uint64_t n_chunks)
{
    uint64_t at = 0;
    for (uint64_t c = 0; lengths && c < n_chunks; ++c)
        at = (c + 1 == n_chunks) ? at + lengths[c] : at + align16(lengths[c]);
    return at;
}

// chunk c of a container in ANY layout (stream at src + offsets[c], lengths[c] bytes) into the file's compact payload
static int copy_chunks_indexed(const uint64_t *offsets, const uint32_t *lengths, uint64_t n_chunks, const void *src, uint64_t src_bytes,
                               uint8_t *dst)
{
    uint64_t at = 0;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        // (the index is DATA: nothing outside [src, src + src_bytes) is read, whatever it says)
        if (offsets[c] > src_bytes || lengths[c] > src_bytes - offsets[c])
            return RANS_AMD_E_CORRUPT;
        memcpy(dst + at, static_cast<const uint8_t *>(src) + offsets[c], lengths[c]);
        const uint64_t next = (c + 1 == n_chunks) ? at + lengths[c] : at + align16(lengths[c]);
        if (next > at + lengths[c])
            memset(dst + at + lengths[c], 0, (size_t)(next - at - lengths[c])); // alignment padding: zeros, so that files compare
        at = next;
    }
    return RANS_AMD_OK;
}

int rans_amd_container_pack_indexed(const rans_amd_container_info *info, const uint32_t *norm_freqs, const uint64_t *offsets,
                                    const uint32_t *lengths, const void *payload, uint64_t payload_bytes, void *dst, uint64_t cap,
                                    uint64_t *out_bytes)
{
    if (!info_sane(info) || !norm_freqs || !dst || (info->n_chunks && (!offsets || !lengths || !payload)))
        return RANS_AMD_E_ARG;
    rans_amd_container_info packed = *info; // (the file's payload is the compact one, whatever the source's extent)
    packed.payload_bytes = rans_amd_packed_payload_bytes(lengths, info->n_chunks);
    const uint64_t meta = meta_bytes(&packed);
    if (cap < meta + packed.payload_bytes)
        return RANS_AMD_E_SPACE;
    uint8_t *out = static_cast<uint8_t *>(dst);
    if (int rc = copy_chunks_indexed(offsets, lengths, info->n_chunks, payload, payload_bytes, out + meta))
        return rc;
    // header, tables and checksum as rans_amd_container_pack writes them (a NULL payload there: the header only)
    uint64_t sum = 0;
    for (uint32_t s = 0; s < info->nsyms; ++s)
        sum += norm_freqs[s];
    if (sum != (1ull << info->scale_bits))
        return RANS_AMD_E_MODEL;
    memset(out, 0, (size_t)meta);
    Header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.format = packed.format;
    h.scale_bits = packed.scale_bits;
    h.nsyms = packed.nsyms;
    h.n_ways = packed.n_ways;
    h.chunk_syms = packed.chunk_syms;
    h.sym_bytes = packed.sym_bytes;
    h.n_symbols = packed.n_symbols;
    h.n_chunks = packed.n_chunks;
    h.payload_bytes = packed.payload_bytes;
    uint64_t ck = fnv1a(0xcbf29ce484222325ull, &h, sizeof(h));
    ck = fnv1a(ck, norm_freqs, 4ull * packed.nsyms);
    ck = fnv1a(ck, lengths, 4ull * packed.n_chunks);
    h.checksum = ck;
    memcpy(out, &h, sizeof(h));
    memcpy(out + kHeaderBytes, norm_freqs, 4ull * packed.nsyms);
    if (packed.n_chunks)
        memcpy(out + kHeaderBytes + 4ull * packed.nsyms, lengths, 4ull * packed.n_chunks);
    if (out_bytes)
        *out_bytes = meta + packed.payload_bytes;
    return RANS_AMD_OK;
}

int rans_amd_container_parse(const void *src, uint64_t bytes, rans_amd_container_info *info, const uint32_t **freqs,
                             const uint32_t **lengths, const void **payload)
{
    if (!src || !info || (reinterpret_cast<uintptr_t>(src) & 3u) != 0) // (the tables are handed back as uint32_t pointers INTO src)
        return RANS_AMD_E_ARG;
    if (bytes < kHeaderBytes)
        return RANS_AMD_E_CORRUPT;
    Header h;
    memcpy(&h, src, sizeof(h));
    if (memcmp(h.magic, kMagic, 8) != 0 || h.version != kVersion)
        return RANS_AMD_E_CORRUPT;
    rans_amd_container_info i;
    i.format = h.format;
    i.scale_bits = h.scale_bits;
    i.nsyms = h.nsyms;
    i.n_ways = h.n_ways;
    i.chunk_syms = h.chunk_syms;
    i.sym_bytes = h.sym_bytes;
    i.n_symbols = h.n_symbols;
    i.n_chunks = h.n_chunks;
    i.payload_bytes = h.payload_bytes;
    if (!info_sane(&i))
        return RANS_AMD_E_CORRUPT;
    const uint64_t meta = meta_bytes(&i);
    if (meta > bytes || i.payload_bytes > bytes - meta)
        return RANS_AMD_E_CORRUPT;
    const uint8_t *p = static_cast<const uint8_t *>(src);
    const uint32_t *f = reinterpret_cast<const uint32_t *>(p + kHeaderBytes);
    const uint32_t *l = reinterpret_cast<const uint32_t *>(p + kHeaderBytes + 4ull * i.nsyms);
    const uint64_t stored = h.checksum;
    h.checksum = 0;
    uint64_t ck = fnv1a(0xcbf29ce484222325ull, &h, sizeof(h));
    ck = fnv1a(ck, f, 4ull * i.nsyms);
    ck = fnv1a(ck, l, 4ull * i.n_chunks);
    if (ck != stored)
        return RANS_AMD_E_CORRUPT;
    uint64_t at = 0;
    for (uint64_t c = 0; c < i.n_chunks; ++c)
        at = (c + 1 == i.n_chunks) ? at + l[c] : at + align16(l[c]);
    if (at != i.payload_bytes)
        return RANS_AMD_E_CORRUPT;
    *info = i;
    if (freqs)
        *freqs = f;
    if (lengths)
        *lengths = l;
    if (payload)
        *payload = p + meta;
    return RANS_AMD_OK;
}


/* ---- version 2: one model per chunk (rans_amd_encode_adaptive) ------------------------------
 *   [ 80-byte header (version 2, reserved = 1) | u16 chunk_freqs[n_chunks][256] | u32 lengths[n_chunks] | pad to 16 | payload ]
 * info->format is RANS_AMD_FMT_BYTE (scale_bits 8..12) or RANS_AMD_FMT_WORD (scale_bits 12), nsyms 256, sym_bytes 1; the header
 * carries the format; the checksum covers header, frequencies and lengths. */
static uint64_t meta_bytes_v2(const rans_amd_container_info *i)
{
    return align16(kHeaderBytes + 512ull * i->n_chunks + 4ull * i->n_chunks);
}

static bool info_sane_v2(const rans_amd_container_info *i)
{
    return info_sane(i) && i->nsyms == 256 && i->sym_bytes == 1 &&
           ((i->format == RANS_AMD_FMT_BYTE && i->scale_bits >= 8 && i->scale_bits <= 12) ||
            (i->format == RANS_AMD_FMT_WORD && i->scale_bits == 12));
}

uint64_t rans_amd_container_bytes_adaptive(const rans_amd_container_info *info)
{
    if (!info_sane_v2(info))
        return 0;
    return meta_bytes_v2(info) + info->payload_bytes;
}

int rans_amd_container_pack_adaptive(const rans_amd_container_info *info, const uint16_t *chunk_freqs,
                                     const uint32_t *lengths, const void *payload, void *dst, uint64_t cap,
                                     uint64_t *out_bytes)
{
    if (!info_sane_v2(info) || !dst || (info->n_chunks && (!chunk_freqs || !lengths || !payload)))
        return RANS_AMD_E_ARG;
    uint64_t at = 0;
    for (uint64_t c = 0; c < info->n_chunks; ++c)
        at = (c + 1 == info->n_chunks) ? at + lengths[c] : at + align16(lengths[c]);
    if (at != info->payload_bytes)
        return RANS_AMD_E_ARG;
    for (uint64_t c = 0; c < info->n_chunks; ++c) { // every chunk's model must be a model
        uint32_t sum = 0;
        for (int s = 0; s < 256; ++s)
            sum += chunk_freqs[c * 256 + s];
        if (sum != (1u << info->scale_bits))
            return RANS_AMD_E_MODEL;
    }
    const uint64_t meta = meta_bytes_v2(info), total = meta + info->payload_bytes;
    if (cap < total)
        return RANS_AMD_E_SPACE;
    uint8_t *out = static_cast<uint8_t *>(dst);
    memset(out, 0, (size_t)meta);
    Header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8);
    h.version = 2;
    h.reserved = 1; // model mode: per-chunk u16 frequencies
    h.format = info->format;
    h.scale_bits = info->scale_bits;
    h.nsyms = info->nsyms;
    h.n_ways = info->n_ways;
    h.chunk_syms = info->chunk_syms;
    h.sym_bytes = info->sym_bytes;
    h.n_symbols = info->n_symbols;
    h.n_chunks = info->n_chunks;
    h.payload_bytes = info->payload_bytes;
    uint64_t ck = fnv1a(0xcbf29ce484222325ull, &h, sizeof(h));
    ck = fnv1a(ck, chunk_freqs, 512ull * info->n_chunks);
    ck = fnv1a(ck, lengths, 4ull * info->n_chunks);
    h.checksum = ck;
    memcpy(out, &h, sizeof(h));
    if (info->n_chunks) {
        memcpy(out + kHeaderBytes, chunk_freqs, 512ull * info->n_chunks);
        memcpy(out + kHeaderBytes + 512ull * info->n_chunks, lengths, 4ull * info->n_chunks);
    }
    if (info->payload_bytes)
        memcpy(out + meta, payload, (size_t)info->payload_bytes);
    if (out_bytes)
        *out_bytes = total;
    return RANS_AMD_OK;
}

int rans_amd_container_pack_indexed_adaptive(const rans_amd_container_info *info, const uint16_t *chunk_freqs, const uint64_t *offsets,
                                             const uint32_t *lengths, const void *payload, uint64_t payload_bytes, void *dst,
                                             uint64_t cap, uint64_t *out_bytes)
{
    if (!info_sane_v2(info) || !dst || (info->n_chunks && (!chunk_freqs || !offsets || !lengths || !payload)))
        return RANS_AMD_E_ARG;
    rans_amd_container_info packed = *info;
    packed.payload_bytes = rans_amd_packed_payload_bytes(lengths, info->n_chunks);
    const uint64_t meta = meta_bytes_v2(&packed);
    if (cap < meta + packed.payload_bytes)
        return RANS_AMD_E_SPACE;
    for (uint64_t c = 0; c < info->n_chunks; ++c) { // every chunk's model must be a model
        uint32_t sum = 0;
        for (int s = 0; s < 256; ++s)
            sum += chunk_freqs[c * 256 + s];
        if (sum != (1u << info->scale_bits))
            return RANS_AMD_E_MODEL;
    }
    uint8_t *out = static_cast<uint8_t *>(dst);
    if (int rc = copy_chunks_indexed(offsets, lengths, info->n_chunks, payload, payload_bytes, out + meta))
        return rc;
    memset(out, 0, (size_t)meta);
    Header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8);
    h.version = 2;
    h.reserved = 1; // model mode: per-chunk u16 frequencies
    h.format = packed.format;
    h.scale_bits = packed.scale_bits;
    h.nsyms = packed.nsyms;
    h.n_ways = packed.n_ways;
    h.chunk_syms = packed.chunk_syms;
    h.sym_bytes = packed.sym_bytes;
    h.n_symbols = packed.n_symbols;
    h.n_chunks = packed.n_chunks;
    h.payload_bytes = packed.payload_bytes;
    uint64_t ck = fnv1a(0xcbf29ce484222325ull, &h, sizeof(h));
    ck = fnv1a(ck, chunk_freqs, 512ull * packed.n_chunks);
    ck = fnv1a(ck, lengths, 4ull * packed.n_chunks);
    h.checksum = ck;
    memcpy(out, &h, sizeof(h));
    if (packed.n_chunks) {
        memcpy(out + kHeaderBytes, chunk_freqs, 512ull * packed.n_chunks);
        memcpy(out + kHeaderBytes + 512ull * packed.n_chunks, lengths, 4ull * packed.n_chunks);
    }
    if (out_bytes)
        *out_bytes = meta + packed.payload_bytes;
    return RANS_AMD_OK;
}

int rans_amd_container_parse_adaptive(const void *src, uint64_t bytes, rans_amd_container_info *info,
                                      const uint16_t **chunk_freqs, const uint32_t **lengths, const void **payload)
{
    if (!src || !info || (reinterpret_cast<uintptr_t>(src) & 3u) != 0)
        return RANS_AMD_E_ARG;
    if (bytes < kHeaderBytes)
        return RANS_AMD_E_CORRUPT;
    Header h;
    memcpy(&h, src, sizeof(h));
    if (memcmp(h.magic, kMagic, 8) != 0 || h.version != 2 || h.reserved != 1)
        return RANS_AMD_E_CORRUPT;
    rans_amd_container_info i;
    i.format = h.format;
    i.scale_bits = h.scale_bits;
    i.nsyms = h.nsyms;
    i.n_ways = h.n_ways;
    i.chunk_syms = h.chunk_syms;
    i.sym_bytes = h.sym_bytes;
    i.n_symbols = h.n_symbols;
    i.n_chunks = h.n_chunks;
    i.payload_bytes = h.payload_bytes;
    if (!info_sane_v2(&i))
        return RANS_AMD_E_CORRUPT;
    const uint64_t meta = meta_bytes_v2(&i);
    if (meta > bytes || i.payload_bytes > bytes - meta)
        return RANS_AMD_E_CORRUPT;
    const uint8_t *p = static_cast<const uint8_t *>(src);
    const uint16_t *f = reinterpret_cast<const uint16_t *>(p + kHeaderBytes);
    const uint32_t *l = reinterpret_cast<const uint32_t *>(p + kHeaderBytes + 512ull * i.n_chunks);
    const uint64_t stored = h.checksum;
    h.checksum = 0;
    uint64_t ck = fnv1a(0xcbf29ce484222325ull, &h, sizeof(h));
    ck = fnv1a(ck, f, 512ull * i.n_chunks);
    ck = fnv1a(ck, l, 4ull * i.n_chunks);
    if (ck != stored)
        return RANS_AMD_E_CORRUPT;
    uint64_t at = 0;
    for (uint64_t c = 0; c < i.n_chunks; ++c)
        at = (c + 1 == i.n_chunks) ? at + l[c] : at + align16(l[c]);
    if (at != i.payload_bytes)
        return RANS_AMD_E_CORRUPT;
    *info = i;
    if (chunk_freqs)
        *chunk_freqs = f;
    if (lengths)
        *lengths = l;
    if (payload)
        *payload = p + meta;
    return RANS_AMD_OK;
}

} // extern "C"
