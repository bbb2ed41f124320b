// api.cpp -- the extern "C" boundary declared in include/ryg_rans_amd.h.
//
// Host-side plumbing only: argument checking, device memory for tables and
// workspaces, kernel launches (the .hip files).  There is no CPU implementation of
// encode/decode behind these entry points: without a usable GPU every call
// fails with RANS_AMD_E_HIP.
#include "../../include/ryg_rans_amd.h"

#include <cstdio>
#include <cstring>
#include <cmath>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "kernels.h"
#include "launchers.hpp"
#include "model.h"

using namespace rans_amd;

namespace {

thread_local std::string g_last_error;

int fail(int status, const char *what)
{
    g_last_error = what ? what : "";
    return status;
}

int hip_fail(hipError_t e, const char *where)
{
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    return RANS_AMD_E_HIP;
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess)                          \
            return hip_fail(e__, #expr);                \
    } while (0)

// Stream capture (hipStreamBeginCapture ... EndCapture around calls of this library: the launches become nodes of a
// hipGraph).  What a captured call may not do is what a graph cannot replay: allocate, wait for the stream, read
// anything back.  The entry points find out once per call (CaptureScope) and the workspaces refuse to grow meanwhile
// -- run the call once outside the capture first, the workspaces are kept.
thread_local bool t_capturing = false;
struct CaptureScope {
    bool prev;
    bool active = false;
    explicit CaptureScope(hipStream_t s) : prev(t_capturing)
    {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &st) == hipSuccess) // (the legacy null stream cannot be captured)
            active = st == hipStreamCaptureStatusActive;
        else
            (void)hipGetLastError();
        t_capturing = active;
    }
    ~CaptureScope() { t_capturing = prev; }
};

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t want)
    {
        if (want <= bytes)
            return RANS_AMD_OK;
        if (t_capturing)
            return fail(RANS_AMD_E_ARG, "a workspace would have to grow while the stream is capturing: make the same call once outside the capture first");
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
        size_t padded = (want + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&ptr, padded);
        if (e != hipSuccess) {
            ptr = nullptr;
            return hip_fail(e, "hipMalloc(workspace)");
        }
        bytes = padded;
        return RANS_AMD_OK;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

} // namespace

struct rans_amd_ctx {
    int device = 0;
    int num_cus = 0;
    // device words: [0] decode error counter (u64), [8] encode flags (u32), [12] histogram flags (u32), [16] flags of
    // rans_amd_container_compact (u32: a compaction behind an asynchronous encode must not wipe that encode's verdict)
    uint8_t *d_words = nullptr;
    DeviceBuffer scratch;   // encode slots
    DeviceBuffer lengths;   // encode lengths when the caller passes none
    DeviceBuffer hist;
    DeviceBuffer layout_sums; // per-block totals of the offset scan (many-chunk containers)
    DeviceBuffer enc_status;  // fused encoder: look-back word per chunk + the claim counters (EncParams::status)
    DeviceBuffer enc_mailboxes; // fused encoder whose tables fill the LDS: one mailbox per block (EncParams::mailbox_global)
    DeviceBuffer wave_scratch; // one 64-byte line per resident decoder wave (DecParams::wave_scratch)
    DeviceBuffer adapt_rcp;    // reciprocals by frequency for the fused per-chunk-model encoder (AdaptEncParams::rcp); built once, kept
    DeviceBuffer host_in, host_out, host_idx; // staging of the *_host wrappers, kept between calls (under host_mu)
    std::mutex host_mu;
    DeviceBuffer trace;       // per-wave clock records (rans_amd_set_timing(ctx, 2) / RANS_AMD_TRACE)
    rans_amd_wave_clocks wave_clocks = {0, 0, 0, 0.0, 0.0};
    bool wave_clocks_on = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; // dec start/stop, enc start/stop
    bool timing = false;
    bool dec_timed = false, enc_timed = false;
    uint32_t launch_seq = 0; // selects one of kWorkSlots chunk counters at d_words + 256
    uint32_t capture_seq = 0; // ... and one of the kCaptureSlots behind them for a launch that is being captured into a graph
    uint32_t variant = 0;    // kVar* bits (rans_amd_ctx_set_option)
    bool unfused = false;    // RANS_AMD_OPT_FUSED_PLACEMENT = 0: k_encode + k_layout + k_compact
    bool scratch_ring = false; // RANS_AMD_OPT_ENC_SCRATCH_RING = 1
    const char *last_kernel = "";
    const char *last_enc_kernel = ""; // the coding kernel of the last encode call
    bool last_enc_fused = false;      // ... and whether it placed its chunks itself (no k_layout / k_compact)
    bool last_enc_slots = false;      // ... or left them in their slots (rans_amd_encode_slots)
    std::mutex mu;

    unsigned long long *d_err() { return reinterpret_cast<unsigned long long *>(d_words); }
    uint32_t *d_enc_flags() { return reinterpret_cast<uint32_t *>(d_words + 8); }
    uint32_t *d_hist_flags() { return reinterpret_cast<uint32_t *>(d_words + 12); }
    uint32_t *d_compact_flags() { return reinterpret_cast<uint32_t *>(d_words + 16); }
};

struct rans_amd_model {
    rans_amd_ctx *ctx = nullptr; // identity only (encode/decode check it); never dereferenced by the model
    int device = -1;             // device that owns the d_* buffers (the context may be gone at destroy time)
    HostModel host;
    void *d_table0 = nullptr;
    void *d_table1 = nullptr;
    void *d_enc = nullptr;
    void *d_word_enc = nullptr;
    void *d_remap = nullptr;
    void *d_alias_recs8 = nullptr, *d_alias_remap16 = nullptr; // alias encoder's LDS tables, when they fit
    void *d_dual0 = nullptr, *d_dual1 = nullptr; // alias: the tables of the two-chunks-per-wave decoder (FMT_ALIAS2)
    void *d_packed = nullptr;                    // rans64: 4-byte slot records (HostModel::r64_packed), when the model has them
    void *d_fused = nullptr;                     // byte format: 8-byte slot records (HostModel::byte_slots), scale_bits <= 13
    uint32_t table0_bytes = 0, table1_bytes = 0, dual0_bytes = 0, dual1_bytes = 0;
};

namespace {

// the encoders' device flags (EncParams::flags) as a status
int encode_flags_status(uint32_t flags)
{
    if (flags & 1u) {
        char msg[160];
        snprintf(msg, sizeof msg, "encode: input holds a symbol with frequency 0 (flags 0x%x)", flags);
        return fail(RANS_AMD_E_MODEL, msg);
    }
    if (flags & 2u)
        return fail(RANS_AMD_E_SPACE, "encode: container does not fit out_cap");
    if (flags & 4u) // (a kernel that addresses its LDS tables by raw offsets found them elsewhere: never code on that)
        return fail(RANS_AMD_E_HIP, "encode: internal error (dynamic LDS does not start at offset 0)");
    if (flags & 512u) // rans_amd_container_compact: an index entry (offset, length) does not lie inside the source buffer
        return fail(RANS_AMD_E_CORRUPT, "container_compact: a chunk of the source index lies outside [0, src_bytes)");
    if (flags & ~7u) { // a wait of the fused placement gave up (device_common.hpp SpinWatch; 256: a coder waiting for its scratch slot)
        char msg[160];
        snprintf(msg, sizeof msg, "encode: internal error (placement protocol timed out, flags 0x%x)", flags);
        return fail(RANS_AMD_E_HIP, msg);
    }
    return RANS_AMD_OK;
}

// Words a launch expects to find zero, cleared by one k_zero launch per three regions (kernels.h ZeroParams says why
// this is a kernel and not hipMemsetAsync).
struct ZeroList {
    ZeroParams p{};
    int count = 0;
    hipStream_t stream;
    explicit ZeroList(hipStream_t s) : stream(s) {}
    hipError_t add(void *ptr, uint64_t bytes)
    {
        if (bytes == 0)
            return hipSuccess;
        if (count == 3) {
            const hipError_t e = flush();
            if (e != hipSuccess)
                return e;
        }
        p.ptr[count] = ptr;
        p.bytes[count++] = bytes;
        return hipSuccess;
    }
    hipError_t flush()
    {
        const hipError_t e = count ? launch_zero(p, stream) : hipSuccess;
        p = ZeroParams{};
        count = 0;
        return e;
    }
};

// counter slot of a launch that is being captured (kernels.h kCaptureSlots: that many captured launches of one context
// may run at the same time)
unsigned int *capture_counters(rans_amd_ctx *ctx)
{
    unsigned int *ring = reinterpret_cast<unsigned int *>(ctx->d_words + 256);
    return ring + (size_t)(kWorkSlots + ctx->capture_seq++ % kCaptureSlots) * kWorkSlotWords;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        if (prev != dev)
            ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

int upload(const void *src, size_t bytes, void **d_out)
{
    *d_out = nullptr;
    if (bytes == 0)
        return RANS_AMD_OK;
    const size_t padded = (bytes + 255) & ~size_t(255);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, padded));
    hipError_t e = hipMemset(p, 0, padded);
    if (e == hipSuccess)
        e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return hip_fail(e, "hipMemcpy(table)");
    }
    *d_out = p;
    return RANS_AMD_OK;
}

// The stream of an empty input: n_ways states, each the coder's initial value L, little endian (lane 0 first,
// all equal).  Returns its size; writes it when dst != NULL.
uint64_t empty_stream(int format, uint32_t n_ways, uint8_t *dst)
{
    const bool r64 = format == RANS_AMD_FMT_R64;
    const uint64_t L = r64 ? (1ull << 31) : format == RANS_AMD_FMT_WORD ? (1ull << 16) : (1ull << 23);
    const uint32_t sbytes = r64 ? 8u : 4u;
    if (dst)
        for (uint32_t l = 0; l < n_ways; ++l)
            for (uint32_t b = 0; b < sbytes; ++b)
                dst[(size_t)l * sbytes + b] = (uint8_t)(L >> (8 * b));
    return (uint64_t)n_ways * sbytes;
}

uint32_t state_bytes(int format) { return format == RANS_AMD_FMT_R64 ? 8u : 4u; }

} // namespace

extern "C" {

int rans_amd_version(void) { return RANS_AMD_VERSION; }

unsigned rans_amd_build_flags(void) { return kMeasureBuild ? RANS_AMD_BUILD_MEASURE : 0u; }

const char *rans_amd_status_string(int status)
{
    switch (status) {
    case RANS_AMD_OK: return "ok";
    case RANS_AMD_E_ARG: return "invalid argument";
    case RANS_AMD_E_MODEL: return "invalid frequency model";
    case RANS_AMD_E_SPACE: return "output buffer too small";
    case RANS_AMD_E_CORRUPT: return "corrupt stream";
    case RANS_AMD_E_UNSUPPORTED: return "unsupported configuration";
    case RANS_AMD_E_HIP: return "HIP runtime error";
    case RANS_AMD_E_NOMEM: return "out of memory";
    default: return "unknown status";
    }
}

const char *rans_amd_last_error(void) { return g_last_error.c_str(); }

int rans_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int rans_amd_ctx_create(int device, rans_amd_ctx **out_ctx)
{
    if (!out_ctx)
        return fail(RANS_AMD_E_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RANS_AMD_E_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count)
        return fail(RANS_AMD_E_ARG, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(RANS_AMD_E_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    rans_amd_ctx *ctx = new (std::nothrow) rans_amd_ctx;
    if (!ctx)
        return fail(RANS_AMD_E_NOMEM, "ctx");
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t words_bytes = 256 + (size_t)(kWorkSlots + kCaptureSlots) * kWorkSlotWords * 4;
    e = hipMalloc(reinterpret_cast<void **>(&ctx->d_words), words_bytes);
    if (e == hipSuccess)
        e = hipMemset(ctx->d_words, 0, words_bytes);
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        e = hipEventCreate(&ctx->ev[i]);
    if (e != hipSuccess) {
        rans_amd_ctx_destroy(ctx);
        return hip_fail(e, "ctx_create");
    }
    *out_ctx = ctx;
    return RANS_AMD_OK;
}

int rans_amd_ctx_destroy(rans_amd_ctx *ctx)
{
    if (!ctx)
        return RANS_AMD_OK;
    DeviceGuard guard(ctx->device);
    ctx->scratch.release();
    ctx->lengths.release();
    ctx->hist.release();
    ctx->layout_sums.release();
    ctx->enc_status.release();
    ctx->enc_mailboxes.release();
    ctx->host_in.release();
    ctx->host_out.release();
    ctx->host_idx.release();
    ctx->trace.release();
    ctx->wave_scratch.release();
    ctx->adapt_rcp.release();
    if (ctx->d_words)
        (void)hipFree(ctx->d_words);
    for (int i = 0; i < 4; ++i)
        if (ctx->ev[i])
            (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
    return RANS_AMD_OK;
}

int rans_amd_ctx_trim(rans_amd_ctx *ctx)
{
    if (!ctx)
        return fail(RANS_AMD_E_ARG, "ctx is NULL");
    DeviceGuard guard(ctx->device);
    // host_mu before mu: the order the *_host wrappers take them in (they hold host_mu across the nested call)
    std::lock_guard<std::mutex> host_lock(ctx->host_mu);
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->scratch.release();
    ctx->lengths.release();
    ctx->hist.release();
    ctx->layout_sums.release();
    ctx->enc_status.release();
    ctx->enc_mailboxes.release();
    ctx->host_in.release();
    ctx->host_out.release();
    ctx->host_idx.release();
    ctx->trace.release();
    return RANS_AMD_OK;
}


int rans_amd_ctx_set_option(rans_amd_ctx *ctx, int option, int value)
{
    if (!ctx)
        return fail(RANS_AMD_E_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    switch (option) {
    case RANS_AMD_OPT_LANE_KERNELS: // (0.6.0: no choice any more -- the first generation is the fallback for the shapes only it serves)
        if (value != 0)
            return fail(RANS_AMD_E_UNSUPPORTED, "set_option: RANS_AMD_OPT_LANE_KERNELS was retired in 0.6.0 (only 0 = automatic is accepted)");
        return RANS_AMD_OK;
    case RANS_AMD_OPT_LANE_FUSED_PLACEMENT:
        ctx->variant = value ? (ctx->variant | kVarLanesFused) : (ctx->variant & ~kVarLanesFused);
        return RANS_AMD_OK;
    case RANS_AMD_OPT_FUSED_PLACEMENT:
        ctx->unfused = value == 0;
        return RANS_AMD_OK;
    case RANS_AMD_OPT_ENC_SCRATCH_RING:
        ctx->scratch_ring = value != 0;
        return RANS_AMD_OK;
    case RANS_AMD_OPT_DUAL_DECODE:
        if (value < 0 || value > 2)
            return fail(RANS_AMD_E_ARG, "set_option: RANS_AMD_OPT_DUAL_DECODE takes 0 (never), 1 (automatic) or 2 (whenever the tables fit)");
        ctx->variant &= ~(kVarNoDual | kVarDualAlways);
        ctx->variant |= value == 0 ? kVarNoDual : value == 2 ? kVarDualAlways : 0u;
        return RANS_AMD_OK;
    default:
        return fail(RANS_AMD_E_ARG, "set_option: unknown option");
    }
}

/* ---- model ------------------------------------------------------------ */

// The model builder's counters are 32 bits wide, as the reference's (SymbolStats: uint32_t freqs[], cum_freqs[], main.cpp:49-57;
// normalize_freqs sums them in 32 bits as well): a histogram of 2^32 symbols and more is refused, not wrapped.  The coders
// take any n (uint64_t); such an input needs its model from a part of it -- or per-chunk models.
static const char *kCountTooMany = "count_freqs: 2^32 symbols and more do not fit the model builder's 32-bit counters (SymbolStats, "
                                   "main.cpp:49-57): count a part of the input, or use per-chunk models";

int rans_amd_count_freqs_host(const void *syms, uint64_t n, int sym_bytes, uint32_t nsyms, uint32_t *freqs)
{
    if (n > 0xffffffffull)
        return fail(RANS_AMD_E_UNSUPPORTED, kCountTooMany);
    int rc = count_freqs_host(syms, n, sym_bytes, nsyms, freqs);
    return rc ? fail(rc, "count_freqs_host: bad argument or symbol outside the alphabet") : rc;
}

int rans_amd_count_freqs(rans_amd_ctx *ctx, const void *d_syms, uint64_t n, int sym_bytes, uint32_t nsyms,
                         uint32_t *freqs, void *stream)
{
    if (!ctx || !freqs || (n && !d_syms) || (sym_bytes != 1 && sym_bytes != 2) || nsyms == 0 || nsyms > 16384)
        return fail(RANS_AMD_E_ARG, "count_freqs: bad argument");
    if (n > 0xffffffffull)
        return fail(RANS_AMD_E_UNSUPPORTED, kCountTooMany);
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = ctx->hist.reserve((size_t)nsyms * 4);
    if (rc)
        return rc;
    {
        ZeroList zero(s);
        HIP_TRY(zero.add(ctx->hist.ptr, (uint64_t)nsyms * 4));
        HIP_TRY(zero.add(ctx->d_hist_flags(), 4));
        HIP_TRY(zero.flush());
    }
    if (n)
        HIP_TRY(launch_histogram(d_syms, n, sym_bytes, nsyms, static_cast<uint32_t *>(ctx->hist.ptr),
                                 ctx->d_hist_flags(), ctx->num_cus, s));
    uint32_t flags = 0;
    HIP_TRY(hipMemcpyAsync(freqs, ctx->hist.ptr, (size_t)nsyms * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&flags, ctx->d_hist_flags(), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (flags & 1u)
        return fail(RANS_AMD_E_ARG, "count_freqs: symbol outside the alphabet");
    return RANS_AMD_OK;
}

int rans_amd_normalize_freqs(uint32_t *freqs, uint32_t *cum_freqs, uint32_t nsyms, uint32_t target_total)
{
    int rc = normalize_freqs(freqs, cum_freqs, nsyms, target_total);
    return rc ? fail(rc, "normalize_freqs failed") : rc;
}

int rans_amd_model_create(rans_amd_ctx *ctx, int format, const uint32_t *norm_freqs, uint32_t nsyms,
                          uint32_t scale_bits, rans_amd_model **out_model)
{
    if (!out_model)
        return fail(RANS_AMD_E_ARG, "model_create: NULL argument");
    *out_model = nullptr;
    rans_amd_model *m = new (std::nothrow) rans_amd_model;
    if (!m)
        return fail(RANS_AMD_E_NOMEM, "model");
    m->ctx = ctx;
    m->device = ctx ? ctx->device : -1;
    int rc = m->host.build(format, norm_freqs, nsyms, scale_bits);
    if (rc) {
        delete m;
        return fail(rc, "model_create: frequencies rejected");
    }
    if (!ctx) { // host-only model: tables can be inspected, encode/decode refuse it
        *out_model = m;
        return RANS_AMD_OK;
    }
    if ((format == RANS_AMD_FMT_BYTE || format == RANS_AMD_FMT_ALIAS) && scale_bits < 8) {
        // the byte-format kernels multiply freq * (x >> scale_bits) in 24 bits (x < 2^31)
        delete m;
        return fail(RANS_AMD_E_UNSUPPORTED, "model_create: the GPU byte/alias coders need scale_bits >= 8");
    }
    DeviceGuard guard(ctx->device);
    const HostModel &h = m->host;
    switch (format) {
    case RANS_AMD_FMT_WORD:
        m->table0_bytes = (uint32_t)(h.word_slots.size() * sizeof(WordSlot));
        rc = upload(h.word_slots.data(), m->table0_bytes, &m->d_table0);
        break;
    case RANS_AMD_FMT_BYTE:
    case RANS_AMD_FMT_R64: {
        if (h.r64_search) { // table0 = padded cum[], table1 = {freq, start} per symbol
            if (h.sym_bytes != 1) {
                rc = fail(RANS_AMD_E_UNSUPPORTED, "rans64 decoders support alphabets up to 256 symbols");
                break;
            }
            m->table0_bytes = (uint32_t)(h.cum_padded.size() * 4);
            rc = upload(h.cum_padded.data(), m->table0_bytes, &m->d_table0);
            if (rc == RANS_AMD_OK) {
                m->table1_bytes = (uint32_t)(h.sym_recs.size() * sizeof(SymRec));
                rc = upload(h.sym_recs.data(), m->table1_bytes, &m->d_table1);
            }
            break;
        }
        if (h.sym_bytes != 1) { // u16 cum2sym does not fit beside the stream windows; use FMT_ALIAS
            rc = fail(RANS_AMD_E_UNSUPPORTED, "cum2sym decoders support alphabets up to 256 symbols");
            break;
        }
        std::vector<uint8_t> c2s(h.cum2sym.size());
        for (size_t i = 0; i < c2s.size(); ++i)
            c2s[i] = (uint8_t)h.cum2sym[i];
        m->table0_bytes = (uint32_t)c2s.size();
        rc = upload(c2s.data(), c2s.size(), &m->d_table0);
        if (rc == RANS_AMD_OK) {
            m->table1_bytes = (uint32_t)(h.sym_recs.size() * sizeof(SymRec));
            rc = upload(h.sym_recs.data(), m->table1_bytes, &m->d_table1);
        }
        if (rc == RANS_AMD_OK && !h.r64_packed.empty())
            rc = upload(h.r64_packed.data(), h.r64_packed.size() * 4, &m->d_packed);
        if (rc == RANS_AMD_OK && !h.byte_slots.empty())
            rc = upload(h.byte_slots.data(), h.byte_slots.size() * sizeof(WordSlot), &m->d_fused);
        break;
    }
    case RANS_AMD_FMT_ALIAS:
        // the device packs a half bucket's frequency into 16 bits: 65536 (one symbol owning the whole 16-bit
        // range) does not fit; such a model exists as a host-only model, or at scale_bits <= 15
        for (uint32_t f : h.slot_freqs)
            if (f > 0xffffu)
                rc = fail(RANS_AMD_E_UNSUPPORTED, "model_create: alias model with a 65536-wide symbol (use scale_bits <= 15)");
        if (rc)
            break;
        m->table0_bytes = (uint32_t)(h.alias_halves.size() * sizeof(AliasHalf));
        rc = upload(h.alias_halves.data(), m->table0_bytes, &m->d_table0);
        if (rc == RANS_AMD_OK) {
            m->table1_bytes = (uint32_t)(h.divider.size() * 4);
            rc = upload(h.divider.data(), m->table1_bytes, &m->d_table1);
        }
        if (rc == RANS_AMD_OK)
            rc = upload(h.alias_remap.data(), h.alias_remap.size() * 4, &m->d_remap);
        if (rc == RANS_AMD_OK && !h.alias2_halves.empty()) {
            m->dual0_bytes = (uint32_t)(h.alias2_halves.size() * sizeof(AliasHalf));
            m->dual1_bytes = (uint32_t)h.alias2_own.size();
            rc = upload(h.alias2_halves.data(), m->dual0_bytes, &m->d_dual0);
            if (rc == RANS_AMD_OK)
                rc = upload(h.alias2_own.data(), m->dual1_bytes, &m->d_dual1);
        }
        if (rc == RANS_AMD_OK && !h.alias_remap16.empty()) {
            rc = upload(h.alias_recs8.data(), h.alias_recs8.size() * 8, &m->d_alias_recs8);
            if (rc == RANS_AMD_OK)
                rc = upload(h.alias_remap16.data(), h.alias_remap16.size() * 2, &m->d_alias_remap16);
        }
        break;
    default:
        rc = RANS_AMD_E_ARG;
    }
    if (rc == RANS_AMD_OK)
        rc = upload(h.enc_recs.data(), h.enc_recs.size() * sizeof(EncRec), &m->d_enc);
    if (rc == RANS_AMD_OK && !h.word_enc_recs.empty())
        rc = upload(h.word_enc_recs.data(), h.word_enc_recs.size() * sizeof(WordEncRec), &m->d_word_enc);
    if (rc == RANS_AMD_OK) {
        const size_t lds = (size_t)((m->table0_bytes + 15u) & ~15u) + ((m->table1_bytes + 15u) & ~15u) +
                           (size_t)(kDecBlockThreads / 64) * kRingStride;
        if (lds > 160 * 1024)
            rc = fail(RANS_AMD_E_UNSUPPORTED, "decode tables do not fit in LDS");
    }
    if (rc) {
        rans_amd_model_destroy(m);
        return rc;
    }
    *out_model = m;
    return RANS_AMD_OK;
}

int rans_amd_model_destroy(rans_amd_model *m)
{
    if (!m)
        return RANS_AMD_OK;
    if (m->device >= 0) { // not m->ctx->device: a model may outlive its context
        DeviceGuard guard(m->device);
        for (void *p : {m->d_table0, m->d_table1, m->d_enc, m->d_word_enc, m->d_remap, m->d_alias_recs8, m->d_alias_remap16,
                        m->d_dual0, m->d_dual1, m->d_packed, m->d_fused})
            if (p)
                (void)hipFree(p);
    }
    delete m;
    return RANS_AMD_OK;
}

int rans_amd_model_format(const rans_amd_model *m) { return m ? m->host.format : -1; }
uint32_t rans_amd_model_scale_bits(const rans_amd_model *m) { return m ? m->host.scale_bits : 0; }
uint32_t rans_amd_model_nsyms(const rans_amd_model *m) { return m ? m->host.nsyms : 0; }
int rans_amd_model_sym_bytes(const rans_amd_model *m) { return m ? m->host.sym_bytes : 0; }

int rans_amd_model_table(const rans_amd_model *m, int which, void *dst, size_t cap, size_t *size)
{
    if (!m || !size)
        return fail(RANS_AMD_E_ARG, "model_table: NULL argument");
    std::vector<uint8_t> img;
    int rc = m->host.export_table(which, img);
    if (rc)
        return fail(rc, "model_table: no such table for this format");
    *size = img.size();
    if (!dst)
        return RANS_AMD_OK;
    if (cap < img.size())
        return fail(RANS_AMD_E_SPACE, "model_table: buffer too small");
    memcpy(dst, img.data(), img.size());
    return RANS_AMD_OK;
}

/* ---- layout ------------------------------------------------------------ */

uint64_t rans_amd_num_chunks(uint64_t n, uint32_t chunk_syms)
{
    if (chunk_syms == 0)
        return 0;
    return (n + chunk_syms - 1) / chunk_syms;
}

uint64_t rans_amd_chunk_bound(int format, uint32_t chunk_syms, uint32_t n_ways)
{
    // units per symbol: <= 2 bytes (byte/alias, scale_bits <= 16), 1 word, 1 dword
    const uint64_t per_sym = format == RANS_AMD_FMT_R64 ? 4 : 2;
    const uint64_t b = (uint64_t)chunk_syms * per_sym + (uint64_t)n_ways * state_bytes(format);
    return (b + 15) & ~uint64_t(15);
}

uint64_t rans_amd_encode_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms)
{
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    if (nchunks == 0)
        return 16;
    const uint64_t last = n - (nchunks - 1) * chunk_syms;
    return (nchunks - 1) * rans_amd_chunk_bound(format, chunk_syms, n_ways) +
           rans_amd_chunk_bound(format, (uint32_t)last, n_ways);
}

int rans_amd_ways_supported(int format, uint32_t n_ways) { return ways_supported(format, n_ways) ? 1 : 0; }

// scratch slot of one chunk: worst-case stream, a whole number of 64-byte lines (the staged lane
// encoder flushes its output ring line by line)
static uint64_t encode_slot_bytes(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms)
{
    const uint64_t b = rans_amd_chunk_bound(format, (uint32_t)(n < chunk_syms ? n : chunk_syms), n_ways);
    return (b + 63) & ~uint64_t(63);
}

uint64_t rans_amd_encode_workspace_bytes(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms)
{
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    const uint64_t slot = encode_slot_bytes(format, n, n_ways, chunk_syms);
    return nchunks * slot + 64;
}

int rans_amd_build_model_o0(rans_amd_ctx *ctx, int format, const void *syms, uint64_t n, int syms_on_device,
                            uint32_t nsyms, uint32_t scale_bits, uint32_t *norm_freqs_out,
                            rans_amd_model **out_model, void *stream)
{
    if (!out_model || nsyms == 0 || nsyms > 65536 || scale_bits == 0 || scale_bits > 31)
        return fail(RANS_AMD_E_ARG, "build_model_o0: bad argument");
    std::vector<uint32_t> freqs(nsyms), cum(nsyms + 1);
    const int sym_bytes = nsyms <= 256 ? 1 : 2;
    int rc = syms_on_device ? rans_amd_count_freqs(ctx, syms, n, sym_bytes, nsyms, freqs.data(), stream)
                            : rans_amd_count_freqs_host(syms, n, sym_bytes, nsyms, freqs.data());
    if (rc == RANS_AMD_OK)
        rc = rans_amd_normalize_freqs(freqs.data(), cum.data(), nsyms, 1u << scale_bits);
    if (rc == RANS_AMD_OK)
        rc = rans_amd_model_create(ctx, format, freqs.data(), nsyms, scale_bits, out_model);
    if (rc == RANS_AMD_OK && norm_freqs_out)
        memcpy(norm_freqs_out, freqs.data(), sizeof(uint32_t) * nsyms);
    return rc;
}

/* ---- encode ------------------------------------------------------------- */

} // extern "C"

// rans_amd_encode (slots == false: the compact layout) and rans_amd_encode_slots (slots == true: every chunk stays in the
// slot it was coded into)
// sized_slot != 0: rans_amd_encode_slots_sized (slots == true; the slot is the caller's, overflowed chunks are coded again
// into worst-case slots behind the sized ones)
static int encode_impl(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n, uint32_t n_ways,
                       uint32_t chunk_syms, void *d_out, uint64_t out_cap, uint64_t *d_offsets, uint32_t *d_lengths,
                       uint64_t *h_total_bytes, void *stream, const bool slots, const uint64_t sized_slot = 0)
{
    if (!ctx || !model || !d_out || !d_offsets || !d_lengths || (n && !d_syms) || chunk_syms == 0)
        return fail(RANS_AMD_E_ARG, "encode: NULL argument or chunk_syms == 0");
    if (model->ctx != ctx)
        return fail(RANS_AMD_E_ARG, "encode: model belongs to another context");
    const int format = model->host.format;
    if (!ways_supported(format, n_ways))
        return fail(RANS_AMD_E_UNSUPPORTED, "encode: n_ways must be in 1..512");
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0)
        return fail(RANS_AMD_E_ARG, "encode: d_out must be 16-byte aligned");
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CaptureScope capture(s);
    if (capture.active && h_total_bytes)
        return fail(RANS_AMD_E_ARG, "encode: h_total_bytes must be NULL while the stream is capturing (d_offsets[n_chunks] holds the total)");

    const uint64_t worst_slot = encode_slot_bytes(format, n, n_ways, chunk_syms);
    if (worst_slot > 0xfffffff0ull) // chunk stream lengths and in-slot cursors are 32-bit
        return fail(RANS_AMD_E_UNSUPPORTED, "encode: chunk_syms too large (a chunk's stream must stay below 4 GiB)");
    // sized slots: the caller's slot, as long as it is smaller than the worst case (else: plain rans_amd_encode_slots)
    const bool sized = slots && sized_slot != 0 && sized_slot < worst_slot && nchunks > 0 && nchunks < (1ull << 32);
    const uint64_t slot = sized ? sized_slot : worst_slot;
    if (slots && (nchunks > (~0ull) / slot || out_cap < nchunks * slot)) // (known up front: nothing is launched)
        return fail(RANS_AMD_E_SPACE, sized ? "encode_slots_sized: out_cap does not hold n_chunks * slot_bytes"
                                            : "encode_slots: out_cap is below rans_amd_encode_slots_bound()");
    int rc = RANS_AMD_OK;
    ZeroList zero(s); // (everything the kernels below expect to find zero: one launch)
    HIP_TRY(zero.add(ctx->d_enc_flags(), 8)); // encode flags and histogram flags (unused by an encode).  NOT the compaction's word:
                                             // a compaction's verdict stays until it has been reported (rans_amd_encode_status, or the
                                             // compaction's own synchronous return) -- an encode queued behind an asynchronous
                                             // compaction must not wipe what nobody has seen yet

    // The wave-per-chunk encoders place and copy their chunks themselves (EncParams::status; encode_wave.hip
    // place_and_copy): no k_layout / k_compact.  The lane-per-chunk encoders (N = 1, 2, 4, 8 with many chunks) and the
    // 160 KiB alias model keep the three-kernel path.
    // (context option RANS_AMD_OPT_FUSED_PLACEMENT = 0 restores the three-kernel path)
    const bool unfused_env = ctx->unfused;
    const int enc_format = model->host.r64_search ? kKernelFormatR64Search
                           : (format == RANS_AMD_FMT_WORD && model->host.sym_bytes == 2) ? kKernelFormatWord16
                           : (format == RANS_AMD_FMT_ALIAS && model->d_alias_remap16) ? kKernelFormatAliasLds
                                                                                                  : format;
    if (enc_format == RANS_AMD_FMT_ALIAS && !encode_uses_lanes(enc_format, nchunks, n_ways))
        return fail(RANS_AMD_E_UNSUPPORTED, "encode: this alias model has no LDS tables (cannot happen for a model rans_amd_model_create accepted)");
    EncParams ep{};
    ep.syms = static_cast<const uint8_t *>(d_syms);
    ep.n = n;
    ep.nchunks = nchunks;
    ep.chunk_syms = chunk_syms;
    ep.n_ways = n_ways;
    ep.scratch = static_cast<uint8_t *>(ctx->scratch.ptr);
    ep.slot_bytes = slot;
    ep.nsyms = model->host.nsyms;
    ep.sym_bytes = (uint32_t)model->host.sym_bytes;
    ep.scale_bits = model->host.scale_bits;
    ep.variant = ctx->variant;
    // Watchdog of the placement protocols: half a minute plus what one wave may legitimately need for the call's largest
    // chunk -- a lane codes a symbol in well under a microsecond, and min(n_ways, 64) lanes share a chunk (a 2^31-symbol
    // chunk of a 1-way stream: the waits for it may last as long as it does).
    ep.wait_ticks = 30ull * 100000000ull + ((uint64_t)chunk_syms / (n_ways < 64u ? n_ways : 64u)) * 100ull;
    {   // (measure build only: a base of milliseconds instead of half a minute, and the per-chunk share switched off -- what
        //  tests/test_gpu_slots.py uses to show that the share is what lets one huge low-way chunk through)
        static const char *base_ms = measure_knob("RANS_AMD_WATCHDOG_BASE_MS");
        static const bool no_scale = measure_knob("RANS_AMD_WATCHDOG_NO_SCALE") != nullptr;
        if (base_ms)
            ep.wait_ticks = (uint64_t)atoi(base_ms) * 100000ull +
                            (no_scale ? 0ull : ((uint64_t)chunk_syms / (n_ways < 64u ? n_ways : 64u)) * 100ull);
    }
    const bool lanes = encode_uses_lanes(enc_format, nchunks, n_ways);
    // context option RANS_AMD_OPT_LANE_FUSED_PLACEMENT: the lane encoders place their chunks themselves as well -- bit-exact,
    // tested, and on config 2 no faster than k_layout + k_compact_small behind them (lanes.hip says why), hence opt-in
    const bool lanes_fused_env = (ctx->variant & kVarLanesFused) != 0;
    const int fits = lanes ? 0 : encode_fused_fits(enc_format, model->host.nsyms, model->host.scale_bits);
    const bool fused = !slots && nchunks > 0 && nchunks < (1ull << 31) && !unfused_env &&
                       (lanes ? lanes_fused_env && encode_lanes_can_fuse(enc_format, ep, ctx->num_cus) : fits != 0);
    // Scratch: one worst-case slot per chunk, or -- context option RANS_AMD_OPT_ENC_SCRATCH_RING, fused wave encoders
    // only -- a small ring of slots per coding wave (kernels.h kEncRingSlots): half the workspace for a 1 GiB shard, and
    // measured 2-4 % slower, hence opt-in.
    const uint64_t ring_waves = (uint64_t)ctx->num_cus * kEncRingMaxWavesPerCu;
    const bool ring = fused && !lanes && fits == 1 && ctx->scratch_ring && slot <= kEncRingMaxSlotBytes && nchunks > ring_waves * kEncRingSlots;
    bool use_lanes = lanes;
    // the 8-way word layout's group encoder (encode_groups.hip): octets of chunks handed out through the claim counters
    const bool groups = lanes && !fused && enc_format == RANS_AMD_FMT_WORD && encode_word_groups_applicable(ep);
    if (slots) { // the caller's container is where the chunks are coded: no scratch at all
        ep.scratch = static_cast<uint8_t *>(d_out);
        ep.slot_layout = 1u;
        ep.offsets = d_offsets;
        // (sized slots: of the lane encoders only the staged ones -- whole-line flushes that stop at their slot's first line --
        //  can tell that a chunk does not fit; a shape they would not take goes to the wave encoders)
        if (sized && lanes && !encode_lanes_sized_ok(enc_format, ep, ctx->num_cus)) {
            use_lanes = false;
            ep.no_lanes = 1u;
        }
        if ((!use_lanes || groups) && nchunks > 0 && nchunks < (1ull << 32)) { // wave / group encoders hand their chunks out dynamically
            // claim counters; sized slots: + the overflow count and the redo launch's claim counter (a line each) + the list
            const size_t claim_bytes = (size_t)(kWorkPools + 2) * kWorkPoolStride * 4;
            rc = ctx->enc_status.reserve(claim_bytes + (sized ? (size_t)nchunks * 4 : 0));
            if (rc)
                return rc;
            HIP_TRY(zero.add(ctx->enc_status.ptr, claim_bytes));
            ep.claims = static_cast<unsigned int *>(ctx->enc_status.ptr);
        } else if (sized) { // lane encoders: only the overflow words
            const size_t claim_bytes = (size_t)(kWorkPools + 2) * kWorkPoolStride * 4;
            rc = ctx->enc_status.reserve(claim_bytes + (size_t)nchunks * 4);
            if (rc)
                return rc;
            HIP_TRY(zero.add(ctx->enc_status.ptr, claim_bytes));
        }
        if (sized) {
            ep.ovf_ctl = static_cast<unsigned int *>(ctx->enc_status.ptr) + kWorkPools * kWorkPoolStride;
            ep.ovf_list = static_cast<uint32_t *>(ctx->enc_status.ptr) + (kWorkPools + 2) * kWorkPoolStride;
        }
        if (nchunks == 0)
            HIP_TRY(zero.add(d_offsets, 8));
    } else {
        rc = ctx->scratch.reserve((size_t)((ring ? ring_waves * kEncRingSlots : nchunks) * slot + 64));
        if (rc)
            return rc;
        ep.scratch = static_cast<uint8_t *>(ctx->scratch.ptr);
        if (groups && nchunks < (1ull << 32)) {
            const size_t claim_bytes = (size_t)(kWorkPools + 2) * kWorkPoolStride * 4;
            rc = ctx->enc_status.reserve(claim_bytes);
            if (rc)
                return rc;
            HIP_TRY(zero.add(ctx->enc_status.ptr, claim_bytes));
            ep.claims = static_cast<unsigned int *>(ctx->enc_status.ptr);
        }
    }
    ep.ring_slots = ring ? kEncRingSlots : 0u;
    if (fused) {
        // (wave encoders: a word per chunk; lane encoders: a word per round of a block, at most one per batch of 64
        //  chunks; then the claim counters)
        const size_t status_bytes = (size_t)(nchunks + 8u * kWorkPools) * 8;
        rc = ctx->enc_status.reserve(status_bytes);
        if (rc)
            return rc;
        HIP_TRY(zero.add(static_cast<uint8_t *>(ctx->enc_status.ptr), status_bytes));
        if (fits == 2) { // the tables fill the LDS: one mailbox per block in global memory
            const size_t mb_bytes = (size_t)ctx->num_cus * kEncMailboxStride;
            rc = ctx->enc_mailboxes.reserve(mb_bytes);
            if (rc)
                return rc;
            HIP_TRY(zero.add(ctx->enc_mailboxes.ptr, mb_bytes));
            ep.mailbox_global = static_cast<uint8_t *>(ctx->enc_mailboxes.ptr);
        }
    }

    if (ctx->timing && !t_capturing)
        HIP_TRY(hipEventRecord(ctx->ev[2], s));
    HIP_TRY(zero.flush());
    if (nchunks) {
        ep.syms = static_cast<const uint8_t *>(d_syms);
        ep.n = n;
        ep.nchunks = nchunks;
        ep.chunk_syms = chunk_syms;
        ep.n_ways = n_ways;
        ep.slot_bytes = slot;
        ep.lengths = d_lengths;
        ep.enc_recs = model->d_enc;
        ep.word_enc_recs = model->d_word_enc;
        ep.word_small = model->host.word_small ? 1u : 0u;
        ep.dense256 = model->host.dense256 ? 1u : 0u;
        ep.alias_remap = static_cast<const uint32_t *>(model->d_remap);
        ep.alias_recs8 = model->d_alias_recs8;
        ep.alias_remap16 = static_cast<const uint16_t *>(model->d_alias_remap16);
        ep.nsyms = model->host.nsyms;
        ep.scale_bits = model->host.scale_bits;
        ep.sym_bytes = (uint32_t)model->host.sym_bytes;
        ep.flags = ctx->d_enc_flags();
        if (fused) {
            ep.status = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(ctx->enc_status.ptr));
            ep.claims = reinterpret_cast<unsigned int *>(ep.status + nchunks); // (wave encoders; the lane encoders' scanners count behind their own status words)
            ep.offsets = d_offsets;
            ep.out = static_cast<uint8_t *>(d_out);
            ep.out_cap = out_cap;
        }
        HIP_TRY(launch_encode(enc_format, ep, ctx->num_cus, s, &ctx->last_enc_kernel));
        ctx->last_enc_fused = fused;
        ctx->last_enc_slots = slots;
        if (sized) {
            // the second launch: the chunks the first one abandoned, coded by the wave encoder into worst-case slots behind
            // the sized ones (normally none: the launch reads one word and ends)
            EncParams redo = ep;
            redo.redo = 1u;
            redo.no_lanes = 1u;
            redo.slot_bytes = worst_slot;
            redo.ovf_base = nchunks * slot;
            const uint64_t room = (out_cap - nchunks * slot) / worst_slot;
            redo.ovf_cap = (uint32_t)(room < nchunks ? room : nchunks);
            redo.claims = static_cast<unsigned int *>(ctx->enc_status.ptr); // (unused by the redo claims; MODE 3 wants it set)
            HIP_TRY(launch_encode(enc_format, redo, ctx->num_cus, s, nullptr));
        }
    }
    if (!fused && !slots) {
        LayoutParams lp;
        lp.lengths = d_lengths;
        lp.offsets = d_offsets;
        lp.nchunks = nchunks;
        lp.out_cap = out_cap;
        lp.flags = ctx->d_enc_flags();
        lp.block_sums = nullptr;
        if (layout_blocks(nchunks) > 1) {
            int rc = ctx->layout_sums.reserve((size_t)layout_blocks(nchunks) * 8);
            if (rc)
                return rc;
            lp.block_sums = static_cast<uint64_t *>(ctx->layout_sums.ptr);
        }
        HIP_TRY(launch_layout(lp, s));
        if (nchunks) {
            CompactParams cp{};
            cp.src_limit = ~0ull;
            cp.scratch = ep.scratch;
            cp.slot_bytes = slot;
            cp.lengths = d_lengths;
            cp.offsets = d_offsets;
            cp.out = static_cast<uint8_t *>(d_out);
            cp.nchunks = nchunks;
            cp.flags = ctx->d_enc_flags();
            HIP_TRY(launch_compact(cp, ctx->num_cus, s));
        }
    }
    if (ctx->timing && !t_capturing) {
        HIP_TRY(hipEventRecord(ctx->ev[3], s));
        ctx->enc_timed = true;
    }

    if (h_total_bytes) {
        uint32_t flags = 0;
        uint64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&flags, ctx->d_enc_flags(), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&total, d_offsets + nchunks, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *h_total_bytes = total;
        return encode_flags_status(flags);
    }
    return RANS_AMD_OK;
}

extern "C" {

int rans_amd_encode(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n,
                    uint32_t n_ways, uint32_t chunk_syms, void *d_out, uint64_t out_cap, uint64_t *d_offsets,
                    uint32_t *d_lengths, uint64_t *h_total_bytes, void *stream)
{
    return encode_impl(ctx, model, d_syms, n, n_ways, chunk_syms, d_out, out_cap, d_offsets, d_lengths, h_total_bytes, stream, false);
}

int rans_amd_encode_slots(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n,
                          uint32_t n_ways, uint32_t chunk_syms, void *d_out, uint64_t out_cap, uint64_t *d_offsets,
                          uint32_t *d_lengths, uint64_t *h_total_bytes, void *stream)
{
    return encode_impl(ctx, model, d_syms, n, n_ways, chunk_syms, d_out, out_cap, d_offsets, d_lengths, h_total_bytes, stream, true);
}

int rans_amd_encode_slots_sized(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_syms, uint64_t n,
                                uint32_t n_ways, uint32_t chunk_syms, uint64_t slot_bytes, void *d_out, uint64_t out_cap,
                                uint64_t *d_offsets, uint32_t *d_lengths, uint64_t *h_total_bytes, void *stream)
{
    if (slot_bytes == 0 || (slot_bytes & 63u) != 0)
        return fail(RANS_AMD_E_ARG, "encode_slots_sized: slot_bytes must be a positive multiple of 64");
    return encode_impl(ctx, model, d_syms, n, n_ways, chunk_syms, d_out, out_cap, d_offsets, d_lengths, h_total_bytes, stream, true,
                       slot_bytes);
}

uint64_t rans_amd_tight_slot_bytes(const rans_amd_model *model, uint32_t n_ways, uint32_t chunk_syms)
{
    if (!model || chunk_syms == 0 || n_ways == 0)
        return 0;
    // expected code length of a symbol drawn from the model, and its variance, in bits: the model coded by itself
    const HostModel &h = model->host;
    const double M = (double)(1ull << h.scale_bits);
    double mean = 0.0, sq = 0.0;
    for (uint32_t f : h.freqs)
        if (f) {
            const double pr = (double)f / M, bits = -std::log2(pr);
            mean += pr * bits;
            sq += pr * bits * bits;
        }
    const double var = sq > mean * mean ? sq - mean * mean : 0.0;
    const double state_bytes = h.format == RANS_AMD_FMT_R64 ? 8.0 : 4.0;
    // The coders that stage their stream in LDS (word / byte / LDS-alias format, byte symbols, 64 lanes) check the slot
    // exactly; the others before every pair of rounds, by what two rounds can emit at most -- their slot gets that margin
    const bool staged = h.sym_bytes == 1 && n_ways == 64 && h.nsyms <= 256 && h.scale_bits <= 16 &&
                        (h.format == RANS_AMD_FMT_WORD || h.format == RANS_AMD_FMT_BYTE || h.format == RANS_AMD_FMT_ALIAS);
    const double lanes = n_ways >= 64 ? (double)((n_ways + 63u) & ~63u) : 0.0; // (narrow interleaves: lane encoders, whole lines)
    const double margin = staged ? 0.0 : 2.0 * lanes * (h.format == RANS_AMD_FMT_R64 ? 4.0 : 2.0); // (two rounds)
    const double bytes = 1.02 * mean * (double)chunk_syms / 8.0 + state_bytes * (double)n_ways +
                         4.0 * std::sqrt(var * (double)chunk_syms) / 8.0 + 16.0 + margin;
    const uint64_t worst = encode_slot_bytes(h.format, chunk_syms, n_ways, chunk_syms);
    const uint64_t tight = ((uint64_t)bytes + 63u) & ~uint64_t(63);
    return tight < worst ? tight : worst;
}

uint64_t rans_amd_encode_sized_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms, uint64_t slot_bytes,
                                     uint64_t overflow_chunks)
{
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    if (nchunks == 0 || chunk_syms == 0)
        return 16;
    const uint64_t worst = encode_slot_bytes(format, n, n_ways, chunk_syms);
    if (slot_bytes == 0 || slot_bytes >= worst)
        return nchunks * worst;
    return nchunks * slot_bytes + (overflow_chunks < nchunks ? overflow_chunks : nchunks) * worst;
}

uint64_t rans_amd_slot_bytes(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms)
{
    return chunk_syms ? encode_slot_bytes(format, n, n_ways, chunk_syms) : 0;
}

uint64_t rans_amd_encode_slots_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms)
{
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    return nchunks ? nchunks * encode_slot_bytes(format, n, n_ways, chunk_syms) : 16;
}

int rans_amd_container_compact(rans_amd_ctx *ctx, const void *d_src, uint64_t src_bytes, const uint64_t *d_src_offsets,
                               const uint32_t *d_lengths, uint64_t n_chunks, void *d_dst, uint64_t dst_cap,
                               uint64_t *d_dst_offsets, uint64_t *h_total_bytes, void *stream)
{
    if (!ctx || !d_dst || !d_dst_offsets || (n_chunks && (!d_src || !d_src_offsets || !d_lengths)))
        return fail(RANS_AMD_E_ARG, "container_compact: NULL argument");
    if (((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15u) != 0)
        return fail(RANS_AMD_E_ARG, "container_compact: d_src and d_dst must be 16-byte aligned");
    if (d_src == d_dst)
        return fail(RANS_AMD_E_ARG, "container_compact: the copy is not in place (d_dst must be another buffer)");
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CaptureScope capture(s);
    if (capture.active && h_total_bytes)
        return fail(RANS_AMD_E_ARG, "container_compact: h_total_bytes must be NULL while the stream is capturing");
    {   // (this compaction's verdict replaces an older compaction's: the kernels below skip their work while the word says
        //  "no room" -- but no ENCODE clears it any more, see encode_impl)
        ZeroList zero(s);
        HIP_TRY(zero.add(ctx->d_compact_flags(), 4));
        HIP_TRY(zero.flush());
    }
    LayoutParams lp;
    lp.lengths = d_lengths;
    lp.offsets = d_dst_offsets;
    lp.nchunks = n_chunks;
    lp.out_cap = dst_cap;
    lp.flags = ctx->d_compact_flags();
    lp.block_sums = nullptr;
    if (layout_blocks(n_chunks) > 1) {
        int rc = ctx->layout_sums.reserve((size_t)layout_blocks(n_chunks) * 8);
        if (rc)
            return rc;
        lp.block_sums = static_cast<uint64_t *>(ctx->layout_sums.ptr);
    }
    HIP_TRY(launch_layout(lp, s));
    if (n_chunks) {
        CompactParams cp{};
        cp.scratch = static_cast<const uint8_t *>(d_src);
        cp.src_offsets = d_src_offsets;
        cp.src_limit = (reinterpret_cast<uint64_t>(d_src) + src_bytes + 15u) & ~uint64_t(15);
        cp.src_bytes = src_bytes;
        // kernel choice (launch_compact: "slots" of at most 8 KiB take 16 lanes per chunk, larger ones a wave): the lengths
        // live on the device, but the source cannot hold chunks longer than src_bytes / n_chunks on average -- round 4 went by
        // the chunk COUNT alone and gave a 1 GiB container of 13 KB chunks the small-chunk kernel (0.82 ms; 0.33 with a wave each)
        cp.slot_bytes = (src_bytes / n_chunks <= 2048) ? 4096 : 1u << 20;
        cp.lengths = d_lengths;
        cp.offsets = d_dst_offsets;
        cp.out = static_cast<uint8_t *>(d_dst);
        cp.nchunks = n_chunks;
        cp.flags = ctx->d_compact_flags();
        HIP_TRY(launch_compact(cp, ctx->num_cus, s));
    }
    if (h_total_bytes) {
        uint32_t flags = 0;
        uint64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&flags, ctx->d_compact_flags(), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&total, d_dst_offsets + n_chunks, 8, hipMemcpyDeviceToHost, s));
        {   // reported with this return: the word starts over
            ZeroList zero(s);
            HIP_TRY(zero.add(ctx->d_compact_flags(), 4));
            HIP_TRY(zero.flush());
        }
        HIP_TRY(hipStreamSynchronize(s));
        *h_total_bytes = total;
        return encode_flags_status(flags);
    }
    return RANS_AMD_OK;
}

int rans_amd_encode_status(rans_amd_ctx *ctx, void *stream)
{
    if (!ctx)
        return fail(RANS_AMD_E_ARG, "encode_status: ctx is NULL");
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t flags = 0, cflags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, ctx->d_enc_flags(), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&cflags, ctx->d_compact_flags(), 4, hipMemcpyDeviceToHost, s));
    {   // the compactions' verdict is reported by this call: the word starts over
        ZeroList zero(s);
        HIP_TRY(zero.add(ctx->d_compact_flags(), 4));
        HIP_TRY(zero.flush());
    }
    HIP_TRY(hipStreamSynchronize(s));
    const int rc = encode_flags_status(flags);
    // (the last compaction's verdict -- destination too small, source index corrupt -- whatever encodes ran behind it)
    return rc != RANS_AMD_OK ? rc : encode_flags_status(cflags & (2u | 512u));
}

/* ---- decode ------------------------------------------------------------- */

int rans_amd_decode(rans_amd_ctx *ctx, const rans_amd_model *model, const void *d_container,
                    uint64_t container_bytes, const uint64_t *d_offsets, const uint32_t *d_lengths, uint64_t n,
                    uint32_t n_ways, uint32_t chunk_syms, void *d_out, uint64_t *h_bad_chunks, void *stream)
{
    if (!ctx || !model || (n && (!d_container || !d_offsets || !d_lengths || !d_out)) || chunk_syms == 0)
        return fail(RANS_AMD_E_ARG, "decode: NULL argument or chunk_syms == 0");
    if (model->ctx != ctx)
        return fail(RANS_AMD_E_ARG, "decode: model belongs to another context");
    const int format = model->host.format;
    if (!ways_supported(format, n_ways))
        return fail(RANS_AMD_E_UNSUPPORTED, "decode: n_ways must be in 1..512");
    if ((reinterpret_cast<uintptr_t>(d_container) & 15u) != 0)
        return fail(RANS_AMD_E_ARG, "decode: d_container must be 16-byte aligned");
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CaptureScope capture(s);
    if (capture.active && h_bad_chunks)
        return fail(RANS_AMD_E_ARG, "decode: h_bad_chunks must be NULL while the stream is capturing (rans_amd_decode_errors after the replay)");

    if (nchunks) {
        DecParams dp{};
        dp.container = static_cast<const uint8_t *>(d_container);
        dp.container_bytes = container_bytes;
        dp.offsets = d_offsets;
        dp.lengths = d_lengths;
        dp.out = static_cast<uint8_t *>(d_out);
        dp.n = n;
        dp.nchunks = nchunks;
        dp.chunk_syms = chunk_syms;
        dp.n_ways = n_ways;
        dp.table0 = model->d_table0;
        dp.table1 = model->d_table1 ? model->d_table1 : model->d_table0;
        dp.table0_bytes = model->table0_bytes;
        dp.table1_bytes = model->table1_bytes;
        dp.packed = model->d_packed;
        dp.packed_bytes = model->d_packed ? (uint32_t)(model->host.r64_packed.size() * 4) : 0u;
        dp.scale_bits = model->host.scale_bits;
        dp.log2nsyms = model->host.log2nsyms;
        if (model->host.r64_search) { // log2 of the padded cum table
            dp.log2nsyms = 0;
            while ((1u << dp.log2nsyms) < model->host.cum_padded.size())
                dp.log2nsyms++;
        }
        dp.sym_bytes = (uint32_t)model->host.sym_bytes;
        dp.err_count = ctx->d_err();
        // dynamic chunk hand-out: a 4-byte counter zeroed in stream order ahead of the kernel
        // Counters form a ring of 64 slots (all zero at context creation); launch i uses slot
        // i % 64 and re-zeroes slot (i + 32) % 64 from inside the kernel, so no memset node is
        // needed and up to 32 decode launches of one context may be in flight at once.
        dp.work_counter = nullptr;
        dp.work_counter_reset = nullptr;
        dp.span = nullptr;
        dp.span_reset = nullptr;
        {   // (small, allocated once, kept: not part of what rans_amd_ctx_trim drops)
            int wrc = ctx->wave_scratch.reserve((size_t)ctx->num_cus * 2u * (kDecBlockThreads / 64) * 64u);
            if (wrc)
                return wrc;
            dp.wave_scratch = static_cast<uint8_t *>(ctx->wave_scratch.ptr);
        }
        dp.variant = ctx->variant;
        if (nchunks < 0xffffffffull) {
            unsigned int *ring = reinterpret_cast<unsigned int *>(ctx->d_words + 256);
            const uint32_t per_slot = kWorkSlotWords;
            if (capture.active) {
                // A launch that becomes a graph node runs again and again with these very arguments: its counters are one
                // of kCaptureSlots slots of their own, zeroed by a k_zero node in front of the kernel (so every replay
                // starts from zero, and the ring of the eager launches never sees a slot a replay has used).
                dp.work_counter = capture_counters(ctx);
                {
                    ZeroList zero(s);
                    HIP_TRY(zero.add(dp.work_counter, (uint64_t)per_slot * 4));
                    HIP_TRY(zero.flush());
                }
                dp.span = reinterpret_cast<unsigned long long *>(dp.work_counter + kWorkPools * kWorkPoolStride);
            } else {
                dp.work_counter = ring + (size_t)(ctx->launch_seq % kWorkSlots) * per_slot;
                dp.work_counter_reset = ring + (size_t)((ctx->launch_seq + kWorkSlots / 2) % kWorkSlots) * per_slot;
                // the slot's last line: first wave start / last wave end of the launch (rans_amd_launch_spans)
                dp.span = reinterpret_cast<unsigned long long *>(dp.work_counter + kWorkPools * kWorkPoolStride);
                dp.span_reset = reinterpret_cast<unsigned long long *>(dp.work_counter_reset + kWorkPools * kWorkPoolStride);
            }
        }
        // wave clocks (rans_amd_set_timing(ctx, 2)) and the debug timeline (RANS_AMD_TRACE=<file>): per-wave
        // start/end ticks, XCD, shader cycles and rounds, read back after a sync
        static const char *trace_path = measure_knob("RANS_AMD_TRACE");
        const bool want_trace = (trace_path || ctx->wave_clocks_on) && !capture.active; // (read back after a sync)
        const size_t trace_words = (size_t)kTraceWords * 2u * 16u * (size_t)ctx->num_cus;
        dp.trace = nullptr;
        if (want_trace) {
            int trc = ctx->trace.reserve(trace_words * 8);
            if (trc)
                return trc;
            dp.trace = static_cast<unsigned long long *>(ctx->trace.ptr);
            HIP_TRY(hipMemsetAsync(dp.trace, 0, trace_words * 8, s));
        }
        if (ctx->timing && !t_capturing)
            HIP_TRY(hipEventRecord(ctx->ev[0], s));
        int dec_format = model->host.r64_search ? kKernelFormatR64Search
                         : (format == RANS_AMD_FMT_WORD && model->host.sym_bytes == 2) ? kKernelFormatWord16
                                                                                     : format;
        // 64-way alias streams with at least a pair of chunks: two chunks per wave, tables in the FMT_ALIAS2 form
        // (decode_dual.hip).  u8 symbols are stored a dword per lane: 4-byte aligned chunks of output.
        // Only models whose tables leave no room for a second block per CU: with two blocks, eight waves per SIMD and one
        // chunk each are faster than four with two (decode_dual.hip).
        const bool one_block_per_cu = 2u * ((size_t)((model->table0_bytes + 15u) & ~15u) + ((model->table1_bytes + 15u) & ~15u) +
                                            (size_t)(kDecBlockThreads / 64) * kRingStride) > 160u * 1024u;
        const bool dual_always = (ctx->variant & kVarDualAlways) != 0;
        if (format == RANS_AMD_FMT_ALIAS && model->d_dual0 && n_ways == 64 && nchunks >= 2 && !(ctx->variant & kVarNoDual) &&
            (one_block_per_cu || dual_always) && decode_dual_fits(model->dual0_bytes, model->dual1_bytes) && !dp.trace &&
            ((reinterpret_cast<uintptr_t>(d_out) | ((uintptr_t)chunk_syms * dp.sym_bytes)) & 3u) == 0) {
            dp.table0 = model->d_dual0;
            dp.table1 = model->d_dual1;
            dp.table0_bytes = model->dual0_bytes;
            dp.table1_bytes = model->dual1_bytes;
            dec_format = model->host.alias2_wide ? kKernelFormatAlias2W : kKernelFormatAlias2;
        }
        // Byte format, wave-per-chunk decoders: the fused slot records (one gather per symbol) where the model has them and
        // the table leaves room for two blocks per CU (scale_bits <= 12: 32 KiB; at 13 bits one block per CU -- four waves
        // per SIMD -- loses to eight with the two-gather tables: profiles/r04_byte_decoder_variants.log).
        // The lane-per-chunk kernels, the 2-way pair kernel (decode_groups.hip) and the two-chunk kernel keep cum2sym + records.
        if (dec_format == RANS_AMD_FMT_BYTE && model->d_fused && !lanes_applicable(nchunks, n_ways) && !decode_byte_pairs_applicable(dp) &&
            model->host.scale_bits <= 12) {
            dp.table0 = model->d_fused;
            dp.table0_bytes = (uint32_t)(model->host.byte_slots.size() * sizeof(WordSlot));
            dp.table1 = model->d_fused;
            dp.table1_bytes = 0;
            dec_format = kKernelFormatByteFused;
        }
        HIP_TRY(launch_decode(dec_format, dp, ctx->num_cus, s, &ctx->last_kernel));
        // the launch that uses slot i zeroes slot i + 32: move on only once it really is in the stream,
        // or a later launch would start from a counter nobody reset
        if (dp.work_counter && !capture.active)
            ctx->launch_seq++;
        if (ctx->timing && !t_capturing) {
            HIP_TRY(hipEventRecord(ctx->ev[1], s));
            ctx->dec_timed = true;
        }
        if (want_trace) {
            std::vector<unsigned long long> host(trace_words);
            HIP_TRY(hipMemcpyAsync(host.data(), dp.trace, trace_words * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            rans_amd_wave_clocks wc = {0, 0, 0, 0.0, 0.0};
            unsigned long long t_first = ~0ull, t_last = 0, longest_ticks = 0, longest_cycles = 0;
            for (size_t w = 0; w < trace_words / kTraceWords; ++w) {
                const unsigned long long *t = &host[kTraceWords * w];
                if (!t[1])
                    continue; // this slot's wave never ran (grid smaller than the buffer) or is not instrumented
                wc.waves++;
                wc.shader_cycles += t[3];
                wc.rounds += t[4];
                t_first = t[0] < t_first ? t[0] : t_first;
                t_last = t[1] > t_last ? t[1] : t_last;
                if (t[1] - t[0] > longest_ticks) {
                    longest_ticks = t[1] - t[0];
                    longest_cycles = t[3];
                }
            }
            if (wc.waves && longest_ticks) {
                wc.sclk_hz = (double)longest_cycles / ((double)longest_ticks * 1e-8); // wall_clock64 ticks at 100 MHz
                wc.kernel_ticks_ms = (double)(t_last - t_first) * 1e-5;
            }
            ctx->wave_clocks = wc;
            if (trace_path) {
                if (FILE *f = fopen(trace_path, "w")) {
                    for (size_t w = 0; w < trace_words / kTraceWords; ++w)
                        if (host[kTraceWords * w + 1])
                            fprintf(f, "%zu %llu %llu %llu %llu %llu\n", w, host[kTraceWords * w], host[kTraceWords * w + 1],
                                    host[kTraceWords * w + 2], host[kTraceWords * w + 3], host[kTraceWords * w + 4]);
                    fclose(f);
                }
            }
        }
    }
    if (h_bad_chunks) {
        unsigned long long bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, ctx->d_err(), 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemsetAsync(ctx->d_err(), 0, 8, s));
        HIP_TRY(hipStreamSynchronize(s));
        *h_bad_chunks = bad;
        if (bad)
            return fail(RANS_AMD_E_CORRUPT, "decode: at least one chunk failed its integrity check");
    }
    return RANS_AMD_OK;
}

int rans_amd_probe_placement(rans_amd_ctx *ctx, const rans_amd_model *model, const void *const *d_containers,
                             uint32_t n_containers, uint64_t container_bytes, const uint64_t *d_offsets,
                             const uint32_t *d_lengths, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                             void *const *d_outs, uint32_t n_outs, uint32_t launches, uint32_t sweeps,
                             uint32_t *best_container, uint32_t *best_out, float *ms_matrix, void *stream)
{
    if (!ctx || !model || !d_containers || !d_outs || n_containers == 0 || n_outs == 0 || !best_container || !best_out ||
        (uint64_t)n_containers * n_outs > 4096)
        return fail(RANS_AMD_E_ARG, "probe_placement: NULL argument, no candidates, or more than 4096 pairs");
    launches = launches ? launches : 6;
    sweeps = sweeps ? sweeps : 2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    {
        DeviceGuard guard(ctx->device);
        HIP_TRY(hipEventCreate(&ev0));
        if (hipError_t e = hipEventCreate(&ev1); e != hipSuccess) {
            (void)hipEventDestroy(ev0);
            HIP_TRY(e);
        }
    }
    std::vector<double> acc((size_t)n_containers * n_outs, 0.0);
    int rc = RANS_AMD_OK;
    // (rans_amd_decode takes the context's lock itself: this loop holds none)
    for (uint32_t sw = 0; sw < sweeps && rc == RANS_AMD_OK; ++sw)
        for (uint32_t i = 0; i < n_containers && rc == RANS_AMD_OK; ++i)
            for (uint32_t j = 0; j < n_outs && rc == RANS_AMD_OK; ++j) {
                for (uint32_t k = 0; k < launches + 2 && rc == RANS_AMD_OK; ++k) {
                    if (k == 2) {
                        DeviceGuard guard(ctx->device);
                        if (hipEventRecord(ev0, s) != hipSuccess)
                            rc = fail(RANS_AMD_E_HIP, "probe_placement: hipEventRecord failed");
                    }
                    if (rc == RANS_AMD_OK)
                        rc = rans_amd_decode(ctx, model, d_containers[i], container_bytes, d_offsets, d_lengths, n, n_ways, chunk_syms,
                                             d_outs[j], nullptr, stream);
                }
                if (rc != RANS_AMD_OK)
                    break;
                DeviceGuard guard(ctx->device);
                float ms = 0.f;
                if (hipEventRecord(ev1, s) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess ||
                    hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess)
                    rc = fail(RANS_AMD_E_HIP, "probe_placement: timing a pair failed");
                acc[(size_t)i * n_outs + j] += (double)ms / launches;
            }
    {
        DeviceGuard guard(ctx->device);
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
    }
    if (rc != RANS_AMD_OK)
        return rc;
    uint64_t bad = 0;
    rc = rans_amd_decode_errors(ctx, &bad, stream);
    if (rc != RANS_AMD_OK)
        return rc;
    size_t best = 0;
    for (size_t k = 0; k < acc.size(); ++k) {
        if (ms_matrix)
            ms_matrix[k] = (float)(acc[k] / sweeps);
        if (acc[k] < acc[best])
            best = k;
    }
    *best_container = (uint32_t)(best / n_outs);
    *best_out = (uint32_t)(best % n_outs);
    return RANS_AMD_OK;
}

int rans_amd_decode_errors(rans_amd_ctx *ctx, uint64_t *h_bad_chunks, void *stream)
{
    if (!ctx || !h_bad_chunks)
        return fail(RANS_AMD_E_ARG, "decode_errors: NULL argument");
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, ctx->d_err(), 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(ctx->d_err(), 0, 8, s));
    HIP_TRY(hipStreamSynchronize(s));
    *h_bad_chunks = bad;
    return bad ? fail(RANS_AMD_E_CORRUPT, "decode: at least one chunk failed its integrity check") : RANS_AMD_OK;
}


/* ---- per-chunk adaptive models ------------------------------ */

uint64_t rans_amd_chunk_freqs_bytes(uint64_t n, uint32_t chunk_syms)
{
    return rans_amd_num_chunks(n, chunk_syms) * 256u * sizeof(uint16_t);
}

} // extern "C"

// format: RANS_AMD_FMT_BYTE (scale_bits 8..12) or RANS_AMD_FMT_WORD (scale_bits 12, the format's own)
static int adaptive_format_check(int format, uint32_t scale_bits, const char *who)
{
    char msg[160];
    if (format != RANS_AMD_FMT_BYTE && format != RANS_AMD_FMT_WORD) {
        snprintf(msg, sizeof msg, "%s: per-chunk models exist for the byte and the word format", who);
        return fail(RANS_AMD_E_UNSUPPORTED, msg);
    }
    if (format == RANS_AMD_FMT_WORD ? scale_bits != 12 : (scale_bits < 8 || scale_bits > 12)) {
        snprintf(msg, sizeof msg, "%s: scale_bits must be 8..12 (a chunk's tables live in one wave's LDS), 12 for the word format", who);
        return fail(RANS_AMD_E_UNSUPPORTED, msg);
    }
    return RANS_AMD_OK;
}

static int encode_adaptive_impl(rans_amd_ctx *ctx, const int format, const void *d_syms, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                                uint32_t scale_bits, void *d_out, uint64_t out_cap, uint64_t *d_offsets, uint32_t *d_lengths,
                                uint16_t *d_chunk_freqs, uint64_t *h_total_bytes, void *stream)
{
    if (!ctx || !d_out || !d_offsets || !d_lengths || !d_chunk_freqs || (n && !d_syms) || chunk_syms == 0)
        return fail(RANS_AMD_E_ARG, "encode_adaptive: NULL argument or chunk_syms == 0");
    if (int rc = adaptive_format_check(format, scale_bits, "encode_adaptive"))
        return rc;
    if ((reinterpret_cast<uintptr_t>(d_chunk_freqs) & 7u) != 0) // (the kernels read a row with 8-byte loads per lane)
        return fail(RANS_AMD_E_ARG, "encode_adaptive: d_chunk_freqs must be 8-byte aligned");
    if (!ways_supported(RANS_AMD_FMT_BYTE, n_ways))
        return fail(RANS_AMD_E_UNSUPPORTED, "encode_adaptive: n_ways must be in 1..512");
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0)
        return fail(RANS_AMD_E_ARG, "encode_adaptive: d_out must be 16-byte aligned");
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CaptureScope capture(s);
    if (capture.active && h_total_bytes)
        return fail(RANS_AMD_E_ARG, "encode_adaptive: h_total_bytes must be NULL while the stream is capturing");
    const uint64_t slot = encode_slot_bytes(format, n, n_ways, chunk_syms);
    if (slot > 0xfffffff0ull)
        return fail(RANS_AMD_E_UNSUPPORTED, "encode_adaptive: chunk_syms too large");
    int rc = ctx->scratch.reserve((size_t)(nchunks * slot + 64));
    if (rc)
        return rc;
    {
        ZeroList zero(s);
        HIP_TRY(zero.add(ctx->d_enc_flags(), 8)); // (encode + histogram flags; a compaction's verdict stays until reported, as in encode_impl)
        HIP_TRY(zero.flush());
    }
    if (nchunks) {
        // 1 + 2. count_freqs and normalize_freqs per chunk ON THE DEVICE (main.cpp:59-129, one wave per chunk; the
        // arithmetic of model.cpp normalize_freqs): the rows go straight into d_chunk_freqs -- no copy to the host, no
        // synchronisation, the call is asynchronous like the rest of the ABI
        HIP_TRY(launch_chunk_models(d_syms, n, chunk_syms, nchunks, scale_bits, d_chunk_freqs, ctx->d_enc_flags(), ctx->num_cus, s));
        // 3. encode: every wave builds the records of the chunk it codes
        if (ctx->timing && !t_capturing)
            HIP_TRY(hipEventRecord(ctx->ev[2], s));
        EncParams ep{};
        ep.syms = static_cast<const uint8_t *>(d_syms);
        ep.n = n;
        ep.nchunks = nchunks;
        ep.chunk_syms = chunk_syms;
        ep.n_ways = n_ways;
        ep.scratch = static_cast<uint8_t *>(ctx->scratch.ptr);
        ep.slot_bytes = slot;
        ep.lengths = d_lengths;
        ep.nsyms = 256;
        ep.scale_bits = scale_bits;
        ep.sym_bytes = 1;
        ep.flags = ctx->d_enc_flags();
        ep.chunk_freqs = d_chunk_freqs;
        HIP_TRY(launch_encode(format == RANS_AMD_FMT_WORD ? kKernelFormatWordAdaptive : kKernelFormatByteAdaptive, ep, ctx->num_cus, s,
                              &ctx->last_enc_kernel));
        ctx->last_enc_fused = false;
        ctx->last_enc_slots = false;
    }
    LayoutParams lp;
    lp.lengths = d_lengths;
    lp.offsets = d_offsets;
    lp.nchunks = nchunks;
    lp.out_cap = out_cap;
    lp.flags = ctx->d_enc_flags();
    lp.block_sums = nullptr;
    if (layout_blocks(nchunks) > 1) {
        rc = ctx->layout_sums.reserve((size_t)layout_blocks(nchunks) * 8);
        if (rc)
            return rc;
        lp.block_sums = static_cast<uint64_t *>(ctx->layout_sums.ptr);
    }
    HIP_TRY(launch_layout(lp, s));
    if (nchunks) {
        CompactParams cp{};
        cp.src_limit = ~0ull;
        cp.scratch = static_cast<const uint8_t *>(ctx->scratch.ptr);
        cp.slot_bytes = slot;
        cp.lengths = d_lengths;
        cp.offsets = d_offsets;
        cp.out = static_cast<uint8_t *>(d_out);
        cp.nchunks = nchunks;
        cp.flags = ctx->d_enc_flags();
        HIP_TRY(launch_compact(cp, ctx->num_cus, s));
    }
    if (ctx->timing && !t_capturing && nchunks) { // (ev[2] is recorded in front of the coding kernel, which an empty input does not launch)
        HIP_TRY(hipEventRecord(ctx->ev[3], s));
        ctx->enc_timed = true;
    }
    if (h_total_bytes) {
        uint32_t flags = 0;
        uint64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&flags, ctx->d_enc_flags(), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&total, d_offsets + nchunks, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *h_total_bytes = total;
        if (flags & 1u)
            return fail(RANS_AMD_E_MODEL, "encode_adaptive: a chunk's counts could not be normalised, or a symbol without a slot was met");
        if (flags & 2u)
            return fail(RANS_AMD_E_SPACE, "encode_adaptive: container does not fit out_cap");
    }
    return RANS_AMD_OK;
}

static int decode_adaptive_impl(rans_amd_ctx *ctx, const int format, const void *d_container, uint64_t container_bytes,
                                const uint64_t *d_offsets, const uint32_t *d_lengths, const uint16_t *d_chunk_freqs, uint64_t n,
                                uint32_t n_ways, uint32_t chunk_syms, uint32_t scale_bits, void *d_out, uint64_t *h_bad_chunks, void *stream)
{
    if (!ctx || (n && (!d_container || !d_offsets || !d_lengths || !d_chunk_freqs || !d_out)) || chunk_syms == 0)
        return fail(RANS_AMD_E_ARG, "decode_adaptive: NULL argument or chunk_syms == 0");
    if (int rc = adaptive_format_check(format, scale_bits, "decode_adaptive"))
        return rc;
    if ((reinterpret_cast<uintptr_t>(d_chunk_freqs) & 7u) != 0)
        return fail(RANS_AMD_E_ARG, "decode_adaptive: d_chunk_freqs must be 8-byte aligned");
    if (!ways_supported(RANS_AMD_FMT_BYTE, n_ways))
        return fail(RANS_AMD_E_UNSUPPORTED, "decode_adaptive: n_ways must be in 1..512");
    if ((reinterpret_cast<uintptr_t>(d_container) & 15u) != 0)
        return fail(RANS_AMD_E_ARG, "decode_adaptive: d_container must be 16-byte aligned");
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CaptureScope capture(s);
    if (capture.active && h_bad_chunks)
        return fail(RANS_AMD_E_ARG, "decode_adaptive: h_bad_chunks must be NULL while the stream is capturing");
    if (nchunks) {
        DecParams dp{};
        dp.container = static_cast<const uint8_t *>(d_container);
        dp.container_bytes = container_bytes;
        dp.offsets = d_offsets;
        dp.lengths = d_lengths;
        dp.out = static_cast<uint8_t *>(d_out);
        dp.n = n;
        dp.nchunks = nchunks;
        dp.chunk_syms = chunk_syms;
        dp.n_ways = n_ways;
        dp.scale_bits = scale_bits;
        dp.log2nsyms = 8;
        dp.sym_bytes = 1;
        dp.err_count = ctx->d_err();
        dp.chunk_freqs = d_chunk_freqs;
        if (nchunks < 0xffffffffull && capture.active) { // (as in rans_amd_decode)
            dp.work_counter = capture_counters(ctx);
            {
                ZeroList zero(s);
                HIP_TRY(zero.add(dp.work_counter, (uint64_t)kWorkSlotWords * 4));
                HIP_TRY(zero.flush());
            }
            dp.span = reinterpret_cast<unsigned long long *>(dp.work_counter + kWorkPools * kWorkPoolStride);
        } else if (nchunks < 0xffffffffull) {
            unsigned int *ring = reinterpret_cast<unsigned int *>(ctx->d_words + 256);
            dp.work_counter = ring + (size_t)(ctx->launch_seq % kWorkSlots) * kWorkSlotWords;
            dp.work_counter_reset = ring + (size_t)((ctx->launch_seq + kWorkSlots / 2) % kWorkSlots) * kWorkSlotWords;
            dp.span = reinterpret_cast<unsigned long long *>(dp.work_counter + kWorkPools * kWorkPoolStride);
            dp.span_reset = reinterpret_cast<unsigned long long *>(dp.work_counter_reset + kWorkPools * kWorkPoolStride);
        }
        if (ctx->timing && !t_capturing)
            HIP_TRY(hipEventRecord(ctx->ev[0], s));
        HIP_TRY(launch_decode(format == RANS_AMD_FMT_WORD ? kKernelFormatWordAdaptive : kKernelFormatByteAdaptive, dp, ctx->num_cus, s,
                              &ctx->last_kernel));
        if (dp.work_counter && !capture.active)
            ctx->launch_seq++;
        if (ctx->timing && !t_capturing) {
            HIP_TRY(hipEventRecord(ctx->ev[1], s));
            ctx->dec_timed = true;
        }
    }
    if (h_bad_chunks) {
        unsigned long long bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, ctx->d_err(), 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemsetAsync(ctx->d_err(), 0, 8, s));
        HIP_TRY(hipStreamSynchronize(s));
        *h_bad_chunks = bad;
        if (bad)
            return fail(RANS_AMD_E_CORRUPT, "decode_adaptive: at least one chunk failed its integrity check");
    }
    return RANS_AMD_OK;
}

// rans_amd_encode_adaptive_sized: the whole per-chunk-model encode as ONE kernel (encode_adaptive.hip)
static int encode_adaptive_sized_impl(rans_amd_ctx *ctx, const int format, const void *d_syms, uint64_t n, uint32_t n_ways,
                                      uint32_t chunk_syms, uint32_t scale_bits, void *d_out, uint64_t out_cap,
                                      uint64_t *d_offsets, uint32_t *d_lengths, uint16_t *d_chunk_freqs, uint64_t *h_total_bytes,
                                      void *stream)
{
    if (!ctx || !d_out || !d_offsets || !d_lengths || !d_chunk_freqs || (n && !d_syms) || chunk_syms == 0)
        return fail(RANS_AMD_E_ARG, "encode_adaptive_sized: NULL argument or chunk_syms == 0");
    if (int rc = adaptive_format_check(format, scale_bits, "encode_adaptive_sized"))
        return rc;
    if ((reinterpret_cast<uintptr_t>(d_chunk_freqs) & 7u) != 0) // (a row is written with 8-byte stores per lane)
        return fail(RANS_AMD_E_ARG, "encode_adaptive_sized: d_chunk_freqs must be 8-byte aligned");
    if (!ways_supported(format, n_ways))
        return fail(RANS_AMD_E_UNSUPPORTED, "encode_adaptive_sized: n_ways must be in 1..512");
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0)
        return fail(RANS_AMD_E_ARG, "encode_adaptive_sized: d_out must be 16-byte aligned");
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CaptureScope capture(s);
    if (capture.active && h_total_bytes)
        return fail(RANS_AMD_E_ARG, "encode_adaptive_sized: h_total_bytes must be NULL while the stream is capturing");
    const uint64_t worst = encode_slot_bytes(format, n, n_ways, chunk_syms);
    if (worst > 0xfffffff0ull || nchunks >= (1ull << 32))
        return fail(RANS_AMD_E_UNSUPPORTED, "encode_adaptive_sized: chunk_syms too large, or 2^32 chunks and more");
    if (!ctx->adapt_rcp.ptr) { // reciprocals by frequency (model.cpp adapt_rcp_tables): built once per context
        if (capture.active)
            return fail(RANS_AMD_E_ARG, "encode_adaptive_sized: make the same call once outside the capture first (a table is uploaded on first use)");
        std::vector<uint32_t> t;
        adapt_rcp_tables(t);
        int rc = ctx->adapt_rcp.reserve(t.size() * 4);
        if (rc)
            return rc;
        const hipError_t e = hipMemcpy(ctx->adapt_rcp.ptr, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ctx->adapt_rcp.release();
            return hip_fail(e, "encode_adaptive_sized: table upload");
        }
    }
    // a look-back word per chunk, then the claim counters (a line each)
    const size_t ctl_bytes = (size_t)nchunks * 8 + (size_t)kWorkPools * kWorkPoolStride * 4;
    int rc = ctx->enc_status.reserve(ctl_bytes);
    if (rc)
        return rc;
    {
        ZeroList zero(s);
        HIP_TRY(zero.add(ctx->d_enc_flags(), 8)); // (encode + histogram flags; a compaction's verdict stays until reported, as in encode_impl)
        HIP_TRY(zero.add(ctx->enc_status.ptr, ctl_bytes));
        if (nchunks == 0)
            HIP_TRY(zero.add(d_offsets, 8));
        HIP_TRY(zero.flush());
    }
    if (nchunks) {
        if (ctx->timing && !t_capturing)
            HIP_TRY(hipEventRecord(ctx->ev[2], s));
        AdaptEncParams ap{};
        ap.syms = static_cast<const uint8_t *>(d_syms);
        ap.n = n;
        ap.nchunks = nchunks;
        ap.chunk_syms = chunk_syms;
        ap.n_ways = n_ways;
        ap.scale_bits = scale_bits;
        ap.worst_slot = (uint32_t)worst;
        ap.out = static_cast<uint8_t *>(d_out);
        ap.out_cap = out_cap;
        ap.offsets = d_offsets;
        ap.lengths = d_lengths;
        ap.chunk_freqs = d_chunk_freqs;
        ap.flags = ctx->d_enc_flags();
        ap.status = static_cast<unsigned long long *>(ctx->enc_status.ptr);
        ap.claims = reinterpret_cast<unsigned int *>(ap.status + nchunks);
        // (watchdog of the look-back: half a minute plus what ONE wave may need to count the call's largest chunk)
        ap.wait_ticks = 30ull * 100000000ull + ((uint64_t)chunk_syms / 64u) * 100ull;
        ap.rcp = static_cast<const uint32_t *>(ctx->adapt_rcp.ptr);
        HIP_TRY(launch_encode_adaptive(format, ap, ctx->num_cus, s, &ctx->last_enc_kernel));
        ctx->last_enc_fused = false;
        ctx->last_enc_slots = true;
        if (ctx->timing && !t_capturing) {
            HIP_TRY(hipEventRecord(ctx->ev[3], s));
            ctx->enc_timed = true;
        }
    }
    if (h_total_bytes) {
        uint32_t flags = 0;
        uint64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&flags, ctx->d_enc_flags(), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&total, d_offsets + nchunks, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *h_total_bytes = total;
        if (flags & 1u)
            return fail(RANS_AMD_E_MODEL, "encode_adaptive_sized: a chunk's counts could not be normalised");
        return encode_flags_status(flags);
    }
    return RANS_AMD_OK;
}

extern "C" {

int rans_amd_encode_adaptive(rans_amd_ctx *ctx, const void *d_syms, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                             uint32_t scale_bits, void *d_out, uint64_t out_cap, uint64_t *d_offsets, uint32_t *d_lengths,
                             uint16_t *d_chunk_freqs, uint64_t *h_total_bytes, void *stream)
{
    return encode_adaptive_impl(ctx, RANS_AMD_FMT_BYTE, d_syms, n, n_ways, chunk_syms, scale_bits, d_out, out_cap, d_offsets, d_lengths,
                                d_chunk_freqs, h_total_bytes, stream);
}

int rans_amd_encode_adaptive_fmt(rans_amd_ctx *ctx, int format, const void *d_syms, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                                 uint32_t scale_bits, void *d_out, uint64_t out_cap, uint64_t *d_offsets, uint32_t *d_lengths,
                                 uint16_t *d_chunk_freqs, uint64_t *h_total_bytes, void *stream)
{
    return encode_adaptive_impl(ctx, format, d_syms, n, n_ways, chunk_syms, scale_bits, d_out, out_cap, d_offsets, d_lengths,
                                d_chunk_freqs, h_total_bytes, stream);
}

int rans_amd_encode_adaptive_sized(rans_amd_ctx *ctx, int format, const void *d_syms, uint64_t n, uint32_t n_ways, uint32_t chunk_syms,
                                   uint32_t scale_bits, void *d_out, uint64_t out_cap, uint64_t *d_offsets,
                                   uint32_t *d_lengths, uint16_t *d_chunk_freqs, uint64_t *h_total_bytes, void *stream)
{
    return encode_adaptive_sized_impl(ctx, format, d_syms, n, n_ways, chunk_syms, scale_bits, d_out, out_cap, d_offsets,
                                      d_lengths, d_chunk_freqs, h_total_bytes, stream);
}

uint64_t rans_amd_encode_adaptive_sized_bound(int format, uint64_t n, uint32_t n_ways, uint32_t chunk_syms)
{
    // every chunk in a worst-case piece (a chunk's own bound is never larger): what no input can exceed
    const uint64_t nchunks = rans_amd_num_chunks(n, chunk_syms);
    return nchunks ? nchunks * encode_slot_bytes(format, n, n_ways, chunk_syms) : 16;
}

int rans_amd_decode_adaptive(rans_amd_ctx *ctx, const void *d_container, uint64_t container_bytes, const uint64_t *d_offsets,
                             const uint32_t *d_lengths, const uint16_t *d_chunk_freqs, uint64_t n, uint32_t n_ways,
                             uint32_t chunk_syms, uint32_t scale_bits, void *d_out, uint64_t *h_bad_chunks, void *stream)
{
    return decode_adaptive_impl(ctx, RANS_AMD_FMT_BYTE, d_container, container_bytes, d_offsets, d_lengths, d_chunk_freqs, n, n_ways,
                                chunk_syms, scale_bits, d_out, h_bad_chunks, stream);
}

int rans_amd_decode_adaptive_fmt(rans_amd_ctx *ctx, int format, const void *d_container, uint64_t container_bytes,
                                 const uint64_t *d_offsets, const uint32_t *d_lengths, const uint16_t *d_chunk_freqs, uint64_t n,
                                 uint32_t n_ways, uint32_t chunk_syms, uint32_t scale_bits, void *d_out, uint64_t *h_bad_chunks,
                                 void *stream)
{
    return decode_adaptive_impl(ctx, format, d_container, container_bytes, d_offsets, d_lengths, d_chunk_freqs, n, n_ways, chunk_syms,
                                scale_bits, d_out, h_bad_chunks, stream);
}

/* ---- host-buffer wrappers: one raw reference-format stream ----------------- */

int rans_amd_encode_host(rans_amd_ctx *ctx, const rans_amd_model *model, const void *syms, uint64_t n,
                         uint32_t n_ways, uint8_t *buf, uint64_t cap, uint64_t *out_len)
{
    if (!ctx || !model || !buf || !out_len || (n && !syms))
        return fail(RANS_AMD_E_ARG, "encode_host: NULL argument");
    if (n > 0xffffffffull - 64)
        return fail(RANS_AMD_E_UNSUPPORTED, "encode_host: a single stream is limited to 2^32 symbols");
    const int format = model->host.format;
    const int sb = model->host.sym_bytes;
    if (n == 0) {
        // nothing to code: the reference's loops run zero times and then flush the N untouched states,
        // lanes N-1 .. 0, each still at its initial value L (rans_byte.h:56-59,93-105; rans64.h:65-68,96-103;
        // rans_word_sse41.h:75-78,96-106) -- a constant, so it is written here without a kernel
        if (!ways_supported(format, n_ways))
            return fail(RANS_AMD_E_UNSUPPORTED, "encode_host: n_ways must be in 1..512");
        const uint64_t total = empty_stream(format, n_ways, nullptr);
        if (total > cap)
            return fail(RANS_AMD_E_SPACE, "encode_host: buffer too small");
        empty_stream(format, n_ways, buf + (cap - total));
        *out_len = total;
        return RANS_AMD_OK;
    }
    const uint32_t chunk = (uint32_t)n;
    const uint64_t bound = rans_amd_chunk_bound(format, chunk, n_ways);
    DeviceGuard guard(ctx->device);
    // staging buffers live in the context (grown on demand, freed with it): no hipMalloc / hipFree per call
    std::lock_guard<std::mutex> host_lock(ctx->host_mu);
    int rc = ctx->host_in.reserve((size_t)(n * sb + 256));
    if (rc == RANS_AMD_OK)
        rc = ctx->host_out.reserve((size_t)bound + 256);
    if (rc == RANS_AMD_OK)
        rc = ctx->host_idx.reserve(128);
    if (rc != RANS_AMD_OK)
        return rc;
    uint8_t *d_in = static_cast<uint8_t *>(ctx->host_in.ptr), *d_out = static_cast<uint8_t *>(ctx->host_out.ptr);
    uint64_t *d_off = static_cast<uint64_t *>(ctx->host_idx.ptr);
    uint32_t *d_len = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->host_idx.ptr) + 64);
    hipError_t e = hipMemcpy(d_in, syms, (size_t)(n * sb), hipMemcpyHostToDevice);
    uint64_t total = 0;
    if (e != hipSuccess)
        rc = hip_fail(e, "encode_host: device staging");
    if (rc == RANS_AMD_OK)
        rc = rans_amd_encode(ctx, model, d_in, n, n_ways, chunk, d_out, bound, d_off, d_len, &total, nullptr);
    if (rc == RANS_AMD_OK) {
        if (total > cap)
            rc = fail(RANS_AMD_E_SPACE, "encode_host: buffer too small");
        else {
            e = hipMemcpy(buf + (cap - total), d_out, (size_t)total, hipMemcpyDeviceToHost);
            if (e != hipSuccess)
                rc = hip_fail(e, "encode_host: copy back");
            *out_len = total;
        }
    }
    return rc;
}

int rans_amd_decode_host(rans_amd_ctx *ctx, const rans_amd_model *model, const uint8_t *stream_bytes, uint64_t len,
                         uint64_t n, uint32_t n_ways, void *out)
{
    if (!ctx || !model || !stream_bytes || (n && !out))
        return fail(RANS_AMD_E_ARG, "decode_host: NULL argument");
    if (n > 0xffffffffull - 64 || len > 0xffffffffull)
        return fail(RANS_AMD_E_UNSUPPORTED, "decode_host: n < 2^32 and len < 2^32 required");
    if (n == 0) { // the stream of an empty input: exactly the N initial states (see encode_host)
        if (!ways_supported(model->host.format, n_ways))
            return fail(RANS_AMD_E_UNSUPPORTED, "decode_host: n_ways must be in 1..512");
        uint8_t want[512 * 8];
        const uint64_t total = empty_stream(model->host.format, n_ways, want);
        if (len != total || memcmp(want, stream_bytes, (size_t)total) != 0)
            return fail(RANS_AMD_E_CORRUPT, "decode_host: n == 0 but the stream is not N untouched states");
        return RANS_AMD_OK;
    }
    const int sb = model->host.sym_bytes;
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> host_lock(ctx->host_mu); // (staging buffers of the context, see encode_host)
    int rc = ctx->host_in.reserve((size_t)len + 256);
    if (rc == RANS_AMD_OK)
        rc = ctx->host_out.reserve((size_t)(n * sb + 256));
    if (rc == RANS_AMD_OK)
        rc = ctx->host_idx.reserve(128);
    if (rc != RANS_AMD_OK)
        return rc;
    uint8_t *d_in = static_cast<uint8_t *>(ctx->host_in.ptr), *d_out = static_cast<uint8_t *>(ctx->host_out.ptr);
    uint64_t *d_off = static_cast<uint64_t *>(ctx->host_idx.ptr);
    uint32_t *d_len = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->host_idx.ptr) + 64);
    const uint64_t offs[2] = {0, len};
    const uint32_t len32 = (uint32_t)len;
    hipError_t e = hipMemcpy(d_in, stream_bytes, (size_t)len, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, offs, sizeof(offs), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_len, &len32, 4, hipMemcpyHostToDevice);
    if (e != hipSuccess)
        rc = hip_fail(e, "decode_host: device staging");
    uint64_t bad = 0;
    if (rc == RANS_AMD_OK)
        rc = rans_amd_decode(ctx, model, d_in, len, d_off, d_len, n, n_ways, (uint32_t)n, d_out, &bad, nullptr);
    if (rc == RANS_AMD_OK || rc == RANS_AMD_E_CORRUPT) {
        e = hipMemcpy(out, d_out, (size_t)(n * sb), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = hip_fail(e, "decode_host: copy back");
    }
    return rc;
}

/* ---- measurement ----------------------------------------------------------- */

int rans_amd_set_timing(rans_amd_ctx *ctx, int enabled)
{
    if (!ctx)
        return fail(RANS_AMD_E_ARG, "ctx is NULL");
    ctx->timing = enabled != 0;
    ctx->wave_clocks_on = enabled == 2;
    return RANS_AMD_OK;
}

int rans_amd_last_kernel_ms(rans_amd_ctx *ctx, float *decode_ms, float *encode_ms)
{
    if (!ctx)
        return fail(RANS_AMD_E_ARG, "ctx is NULL");
    DeviceGuard guard(ctx->device);
    if (decode_ms) {
        *decode_ms = -1.0f;
        if (ctx->dec_timed) {
            HIP_TRY(hipEventSynchronize(ctx->ev[1]));
            HIP_TRY(hipEventElapsedTime(decode_ms, ctx->ev[0], ctx->ev[1]));
        }
    }
    if (encode_ms) {
        *encode_ms = -1.0f;
        if (ctx->enc_timed) {
            HIP_TRY(hipEventSynchronize(ctx->ev[3]));
            HIP_TRY(hipEventElapsedTime(encode_ms, ctx->ev[2], ctx->ev[3]));
        }
    }
    return RANS_AMD_OK;
}

const char *rans_amd_last_decode_kernel(rans_amd_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

const char *rans_amd_last_encode_kernel(rans_amd_ctx *ctx, int *fused_placement)
{
    if (fused_placement)
        *fused_placement = ctx && ctx->last_enc_slots ? 2 : (ctx && ctx->last_enc_fused ? 1 : 0);
    return ctx ? ctx->last_enc_kernel : "";
}

int rans_amd_launch_spans(rans_amd_ctx *ctx, uint32_t count, double *span_ms, void *stream)
{
    if (!ctx || !span_ms || count == 0 || count > kWorkSlots / 2)
        return fail(RANS_AMD_E_ARG, "launch_spans: 1 <= count <= 32 launches");
    DeviceGuard guard(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<uint32_t> host((size_t)kWorkSlots * kWorkSlotWords);
    HIP_TRY(hipMemcpyAsync(host.data(), ctx->d_words + 256, host.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < count; ++i) {
        span_ms[i] = -1.0;
        if (ctx->launch_seq < count - i)
            continue; // fewer launches than asked for
        const uint32_t seq = ctx->launch_seq - (count - i); // oldest first
        unsigned long long rec[2];
        memcpy(rec, &host[(size_t)(seq % kWorkSlots) * kWorkSlotWords + kWorkPools * kWorkPoolStride], sizeof(rec));
        if (rec[1])
            span_ms[i] = (double)(rec[1] - ~rec[0]) * 1e-5; // 100 MHz ticks
    }
    return RANS_AMD_OK;
}

int rans_amd_last_wave_clocks(rans_amd_ctx *ctx, rans_amd_wave_clocks *out)
{
    if (!ctx || !out)
        return fail(RANS_AMD_E_ARG, "last_wave_clocks: NULL argument");
    *out = ctx->wave_clocks;
    return out->waves ? RANS_AMD_OK : fail(RANS_AMD_E_ARG, "last_wave_clocks: no instrumented decode has run");
}

} // extern "C"
