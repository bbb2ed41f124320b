// encode_wave.hip -- the wave-per-chunk encoder (see decode_wave.hip for the mapping of the
// reference's loops onto a wavefront).

#include "device_common.hpp"
#include "launchers.hpp"
#include <type_traits>

namespace rans_amd {

namespace {

// ===========================================================================
// Encoder: mirror image of the decoder.  One wave per chunk, symbols visited
// last round first; within a sub-step the lanes that must emit compact their
// units below the write cursor in ascending lane order (the decoder will read
// them back in exactly that order).
// ===========================================================================

#include "encode_common.hpp"

constexpr int kEncAliasLdsThreads = 1024; // FMT_ALIAS_LDS: 16 waves share the (up to 160 KiB) tables of a CU

// ---------------------------------------------------------------------------
// Fused placement (EncParams::status).  The container's layout is the oracle's: chunk c starts at the sum of the
// 16-byte aligned lengths of the chunks before it.  A chunk's length is known when its last symbol is coded; the
// three-kernel path therefore codes into per-chunk scratch slots, scans the lengths (k_layout) and copies everything
// once more (k_compact: 0.33 ms per GiB on top of 0.66 ms of coding).  Here the copy runs INSIDE the coding kernel,
// beside the arithmetic: the last wave(s) of every block are copiers.
//   encoder wave, chunk done:  stores complete (s_waitcnt), status[c] = AGGREGATE | align16(len), {c, len} into the
//                              block's mailbox (LDS);
//   copier wave:               takes {c, len} from the mailbox; looks back over status[c-1], status[c-2], ... 64 at a
//                              time until a published inclusive PREFIX is met (chunk 0's virtual predecessor is one),
//                              waiting for predecessors that are still being coded; publishes status[c] = PREFIX |
//                              (base + align16(len)), writes offsets[c]; copies the stream from the scratch slot to
//                              out + base, 8 KiB per trip (eight 16-byte loads per lane in flight, source unaligned).
// Why copiers of the SAME block: the L2 caches of different XCDs are not coherent for ordinary stores -- a reader on
// another XCD would need the encoder's stream stores written through (measured: +0.33 ms on the word encoder) -- but
// a block lives on one CU, so its copier reads what its encoders wrote through the same L2.  The status words are
// the only data that cross XCDs: agent-scope atomics.
// Forward progress: chunks are claimed in ascending order per pool (blockIdx % kWorkPools) by running waves only, an
// encoder waits for nothing but a free mailbox entry of its own block, and a copier only for chunks smaller than its
// own -- every chain of waits ends at the smallest unfinished chunk, whose encoder is running.
// ---------------------------------------------------------------------------
// (kStAggregate / kStPrefix / kStValue: device_common.hpp -- the lane encoders place their batches the same way)

__device__ __forceinline__ void place_and_copy(const EncParams &p, uint64_t chunk, uint64_t sa, uint32_t len, uint32_t lane)
{
    const unsigned long long alen = (len + 15u) & ~15u;
    unsigned long long base = 0;
    status_lookback(p.status, chunk, lane, p.flags, p.wait_ticks, 32u, base); // (a wait that gave up: bit 5, the launch has failed)
    if (lane == 0) {
        __hip_atomic_store(p.status + chunk, kStPrefix | (base + alen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.offsets[chunk] = base;
        if (chunk + 1 == p.nchunks)
            p.offsets[p.nchunks] = base + len;
    }
    if (base + len > p.out_cap) { // wave-uniform
        if (lane == 0)
            atomicOr(p.flags, 2u);
        return;
    }
    u32x4 RANS_GLOBAL *dst = reinterpret_cast<u32x4 RANS_GLOBAL *>(reinterpret_cast<uint64_t>(p.out) + base);
    const uint32_t n16 = (len + 15u) >> 4; // (the last piece may read up to 15 bytes of the next slot: scratch is padded)
    constexpr int kCopyDepth = 8; // (16-byte loads a lane keeps in flight)
    for (uint32_t i0 = lane; i0 < n16; i0 += 64u * kCopyDepth) { // (nt: the slot is read once, the container written once)
        u32x4 v[kCopyDepth];
#pragma unroll
        for (int j = 0; j < kCopyDepth; ++j)
            if (i0 + 64u * j < n16) {
                v[j] = __builtin_nontemporal_load(reinterpret_cast<gvec_cptr>(sa + 16ull * (i0 + 64u * j)));
            }
#pragma unroll
        for (int j = 0; j < kCopyDepth; ++j)
            if (i0 + 64u * j < n16) {
                __builtin_nontemporal_store(v[j], dst + i0 + 64u * j);
            }
    }
}

// waves per SIMD the fused kernel is compiled for: 8 (64 VGPRs: 4 blocks of 8 waves per CU) with one state per lane, 4 and 2
// with 2-4 and 8 states per lane -- at 64 VGPRs those spilled 20 to 785 registers (the 512-way rans64 encoder)
constexpr int enc_fused_waves_per_simd(int K) { return K == 1 ? 8 : (K <= 4 ? 4 : 2); }

// MODE 0: static chunk striding into scratch slots (k_layout + k_compact follow; or, with EncParams::slot_layout, nothing
//         follows: the slots ARE the container).
// MODE 1: fused placement -- dynamic chunk claims, the last wave(s) of a block copy the finished streams to their place.
// MODE 2: slot layout -- dynamic chunk claims, every wave codes, a chunk stays in its slot (EncParams::slot_layout):
//         what the reference does with each of its buffers (main.cpp:176-188), every stream byte written exactly once.
// MODE 3: MODE 2 with slots of the CALLER's size (EncParams::ovf_ctl): before anything is stored the coder makes sure it
//         still lies inside the slot -- exactly where the stream is staged in LDS (the flush knows what it is about to
//         write), by the round's worst case elsewhere -- and a chunk that does not fit is abandoned and listed for the
//         second launch (EncParams::redo), which codes the listed chunks into worst-case slots behind the sized ones.
template <int FMT, int K, int MODE>
__global__ void __launch_bounds__(FMT == FMT_ALIAS_LDS ? kEncAliasLdsThreads : (MODE != 0 ? kEncFusedThreads : kEncBlockThreads),
                                  (MODE != 0 && FMT != FMT_ALIAS_LDS) ? enc_fused_waves_per_simd(K) : 1)
    k_encode(const EncParams p)
{
    using Tr = FmtTraits<FMT>;
    using state_t = typename Tr::state_t;
    constexpr bool FUSED = MODE == 1;
    constexpr bool DYNAMIC = MODE != 0; // chunks are claimed from EncParams::claims
    constexpr bool SIZED = MODE == 3;   // slots of the caller's size: overflow checks, abandoned chunks, the redo launch
    constexpr uint32_t kMaxEmit = kIsR64<FMT> ? 4u : 2u; // bytes one symbol can push out of one state (scale_bits <= 16 for the byte formats)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t redo_todo = 0; // redo launch: listed chunks this launch codes
    if constexpr (SIZED) {
        if (p.redo) { // (the first launch is complete: stream order)
            const uint32_t listed = __hip_atomic_load(p.ovf_ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            redo_todo = uniform(listed < p.ovf_cap ? listed : p.ovf_cap);
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                p.offsets[p.nchunks] = p.ovf_base + (uint64_t)redo_todo * p.slot_bytes; // the container's end
                if (listed > p.ovf_cap)
                    atomicOr(p.flags, 2u); // the overflow region is too small: RANS_AMD_E_SPACE
            }
            if (redo_todo == 0u) // the usual case: nothing overflowed, the launch ends before it has loaded a table
                return;
        }
    }

    // word format: the 256 WordEncRec of the full-wave path come first (LDS address = sym << 4),
    // the per-symbol EncRec table of the general path behind them
    // (byte format: 256 records {rcp, cmpl | rshift << 24, bias, x_max} of enc_byte_full, built from the EncRec table below)
    constexpr uint32_t kWordRecBytes = (FMT == FMT_WORD || FMT == FMT_BYTE) ? 256u * 16u : 0u; // (word: 2 KiB of it in use)
    if constexpr (FMT == FMT_WORD) {
        if (p.word_enc_recs) { // (absent for alphabets beyond 256 symbols: they never take the full-wave path)
            const uint4 *g = reinterpret_cast<const uint4 *>(p.word_enc_recs); // 256 records of 16 bytes
            uint4 *l = reinterpret_cast<uint4 *>(smem);
            for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x)
                l[i] = g[i];
        }
    }
    const uint32_t nrecs = p.nsyms < 256u ? 256u : p.nsyms; // byte alphabets: 256 entries, freq 0 behind nsyms
    if constexpr (FMT == FMT_ALIAS_LDS) {
        // 8-byte records (the host pads them to nrecs entries), then alias_remap as u16[M]; both in 16-byte pieces
        const uint4 *g = reinterpret_cast<const uint4 *>(p.alias_recs8);
        uint4 *l = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < nrecs / 2u; i += blockDim.x)
            l[i] = g[i];
        const uint4 *gr = reinterpret_cast<const uint4 *>(p.alias_remap16);
        uint4 *lr = reinterpret_cast<uint4 *>(smem + (size_t)nrecs * 8u);
        for (uint32_t i = threadIdx.x; i < (2u << p.scale_bits) / 16u; i += blockDim.x)
            lr[i] = gr[i];
    } else if (!(FMT == FMT_BYTE && p.chunk_freqs) && FMT != FMT_WORDA) { // (per-chunk models: every wave builds its own records below)
        const uint4 *g = reinterpret_cast<const uint4 *>(p.enc_recs);
        uint4 *l = reinterpret_cast<uint4 *>(smem + kWordRecBytes);
        for (uint32_t i = threadIdx.x; i < p.nsyms; i += blockDim.x)
            l[i] = g[i];
        for (uint32_t i = p.nsyms + threadIdx.x; i < 256u; i += blockDim.x)
            l[i] = uint4{0u, 0u, 0u, 0u};
        if constexpr (FMT == FMT_BYTE) { // EncRec {freq | rshift << 24, bias, rcp, -} -> the full-wave path's records
            uint4 *f = reinterpret_cast<uint4 *>(smem);
            for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) {
                const uint4 r = i < p.nsyms ? g[i] : uint4{0u, 0u, 0u, 0u};
                const uint32_t freq = r.x & 0xffffffu;
                f[i] = freq ? uint4{r.z, (((1u << p.scale_bits) - freq) & 0xffffffu) | (r.x & 0xff000000u), r.y, freq << (31u - p.scale_bits)}
                            : uint4{0u, 0xffffffffu, 0u, 0xffffffffu}; // no frequency: nothing leaves, x stays, v_max sees it
            }
        }
    }
    // (the mailbox sits behind the tables in LDS, or -- tables that fill the LDS -- in this block's slot of a global array:
    //  coders and copier of a block run on one CU, i.e. behind one L2)
    EncMailbox *mb = p.mailbox_global ? reinterpret_cast<EncMailbox *>(p.mailbox_global + (size_t)blockIdx.x * kEncMailboxStride)
                                      : reinterpret_cast<EncMailbox *>(smem + p.mailbox_off);
    // scratch ring: drained[w] = chunks of coding wave w that the copier has moved out of their slots.  Raw LDS address,
    // explicit DS instructions on both sides: the protocol must not depend on how the compiler threads a lane-0 branch
    // through the code around it
    const uint32_t drained_lds = (uint32_t)(uintptr_t)(RANS_LDS uint8_t *)(smem + p.mailbox_off + kEncMailboxBytes);
    if constexpr (FUSED) {
        if (!p.mailbox_global && threadIdx.x < kEncFusedLdsBytes / 4u) // (the global ones are zeroed by the host)
            reinterpret_cast<uint32_t *>(mb)[threadIdx.x] = 0u;
    }
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    // fused: the last wave of a block (the last two of a 16-wave block) copies, the others encode
    const uint32_t copiers = FUSED ? ((blockDim.x >> 6) >= 16u ? kEncFusedCopiers16 : 1u) : 0u;
    const uint32_t waves_per_block = (blockDim.x >> 6) - copiers;
    const uint32_t N = p.n_ways; // <= 64 * K; lanes idx >= N idle

    if constexpr (FUSED) {
        if (wave >= waves_per_block) { // ---- copier wave
            for (;;) {
                uint32_t ex = 0, ey = 0;
                if (!mailbox_pop(mb, lane, waves_per_block, ex, ey, p.flags, p.wait_ticks))
                    break;
                const uint64_t chunk = ex - 1u;
                if (p.ring_slots) { // entry: length | slot of the ring << 26 | coding wave << 28
                    const uint32_t len = ey & 0x3ffffffu, j = (ey >> 26) & 3u, w = ey >> 28;
                    const uint64_t slot = ((uint64_t)blockIdx.x * waves_per_block + w) * p.ring_slots + j;
                    // a ring slot is read again and again, each time with another chunk's bytes in it: whatever this CU's
                    // vector L1 still holds of the slot's previous occupant must go (the coders' stores are in L2)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    place_and_copy(p, chunk, reinterpret_cast<uint64_t>(p.scratch) + (slot + 1u) * p.slot_bytes - len, len, lane);
                    // the copy's loads have returned (its stores could not have been issued otherwise): the slot is free.
                    // (every lane executes the add, lane 0 with 1 and the others with 0)
                    asm volatile("ds_add_u32 %0, %1" ::"v"(drained_lds + 4u * w), "v"(lane == 0 ? 1u : 0u) : "memory");
                } else {
                    const uint64_t sa = reinterpret_cast<uint64_t>(p.scratch) + (chunk + 1u) * p.slot_bytes - ey;
                    place_and_copy(p, chunk, sa, ey, lane);
                }
            }
            return;
        }
    }

    EncTables<FMT> T;
    T.recs = reinterpret_cast<const uint4 *>(smem + kWordRecBytes);
    const bool adaptive = (FMT == FMT_BYTE && p.chunk_freqs) || FMT == FMT_WORDA; // one model per chunk
    if (adaptive)
        T.recs = reinterpret_cast<const uint4 *>(smem + wave * kAdaptEncWaveLds);
    T.remap16 = reinterpret_cast<const uint16_t *>(smem + (size_t)nrecs * 8u);
    T.scale_bits = p.scale_bits;
    T.nsyms = p.nsyms;
    T.swap_sel = 0x0c0c0001u; // (the low two bytes swapped, zeros above)
    asm volatile("" : "+v"(T.swap_sel));
    T.split_sel = 0x0c000c01u; // (byte 1 of x in byte 0, byte 0 of x in byte 2)
    asm volatile("" : "+v"(T.split_sel));

    bool bad = false;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    // per-lane constants of the 4x4 byte transpose (same lane mapping as the decoder's stores)
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t in_lane_off = (lane & 3u) * N + (lane & ~3u);

    uint32_t coded = 0; // chunks this wave has coded (scratch ring: chunk number `coded` goes into slot coded % R)
    uint32_t redo_slot = 0; // redo launch: the overflow slot of the chunk in hand
    for (uint64_t chunk_v = (uint64_t)blockIdx.x * waves_per_block + wave;; chunk_v += total_waves) {
        if constexpr (DYNAMIC) {
            // scratch ring: the slot about to be reused must have been drained -- BEFORE the claim, so that a claimed chunk
            // is always in the hands of a running wave (the forward-progress argument above: the copier that drains this
            // wave's old chunk waits only for chunks smaller than ones its own block has claimed, and those are being coded)
            if (FUSED && p.ring_slots && coded >= p.ring_slots) {
                SpinWatch watch(p.wait_ticks);
                for (;;) {
                    uint32_t got_drained;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got_drained) : "v"(drained_lds + 4u * wave) : "memory");
                    if (uniform(got_drained) + p.ring_slots > coded)
                        break;
                    if (watch.expired(p.flags)) {
                        if (lane == 0)
                            atomicOr(p.flags, 256u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // ascending claims, one counter per pool of blocks (a 64-byte line each, behind the status words; pool q
            // hands out the chunks c with c % npools == q): a single counter retires ~90 claims per microsecond
            const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
            const uint32_t pool = blockIdx.x % npools;
            uint32_t got = 0;
            if (SIZED && p.redo) { // the i-th listed chunk goes into the i-th overflow slot
                if (lane == 0)
                    got = atomicAdd(p.ovf_ctl + kWorkPoolStride, 1u);
                redo_slot = uniform(got);
                if (redo_slot >= redo_todo)
                    break;
                chunk_v = p.ovf_list[redo_slot];
            } else {
                if (lane == 0)
                    got = atomicAdd(p.claims + kWorkPoolStride * pool, 1u);
                chunk_v = (uint64_t)uniform(got) * npools + pool;
            }
        }
        if (chunk_v >= p.nchunks)
            break;
        const uint64_t chunk = uniform64(chunk_v);
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + first * p.sym_bytes;
        const uint32_t ring_j = (FUSED && p.ring_slots) ? coded % p.ring_slots : 0u;
        const uint64_t slot_no = (FUSED && p.ring_slots) ? ((uint64_t)blockIdx.x * waves_per_block + wave) * p.ring_slots + ring_j : chunk;
        // (sized slots, redo launch: the overflow region behind the sized slots, one worst-case slot per listed chunk)
        const uint64_t slot_at = (SIZED && p.redo) ? p.ovf_base + (uint64_t)redo_slot * p.slot_bytes : uniform64(slot_no) * p.slot_bytes;
        uint8_t RANS_GLOBAL *slot = (uint8_t RANS_GLOBAL *)p.scratch + slot_at;
        uint32_t wp = (uint32_t)p.slot_bytes;
        bool ovf = false; // SIZED, wave-uniform: the chunk's stream does not fit its slot
        // SIZED, the coders that store every round's units themselves: 0, or ~0 from the pair of rounds on before which the slot no
        // longer had room for what two rounds can emit at most -- OR-ed into the renormalisation thresholds, so that nothing
        // leaves the states any more (no branch out of the unrolled loops: a `break` there cost the 4096-symbol alias coder
        // a third of its speed)
        uint32_t dead = 0u;
        ++coded;

        if (adaptive) { // this chunk's model -> this wave's records (RansEncSymbolInit per chunk, main.cpp:159-162)
            if constexpr (FMT == FMT_WORDA)
                adapt_build_enc_word(p.chunk_freqs + chunk * 256u, lane, const_cast<uint4 *>(T.recs));
            else
                adapt_build_enc(p.chunk_freqs + chunk * 256u, p.scale_bits, lane, const_cast<uint4 *>(T.recs));
        }

        state_t x[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            x[k] = Tr::kL; // RansEncInit / RansWordEncInit / Rans64EncInit

        const uint32_t rounds = uniform(nsym / N);
        const uint32_t tail = uniform(nsym - rounds * N);
        // Fast input path: full waves, u8 symbols, dword-aligned rows -> symbols of 4 rounds
        // arrive as one coalesced dword per lane and are transposed in registers; loads run
        // one super-group (16 rounds) ahead of the arithmetic.
        // (N == 64 K exactly: with N = 192 on the K = 4 kernel the fourth sub-step has no lanes)
        const bool fast_in = p.sym_bytes == 1 && N == 64u * K &&
                             ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 3u) == 0;
        // (the word path addresses its record table by raw LDS address: dynamic LDS must start at 0)
        const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem == 0u;
        // The same for u16 symbols (alias coding over more than 256 symbols): two rounds per dword, lane pairs
        // swap halves (the mirror image of the decoder's OUT_FAST16 stores).
        // (K <= 2: sixteen rounds in flight twice over are 32 K registers)
        const bool fast_in16 = (kIsAlias<FMT> || FMT == FMT_WORD) && K <= 2 && p.sym_bytes == 2 && N == 64u * K &&
                               ((reinterpret_cast<uintptr_t>(p.syms) | (p.chunk_syms * 2u)) & 3u) == 0;
        const uint32_t fast_rounds =
            ((fast_in && (FMT != FMT_WORD || lds_at_zero)) || fast_in16) ? (rounds & ~15u) : 0u;

        // rounds from last to first; round `rounds` is the partial one
        for (uint32_t rr = rounds + 1; rr-- > fast_rounds;) {
            const uint32_t cnt = (rr < rounds) ? N : tail;
            if (cnt == 0)
                continue;
            if constexpr (SIZED) {
                if (wp < N * kMaxEmit) { // (what a round can emit at most)
                    ovf = true;
                    break;
                }
            }
            const uint8_t RANS_GLOBAL *rsrc = src + (uint64_t)rr * N * p.sym_bytes;
            uint32_t sym[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t idx = k * 64u + lane;
                sym[k] = 0;
                if (idx < cnt)
                    sym[k] = p.sym_bytes == 1 ? (uint32_t)rsrc[idx]
                                              : (uint32_t) reinterpret_cast<const uint16_t RANS_GLOBAL *>(rsrc)[idx];
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const uint32_t idx = k * 64u + lane;
                enc_substep<FMT>(T, x[k], sym[k], idx < cnt, slot, wp, bad);
            }
        }

        uint32_t worst = 0; // word fast path: max of cmpl_sh, > 0x0fffffff iff a symbol has no record
        bool mirrored = false; // byte format, staged: lane l holds stream 63 - l during the fast loop (enc_byte_full_staged)
        if (fast_rounds && fast_in16) {
            if constexpr ((kIsAlias<FMT> || FMT == FMT_WORD) && K <= 2) {
                // lane l of a pair loads the dword {row 2j + (l & 1), columns l & ~1 and (l & ~1) + 1}
                const uint32_t sel16 = (lane & 1u) ? 0x03020706u : 0x05040100u;
                const uint32_t in_off16 = ((lane & 1u) * N + (lane & ~1u)) * 2u;
                uint32_t cur[8][K], nxt[8][K];
                auto load_super16 = [&](uint32_t (&dstq)[8][K], uint32_t sg) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            dstq[j][k] = *reinterpret_cast<const uint32_t RANS_GLOBAL *>(
                                src + (uint64_t)(sg * 16u + j * 2u) * N * 2u + in_off16 + k * 128u);
                };
                uint32_t sg = fast_rounds >> 4;
                load_super16(cur, sg - 1);
                while (sg-- > 0) {
                    if (SIZED && dead)
                        break;
                    if (sg > 0)
                        load_super16(nxt, sg - 1);
#pragma unroll
                    for (int j = 7; j >= 0; --j) {
                        uint32_t t[K]; // this lane's symbol of row 2j (low half) and of row 2j + 1 (high half)
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            t[k] = __builtin_amdgcn_perm(quad_perm<1, 0, 3, 2>(cur[j][k]), cur[j][k], sel16);
                        if constexpr (SIZED) // (before every PAIR of rounds: a check per round is 4 of the loop's 36 instructions)
                            dead = uniform(wp) < 2u * 64u * K * kMaxEmit ? ~0u : dead;
#pragma unroll
                        for (int h = 1; h >= 0; --h) {
#pragma unroll
                            for (int k = K - 1; k >= 0; --k)
                                enc_substep<FMT, false, true>(T, x[k], (t[k] >> (16 * h)) & 0xffffu, true, slot, wp, bad, dead);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            cur[j][k] = nxt[j][k];
                }
            }
        } else if (fast_rounds) {
            uint32_t rec_mask = 0xff0u, swap_sel = 0x0c0c0001u; // (v_perm selector: the low two bytes swapped, zeros above)
            uint32_t k3v = 4u; // (word and byte format: 16-byte records; the name is round 3's, when the word format had eight)
            asm volatile("" : "+v"(k3v)); // (SDWA takes no literal; a VGPR operand is also the faster VALU form)
            (void)k3v;
            asm volatile("" : "+v"(rec_mask)); // keep the mask in a VGPR (a literal operand costs a slower VALU form)
            asm volatile("" : "+v"(swap_sel));
            // byte format: the hand-written sub-step needs its records at LDS address 0 and the one model of the launch
            const bool byte_asm = FMT == FMT_BYTE && lds_at_zero && !adaptive;
            (void)swap_sel;
            (void)byte_asm;
            // word format, one state per lane: stream staging (enc_word_full_staged).  Invariant between groups of four
            // rounds: memory holds the stream from wp up; what a flush writes below wp (< 16 bytes, whatever the window
            // held) is written again, correctly, by the next flush, and at the end by the state flush.
            constexpr bool kStageW = FMT == FMT_WORD && K == 1;
            // (byte format: 16 rounds emit at most (8 + 16 scale_bits) / 8 bytes per lane, 31 at 15 bits: 1984 bytes fit the
            //  window, the 33 x 64 of 16-bit probabilities do not -- those models flush every eight rounds: 17 x 64)
            constexpr bool kStageB = FMT == FMT_BYTE && K == 1;
            const bool stage_b = kStageB && byte_asm && p.scale_bits <= 16u;
            // (alias tables in LDS, byte symbols: windows where the launcher found room behind the tables, EncParams::stage_off)
            constexpr bool kStageA = FMT == FMT_ALIAS_LDS && K == 1;
            const bool stage_a = kStageA && p.stage_off != 0u && p.scale_bits <= 16u;
            const bool flush8 = kStageA || (kStageB && p.scale_bits == 16u); // (alias: two bytes per lane and round at most, whatever the input)
            (void)flush8;
            const bool stage_on = kStageW || stage_b || stage_a;
            uint32_t split_sel = 0x0c000c01u; // (v_perm selector: byte 1 of x in byte 0, byte 0 of x in byte 2)
            asm volatile("" : "+v"(split_sel));
            (void)split_sel;
            // The window: LDS [win_base, win_base + 2048).  Its last 16 bytes are the piece of the slot the write offset stands
            // in (what lies above the offset there was produced before and is in memory already); a super-group of sixteen
            // rounds writes downwards from win_top = win_base + 2032 + (wp & 15), at most 1664 bytes (13 words per lane: a
            // symbol adds at most 12 bits to a state of 16..32 bits, a word takes 16 out), so LDS address and slot offset
            // stay congruent modulo 16 and nothing wraps.
            const uint32_t win_base = (FMT == FMT_ALIAS_LDS ? p.stage_off : kWordRecBytes + 256u * (uint32_t)sizeof(EncRec)) + wave * kEncStageBytes;
            constexpr uint32_t kTopPiece = kEncStageBytes - 16u;
            auto lds_u32x4 = [](uint32_t at) { return reinterpret_cast<__attribute__((address_space(3))) u32x4 *>((uintptr_t)at); };
            if (stage_on && (wp & 15u)) { // the tail rounds have stored words themselves: the piece that holds wp goes into the window
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < 4u) {
                    const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t RANS_GLOBAL *>(slot + (wp & ~15u) + 4u * lane));
                    *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)(win_base + kTopPiece + 4u * lane)) = v;
                }
            }
            // LDS [lp, top) -> slot [wp - (top - lp), wp), in whole 16-byte pieces: at most 106 of them -- two passes of the
            // wave, the second one rarely has any lanes (the average is 50 pieces).  What a piece holds below the new write
            // offset is whatever the window held: it is written again, correctly, by the next flush, and at the end by the
            // state flush (same wave, same address: in order).  The lowest piece then becomes the window's top piece.
            auto stage_flush = [&](uint32_t top, uint32_t lp) {
                if constexpr (SIZED) {
                    if (top - lp > wp) { // the bytes in the window do not fit below the write offset: nothing more is stored
                        ovf = true;
                        return;
                    }
                }
                const uint32_t hi = (top + 15u) & ~15u, lo = lp & ~15u;
                const uint32_t to_slot = wp - top; // (wraps; congruent to 0 modulo 16)
                const int32_t a = (int32_t)hi - 16 * (int32_t)(lane + 1u);
                auto piece = [&](int32_t at) {
                    if (at >= (int32_t)lo) {
                        const u32x4 v = *lds_u32x4((uint32_t)at);
                        *reinterpret_cast<u32x4 RANS_GLOBAL *>(const_cast<uint8_t RANS_GLOBAL *>(slot) + ((uint32_t)at + to_slot)) = v;
                        if ((uint32_t)at == lo)
                            *lds_u32x4(win_base + kTopPiece) = v;
                    }
                };
                piece(a);
                if (hi - lo > 1024u) // (wave-uniform)
                    piece(a - 1024);
                wp -= top - lp;
            };
            (void)stage_flush;
            // byte format, staged: the lanes in reverse order.  Lane q of a quad still loads row q of its quad's four columns
            // -- the quad of lanes 4g .. 4g+3 takes columns 60-4g .. 63-4g -- and the transpose hands lane q column 3 - q:
            // its first v_perm reads both dwords byte-reversed (selector indices i -> 3 - i, 4 + i -> 7 - i).
            uint32_t in_off = in_lane_off, tsel1 = sel1;
            if constexpr (kStageB) {
                if (stage_b) {
                    mirrored = true;
                    x[0] = (state_t)__builtin_amdgcn_ds_bpermute((int)((63u - lane) * 4u), (int)x[0]);
                    in_off = (lane & 3u) * N + (60u - (lane & ~3u));
                    tsel1 = (lane & 1u) ? 0x00040206u : 0x05010703u;
                }
            }
            uint32_t cur[4][K], nxt[4][K];
            auto load_super = [&](uint32_t (&dstq)[4][K], uint32_t sg) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        dstq[j][k] = *reinterpret_cast<const uint32_t RANS_GLOBAL *>(
                            src + (uint64_t)(sg * 16u + j * 4u) * N + in_off + k * 64u);
            };
            // (the loop once per reciprocal method of the word format: a branch inside it costs register copies at every join)
            auto fast_loop = [&](auto small_tag, auto track_tag) {
            constexpr bool kSmall = decltype(small_tag)::value; // word: the reciprocal method; byte: staged or not
            constexpr bool kTrack = decltype(track_tag)::value; // the model has byte values without a record (EncParams::dense256 = 0)
            (void)kTrack;
            constexpr bool kStage = kStageW || ((kStageB || kStageA) && kSmall);
            uint32_t sg = fast_rounds >> 4;
            load_super(cur, sg - 1);
            while (sg-- > 0) {
                if (SIZED && (ovf || dead))
                    break;
                if (sg > 0)
                    load_super(nxt, sg - 1);
                uint32_t win_top = win_base + kTopPiece + (uniform(wp) & 15u);
                uint32_t lp = kStageW ? win_top >> 1 : win_top; // (staged path: the coding loop moves this LDS pointer -- the word
                                                                // format's counts 16-bit words -- and stage_flush() sets wp)
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    uint32_t t[K];
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        t[k] = quad_transpose(cur[j][k], tsel1, sel2);
                if constexpr (FMT == FMT_WORD) {
                    // symbol byte J -> LDS address of its record, (sym << 4) + table offset; the record
                    // of the next sub-step is read before the current one is worked on (the asm block is
                    // a scheduling barrier for the compiler)
                    auto rec_at = [&](int step) { // step 0 = (J 3, k K-1), descending
                        const int J = 3 - step / K, k = K - 1 - step % K;
                        uint32_t at; // byte J of t[k], times the eight bytes of a record: one SDWA shift
                        if (J == 3)
                            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(at) : "v"(k3v), "v"(t[k]));
                        else if (J == 2)
                            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(at) : "v"(k3v), "v"(t[k]));
                        else if (J == 1)
                            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(at) : "v"(k3v), "v"(t[k]));
                        else
                            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(at) : "v"(k3v), "v"(t[k]));
                        return *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>((uintptr_t)at); // table at LDS address 0
                    };
                    u32x4 rec = rec_at(0);
                    wp = uniform(wp); // (asm results count as divergent: say what they are, or the "s" operands below get VGPRs)
                    lp = uniform(lp);
#pragma unroll
                    for (int step = 0; step < 4 * K; ++step) {
                        u32x4 now = rec;
                        if (step + 1 < 4 * K)
                            rec = rec_at(step + 1);
                        if constexpr (kStage) {
                            enc_word_full_staged<kSmall, kTrack>(x[K - 1 - step % K], now, lp, worst);
                        } else {
                            if constexpr (SIZED) { // (the staged form is checked, exactly, where it flushes)
                                if (step % (2 * K) == 0)
                                    dead = uniform(wp) < 2u * 64u * K * kMaxEmit ? ~0u : dead; // (two rounds of K x 64 states)
                                now.y |= dead; // x > threshold never holds: no word leaves
                            }
                            enc_word_full<kSmall, kTrack>(x[K - 1 - step % K], now, wp, slot, worst);
                        }
                    }
                    wp = uniform(wp);
                    lp = uniform(lp);
                } else if (FMT == FMT_BYTE && byte_asm) {
                    if constexpr (FMT == FMT_BYTE) { // (the same walk over the 4 K symbols as the word format's)
                        auto rec_at = [&](int step) {
                            const int J = 3 - step / K, k = K - 1 - step % K;
                            uint32_t at; // byte J of t[k] times the sixteen bytes of a record: one SDWA shift (round 4; shift + mask before)
                            if (J == 3)
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(at) : "v"(k3v), "v"(t[k]));
                            else if (J == 2)
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(at) : "v"(k3v), "v"(t[k]));
                            else if (J == 1)
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(at) : "v"(k3v), "v"(t[k]));
                            else
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(at) : "v"(k3v), "v"(t[k]));
                            return *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>((uintptr_t)at); // table at LDS address 0
                        };
                        u32x4 rec = rec_at(0);
                        lp = uniform(lp);
#pragma unroll
                        for (int step = 0; step < 4 * K; ++step) {
                            u32x4 now = rec;
                            if (step + 1 < 4 * K)
                                rec = rec_at(step + 1);
                            if constexpr (kStageB && kSmall) {
                                enc_byte_full_staged<kTrack>(x[K - 1 - step % K], now, lp, worst);
                            } else {
                                if constexpr (SIZED) {
                                    if (step % (2 * K) == 0)
                                        dead = uniform(wp) < 2u * 64u * K * kMaxEmit ? ~0u : dead;
                                    now.w |= dead; // x >= x_max never holds: no byte leaves
                                }
                                enc_byte_full(x[K - 1 - step % K], now, wp, slot, worst, swap_sel);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int J = 3; J >= 0; --J) {
                        if constexpr (SIZED && !(kStageA && kSmall)) {
                            if (J & 1)
                                dead = uniform(wp) < 2u * 64u * K * kMaxEmit ? ~0u : dead; // (two rounds of K x 64 states)
                        }
#pragma unroll
                        for (int k = K - 1; k >= 0; --k)
                            if constexpr (kStageA && kSmall)
                                enc_substep<FMT, true, true, true>(T, x[k], (t[k] >> (8 * J)) & 0xffu, true, slot, lp, bad);
                            else
                                enc_substep<FMT, true, kIsAlias<FMT>>(T, x[k], (t[k] >> (8 * J)) & 0xffu, true, slot, wp, bad, dead);
                    }
                }
                if constexpr ((kStageB || kStageA) && kStage) { // (16-bit models: half a super-group fills the window)
                    if (j == 2 && flush8) {
                        stage_flush(win_top, uniform(lp));
                        win_top = win_base + kTopPiece + (uniform(wp) & 15u);
                        lp = win_top;
                    }
                }
                }
                if constexpr (kStage)
                    stage_flush(win_top, kStageW ? uniform(lp) << 1 : uniform(lp));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        cur[j][k] = nxt[j][k];
            }
            };
            // (the variants without the search for record-less symbols: the word format and the staged byte format over a
            //  model in which every byte value has a frequency -- one VALU instruction of 10 resp. 15 less)
            const bool small = (FMT == FMT_WORD && p.word_small) || (FMT == FMT_BYTE && stage_b) || (FMT == FMT_ALIAS_LDS && stage_a);
            const bool untracked = p.dense256 && (FMT == FMT_WORD || (FMT == FMT_BYTE && stage_b));
            if constexpr (FMT == FMT_WORD || (FMT == FMT_BYTE && K == 1)) {
                if (small && untracked)
                    fast_loop(std::true_type{}, std::false_type{});
                else if (small)
                    fast_loop(std::true_type{}, std::true_type{});
                else if (untracked)
                    fast_loop(std::false_type{}, std::false_type{});
                else
                    fast_loop(std::false_type{}, std::true_type{});
            } else {
                if (small)
                    fast_loop(std::true_type{}, std::true_type{});
                else
                    fast_loop(std::false_type{}, std::true_type{});
            }
        }

        // (a symbol without a record: word format -- bit 31 of the OR over the records' cmpl_sh words; byte format -- its
        //  second word is all ones where a real one keeps cmpl | rshift << 24 below 2^28: the OR of them shows it too, and
        //  v_or issues in the fast VALU class where v_max does not)
        if (worst > (FMT == FMT_WORD ? 0x7fffffffu : 0x0fffffffu))
            bad = true;
        if constexpr (FMT == FMT_BYTE && K == 1) {
            if (mirrored) // back to lane l = stream l
                x[0] = (state_t)__builtin_amdgcn_ds_bpermute((int)((63u - lane) * 4u), (int)x[0]);
        }
        if constexpr (SIZED) {
            if (ovf || dead || wp < N * Tr::kStateBytes) { // abandoned: the redo launch codes this chunk into a worst-case slot
                if (lane == 0)
                    p.ovf_list[atomicAdd(p.ovf_ctl, 1u)] = (uint32_t)chunk;
                continue;
            }
        }
        // flush: lane N-1 first, i.e. lane 0's state ends up first in memory
        // (main.cpp:244-245, main_simd.cpp:298-299)
        wp -= N * Tr::kStateBytes;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = k * 64u + lane;
            if (idx < N) {
                uint8_t RANS_GLOBAL *at = slot + wp + idx * Tr::kStateBytes;
                if constexpr (kIsR64<FMT>) {
                    reinterpret_cast<uint32_t RANS_GLOBAL *>(at)[0] = (uint32_t)x[k];
                    reinterpret_cast<uint32_t RANS_GLOBAL *>(at)[1] = (uint32_t)(x[k] >> 32);
                } else if constexpr (kIsWord<FMT>) {
                    reinterpret_cast<uint16_t RANS_GLOBAL *>(at)[0] = (uint16_t)x[k];
                    reinterpret_cast<uint16_t RANS_GLOBAL *>(at)[1] = (uint16_t)(x[k] >> 16);
                } else {
                    at[0] = (uint8_t)x[k];
                    at[1] = (uint8_t)(x[k] >> 8);
                    at[2] = (uint8_t)(x[k] >> 16);
                    at[3] = (uint8_t)(x[k] >> 24);
                }
            }
        }
        const uint32_t len = (uint32_t)p.slot_bytes - wp;
        if (lane == 0)
            p.lengths[chunk] = len;
        if (p.slot_layout && lane == 0) { // the slot is the chunk's place: the stream is [slot end - len, slot end)
            p.offsets[chunk] = slot_at + wp;
            if (chunk + 1 == p.nchunks && !(SIZED && p.redo)) // (sized slots: the redo launch has the last word on the end)
                p.offsets[p.nchunks] = p.nchunks * p.slot_bytes;
        }
        if constexpr (FUSED) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // every lane's stream stores have reached L2
            if (lane == 0) {
                __hip_atomic_store(p.status + chunk, kStAggregate | (unsigned long long)((len + 15u) & ~15u), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                // (the copier will overwrite that word with the PREFIX: it must not learn of the chunk before the store is done)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (a workgroup-scope fence emits no wait for global stores on this target)
                mailbox_push(mb, (uint32_t)chunk, p.ring_slots ? (len | (ring_j << 26) | (wave << 28)) : len, p.flags, p.wait_ticks);
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
    if constexpr (FUSED) {
        if (lane == 0)
            atomicAdd(&mb->finished, 1u);
        // (letting the encoders that have run dry help the copier was tried: 0.867 ms against 0.817 for the 1 GiB word
        //  encode -- a second copy loop in the kernel costs 13 more spilled SGPRs in the coding loop)
    }
}

template <int FMT, int K> hipError_t launch_encode_t(const EncParams &p, int num_cus, hipStream_t stream)
{
    const bool fused = p.status != nullptr;
    const bool slots = !fused && p.slot_layout && p.claims; // MODE 2: dynamic claims, no copiers
    const bool sized = slots && p.ovf_ctl;                  // MODE 3: ... slots of the caller's size
    const bool dynamic = fused || slots;
    const uint32_t threads = FMT == FMT_ALIAS_LDS ? kEncAliasLdsThreads : (dynamic ? kEncFusedThreads : kEncBlockThreads);
    const uint32_t waves = threads / 64;
    const uint32_t enc_waves = fused ? waves - (waves >= 16 ? kEncFusedCopiers16 : 1) : waves;
    const size_t nrecs = p.nsyms < 256 ? 256 : p.nsyms;
    size_t lds = FMT == FMT_ALIAS_LDS ? nrecs * 8 + ((size_t)2 << p.scale_bits)
                 : ((FMT == FMT_BYTE && p.chunk_freqs) || FMT == FMT_WORDA) ? (size_t)waves * kAdaptEncWaveLds
                                                      : nrecs * sizeof(EncRec) + ((FMT == FMT_WORD || FMT == FMT_BYTE) ? 256 * 16 : 0);
    if ((FMT == FMT_WORD || (FMT == FMT_BYTE && !p.chunk_freqs)) && K == 1 && p.sym_bytes == 1 && nrecs == 256)
        lds += (size_t)waves * kEncStageBytes; // stream staging windows (4 + 4 KiB of tables in front)
    EncParams q = p;
    if (fused && !p.mailbox_global) {
        lds = (lds + 15) & ~(size_t)15;
        q.mailbox_off = (uint32_t)lds;
        lds += kEncFusedLdsBytes;
    }
    q.stage_off = 0;
    if (FMT == FMT_ALIAS_LDS && K == 1 && p.sym_bytes == 1) { // windows of the coding waves, where there is room
        const size_t at = (lds + 15) & ~(size_t)15;
        if (at + (size_t)enc_waves * kEncStageBytes <= 160 * 1024) {
            q.stage_off = (uint32_t)at;
            lds = at + (size_t)enc_waves * kEncStageBytes;
        }
    }
    const size_t lds_cap = FMT == FMT_ALIAS_LDS ? 160 * 1024 : 128 * 1024;
    if (lds > lds_cap || (FMT == FMT_WORD && !p.word_enc_recs && p.sym_bytes == 1) ||
        (FMT == FMT_ALIAS_LDS && (!p.alias_recs8 || !p.alias_remap16)))
        return hipErrorInvalidValue;
    uint64_t want = (p.nchunks + enc_waves - 1) / enc_waves;
    // blocks per CU: what the LDS allows, within the 32 resident waves of a CU
    uint64_t per_cu = lds ? (160 * 1024) / lds : 8;
    per_cu = per_cu < 1 ? 1 : per_cu;
    per_cu = per_cu * waves > 32 ? 32 / waves : per_cu;
    if (dynamic && FMT != FMT_ALIAS_LDS) { // ... and within the waves per SIMD the kernel's register budget was chosen for
        const uint64_t fit = (uint64_t)enc_fused_waves_per_simd(K) * 4 / waves;
        per_cu = per_cu > fit ? (fit ? fit : 1) : per_cu;
    }
    uint64_t cap = (uint64_t)num_cus * (FMT == FMT_ALIAS_LDS || dynamic ? per_cu : 8);
    if (sized && p.redo) // (a handful of chunks at most, usually none: one block per CU finds that out quickly)
        cap = (uint64_t)num_cus;
    const uint32_t grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    if constexpr (FMT != FMT_WORDA) { // (per-chunk word models: the scratch-slot mode only, as rans_amd_encode_adaptive launches it)
    if (fused) {
        auto kern = k_encode<FMT, K, 1>;
        static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), (int)lds_cap, lds_ok); e != hipSuccess)
            return e;
        RANS_LAUNCH(kern, dim3(grid), dim3(threads), lds, stream, q);
        return hipGetLastError();
    }
    if (sized) {
        auto kern = k_encode<FMT, K, 3>;
        static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), (int)lds_cap, lds_ok); e != hipSuccess)
            return e;
        RANS_LAUNCH(kern, dim3(grid), dim3(threads), lds, stream, q);
        return hipGetLastError();
    }
    if (slots) {
        auto kern = k_encode<FMT, K, 2>;
        static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), (int)lds_cap, lds_ok); e != hipSuccess)
            return e;
        RANS_LAUNCH(kern, dim3(grid), dim3(threads), lds, stream, q);
        return hipGetLastError();
    }
    } else if (fused || slots) {
        return hipErrorInvalidValue;
    }
    auto kern = k_encode<FMT, K, 0>;
    static std::atomic<uint64_t> lds_ok{0}; // per instantiation, one bit per device
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), (int)lds_cap, lds_ok); e != hipSuccess)
        return e;
    RANS_LAUNCH(kern, dim3(grid), dim3(threads), lds, stream, q);
    return hipGetLastError();
}

template <int FMT> hipError_t launch_encode_f(const EncParams &p, int num_cus, hipStream_t s)
{
    // K = ceil(N / 64) states per lane; lane counts that are not a multiple of 64 leave lanes idle
    if (p.n_ways >= 1 && p.n_ways <= 64)
        return launch_encode_t<FMT, 1>(p, num_cus, s);
    if (p.n_ways <= 128)
        return launch_encode_t<FMT, 2>(p, num_cus, s);
    if (p.n_ways <= 256)
        return launch_encode_t<FMT, 4>(p, num_cus, s);
    if (p.n_ways <= 512)
        return launch_encode_t<FMT, 8>(p, num_cus, s);
    return hipErrorInvalidValue;
}


} // namespace

hipError_t launch_encode_wave(int format, const EncParams &p, int num_cus, hipStream_t stream, const char **name)
{
    if (name)
        *name = format == FMT_WORD    ? "k_encode<word>"
                : format == FMT_BYTE  ? (p.chunk_freqs ? "k_encode<byte, per-chunk models>" : "k_encode<byte>")
                : format == FMT_WORDA ? "k_encode<word, per-chunk models>"
                : format == FMT_R64   ? "k_encode<r64>"
                : format == FMT_R64S  ? "k_encode<r64 full-width>"
                                      : "k_encode<alias, LDS remap>";
    switch (format) {
    case FMT_WORD: return launch_encode_f<FMT_WORD>(p, num_cus, stream);
    case FMT_WORDA: return p.chunk_freqs ? launch_encode_f<FMT_WORDA>(p, num_cus, stream) : hipErrorInvalidValue;
    case FMT_BYTE: return launch_encode_f<FMT_BYTE>(p, num_cus, stream);
    case FMT_R64: return launch_encode_f<FMT_R64>(p, num_cus, stream);
    case FMT_R64S: return launch_encode_f<FMT_R64S>(p, num_cus, stream);
    case FMT_ALIAS_LDS: return launch_encode_f<FMT_ALIAS_LDS>(p, num_cus, stream);
    // (FMT_ALIAS, alias_remap gathered from global memory: retired in round 6 -- every alias model the library can create has
    //  its encoder tables in LDS form, model.cpp; the lane encoders keep their own gather for narrow interleaves)
    default: return hipErrorInvalidValue;
    }
}

} // namespace rans_amd
