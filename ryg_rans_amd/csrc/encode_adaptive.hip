// encode_adaptive.hip -- per-chunk models in ONE launch (rans_amd_encode_adaptive_sized).
//
// The reference builds its model from the input it is about to code (main.cpp:139-162, main_simd.cpp:138-143: count_freqs,
// normalize_freqs, the encoder's symbol records, then the coding loop).  Here that whole sequence is the work of one wave on
// one chunk:
//   1. count      the chunk's bytes into 4 x 256 counters of the wave's LDS (ds_add; a lane counts into copy lane & 3)
//   2. normalise  exactly as normalize_freqs does (device_common.hpp adapt_normalize); the row goes to chunk_freqs[chunk]
//   3. place      an upper bound of the chunk's stream follows from its own histogram -- sum count[s] * log2(M / freq[s])
//                 bits plus what the floor in C(s, x) can add per symbol, plus the flushed states -- so the wave knows how
//                 much room it needs BEFORE it codes.  It publishes that size and finds its place by a decoupled look-back
//                 over the sizes of the chunks before it (device_common.hpp status_lookback; the predecessors it may have to
//                 wait for are still COUNTING, not coding): chunk c's piece starts at the sum of the pieces before it.  No
//                 scratch trip, no k_layout, no k_compact, no second launch; the container is in index order, the same
//                 from run to run, and about as large as the streams themselves (+ ~1.5 %).
//                 (A bump pointer -- one atomic add per chunk on one word -- was the first version: a single word retires
//                 ~88 atomics per microsecond and 65 536 chunks made that 0.75 ms of a 1.1 ms launch, whatever the occupancy:
//                 profiles/r06_adaptive_encoder.md.)
//   4. records    the hand-written sub-steps' 16-byte records of the chunk's 256 symbols at LDS address 0 (reciprocals by
//                 frequency from a 32 KiB table in global memory: no division on the device)
//   5. code       the chunk a second time through (the second read is served by L2 / the memory-side cache), the same
//                 staged sub-steps as the one-model encoders (encode_common.hpp), rounds last to first
// One wave per WORKGROUP: the record table then sits at LDS address 0 whatever the wave (the sub-steps form a record's address
// from the symbol byte with one SDWA shift), and 6 KiB of LDS per workgroup keep 24 of them resident per CU
// (profiles/r06_wg_residency.log: 25 by the LDS; 24 = six waves per SIMD by the kernel's 80 registers).  Every chunk's stream is the oracle's stream for the model of that chunk alone.

#include "device_common.hpp"
#include "launchers.hpp"
#include <type_traits>

namespace rans_amd {

namespace {

#include "encode_common.hpp"

// six waves per SIMD (80 registers), 24 one-wave workgroups per CU (the LDS would allow 25): measured the same from 20 to 25
// per CU, and with 1, 4 or 8 count loads in flight (profiles/r06_adaptive_encoder.md: the launch is bound by the LDS pipe --
// one ds_add per symbol on top of the coder's record gather -- not by latency)
constexpr int kAdaptWavesPerSimd = 6, kAdaptPerCu = 24;
constexpr uint32_t kAdaptRecBytes = 256u * 16u;                     // the records, at LDS address 0
constexpr uint32_t kAdaptWinBase = kAdaptRecBytes;                  // the stream staging window behind them
constexpr uint32_t kAdaptEncLds = kAdaptRecBytes + kEncStageBytes; // 6 KiB; the counters of step 1 lie over the records

// One sub-step for any lane mask on the FAST records (compiler-scheduled; the rounds in front of the first whole
// super-group of sixteen, interleaves other than 64, unaligned inputs), units stored straight to memory.
//   word: {m', (freq << 20) - 1, cmpl | sh << 24, bias}   rans_word_sse41.h:81-93
//   byte: {rcp, cmpl | rshift << 24, bias, x_max}         rans_byte.h:62-74, :258-280
// `always` (word format, wave-uniform): the chunk holds ONE symbol value, its frequency is 4096 = M and the reference's 32-bit
// bound ((L >> 12) << 16) * freq wraps to 0 (rans_word_sse41.h:85): a word leaves with every symbol.  The record's
// threshold (freq << 20) - 1 cannot say that; such chunks are coded by this sub-step alone.
template <int FMT>
__device__ __forceinline__ void adapt_substep(uint32_t &x, const u32x4 rec, bool active, bool small, bool always,
                                              uint8_t RANS_GLOBAL *slot, uint32_t &wp)
{
    if constexpr (FMT == FMT_WORD) {
        const bool emit = active && (always || x > rec.y);
        const uint64_t m = __builtin_amdgcn_ballot_w64(emit);
        wp -= 2u * (uint32_t)__builtin_popcountll(m);
        if (emit)
            *reinterpret_cast<uint16_t RANS_GLOBAL *>(slot + wp + 2u * rank_below(m)) = (uint16_t)(x & 0xffffu);
        const uint32_t y = emit ? (x >> 16) : x;
        const uint32_t t = __umulhi(y, rec.x);
        const uint32_t sh = rec.z >> 24;
        const uint32_t q = small ? (t >> sh) : ((t + ((y - t) >> 1)) >> sh);
        const uint32_t xn = y + rec.w + __umul24(q, rec.z);
        x = active ? xn : x;
    } else {
        const uint32_t x_max = rec.w;
        const bool e1 = active && x >= x_max;
        const bool e2 = e1 && (x >> 8) >= x_max;
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(e1);
        const uint64_t m2 = __builtin_amdgcn_ballot_w64(e2);
        wp -= (uint32_t)__builtin_popcountll(m1) + (uint32_t)__builtin_popcountll(m2);
        const uint32_t at = wp + rank_below(m1) + rank_below(m2);
        if (e2) {
            slot[at] = (uint8_t)(x >> 8);
            slot[at + 1] = (uint8_t)x;
        } else if (e1) {
            slot[at] = (uint8_t)x;
        }
        const uint32_t y = e2 ? (x >> 16) : (e1 ? (x >> 8) : x);
        const uint32_t q = __umulhi(y, rec.x) >> (rec.y >> 24);
        const uint32_t xn = y + rec.z + __umul24(q, rec.y);
        x = active ? xn : x;
    }
}

// RR > 0: REGISTER-RESIDENT chunks -- a full chunk of RR x 1024 symbols (RR = 4, 8, 16) is loaded ONCE, four dwords per lane and
// super-group of sixteen rounds in exactly the layout the coding loop wants them (the 4 x 4 transpose's input), all RR x 4 loads
// in flight together; the counters are fed from those registers and so is the coding loop, unrolled over the super-groups: the
// chunk crosses the fabric once (the two-pass form reads it twice: 1.62 x the algorithmic bytes) and the coding loop holds no
// loads at all.  The price is occupancy (64 more VGPRs at RR = 16: three waves per SIMD, 12 per CU -- and still 2 % faster than the
// two-pass form at 24, whose second read waits on the memory side).  Ragged chunks (the last one of an
// input) and one-symbol word chunks take the two-pass form inside the same kernel.  The word format codes with the round-up
// reciprocals here whatever the frequencies (exact for every frequency; 3 VALU more than Alverson's, in a launch the LDS pipe
// bounds): one unrolled loop instead of two.
// (waves per SIMD by the registers a resident chunk needs -- 64 / 32 / 16 VGPRs of symbols beside ~90 of working set, no spills:
//  at RR = 16 three spill-free waves beat four that spill 22 registers, 0.874 against 0.984 ms for the word format)
constexpr int adapt_waves_per_simd(int K, int RR) { return K != 1 ? (K <= 4 ? 4 : 2) : RR >= 16 ? 3 : RR >= 4 ? 4 : kAdaptWavesPerSimd; }

template <int FMT, int K, int RR>
__global__ void __launch_bounds__(64, adapt_waves_per_simd(K, RR)) k_encode_adaptive(const AdaptEncParams p)
{
    static_assert(FMT == FMT_WORD || FMT == FMT_BYTE, "per-chunk models: the byte and the word format");
    static_assert(RR == 0 || K == 1, "register-resident chunks: one state per lane");
    constexpr bool kRR = RR > 0;
    using Tr = FmtTraits<FMT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = lane_id();
    const uint32_t N = p.n_ways;
    const uint32_t M = 1u << p.scale_bits;
    const bool lds_at_zero = lds_starts_at_zero(smem);
    auto lds_u32x4 = [](uint32_t at) { return reinterpret_cast<RANS_LDS u32x4 *>((uintptr_t)at); };
    // per-lane constants of the 4x4 byte transpose (encode_wave.hip; the decoder's stores mirrored)
    const uint32_t sel1 = (lane & 1u) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t in_lane_off = (lane & 3u) * N + (lane & ~3u);
    const uint32_t npools = gridDim.x < kWorkPools ? gridDim.x : kWorkPools;
    const uint32_t pool = blockIdx.x % npools;
    bool failed = false;

    for (;;) {
        // ascending claims, one counter per pool of workgroups (encode_wave.hip: a single counter retires ~90 claims / us)
        uint32_t got = 0;
        if (lane == 0)
            got = atomicAdd(p.claims + kWorkPoolStride * pool, 1u);
        const uint64_t chunk = (uint64_t)uniform(got) * npools + pool;
        if (chunk >= p.nchunks)
            break;
        const uint64_t first = chunk * p.chunk_syms;
        const uint32_t nsym = (uint32_t)((p.n - first) < p.chunk_syms ? (p.n - first) : p.chunk_syms);
        const uint8_t RANS_GLOBAL *src = (const uint8_t RANS_GLOBAL *)p.syms + first;

        // (register-resident form: the whole chunk, in the coding loop's layout -- lane l holds row 4 j + (l & 3), columns
        //  4 (l >> 2) .. + 3 of super-group sg; the byte format's lanes mirrored, enc_byte_full_staged)
        const bool rr_chunk = kRR && N == 64u && lds_at_zero && nsym == (uint32_t)RR * 1024u && p.chunk_syms == (uint32_t)RR * 1024u &&
                              (reinterpret_cast<uintptr_t>(p.syms) & 3u) == 0;
        uint32_t d[kRR ? RR : 1][4];
        if constexpr (kRR) {
            if (rr_chunk) {
                const uint32_t off = FMT == FMT_BYTE ? (lane & 3u) * 64u + (60u - (lane & ~3u)) : in_lane_off;
#pragma unroll
                for (int sg = 0; sg < RR; ++sg)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        d[sg][j] = *reinterpret_cast<const uint32_t RANS_GLOBAL *>(src + (uint32_t)(sg * 16 + j * 4) * 64u + off);
            }
        }

        // ---- 1. count (count_freqs, main.cpp:59-66).  Counter (s, k) at byte 16 s + 4 k: the four copies of a symbol side by
        // side, so that the sums below are one ds_read_b128 per symbol.
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *lds_u32x4(1024u * i + 16u * lane) = u32x4{0u, 0u, 0u, 0u};
        const uint32_t copy_off = (lane & 3u) << 2;
        auto count1 = [&](uint32_t s) {
            asm volatile("ds_add_u32 %0, %1" ::"v"((s << 4) | copy_off), "v"(1u) : "memory");
        };
        auto count4 = [&](uint32_t v) {
            count1(v & 0xffu);
            count1((v >> 8) & 0xffu);
            count1((v >> 16) & 0xffu);
            count1(v >> 24);
        };
        uint32_t done = 0;
        if constexpr (kRR) {
            if (rr_chunk) { // from the registers (the order does not matter to a histogram)
#pragma unroll
                for (int sg = 0; sg < RR; ++sg)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        count4(d[sg][j]);
                done = nsym;
            }
        }
        if (done == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
            // 16 bytes per lane and load, kCountDepth loads in flight and the next kCountDepth issued before this trip's counting:
            // a wave that keeps ONE line per lane in flight waits a memory latency per KiB (2 us: a third of the chunk's time)
            constexpr uint32_t kCountDepth = 4, kTrip = 1024u * kCountDepth;
            auto at16 = [&](uint32_t byte) { return *reinterpret_cast<gvec_cptr>(reinterpret_cast<uint64_t>(src) + byte + lane * 16u); };
            const uint32_t big = nsym & ~(kTrip - 1u);
            if (big) {
                u32x4 v[kCountDepth];
#pragma unroll
                for (uint32_t d = 0; d < kCountDepth; ++d)
                    v[d] = at16(1024u * d);
                for (uint32_t i = 0; i < big; i += kTrip) {
                    u32x4 nx[kCountDepth];
#pragma unroll
                    for (uint32_t d = 0; d < kCountDepth; ++d)
                        nx[d] = v[d];
                    if (i + kTrip < big) {
#pragma unroll
                        for (uint32_t d = 0; d < kCountDepth; ++d)
                            nx[d] = at16(i + kTrip + 1024u * d);
                    }
#pragma unroll
                    for (uint32_t d = 0; d < kCountDepth; ++d) {
                        count4(v[d].x);
                        count4(v[d].y);
                        count4(v[d].z);
                        count4(v[d].w);
                    }
#pragma unroll
                    for (uint32_t d = 0; d < kCountDepth; ++d)
                        v[d] = nx[d];
                }
            }
            const uint32_t body16 = nsym & ~1023u;
            for (uint32_t i = big; i < body16; i += 1024u) {
                const u32x4 v = at16(i);
                count4(v.x);
                count4(v.y);
                count4(v.z);
                count4(v.w);
            }
            done = body16;
        }
        const bool aligned4 = (reinterpret_cast<uintptr_t>(src) & 3u) == 0;
        const uint32_t body = (kRR && rr_chunk) ? nsym : aligned4 ? (nsym & ~3u) : done;
        for (uint32_t i = done + lane * 4u; i < body; i += 256u)
            count4(*reinterpret_cast<const uint32_t RANS_GLOBAL *>(src + i));
        for (uint32_t i = body + lane; i < nsym; i += 64u)
            count1(src[i]);
        // (LDS operations of one wave execute in order: the reads below see every lane's increments)
        uint32_t cnt[4], width[4], cum[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 c4 = *lds_u32x4(64u * lane + 16u * i);
            cnt[i] = c4.x + c4.y + c4.z + c4.w;
        }

        // ---- 2. normalise (normalize_freqs, main.cpp:75-129) and hand the row out
        if (!adapt_normalize(cnt, nsym, M, lane, width))
            failed = true;
        const u32x2 packed = {width[0] | (width[1] << 16), width[2] | (width[3] << 16)};
        *reinterpret_cast<u32x2 RANS_GLOBAL *>(reinterpret_cast<uint64_t>(p.chunk_freqs) + chunk * 512u + 8u * lane) = packed;
        adapt_cum(width, lane, cum);

        // ---- 3. how long can the stream get?  Per lane and symbol, y = x >> (units out) and C(s, y) <= y (M / f)(1 + 1 / q)
        // with q = y / f >= L >> scale_bits (16 in the word format, 2^11 and more in the byte format); the last state is
        // >= L, the first is L: the units a lane emits hold at most sum log2(M / f) + n log2(1 + 1 / q_min) bits.
        float bits = 0.0f;
        uint32_t wmax = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (cnt[i])
                bits += (float)cnt[i] * ((float)p.scale_bits - __log2f((float)width[i]));
            wmax = width[i] > wmax ? width[i] : wmax;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            bits += __shfl_xor(bits, d, 64);
            const uint32_t o = (uint32_t)__shfl_xor((int)wmax, d, 64);
            wmax = o > wmax ? o : wmax;
        }
        bits = __builtin_bit_cast(float, uniform(__builtin_bit_cast(uint32_t, bits)));
        // word format: Alverson reciprocals while the renormalised state stays below 2^31 (no frequency above 2048); the
        // register-resident kernels code with the round-up reciprocals throughout
        const bool small = !kRR && uniform(wmax) <= 2048u;
        const bool always = FMT == FMT_WORD && uniform(wmax) == M; // (one symbol value: see adapt_substep)
        // (f32: 2^-12 of the sum and a line cover the rounding of 256 products and their sum)
        const float est = bits * (0.125f * (1.0f + 1.0f / 4096.0f));
        const uint64_t slack = (FMT == FMT_WORD ? ((uint64_t)nsym * 23u) >> 11 : (uint64_t)nsym >> 12) + N * Tr::kStateBytes + 64u;
        uint64_t need = (est < 4.0e9f ? (uint64_t)est : 0xffffffffull) + slack;
        need = (need + 63u) & ~63ull;
        need = (need < p.worst_slot && !always) ? need : p.worst_slot;
        if (lane == 0) // (the successors' look-back adds this up while the records below are being built)
            __hip_atomic_store(p.status + chunk, kStAggregate | need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ---- 4. the records of this chunk (RansEncSymbolInit / the word format's round-up reciprocals, model.cpp)
        {
            const uint32_t RANS_GLOBAL *rcp = (const uint32_t RANS_GLOBAL *)p.rcp + ((FMT == FMT_WORD && !small) ? kAdaptRcpEntries : 0u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t f = width[i];
                u32x4 r;
                if (f == 0u) { // no frequency: nothing leaves, the state stays (never met: every symbol that occurs has a slot)
                    r = FMT == FMT_WORD ? u32x4{0u, 0xffffffffu, 0x80000000u, 0u} : u32x4{0u, 0xffffffffu, 0u, 0xffffffffu};
                } else {
                    const uint32_t sh = f >= 2u ? 31u - (uint32_t)__builtin_clz(f - 1u) : 0u; // ceil(log2 f) - 1
                    const uint32_t bias = f >= 2u ? cum[i] : cum[i] + M - 1u;
                    const uint32_t cs = (M - f) | (sh << 24);
                    const uint32_t rc = rcp[f];
                    if constexpr (FMT == FMT_WORD)
                        r = u32x4{rc, (f << 20) - 1u, cs, bias};
                    else
                        r = u32x4{rc, cs, bias, f << (31u - p.scale_bits)};
                }
                *lds_u32x4(64u * lane + 16u * i) = r;
            }
        }

        // ---- 5. code: rounds last to first, into the place the bound bought
        const uint32_t rounds = uniform(nsym / N);
        const uint32_t tail = uniform(nsym - rounds * N);
        const bool fast_in = K == 1 && N == 64u && lds_at_zero && ((reinterpret_cast<uintptr_t>(p.syms) | p.chunk_syms) & 3u) == 0;
        const uint32_t fast_rounds = (fast_in && !always) ? (rounds & ~15u) : 0u;
        {
            unsigned long long base = 0;
            const bool found = status_lookback(p.status, chunk, lane, p.flags, p.wait_ticks, 32u, base);
            if (lane == 0) {
                __hip_atomic_store(p.status + chunk, kStPrefix | (base + need), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (chunk + 1 == p.nchunks)
                    p.offsets[p.nchunks] = base + need; // the container's end
            }
            const uint64_t slot_at = base;
            const uint64_t room = need;
            if (!found || slot_at + room > p.out_cap) { // (wave-uniform)
                if (lane == 0) {
                    if (found)
                        atomicOr(p.flags, 2u);
                    p.lengths[chunk] = 0u;
                    p.offsets[chunk] = 0u;
                }
                continue;
            }
            uint8_t RANS_GLOBAL *slot = (uint8_t RANS_GLOBAL *)p.out + slot_at;
            uint32_t wp = (uint32_t)room;
            bool ovf = false;
            uint32_t x[K];
#pragma unroll
            for (int k = 0; k < K; ++k)
                x[k] = Tr::kL;

            for (uint32_t rr = rounds + 1; rr-- > fast_rounds;) {
                const uint32_t cntr = (rr < rounds) ? N : tail;
                if (cntr == 0)
                    continue;
                if (wp < N * 2u) { // (what a round can emit at most)
                    ovf = true;
                    break;
                }
                const uint8_t RANS_GLOBAL *rsrc = src + (uint64_t)rr * N;
                uint32_t sym[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t idx = k * 64u + lane;
                    sym[k] = idx < cntr ? (uint32_t)rsrc[idx] : 0u;
                }
#pragma unroll
                for (int k = K - 1; k >= 0; --k)
                    adapt_substep<FMT>(x[k], *lds_u32x4(sym[k] << 4), k * 64u + lane < cntr, small, always, slot, wp);
            }

            if constexpr (K == 1) {
                if (fast_rounds && !ovf) {
                    uint32_t worst = 0;
                    uint32_t k3v = 4u;
                    asm volatile("" : "+v"(k3v)); // (SDWA takes no literal; a VGPR operand is also the faster VALU form)
                    constexpr uint32_t kTopPiece = kEncStageBytes - 16u;
                    if (wp & 15u) { // the rounds above have stored units themselves: the piece that holds wp goes into the window
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane < 4u) {
                            const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t RANS_GLOBAL *>(slot + (wp & ~15u) + 4u * lane));
                            *reinterpret_cast<RANS_LDS uint32_t *>((uintptr_t)(kAdaptWinBase + kTopPiece + 4u * lane)) = v;
                        }
                    }
                    // LDS [lp, top) -> slot [wp - (top - lp), wp) in whole 16-byte pieces (encode_wave.hip stage_flush)
                    auto stage_flush = [&](uint32_t top, uint32_t lp) {
                        if (top - lp > wp) { // the bytes in the window do not fit below the write offset: nothing more is stored
                            ovf = true;
                            return;
                        }
                        const uint32_t hi = (top + 15u) & ~15u, lo = lp & ~15u;
                        const uint32_t to_slot = wp - top; // (wraps; congruent to 0 modulo 16)
                        const int32_t a = (int32_t)hi - 16 * (int32_t)(lane + 1u);
                        auto piece = [&](int32_t at) {
                            if (at >= (int32_t)lo) {
                                const u32x4 v = *lds_u32x4((uint32_t)at);
                                // (nt: the stream is written once and not read by this launch -- it should not push the
                                //  chunks that wait for their second read out of L2)
                                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 RANS_GLOBAL *>(slot + ((uint32_t)at + to_slot)));
                                if ((uint32_t)at == lo)
                                    *lds_u32x4(kAdaptWinBase + kTopPiece) = v;
                            }
                        };
                        piece(a);
                        if (hi - lo > 1024u) // (wave-uniform)
                            piece(a - 1024);
                        wp -= top - lp;
                    };
                    // byte format: the lanes in reverse order (enc_byte_full_staged: lane l codes stream 63 - l)
                    uint32_t in_off = in_lane_off, tsel1 = sel1;
                    if constexpr (FMT == FMT_BYTE) {
                        x[0] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((63u - lane) * 4u), (int)x[0]);
                        in_off = (lane & 3u) * N + (60u - (lane & ~3u));
                        tsel1 = (lane & 1u) ? 0x00040206u : 0x05010703u;
                    }
                    uint32_t cur[4], nxt[4];
                    auto load_super = [&](uint32_t (&dstq)[4], uint32_t sg) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            // (nt: the chunk's LAST use; the count pass above reads with plain loads so that the lines stay)
                            dstq[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t RANS_GLOBAL *>(src + (uint64_t)(sg * 16u + j * 4u) * N + in_off));
                    };
                    auto fast_loop = [&](auto small_tag) {
                        constexpr bool kSmall = decltype(small_tag)::value;
                        uint32_t sg = fast_rounds >> 4;
                        load_super(cur, sg - 1);
                        while (sg-- > 0) {
                            if (ovf)
                                break;
                            if (sg > 0)
                                load_super(nxt, sg - 1);
                            const uint32_t win_top = kAdaptWinBase + kTopPiece + (uniform(wp) & 15u);
                            uint32_t lp = FMT == FMT_WORD ? win_top >> 1 : win_top; // (the word format's pointer counts 16-bit words)
#pragma unroll
                            for (int j = 3; j >= 0; --j) {
                                const uint32_t t = quad_transpose(cur[j], tsel1, sel2);
                                auto rec_at = [&](int step) { // symbol byte 3 - step -> its record (table at LDS address 0)
                                    uint32_t at;
                                    if (step == 0)
                                        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(at) : "v"(k3v), "v"(t));
                                    else if (step == 1)
                                        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(at) : "v"(k3v), "v"(t));
                                    else if (step == 2)
                                        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(at) : "v"(k3v), "v"(t));
                                    else
                                        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(at) : "v"(k3v), "v"(t));
                                    return *reinterpret_cast<const RANS_LDS u32x4 *>((uintptr_t)at);
                                };
                                u32x4 rec = rec_at(0);
                                lp = uniform(lp);
#pragma unroll
                                for (int step = 0; step < 4; ++step) {
                                    const u32x4 now = rec;
                                    if (step + 1 < 4)
                                        rec = rec_at(step + 1);
                                    if constexpr (FMT == FMT_WORD)
                                        enc_word_full_staged<kSmall, false>(x[0], now, lp, worst);
                                    else
                                        enc_byte_full_staged<false>(x[0], now, lp, worst);
                                }
                                lp = uniform(lp);
                            }
                            stage_flush(win_top, FMT == FMT_WORD ? uniform(lp) << 1 : uniform(lp));
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                cur[j] = nxt[j];
                        }
                    };
                    // symbol byte 3 - step of t -> its record (table at LDS address 0), the next record read before the current one is used
                    auto code4 = [&](uint32_t t, uint32_t &lp) {
                        auto rec_at = [&](int step) {
                            uint32_t at;
                            if (step == 0)
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(at) : "v"(k3v), "v"(t));
                            else if (step == 1)
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(at) : "v"(k3v), "v"(t));
                            else if (step == 2)
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(at) : "v"(k3v), "v"(t));
                            else
                                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(at) : "v"(k3v), "v"(t));
                            return *reinterpret_cast<const RANS_LDS u32x4 *>((uintptr_t)at);
                        };
                        u32x4 rec = rec_at(0);
                        lp = uniform(lp);
#pragma unroll
                        for (int step = 0; step < 4; ++step) {
                            const u32x4 now = rec;
                            if (step + 1 < 4)
                                rec = rec_at(step + 1);
                            if constexpr (FMT == FMT_WORD)
                                enc_word_full_staged<false, false>(x[0], now, lp, worst);
                            else
                                enc_byte_full_staged<false>(x[0], now, lp, worst);
                        }
                        lp = uniform(lp);
                    };
                    bool from_registers = false;
                    if constexpr (kRR) {
                        if (rr_chunk) { // every super-group from the registers, last to first; nothing is loaded in this loop
                            from_registers = true;
#pragma unroll
                            for (int sg = RR - 1; sg >= 0; --sg) {
                                if (!ovf) {
                                    const uint32_t win_top = kAdaptWinBase + kTopPiece + (uniform(wp) & 15u);
                                    uint32_t lp = FMT == FMT_WORD ? win_top >> 1 : win_top;
#pragma unroll
                                    for (int j = 3; j >= 0; --j)
                                        code4(quad_transpose(d[sg][j], tsel1, sel2), lp);
                                    stage_flush(win_top, FMT == FMT_WORD ? uniform(lp) << 1 : uniform(lp));
                                }
                            }
                        }
                    }
                    if (from_registers) {
                    } else if (FMT == FMT_WORD && !small)
                        fast_loop(std::false_type{});
                    else if (!kRR || FMT == FMT_BYTE) // (register-resident word kernels: the round-up records only)
                        fast_loop(std::true_type{});
                    if constexpr (FMT == FMT_BYTE) // back to lane l = stream l
                        x[0] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((63u - lane) * 4u), (int)x[0]);
                }
            }

            if (ovf || wp < N * Tr::kStateBytes) { // the bound was none (never seen; the checks keep every store inside the piece)
                if (lane == 0) {
                    atomicOr(p.flags, 1024u);
                    p.lengths[chunk] = 0u;
                    p.offsets[chunk] = 0u;
                }
                continue;
            }
            // flush: lane N-1 first, i.e. lane 0's state ends up first in memory (main.cpp:244-245, main_simd.cpp:298-299)
            wp -= N * Tr::kStateBytes;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t idx = k * 64u + lane;
                if (idx < N) {
                    uint8_t RANS_GLOBAL *at = slot + wp + idx * Tr::kStateBytes;
                    if constexpr (FMT == FMT_WORD) {
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
            if (lane == 0) {
                p.lengths[chunk] = (uint32_t)room - wp;
                p.offsets[chunk] = slot_at + wp;
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(failed) != 0 && lane == 0)
        atomicOr(p.flags, 1u);
}

template <int FMT, int K, int RR> hipError_t launch_t(const AdaptEncParams &p, int num_cus, hipStream_t stream)
{
    // workgroups per CU: 6 KiB of LDS each allow 25 (profiles/r06_wg_residency.log), the registers kAdaptPerCu and fewer
    const uint64_t per_cu = K != 1 ? (K <= 4 ? 16 : 8) : (uint64_t)adapt_waves_per_simd(K, RR) * 4;
    const uint64_t cap = (uint64_t)num_cus * (per_cu < (uint64_t)kAdaptPerCu ? per_cu : (uint64_t)kAdaptPerCu);
    const uint32_t grid = (uint32_t)(p.nchunks < cap ? (p.nchunks ? p.nchunks : 1) : cap);
    RANS_LAUNCH((k_encode_adaptive<FMT, K, RR>), dim3(grid), dim3(64), kAdaptEncLds, stream, p);
    return hipGetLastError();
}

template <int FMT> hipError_t launch_f(const AdaptEncParams &p, int num_cus, hipStream_t s)
{
    if (p.n_ways == 64 && (reinterpret_cast<uintptr_t>(p.syms) & 3u) == 0 && p.n >= p.chunk_syms) { // register-resident chunks
        if (p.chunk_syms == 16384u)
            return launch_t<FMT, 1, 16>(p, num_cus, s);
        if (p.chunk_syms == 8192u)
            return launch_t<FMT, 1, 8>(p, num_cus, s);
        if (p.chunk_syms == 4096u)
            return launch_t<FMT, 1, 4>(p, num_cus, s);
    }
    if (p.n_ways >= 1 && p.n_ways <= 64)
        return launch_t<FMT, 1, 0>(p, num_cus, s);
    if (p.n_ways <= 128)
        return launch_t<FMT, 2, 0>(p, num_cus, s);
    if (p.n_ways <= 256)
        return launch_t<FMT, 4, 0>(p, num_cus, s);
    if (p.n_ways <= 512)
        return launch_t<FMT, 8, 0>(p, num_cus, s);
    return hipErrorInvalidValue;
}

} // namespace

hipError_t launch_encode_adaptive(int format, const AdaptEncParams &p, int num_cus, hipStream_t stream, const char **name)
{
    if (!p.rcp || !p.claims || !p.status || p.scale_bits < 8 || p.scale_bits > kAdaptMaxScaleBits ||
        (format == FMT_WORD && p.scale_bits != 12))
        return hipErrorInvalidValue;
    if (name)
        *name = format == FMT_WORD ? "k_encode_adaptive<word>" : "k_encode_adaptive<byte>";
    switch (format) {
    case FMT_WORD: return launch_f<FMT_WORD>(p, num_cus, stream);
    case FMT_BYTE: return launch_f<FMT_BYTE>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace rans_amd
