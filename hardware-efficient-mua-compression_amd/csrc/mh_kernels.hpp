// mh_kernels.hpp -- hand-written gfx950 kernels of the hot path.
//
//   k_calibrate      one wave per channel: cutoff, calibration histogram, peak, permutation,
//                    first-min encoder, per-channel encode LUT (also clears per-call scratch)
//   k_hist<NS>       window histogram for S <= 3, byte-parallel compares, 16 B/lane reads
//   k_finalize       rank-map the histogram, bits = SCLV[enc] . post
//   decode_chunk     per-symbol decoder straight from global memory (partial / oversize chunks)
//   k_scan_* / k_compact       dense re-packing of the segment slots (<= 2048 segments: k_compact scans for itself)
//   k_synth          synthetic MUA generator
// The encoder / decoder proper are in mh_codec2.hpp, layout kernels in mh_layout.hpp.
//
// All integer / bit work, HBM-bound: no MFMA anywhere.
#pragma once
#include "mh_device.hpp"

namespace mh {

// ------------------------------------------------------------------------------------------
// calibrate
// ------------------------------------------------------------------------------------------
struct CalArgs {
    const uint8_t *data;
    const uint64_t *ch_off, *ch_len;
    const uint8_t *sclv;       // K*S lengths
    const uint32_t *sclv16;    // the same rows padded to 16 bytes (4 dwords per row): one load prices one encoder
    const uint32_t *codes;     // K*16 : bit-reversed code | len << 16, by rank
    uint32_t C, S, h, mode, K;
    // outputs
    uint64_t *cutoff;          // may be NULL
    uint32_t *cal_sorted;      // C*S, may be NULL
    uint8_t *peak, *enc;       // never NULL (plan scratch when the caller passes NULL)
    uint2 *lut;                // C*16 {code_rev, len} indexed by min(raw value, 15)
    // folded-in initialisation (saves separate memset / memcpy launches); each may be NULL
    unsigned long long *zero_hist;  // [C][16] window-histogram scratch to clear
    unsigned long long *zero_bits;  // [C] per-channel bit totals to clear
    const uint8_t *skip_src;        // plan's skip flags ...
    uint8_t *skip_dst;              // ... copied to the caller's array
    // calibration windows longer than a few KiB are histogrammed by the tiled window-histogram
    // kernel first ([C][16] counts by clipped symbol); NULL = scan the window here
    const unsigned long long *pre_hist;
};

// One channel by one wave; every lane returns the channel's (peak, encoder).
__device__ __forceinline__ void calibrate_channel(const CalArgs &a, uint32_t ch, int lane, int &p_out, uint32_t &k_out)
{
    const int S = (int)a.S;
    const uint64_t T = a.ch_len[ch];
    const uint64_t lim = (uint64_t)1 << a.h;
    const uint64_t c = T < lim ? T : lim;  // functions_1.py:59-64
    const uint8_t *x = a.data + a.ch_off[ch];
    uint32_t cnt[MH_LUT_SYMS];
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] = 0;
    if (a.pre_hist) {
        if (lane == 0) {  // bins 0..S-2 were counted, the top bin is the window length minus the rest
            uint32_t rest = 0;
#pragma unroll
            for (int s = 0; s < MH_LUT_SYMS; ++s)
                if (s < S - 1) {
                    cnt[s] = (uint32_t)a.pre_hist[(size_t)ch * kHistStride + s];
                    rest += cnt[s];
                }
#pragma unroll
            for (int s = 0; s < MH_LUT_SYMS; ++s)
                if (s == S - 1) cnt[s] = (uint32_t)c - rest;
        }
    } else {
        for (uint64_t i = lane; i < c; i += 64) {
            int v = x[i];
            v = v > S - 1 ? S - 1 : v;  // clip, get_BR_with_approx_sort.py:164
#pragma unroll
            for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] += (v == s);
        }
    }
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] = wave_sum_u32(cnt[s]);
    // every lane now holds the full calibration histogram
    int p = 0;
    if (a.mode == MH_MODE_APPROX) {
        uint32_t best = cnt[0];
#pragma unroll
        for (int s = 1; s < MH_LUT_SYMS; ++s)
            if (s < S && cnt[s] > best) {  // first max wins (np.argmax)
                best = cnt[s];
                p = s;
            }
    }
    // calibration histogram in rank order: sorted[k] = cal[symbol_of_rank(k)]
    uint32_t sorted[MH_LUT_SYMS];
#pragma unroll
    for (int k = 0; k < MH_LUT_SYMS; ++k) {
        const int sym = k < S ? symbol_of_rank((int)a.mode, S, p, k) : 0;
        uint32_t v = 0;
#pragma unroll
        for (int s = 0; s < MH_LUT_SYMS; ++s) v = (s == sym) ? cnt[s] : v;
        sorted[k] = k < S ? v : 0;
    }
    // first argmin over the K encoders of sum_r SCLV[k][r] * sorted[r]  (:254,281): lane k prices encoder k from
    // its 16-byte padded row -- K = 35 costs what K = 1 does (a serial loop over the encoders, each iteration a
    // batch of dependent byte loads, took 0.75 us per encoder: S = 10 measure 25 -> 51 us on 2400 short channels)
    uint32_t best_k = 0;
    uint64_t best_cost = ~(uint64_t)0;
    for (uint32_t k0 = 0; k0 < a.K; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        const u32x4 len = *reinterpret_cast<const u32x4 *>(a.sclv16 + (size_t)(k < a.K ? k : a.K - 1) * 4);
        uint64_t cost = 0;
#pragma unroll
        for (int r = 0; r < MH_LUT_SYMS; ++r) cost += (uint64_t)((len[r >> 2] >> (8 * (r & 3))) & 0xFFu) * sorted[r];  // sorted[] is 0 beyond S
        if (k >= a.K) cost = ~(uint64_t)0;
        const uint64_t lo = wave_min_u64(cost);
        const unsigned long long first = __ballot(cost == lo);  // lowest lane = lowest encoder index (np.argmin)
        if (lo < best_cost) {
            best_cost = lo;
            best_k = k0 + (uint32_t)__ffsll((long long)first) - 1u;
        }
    }
    if (lane == 0) {
        if (a.cutoff) a.cutoff[ch] = c;
        a.peak[ch] = (uint8_t)p;
        a.enc[ch] = (uint8_t)best_k;
        if (a.zero_bits) a.zero_bits[ch] = 0;
        if (a.skip_dst) a.skip_dst[ch] = a.skip_src[ch];
    }
    if (a.zero_hist && lane < kHistStride) a.zero_hist[(size_t)ch * kHistStride + lane] = 0;
    if (a.cal_sorted && lane < S) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < MH_LUT_SYMS; ++k) v = (k == lane) ? sorted[k] : v;
        a.cal_sorted[(size_t)ch * S + lane] = v;
    }
    if (lane < kLut) {
        const int sym = lane > S - 1 ? S - 1 : lane;
        const int r = rank_of_symbol((int)a.mode, S, p, sym);
        const uint32_t e = a.codes[best_k * 16 + r];
        a.lut[(size_t)ch * kLut + lane] = make_uint2(e & 0xFFFFu, e >> 16);
    }
    p_out = p;
    k_out = best_k;
}

__global__ __launch_bounds__(256) void k_calibrate(CalArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= a.C) return;
    int p;
    uint32_t k;
    calibrate_channel(a, ch, lane, p, k);
}

// Encode LUTs from a PRESET per-channel (peak, encoder) word instead of a calibration pass: the
// compression phase of the reference's RTL reads exactly this word from its RAM
// (FPGA implementation/RAM.v:4, README.md:50-66).  16 lanes per channel.
__global__ __launch_bounds__(256) void k_lut_preset(const uint8_t *peak, const uint8_t *enc,
                                                    const uint32_t *codes, uint32_t C, uint32_t S,
                                                    uint32_t mode, uint32_t K, uint2 *lut,
                                                    unsigned long long *zero_bits, uint8_t *peak_out,
                                                    uint8_t *enc_out)
{
    const uint32_t ch = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int v = threadIdx.x & 15;
    if (ch >= C) return;
    const int p = peak[ch] < S ? peak[ch] : 0;
    const uint32_t k = enc[ch] < K ? enc[ch] : 0;
    const int sym = v > (int)S - 1 ? (int)S - 1 : v;
    const int r = rank_of_symbol((int)mode, (int)S, p, sym);
    const uint32_t e = codes[k * 16 + r];
    lut[(size_t)ch * kLut + v] = make_uint2(e & 0xFFFFu, e >> 16);
    if (v == 0) {
        if (zero_bits) zero_bits[ch] = 0;
        if (peak_out) peak_out[ch] = (uint8_t)p;
        if (enc_out) enc_out[ch] = (uint8_t)k;
    }
}

struct FinArgs {
    const unsigned long long *hist;
    const uint64_t *w0, *w1;
    const uint8_t *skipflag;
    const uint8_t *peak, *enc, *sclv;
    const uint32_t *sclv16;  // the SCLV rows padded to 16 bytes (CalArgs::sclv16)
    uint32_t C, S, mode;
    uint64_t *post;   // C*S rank order, may be NULL
    uint64_t *bits;   // may be NULL
    uint8_t *skipped; // may be NULL
};

// Rank-map the window histogram of one channel and price it (one thread).  COHERENT: the counts were added by
// other workgroups of the SAME launch (fused measure): read them at device scope.
template <bool COHERENT>
__device__ __forceinline__ void finalize_channel(const FinArgs &a, uint32_t ch, int p, uint32_t enc)
{
    const int S = (int)a.S;
    // Everything this needs is loaded BEFORE the first store: the byte-typed outputs may alias anything, so a load behind a
    // store is issued only once the store is -- one serialised L2 round trip per code length (S = 10: 7 us per channel
    // in the one-launch measure, where this runs on one lane at the end of a workgroup).
    const uint64_t n = a.w1[ch] - a.w0[ch];
    const u32x4 row4 = *reinterpret_cast<const u32x4 *>(a.sclv16 + (size_t)enc * 4);
    const uint64_t row_lo = (uint64_t)row4.x | ((uint64_t)row4.y << 32), row_hi = (uint64_t)row4.z | ((uint64_t)row4.w << 32);
    const uint8_t skipf = a.skipped ? a.skipflag[ch] : (uint8_t)0;
    uint64_t h[MH_LUT_SYMS];
    uint64_t rest = 0, b = 0;
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) {
        h[s] = 0;
        if (s < S - 1) {
            const unsigned long long *q = &a.hist[(size_t)ch * kHistStride + s];
            h[s] = COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
            rest += h[s];
        }
    }
    for (int k = 0; k < S; ++k) {
        const int sym = symbol_of_rank((int)a.mode, S, p, k);
        uint64_t v = n - rest;  // the top bin is the window length minus the rest
#pragma unroll
        for (int s = 0; s < MH_LUT_SYMS; ++s)
            if (s == sym && s < S - 1) v = h[s];
        if (a.post) a.post[(size_t)ch * S + k] = v;  // get_BR_with_approx_sort.py:193
        b += (((k < 8 ? row_lo : row_hi) >> (8 * (k & 7))) & 0xFFu) * v;  // :289 numerator
    }
    if (a.bits) a.bits[ch] = b;
    if (a.skipped) a.skipped[ch] = skipf;
}

// ------------------------------------------------------------------------------------------
// window histogram.  For symbols s = 0..S-2 count the bytes equal to s (the top bin is the
// window length minus the rest, so no clip pass is needed).  Equality is tested on four
// bytes at a time with the exact zero-byte trick; match flags accumulate in packed byte
// counters that are drained with v_sad_u8 every 63 vectors.
// ------------------------------------------------------------------------------------------
struct HistArgs {
    const uint8_t *data;
    const uint64_t *ch_off;
    const uint32_t *tile_ch;
    const uint64_t *tile_start;  // relative to the channel start
    const uint32_t *tile_n;
    unsigned long long *hist;    // [slot][16], zeroed before the launch
    const uint32_t *tile_slot;   // histogram slot of each tile; NULL = the tile's channel
    // Fused measure (mh_measure in ONE launch): the workgroup that adds a channel's LAST tile also calibrates
    // and prices the channel, then leaves histogram and ticket zero for the next call.  tile_cnt == NULL: off.
    const uint32_t *tile_cnt;    // tiles per channel (>= 1: channels with an empty window get an empty tile)
    uint32_t *tile_done;         // [C] arrival tickets, zero between launches
    CalArgs cal;
    FinArgs fin;
};

// the workgroup's contribution to one histogram bin; wait: return only when the add has been performed
__device__ __forceinline__ void hist_add(unsigned long long *bin, uint32_t v, bool wait)
{
    if (!wait) {
        atomicAdd(bin, (unsigned long long)v);
        return;
    }
    const unsigned long long before = atomicAdd(bin, (unsigned long long)v);
    asm volatile("" ::"v"(before));  // consume the returned value: the wave waits for it
}

// call at the very end of a window-histogram kernel, after the workgroup's hist_adds into a.hist
//
// The one-launch measure passes counts between workgroups -- on any of the 8 XCDs, whose L2s are not coherent with
// each other for ordinary accesses -- WITHOUT fences.  It is the hand-off the MI355X guide lists as valid with
// "8-byte agent-scope atomics on both sides" and "the workgroup whose add came last, told by the value its add
// returned" (MI355X_MICROARCH.md, inter-workgroup visibility).  Why a count can never be read before it has landed:
//
//  (1) Every add into hist[ch][bin] and every ticket on tile_done[ch] is an agent-scope atomic RMW: the operations
//      on one address form a single total order, whichever XCD issued them (they are not satisfied from a line an
//      XCD's L2 happens to hold; an atomic drops the line from the issuing L2).
//  (2) hist_add asks for the RETURNING form and consumes the result (`asm volatile("" :: "v"(before))`): the
//      compiler has to wait for it there, and the pre-op value only comes back once the add HAS BEEN PERFORMED.
//      ISA of k_hist<2> (hipcc 7.2, gfx950):   global_atomic_add_x2 v[2:3], v1, v[2:3], s[4:5] sc0
//                                              s_waitcnt vmcnt(0)
//                                              ...  s_barrier
//                                              global_atomic_add v2, v2, v3, s[28:29] sc0        <- the ticket
//  (3) The s_barrier therefore separates "all 256 threads' adds are performed" from the ticket: thread 0 ISSUES the
//      ticket atomic only after the barrier, i.e. after (2) held for every add of this workgroup.
//  (4) Tickets on one address are totally ordered (1).  The workgroup whose ticket returns tile_cnt - 1 issued it
//      after every other workgroup's ticket was performed, each of which was issued after that workgroup's adds
//      were performed (3): when the last arriver's ticket RETURNS, every add of the channel precedes, in the total
//      order of its bin, anything the last arriver does next.
//  (5) The last arriver reads the bins with agent-scope atomic LOADS (finalize_channel<true>:
//      __hip_atomic_load(.., __HIP_MEMORY_SCOPE_AGENT) -> `global_load_dwordx2 .. sc1`), issued after (4) by data
//      dependence (s_last through LDS + barrier).  An sc1 load is not served from this CU's L1 nor from a stale
//      line: it observes all adds of (4).  (The plain loads in calibrate_channel read the caller's input data only.)
//  (6) Re-arming: the last arriver zeroes the channel's 16 bins (one 128-byte line per channel: nobody else's) and its
//      ticket with PLAIN stores.  Nothing of this launch touches them afterwards (all tiles have arrived); the
//      stores are written back at the end of the kernel, which is ordered before the next launch on the stream.
//
// What is NOT relied on: any ordering between plain stores and atomics of different workgroups, any L2 state across
// XCDs, or in-order retirement across addresses beyond "a returned value means performed".  The textbook form -- an
// agent-scope release before the ticket, an acquire in the last arriver -- writes back / invalidates caches per
// workgroup and made this kernel 10x slower.
// tests/test_gpu_parity.py::test_fused_measure_at_its_limits runs it at 4096 channels x 4 tiles, 2^12 calibration.
__device__ __forceinline__ void measure_tail(const HistArgs &a, uint32_t ch)
{
    if (!a.tile_cnt) return;
    __shared__ uint32_t s_last;
    __syncthreads();   // every count of this workgroup has been added (and acknowledged)
    if (threadIdx.x == 0) s_last = atomicAdd(&a.tile_done[ch], 1u) + 1u == a.tile_cnt[ch] ? 1u : 0u;
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    int p;
    uint32_t k;
    calibrate_channel(a.cal, ch, lane, p, k);
    if (lane == 0) finalize_channel<true>(a.fin, ch, p, k);
    if (lane < kHistStride) a.hist[(size_t)ch * kHistStride + lane] = 0;  // (read above by lane 0 of this wave)
    if (lane == 0) a.tile_done[ch] = 0;
}

template <int NS>
__device__ __forceinline__ void hist_word(uint32_t x, uint32_t (&acc)[NS])
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t y = x ^ (0x01010101u * (uint32_t)s);
        const uint32_t t = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;  // bit7 set <=> byte != 0
        acc[s] += (~t >> 7) & 0x01010101u;
    }
}

template <int NS>
__global__ __launch_bounds__(256) void k_hist(HistArgs a)
{
    const uint32_t tile = blockIdx.x;
    const uint32_t ch = a.tile_ch[tile];
    const uint8_t *p = a.data + a.ch_off[ch] + a.tile_start[tile];
    const uint32_t n = a.tile_n[tile];
    const int tid = threadIdx.x;
    uint32_t cnt[NS], acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) cnt[s] = acc[s] = 0;

    uint32_t head = (uint32_t)((16 - ((uintptr_t)p & 15)) & 15);
    head = head < n ? head : n;
    const uint32_t nvec = (n - head) >> 4;
    const uint32_t tail = (n - head) & 15;
    if ((uint32_t)tid < head) {
        const int b = p[tid];
#pragma unroll
        for (int s = 0; s < NS; ++s) cnt[s] += (b == s);
    }
    if ((uint32_t)tid < tail) {
        const int b = p[head + (nvec << 4) + tid];
#pragma unroll
        for (int s = 0; s < NS; ++s) cnt[s] += (b == s);
    }
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p + head);
    int pending = 0;
#pragma unroll 4
    for (uint32_t i = tid; i < nvec; i += 256) {
        const u32x4 x = __builtin_nontemporal_load(q + i);
        hist_word<NS>(x.x, acc);
        hist_word<NS>(x.y, acc);
        hist_word<NS>(x.z, acc);
        hist_word<NS>(x.w, acc);
        if (++pending == 63) {  // 4 adds of <=1 per byte field and vector: 252 <= 255
            pending = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                cnt[s] = __builtin_amdgcn_sad_u8(acc[s], 0u, cnt[s]);
                acc[s] = 0;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) cnt[s] = __builtin_amdgcn_sad_u8(acc[s], 0u, cnt[s]);

    __shared__ uint32_t red[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t v = wave_sum_u32(cnt[s]);
        if ((tid & 63) == 0) red[s][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < NS) {
        const uint32_t v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        const uint32_t slot = a.tile_slot ? a.tile_slot[tile] : ch;
        if (v) hist_add(&a.hist[(size_t)slot * kHistStride + tid], v, a.tile_cnt != nullptr);
    }
    measure_tail(a, ch);
}


__global__ __launch_bounds__(256) void k_finalize(FinArgs a)
{
    const uint32_t ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= a.C) return;
    finalize_channel<false>(a, ch, a.peak[ch], a.enc[ch]);
}

// ------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------
struct EncArgs {
    const uint8_t *data;
    const uint64_t *ch_off, *w0;
    const uint32_t *seg_ch;
    const uint64_t *seg_first, *seg_n, *seg_off;
    const uint2 *lut;
    uint32_t *payload;
    uint64_t *seg_words;
    unsigned long long *ch_bits;
    uint32_t nseg;
    uint32_t stage_dw;  // staging dwords per lane = 8 * maxlen (256 samples * maxlen / 32)
    uint64_t chunk_stride;  // bytes between consecutive 16384-sample chunks of a channel; 0 = contiguous
    // per-wave-table kernel only: where the wave gets its channel's (peak, encoder) word from
    //   0  the LUT that k_calibrate / k_lut_preset wrote (`lut`)
    //   1  calibrates in the wave (window of <= kCalDirect samples) -- mh_encode is one launch
    //   2  the caller's preset word (peak_in / enc_in) -- mh_encode_preset is one launch
    uint32_t cal_mode;
    uint32_t S, mode, K;
    const uint8_t *sclv;          // K*S lengths
    const uint32_t *sclv16;       // the same rows padded to 16 bytes (4 dwords per row)
    const uint32_t *codes;        // K*16 bit-reversed code | len << 16, by rank
    const uint8_t *peak_in, *enc_in;
    uint8_t *peak_out, *enc_out, *skip_out;  // published by the channel's first record; may be NULL
    // cal_mode != 0: per-channel {bit total << 24 | finished records} words in plan scratch (zero
    // between launches); the LAST record of a channel to finish stores the total to ch_bits and re-zeroes
    unsigned long long *acc;
};

// Calibration of one channel by one wave (the body of k_calibrate for windows of <= 4096 samples):
// histogram of min(x, S-1) over x[0, c), first-max peak, approx-sort ranks, first-min encoder over
// the K rows -- lane k prices encoder k, so K = 35 costs as much as K = 1.
// Compressing data/get_BR_with_approx_sort.py:164-176, 254, 281; functions_1.py:75-90.
// Split in two so that the caller can put its own (long) loads between the halves: the vector-memory counter
// retires in order, so what the calibration needs is requested FIRST and waited for with the rest in flight.
struct CalLoads {
    int v0;      // sample `lane` of the window (masked in the second half when past its end)
    u32x4 len;   // code lengths of encoder `lane` by rank, one byte each (lanes >= K: a valid row, unused)
};

__device__ __forceinline__ CalLoads wave_calibrate_issue(const uint8_t *x, uint32_t c, uint32_t K,
                                                         const uint32_t *sclv16, int lane)
{
    // unconditional loads from clamped (always valid) addresses; what does not apply is masked in the second
    // half -- a load inside a branch would be waited for on the spot
    CalLoads l;
    const uint32_t i = (uint32_t)lane < c ? (uint32_t)lane : (c ? c - 1 : 0u);
    l.v0 = (int)x[i];
    const uint32_t row = (uint32_t)lane < K ? (uint32_t)lane : K - 1;
    l.len = *reinterpret_cast<const u32x4 *>(sclv16 + (size_t)row * 4);
    return l;
}

__device__ __forceinline__ void wave_calibrate_finish(const CalLoads &l, const uint8_t *x, uint32_t c, int S,
                                                      uint32_t mode, uint32_t K, const uint8_t *sclv, int lane,
                                                      int &p_out, uint32_t &k_out)
{
    // wave-uniform counts from ballots: they live in scalar registers, and so does everything derived from
    // them up to the pricing -- a record of a short channel spends ~150 instructions here, not ~700
    uint32_t cnt[MH_LUT_SYMS];
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] = 0;
    for (uint32_t i0 = 0; i0 < c; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        int v = i0 == 0 ? l.v0 : (int)x[i < c ? i : c - 1];
        v = i < c ? v : -1;  // -1 matches no symbol
        v = v > S - 1 ? S - 1 : v;
#pragma unroll
        for (int s = 0; s < MH_LUT_SYMS; ++s)
            if (s < S) cnt[s] += (uint32_t)__popcll(__ballot(v == s));
    }
    int p = 0;
    if (mode == MH_MODE_APPROX) {
        uint32_t best = cnt[0];
#pragma unroll
        for (int s = 1; s < MH_LUT_SYMS; ++s)
            if (s < S && cnt[s] > best) {  // first max wins (np.argmax)
                best = cnt[s];
                p = s;
            }
    }
    uint32_t sorted[MH_LUT_SYMS];
#pragma unroll
    for (int k = 0; k < MH_LUT_SYMS; ++k) {
        const int sym = k < S ? symbol_of_rank((int)mode, S, p, k) : 0;
        uint32_t v = 0;
#pragma unroll
        for (int s = 0; s < MH_LUT_SYMS; ++s) v = (s == sym) ? cnt[s] : v;
        sorted[k] = k < S ? v : 0;
    }
    // cost <= 9 * 4096 < 2^24 and k < 256: (cost << 8 | k) orders by cost, then by encoder index
    uint32_t key = 0xFFFFFFFFu;
    if ((uint32_t)lane < K) {
        uint32_t cost = 0;
#pragma unroll
        for (int r = 0; r < MH_LUT_SYMS; ++r) cost += ((l.len[r >> 2] >> (8 * (r & 3))) & 0xFFu) * sorted[r];  // sorted[] is 0 beyond S
        key = (cost << 8) | (uint32_t)lane;
    }
    for (uint32_t k = 64 + lane; k < K; k += 64) {  // more than 64 encoders: the rest from memory
        uint32_t cost = 0;
#pragma unroll
        for (int r = 0; r < MH_LUT_SYMS; ++r)
            if (r < S) cost += (uint32_t)sclv[k * S + r] * sorted[r];
        const uint32_t kk = (cost << 8) | k;
        key = kk < key ? kk : key;
    }
    key = wave_min_dpp(key);
    p_out = p;
    k_out = key & 0xFFu;
}

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------
struct DecArgs {
    const uint32_t *payload;
    const uint64_t *ch_off, *w0;
    const uint32_t *seg_ch;
    const uint64_t *seg_first, *seg_n, *seg_off;
    uint8_t *out;
    uint32_t nseg;
    // every read of the stream stays below payload + payload_words whatever the stream holds; a
    // segment whose headers point outside is abandoned and *err raised to this call's epoch
    // (mh_decode_status compares; nothing has to be cleared between calls)
    uint64_t payload_words;
    uint32_t *err;
    uint32_t epoch;
};

template <int FI, bool FULL>
__device__ __forceinline__ uint32_t decode_chunk(const uint32_t *__restrict__ in, uint32_t m,
                                                 const uint8_t *dtab, uint32_t mask,
                                                 uint8_t *__restrict__ out, int lane)
{
    // header (format revision 2, mh_device.hpp): two dependent reads, only words of this chunk
    const uint32_t w0 = in[0];
    const uint32_t mn = w0 & 0xFFFu, hwid = (w0 >> 12) & 15u, hw = hdr_words(hwid);
    uint32_t len = mn;
    if (hwid) {
        const uint32_t fb = 16u + (uint32_t)lane * hwid;
        uint64_t v = in[fb >> 5];
        if ((fb & 31) + hwid > 32) v |= (uint64_t)in[(fb >> 5) + 1] << 32;
        len += (uint32_t)(v >> (fb & 31)) & ((1u << hwid) - 1u);
    }
    const uint32_t incl = wave_scan_incl(len, lane);
    const uint32_t P = incl - len;
    const uint32_t B = __shfl(incl, 63, 64);
    const uint32_t *pay = in + hw;
    uint32_t wi = P >> 5, bp = P & 31;
    // 64-bit window + one word of read-ahead; reads run <= 3 words past the chunk's own words even
    // when a corrupt stream decodes to more bits than its header announces (index clamp below)
    const uint32_t nw = (B + 31) >> 5;
    uint64_t buf = (uint64_t)pay[wi] | ((uint64_t)pay[wi + 1] << 32);
    uint32_t nxt = pay[wi + 2];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const uint32_t base = ((uint32_t)k * kLanes + lane) * MH_PIECE;
        int cnt = MH_PIECE;
        if (!FULL) {
            const int c = (int)m - (int)base;
            cnt = c < 0 ? 0 : (c > MH_PIECE ? MH_PIECE : c);
        }
        u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < MH_PIECE; ++i) {
            if (FULL || i < cnt) {
                const uint32_t w = (uint32_t)(buf >> bp) & mask;
                const uint32_t e = dtab[w];
                o[i >> 2] |= (e & 15u) << (8 * (i & 3));
                bp += e >> 4;
            }
            if ((i + 1) % FI == 0 || i == MH_PIECE - 1) {
                if (bp >= 32) {
                    buf = (buf >> 32) | ((uint64_t)nxt << 32);
                    bp -= 32;
                    ++wi;
                    nxt = pay[(wi < nw ? wi : nw) + 2];
                }
            }
        }
        if (FULL || cnt == MH_PIECE) {
            *reinterpret_cast<u32x4_u *>(out + base) = o;
        } else {
#pragma unroll
            for (int i = 0; i < MH_PIECE; ++i)
                if (i < cnt) out[base + i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
        }
    }
    return hw + nw;
}

// ------------------------------------------------------------------------------------------
// dense re-packing
// ------------------------------------------------------------------------------------------
// Exclusive scan of the per-segment word counts in three small launches: block sums (coalesced,
// 2048 segments per workgroup), a one-workgroup scan of the block sums, and the in-block scan
// that writes the offsets.  (A single workgroup striding over every segment took longer than
// the copy itself at 3e5 segments.)
constexpr uint32_t kScanBlock = 2048;  // segments per workgroup = 256 threads x 8

__global__ __launch_bounds__(256) void k_scan_block_sums(const uint64_t *seg_words, uint64_t nseg, uint64_t *block_sum)
{
    __shared__ uint64_t red[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t i = base + (uint64_t)j * 256 + threadIdx.x;
        if (i < nseg) s += seg_words[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// in place: block_sum[b] -> sum of the blocks before b; total[0] = everything
__global__ __launch_bounds__(1024) void k_scan_top(uint64_t *block_sum, uint64_t nblocks, uint64_t *total)
{
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const uint64_t per = (nblocks + 1023) / 1024;
    const uint64_t lo = (uint64_t)tid * per, hi = lo + per < nblocks ? lo + per : nblocks;
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += block_sum[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t t = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint64_t run = part[tid] - s;
    for (uint64_t i = lo; i < hi; ++i) {
        const uint64_t v = block_sum[i];
        block_sum[i] = run;
        run += v;
    }
    if (tid == 1023) total[0] = part[1023];
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint64_t *seg_words, uint64_t nseg, const uint64_t *block_base,
                                                    uint64_t *dense_off)
{
    __shared__ uint64_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
    uint64_t run = block_base[blockIdx.x];
    for (int j = 0; j < 8; ++j) {  // rows of 256 consecutive segments
        const uint64_t i = base + (uint64_t)j * 256 + threadIdx.x;
        const uint64_t v = i < nseg ? seg_words[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        __syncthreads();  // wsum of the previous row consumed
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < nseg) dense_off[i] = run + before + incl - v;
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// One wave per segment, kCompactSegs segments per workgroup (a segment is only a few KiB: one workgroup per
// segment is dispatch-bound; more than one segment per WAVE serialises their round trips -- 16 / 8 / 4 per
// workgroup: 0.82 / 0.78 / 0.72 ms for 1.85 GB).  The destination is word-aligned only, so up to 3
// head words are peeled off to make the 16-byte stores aligned; the loads take the misalignment.
#ifndef MH_COMPACT_SEGS
#define MH_COMPACT_SEGS 4
#endif
constexpr uint32_t kCompactSegs = MH_COMPACT_SEGS;

// SELF_SCAN (<= kScanBlock segments: small recordings, stream blocks): no scan kernels in front -- every wave
// sums the word counts of the segments before its own (<= 32 loads per lane), writes its dense offset, and the
// wave of the last segment the total: ONE launch for the whole compaction, which is what such calls are bound by.
template <bool SELF_SCAN>
__global__ __launch_bounds__(256) void k_compact(const uint32_t *__restrict__ payload, const uint64_t *seg_off,
                                                 const uint64_t *seg_words, uint64_t *dense_off,
                                                 uint32_t *__restrict__ dense, uint64_t dense_cap, uint64_t nseg,
                                                 uint64_t *total)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (SELF_SCAN && nseg == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) total[0] = 0;
        return;
    }
    for (uint32_t s = wave; s < kCompactSegs; s += 4) {
        const uint64_t seg = (uint64_t)blockIdx.x * kCompactSegs + s;
        if (seg >= nseg) return;
        const uint64_t n = seg_words[seg];
        uint64_t d0;
        if (SELF_SCAN) {
            uint64_t part = 0;
            for (uint64_t i = lane; i < seg; i += 64) part += seg_words[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
            d0 = part;
            if (lane == 0) {
                dense_off[seg] = d0;
                if (seg + 1 == nseg) total[0] = d0 + n;
            }
        } else {
            d0 = dense_off[seg];
        }
        if (d0 + n > dense_cap) continue;  // host checks total_words afterwards
        const uint32_t *src = payload + seg_off[seg];  // slots start on 128-byte lines
        uint32_t *dst = dense + d0;
        uint64_t head = (4 - (d0 & 3)) & 3;
        if (head > n) head = n;
        if (lane < head) dst[lane] = src[lane];
        const uint64_t nv = (n - head) >> 2;
        for (uint64_t i0 = 0; i0 < nv; i0 += 8 * 64) {  // 8 KiB per wave in flight (a typical segment is < 6 KiB)
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t i = i0 + (uint64_t)j * 64 + lane;
                if (i < nv) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(src + head + 4 * i));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t i = i0 + (uint64_t)j * 64 + lane;
                if (i < nv) __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4 *>(dst + head + 4 * i));
            }
        }
        const uint64_t done = head + 4 * nv;
        if (lane < n - done) dst[done + lane] = src[done + lane];
    }
}

// ------------------------------------------------------------------------------------------
// synthetic MUA, re-binning
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t ch, uint64_t q)
{
    uint64_t z = (seed + 1) * 0x9E3779B97F4A7C15ULL + ch * 0xD1B54A32D192ED03ULL +
                 q * 0x8CB92BA72F3D8DD7ULL;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

// one thread = one 16-sample piece (4 hashes)
__global__ __launch_bounds__(256) void k_synth(uint8_t *data, const uint64_t *ch_off,
                                               const uint64_t *ch_len, uint32_t C,
                                               const uint32_t *thr, uint64_t seed)
{
    for (uint32_t ch = blockIdx.y; ch < C; ch += gridDim.y) {
        const uint64_t T = ch_len[ch];
        uint8_t *x = data + ch_off[ch];
        uint32_t th[15];
#pragma unroll
        for (int s = 0; s < 15; ++s) th[s] = thr[(size_t)ch * 15 + s];
        for (uint64_t pc = (uint64_t)blockIdx.x * 256 + threadIdx.x; pc * 16 < T;
             pc += (uint64_t)gridDim.x * 256) {
            u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint64_t z = mix64(seed, ch, pc * 4 + g);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const uint32_t u = (uint32_t)(z >> (16 * f)) & 0xFFFFu;
                    uint32_t v = 0;
#pragma unroll
                    for (int s = 0; s < 15; ++s) v += (u >= th[s]);
                    o[g] |= v << (8 * f);
                }
            }
            const uint64_t t0 = pc * 16;
            if (t0 + 16 <= T) {
                *reinterpret_cast<u32x4_u *>(x + t0) = o;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t0 + i < T) x[t0 + i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
}

}  // namespace mh
