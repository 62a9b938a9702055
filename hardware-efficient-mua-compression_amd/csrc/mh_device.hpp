// mh_device.hpp -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "muahuff.h"

namespace mh {

// 16-byte vectors that may sit at any byte address: gfx950 global memory handles unaligned
// dwordx4 accesses in hardware and hipcc emits global_load/store_dwordx4 for them.
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLanes = MH_LANES;
constexpr int kRows = MH_ROWS;
constexpr int kChunk = MH_CHUNK;
constexpr int kHdrWords = MH_HDR_WORDS;
constexpr int kLut = 16;      // symbol LUT entries: index min(raw value, 15)
constexpr int kDtab = 512;    // decode table bytes per channel (2^maxlen <= 512)
constexpr int kHistStride = 16;
#define MH_LUT_SYMS 10  // S <= 10 (the reference sweeps S = 2..10)

// rank of symbol s for a calibration histogram peaking at p (closed form of the reference's
// approx_sort, Compressing data/functions_1.py:75-90: order p, p-1, p+1, p-2, p+2, ... with
// the exhausted side skipped); identity for the no-sort mapper.
__host__ __device__ inline int rank_of_symbol(int mode, int S, int p, int s)
{
    if (mode == MH_MODE_NOSORT) return s;
    const int d = s - p, ad = d < 0 ? -d : d;
    const int a = p, b = S - 1 - p, m = a < b ? a : b;
    if (d == 0) return 0;
    if (ad <= m) return d < 0 ? 2 * ad - 1 : 2 * ad;
    return m + ad;
}

// inverse: symbol that holds rank k
__host__ __device__ inline int symbol_of_rank(int mode, int S, int p, int k)
{
    if (mode == MH_MODE_NOSORT) return k;
    const int a = p, b = S - 1 - p, m = a < b ? a : b;
    if (k == 0) return p;
    if (k <= 2 * m) {
        const int j = (k + 1) >> 1;
        return (k & 1) ? p - j : p + j;
    }
    const int j = k - m;
    return a > b ? p - j : p + j;
}

// Sum over the 64 lanes, the same value in every lane.  Six DPP adds (within a row of 16 lanes: row_shr 1, 2, 4, 8
// under bank masks; across rows: row_bcast 15 and 31) leave the total in lane 63 and v_readlane broadcasts it: no LDS
// crossbar round trips (six dependent ds_bpermute per sum before; the histogram kernels and the calibration reduce
// ten counters each).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, true);   // row_shr:4, lanes 4..15 of a row
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, true);   // row_shr:8, lanes 8..15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// The same scan in six DPP adds (a Hillis-Steele scan inside each row of 16 lanes, then row_bcast 15 / 31 carry the row
// totals upwards; lanes without a source add the identity 0).  All 64 lanes must be active.
// WHICH ONE: the DPP forms keep a reduction off the LDS crossbar -- a sixth of the latency, which is what a short
// launch waits for (wave-task encoders, partial chunks, histogram and calibration tails) -- but they are VALU work,
// and the long-channel S <= 5 encoders have none to spare: 1024 ch x 1e7 bins, S = 3 encode 1.995 -> 2.045 ms with
// DPP scan / min / max in the merge, where the ds_bpermute forms wait on an otherwise idle pipe
// (profiles/r03_dpp_reductions.txt).
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// the value lane 63 holds (e.g. the total behind wave_scan_incl), in every lane: one v_readlane, no LDS crossbar
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// min / max over the 64 lanes in every lane: the DPP ladder of wave_sum_u32 with the operation's identity for lanes
// without a source
#define MH_DPP_STEP(v, ident, ctrl, rmask, op)                                                                     \
    {                                                                                                              \
        const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rmask, 0xf, false); \
        v = op(t_, v);                                                                                             \
    }
#define MH_DPP_REDUCE(v, ident, op)        \
    MH_DPP_STEP(v, ident, 0x111, 0xf, op)  \
    MH_DPP_STEP(v, ident, 0x112, 0xf, op)  \
    MH_DPP_STEP(v, ident, 0x114, 0xf, op)  \
    MH_DPP_STEP(v, ident, 0x118, 0xf, op)  \
    MH_DPP_STEP(v, ident, 0x142, 0xa, op)  \
    MH_DPP_STEP(v, ident, 0x143, 0xc, op)

__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v)
{
    MH_DPP_REDUCE(v, 0xFFFFFFFFu, min_u32)
    return wave_last(v);
}

__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v)
{
    MH_DPP_REDUCE(v, 0u, max_u32)
    return wave_last(v);
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t t = __shfl_xor(v, d, 64);
        v = t < v ? t : v;
    }
    return v;
}

// (64-bit minimum in every lane: the same DPP ladder on both halves; used once per channel by the calibration)
#define MH_DPP_STEP64_MIN(v, ctrl, rmask)                                                                                    \
    {                                                                                                                        \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v), ctrl, rmask, 0xf, false);         \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)((v) >> 32), ctrl, rmask, 0xf, false); \
        const uint64_t t_ = (uint64_t)lo_ | ((uint64_t)hi_ << 32);                                                           \
        v = t_ < v ? t_ : v;                                                                                                 \
    }
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
    MH_DPP_STEP64_MIN(v, 0x111, 0xf)
    MH_DPP_STEP64_MIN(v, 0x112, 0xf)
    MH_DPP_STEP64_MIN(v, 0x114, 0xf)
    MH_DPP_STEP64_MIN(v, 0x118, 0xf)
    MH_DPP_STEP64_MIN(v, 0x142, 0xa)
    MH_DPP_STEP64_MIN(v, 0x143, 0xc)
    return (uint64_t)wave_last((uint32_t)v) | ((uint64_t)wave_last((uint32_t)(v >> 32)) << 32);
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t t = __shfl_xor(v, d, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---- chunk header, format revision 2 ---------------------------------------------------------
//   bits [0,12)  min  = shortest sub-stream length of the chunk (<= 256 * 9)
//   bits [12,16) w    = bits needed for (longest - shortest), 0..12
//   then 64 fields of w bits (sub-stream length - min), LSB-first from bit 16;
//   header words = ceil((16 + 64 w) / 32) = 1..25 (kHdrWords = 32 stays the bound slots are sized by)
__host__ __device__ __forceinline__ uint32_t hdr_words(uint32_t w) { return (16u + 64u * w + 31u) >> 5; }

// field width for sub-stream lengths spanning `range` = longest - shortest
__device__ __forceinline__ uint32_t hdr_width(uint32_t range) { return range ? 32u - (uint32_t)__builtin_clz(range) : 0u; }

// Sub-stream length of `lane` from the first 32 words of a chunk, lane i holding word i & 31
// (a full chunk is always longer than 32 words, so that load never leaves the chunk).
// Returns the length; w_out = field width of this chunk.
__device__ __forceinline__ uint32_t hdr_len_from_wave(uint32_t word, int lane, uint32_t &w_out)
{
    const uint32_t w0 = __shfl(word, 0, 64);
    const uint32_t mn = w0 & 0xFFFu, w = (w0 >> 12) & 15u;
    const uint32_t fb = 16u + (uint32_t)lane * w;
    const uint32_t lo = __shfl(word, (int)(fb >> 5), 64), hi = __shfl(word, (int)((fb >> 5) + 1) & 31, 64);
    const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
    w_out = w;
    return mn + ((uint32_t)(v >> (fb & 31)) & ((1u << w) - 1u));
}

}  // namespace mh
