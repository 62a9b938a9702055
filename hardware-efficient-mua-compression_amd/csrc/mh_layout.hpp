// mh_layout.hpp -- data-layout kernels either side of the codec.
//
//   k_rebin3<SAT>    fast path of the same for r >= 4 (see below)
//   k_rebin2<SAT>    per-channel re-binning (ref: Compressing data/functions_1.py:11-24 and the
//                    MATLAB histogram2 binning, Data/Load_and_bin_Sabes_store_as_mat_file.m:50-54):
//                    a workgroup stages a contiguous 32 KiB span of one channel in LDS with
//                    16-byte loads, then each thread sums its r bytes with v_sad_u8 on dwords.
//   k_deinterleave2  time-major interleaved samples |CH1|CH2|...|CHN| per time step (the FPGA's
//                    compression-phase input order, ref: FPGA implementation/README.md:31) ->
//                    the channel-major layout the codec reads: 256(t) x 128(c) byte tiles,
//                    dword-only LDS traffic, 4x4 byte transposes with v_perm_b32.
//   k_interleave     channel-major -> time-major, the inverse
#pragma once
#include "mh_device.hpp"

namespace mh {

constexpr uint32_t kRebinTileBytes = 32768;

__device__ __forceinline__ uint32_t sum_bytes_lds(const uint32_t *lds, uint32_t lo, uint32_t hi)
{
    // sum of bytes [lo, hi) of the LDS byte image `lds`
    uint32_t s = 0;
    const uint32_t w0 = lo >> 2, w1 = (hi + 3) >> 2;
    for (uint32_t w = w0; w < w1; ++w) {
        uint32_t v = lds[w];
        const uint32_t b0 = w << 2;
        if (b0 < lo) v &= 0xFFFFFFFFu << (8 * (lo - b0));
        if (b0 + 4 > hi) v &= 0xFFFFFFFFu >> (8 * (b0 + 4 - hi));
        s = __builtin_amdgcn_sad_u8(v, 0u, s);
    }
    return s;
}

template <bool SAT>
__global__ __launch_bounds__(256) void k_rebin2(const uint8_t *__restrict__ data, const uint64_t *in_off,
                                                const uint64_t *in_len, uint32_t C, uint32_t r,
                                                void *__restrict__ out, const uint64_t *out_off)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kRebinTileBytes / 4 + 4];
    const uint32_t bins_per_tile = kRebinTileBytes / r;
    for (uint32_t ch = blockIdx.y; ch < C; ch += gridDim.y) {
        const uint64_t T = in_len[ch], nb = (T + r - 1) / r;
        const uint8_t *x = data + in_off[ch];
        for (uint64_t b0 = (uint64_t)blockIdx.x * bins_per_tile; b0 < nb; b0 += (uint64_t)gridDim.x * bins_per_tile) {
            const uint64_t t0 = b0 * r;
            const uint64_t t1 = (b0 + bins_per_tile) * r < T ? (b0 + bins_per_tile) * r : T;
            const uint32_t nbytes = (uint32_t)(t1 - t0);
            __syncthreads();  // previous tile fully consumed
            const uint32_t nvec = nbytes >> 4;
            for (uint32_t i = threadIdx.x; i < nvec; i += 256)
                reinterpret_cast<u32x4 *>(tile)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(x + t0) + i);
            if (threadIdx.x < (nbytes & 15u))
                reinterpret_cast<uint8_t *>(tile)[(nvec << 4) + threadIdx.x] = x[t0 + (nvec << 4) + threadIdx.x];
            __syncthreads();
            const uint32_t nbins = (uint32_t)((nbytes + r - 1) / r);
            for (uint32_t b = threadIdx.x; b < nbins; b += 256) {
                const uint32_t lo = b * r, hi = lo + r < nbytes ? lo + r : nbytes;
                const uint32_t s = sum_bytes_lds(tile, lo, hi);
                if (SAT)
                    reinterpret_cast<uint8_t *>(out)[out_off[ch] + b0 + b] = (uint8_t)(s > 255u ? 255u : s);
                else
                    reinterpret_cast<uint32_t *>(out)[out_off[ch] + b0 + b] = s;
            }
        }
    }
}

// k_rebin3: the fast path for r >= 4.  A "unit" is g = 4 / gcd(r, 4) bins = u = g * r / 4 whole
// dwords, so every unit has the same dword/bin structure and the walk over it is wave-uniform:
// one v_sad_u8 per dword, one masked split where a bin boundary cuts a dword (scalar masks).
// Thread i of a pass owns unit i (u is odd for r = 5, 10, 20, 50, 100: conflict-free LDS reads)
// and stores its g results as one 4/2/1-byte (uint8 output) or 16/8/4-byte (uint32) access.
// A workgroup walks `tpw` consecutive tiles of one channel; the next tile's 16-byte loads are
// issued into registers before the current tile is summed.
// RC: the bin factor as a compile-time constant (the reference's periods 5, 10, 20, 50, 100: the
// walk unrolls into straight-line code with literal masks) or 0 for any r at run time.
template <bool SAT, int RC>
__global__ __launch_bounds__(256) void k_rebin3(const uint8_t *__restrict__ data, const uint64_t *in_off,
                                                const uint64_t *in_len, uint32_t C, uint32_t r_arg, uint32_t g_arg,
                                                uint32_t u_arg, uint32_t upt, uint32_t tpw, void *__restrict__ out,
                                                const uint64_t *out_off)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kRebinTileBytes / 4];
    constexpr uint32_t kG = RC == 0 ? 0u : RC % 4 == 0 ? 1u : RC % 2 == 0 ? 2u : 4u;
    const uint32_t r = RC ? (uint32_t)RC : r_arg, g = RC ? kG : g_arg, u = RC ? kG * (uint32_t)RC / 4 : u_arg;
    const uint32_t tile_bytes = upt * u * 4;  // <= kRebinTileBytes
    constexpr int kVec = kRebinTileBytes / 16 / 256;  // 16-byte vectors per thread and tile
    for (uint32_t ch = blockIdx.y; ch < C; ch += gridDim.y) {
        const uint64_t T = in_len[ch], nb = (T + r - 1) / r;
        const uint8_t *x = data + in_off[ch];
        const uint64_t ntiles = (T + tile_bytes - 1) / tile_bytes;
        for (uint64_t tile0 = (uint64_t)blockIdx.x * tpw; tile0 < ntiles; tile0 += (uint64_t)gridDim.x * tpw) {
            const uint64_t tend = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
            u32x4 R[kVec];
            auto fetch = [&](uint64_t t) {
                const uint64_t b0 = t * tile_bytes;
                const uint32_t nbytes = T - b0 < tile_bytes ? (uint32_t)(T - b0) : tile_bytes;
#pragma unroll
                for (int j = 0; j < kVec; ++j) {
                    const uint32_t o = ((uint32_t)j * 256 + threadIdx.x) * 16;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (o + 16 <= nbytes) {
                        v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(x + b0 + o));
                    } else if (o < nbytes) {  // ragged end of the channel: byte by byte, zero padded
                        for (uint32_t k = 0; o + k < nbytes; ++k) v[k >> 2] |= (uint32_t)x[b0 + o + k] << (8 * (k & 3));
                    }
                    R[j] = v;
                }
            };
            fetch(tile0);
            for (uint64_t t = tile0; t < tend; ++t) {
                __syncthreads();  // previous tile fully summed
#pragma unroll
                for (int j = 0; j < kVec; ++j) {
                    const uint32_t i = (uint32_t)j * 256 + threadIdx.x;
                    if (i * 16 < tile_bytes) reinterpret_cast<u32x4 *>(tile)[i] = R[j];
                }
                __syncthreads();
                if (t + 1 < tend) fetch(t + 1);
                const uint64_t bin0 = t * upt * g;  // first bin of this tile
                for (uint32_t idx = threadIdx.x; idx < upt; idx += 256) {
                    const uint64_t b = bin0 + (uint64_t)idx * g;
                    if (b >= nb) break;
                    const uint32_t *p = tile + idx * u;
                    uint32_t acc = 0, rem = r, k = 0, packed = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
                    auto emit = [&](uint32_t v) {
                        if (SAT) {
                            packed |= (v > 255u ? 255u : v) << (8 * k);
                        } else {
                            o0 = k == 0 ? v : o0;
                            o1 = k == 1 ? v : o1;
                            o2 = k == 2 ? v : o2;
                            o3 = k == 3 ? v : o3;
                        }
                        ++k;
                    };
                    auto step = [&](uint32_t v) {
                        if (rem >= 4) {
                            acc = __builtin_amdgcn_sad_u8(v, 0u, acc);
                            rem -= 4;
                        } else {  // bin boundary inside this dword (rem = 1..3 low bytes finish the bin)
                            const uint32_t m = (1u << (8 * rem)) - 1u;
                            emit(__builtin_amdgcn_sad_u8(v & m, 0u, acc));
                            acc = __builtin_amdgcn_sad_u8(v & ~m, 0u, 0u);
                            rem = r - (4 - rem);
                        }
                        if (rem == 0) {
                            emit(acc);
                            acc = 0;
                            rem = r;
                        }
                    };
                    if (RC) {  // compile-time unit: all reads first, then a fully unrolled walk
                        constexpr uint32_t kU = RC ? kG * (uint32_t)RC / 4 : 1u;
                        uint32_t vv[kU];
#pragma unroll
                        for (uint32_t j = 0; j < kU; ++j) vv[j] = p[j];
#pragma unroll
                        for (uint32_t j = 0; j < kU; ++j) step(vv[j]);
                    } else {
                        // LDS reads batched 8 ahead of the (wave-uniform) walk so their latency overlaps
                        for (uint32_t w0 = 0; w0 < u; w0 += 8) {
                            uint32_t vv[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) vv[j] = w0 + j < u ? p[w0 + j] : 0u;
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (w0 + j < u) step(vv[j]);
                        }
                    }
                    const uint32_t valid = nb - b < g ? (uint32_t)(nb - b) : g;
                    if (SAT) {
                        uint8_t *q = reinterpret_cast<uint8_t *>(out) + out_off[ch] + b;
                        if (valid == 4) {
                            *reinterpret_cast<uint32_t __attribute__((aligned(1))) *>(q) = packed;
                        } else if (valid == 2 && g == 2) {
                            *reinterpret_cast<uint16_t __attribute__((aligned(1))) *>(q) = (uint16_t)packed;
                        } else {
                            for (uint32_t j = 0; j < valid; ++j) q[j] = (uint8_t)(packed >> (8 * j));
                        }
                    } else {
                        uint32_t *q = reinterpret_cast<uint32_t *>(out) + out_off[ch] + b;
                        if (valid == 4) {
                            const u32x4 v4 = {o0, o1, o2, o3};
                            *reinterpret_cast<u32x4_u *>(q) = v4;
                        } else {
                            if (valid > 0) q[0] = o0;
                            if (valid > 1) q[1] = o1;
                            if (valid > 2) q[2] = o2;
                        }
                    }
                }
            }
        }
    }
}

// k_deinterleave2: 256(t) x 128(c) byte tiles, dword-granular LDS traffic only.
//   load : 8 x 16-byte global reads per thread (128 contiguous bytes per time step), written to
//          LDS as dwords (dword column = 4 channels) at column ^ swz(row);
//   turn : a thread owns 4 channels x 16 time steps: 16 ds_read_b32 (one per row), four 4x4 byte
//          transposes with v_perm_b32, 4 x 16-byte stores; 16 consecutive lanes cover 256
//          contiguous bytes of one channel.
//   swz(row) = 2 * (row / 16 % 16) ^ (row % 4) makes both LDS phases conflict-free: the reads
//   of a half-wave (2 column groups x 16 time blocks, same row % 16) and the writes of a
//   half-wave (4 consecutive rows x 8 quarters) each hit 32 distinct banks.
//   A workgroup walks `tpw` consecutive tiles of its 128-channel strip; other workgroups of the
//   CU cover its load latency (holding the next tile in registers across the turn costs 180
//   VGPRs and the occupancy that hides more).
constexpr int kTr2T = 256, kTr2C = 128;

// ---- packed pieces: the intermediate of the time-major (stream) path ----------------------------
// A piece = 16 consecutive samples of one channel.  The stream encoder clips at S-1 <= 9 anyway, so
// the de-interleaver may hand it min(x, 15) in 4 bits per sample -- or min(x, 3) in 2 bits when
// S <= 4 -- and halve / quarter the intermediate's trip through HBM.  Layout: plain little-endian bit
// packing, sample i of a channel's stream in bits [i * bits, (i + 1) * bits) -- so a byte of the 4-bit
// stream IS the encoder's pair-table index (s[2j] | s[2j+1] << 4) and a byte of the 2-bit stream indexes a
// four-symbol table directly; the encoder never spreads the samples back to bytes.  The de-interleaver
// gets this order for free by choosing which time rows feed which dword of its byte transposition.
__device__ __forceinline__ uint32_t clip_bytes(uint32_t d, uint32_t lim)
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t b = (d >> (8 * i)) & 0xFFu;
        b = b > lim ? lim : b;
        r |= b << (8 * i);
    }
    return r;
}

// o = one byte per sample in TURNED order (see tr2_row): PK = 4: o.x / o.y = even / odd samples of the
// piece's first 8, o.z / o.w of its last 8; PK = 2: o[f] byte j = sample 4j + f
template <int PK>
__device__ __forceinline__ void pack_piece(u32x4 o, uint32_t &p0, uint32_t &p1)
{
    constexpr uint32_t lim = (1u << PK) - 1u;
    constexpr uint32_t hi = 0x01010101u * (0xFFu & ~lim);
    if ((o.x | o.y | o.z | o.w) & hi) {  // rare: a count above the field's range
        o.x = clip_bytes(o.x, lim);
        o.y = clip_bytes(o.y, lim);
        o.z = clip_bytes(o.z, lim);
        o.w = clip_bytes(o.w, lim);
    }
    if (PK == 4) {
        p0 = o.x | (o.y << 4);
        p1 = o.z | (o.w << 4);
    } else {
        p0 = o.x | (o.y << 2) | (o.z << 4) | (o.w << 6);
        p1 = 0;
    }
}

// ragged-edge helpers of k_deinterleave2, kept out of line so the hot path's register
// allocation is not shaped by them
__device__ __noinline__ u32x4 tr2_load_partial(const uint8_t *src, uint32_t n)
{
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (uint32_t k = 0; k < n; ++k) w[k >> 2] |= (uint32_t)src[k] << (8 * (k & 3));
    const u32x4 v = {w[0], w[1], w[2], w[3]};
    return v;
}

__device__ __noinline__ void tr2_store_partial(uint8_t *dst, u32x4 o, uint32_t n)
{
    const uint32_t w[4] = {o.x, o.y, o.z, o.w};
    for (uint32_t t = 0; t < n; ++t) dst[t] = (uint8_t)(w[t >> 2] >> (8 * (t & 3)));
}

__device__ __forceinline__ uint32_t tr2_swz(uint32_t row) { return (((row >> 4) & 15u) << 1) ^ (row & 3u); }

// time row (of a 16-step block) whose byte becomes byte j of dword m of a channel's turned piece: bytes in
// time order for byte output; for the packed outputs the order that makes pack_piece's shifts-and-ors
// produce plain little-endian bit packing
template <int PK>
__device__ __forceinline__ constexpr int tr2_row(int m, int j)
{
    return PK == 0 ? 4 * m + j : PK == 2 ? 4 * j + m : (m >> 1) * 8 + 2 * j + (m & 1);
}

// PK = 0: bytes out (channel c = T bytes at out + out_off[c]); PK = 4 / 2: packed pieces out
// (channel c = ceil(T / 16) pieces of 8 / 4 bytes at out + out_off[c]; a cut last piece is zero-padded)
// abl (tuning builds pass it; 0 in production): 1 = no global stores, 2 = also no turn (loads + LDS writes only),
// 3 = loads only
template <int PK>
__global__ __launch_bounds__(256) void k_deinterleave2(const uint8_t *__restrict__ in, uint64_t T, uint32_t C,
                                                       uint32_t tpw, uint8_t *__restrict__ out,
                                                       const uint64_t *out_off, uint32_t abl = 0, uint64_t blk_stride = 0)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTr2T * (kTr2C / 4)];
    // 1-D grid, channel strip fastest: consecutive workgroups -- dealt round-robin over the XCDs, running at
    // the same time -- take the 128-byte strips of the SAME rows, so every 1 KiB-ish row of the time-major
    // matrix is consumed while its DRAM page is open.  (Strip-major order walks one strip through all of time
    // first: each page is then opened once per strip, and the kernel runs at 2.6 TB/s instead of 5+.)
    const uint32_t nstrip = (C + kTr2C - 1) / kTr2C;
    const uint32_t c0 = (blockIdx.x % nstrip) * kTr2C;
    const uint32_t cw = C - c0 < (uint32_t)kTr2C ? C - c0 : (uint32_t)kTr2C;
    const uint64_t ntiles = (T + kTr2T - 1) / kTr2T;
    const uint64_t gx = gridDim.x / nstrip;
    // channel bases of this thread's work items, loaded once: a global load inside the turn would sit
    // behind the prefetched tile in the in-order memory counter and drain it
    constexpr int kG = PK == 0 ? 1 : PK == 4 ? 2 : 4;
    constexpr int kUU = kG == 4 ? 1 : 2 / kG;
    uint64_t obase[4] = {0, 0, 0, 0};  // (packed modes: one work item per thread and tile)
    if (PK != 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = (threadIdx.x / (16u / kG)) * 4 + k;
            obase[k] = c < cw ? out_off[c0 + c] : 0;
        }
    }
    for (uint64_t tile0 = (uint64_t)(blockIdx.x / nstrip) * tpw; tile0 < ntiles; tile0 += gx * tpw) {
        const uint64_t tend = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
        u32x4 V[8];
        auto fetch = [&](uint64_t tl) {
            const uint64_t t0 = tl * kTr2T;
            const uint32_t th = T - t0 < (uint64_t)kTr2T ? (uint32_t)(T - t0) : (uint32_t)kTr2T;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = (uint32_t)j * 256 + threadIdx.x, row = i >> 3, q = (i & 7) * 16;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < th) {
                    const uint8_t *src = in + (t0 + row) * C + c0 + q;
                    if (q + 16 <= cw) {
                        v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(src));
                    } else if (q < cw) {
                        v = tr2_load_partial(src, cw - q);
                    }
                }
                V[j] = v;
            }
        };
        // PIPE: the next tile's loads are issued before this tile is turned (its registers stay live across
        // the turn); measured per variant
        constexpr bool PIPE = PK != 0;
        if (PIPE) fetch(tile0);
        for (uint64_t tl = tile0; tl < tend; ++tl) {
            const uint64_t t0 = tl * kTr2T;
            const uint32_t th = T - t0 < (uint64_t)kTr2T ? (uint32_t)(T - t0) : (uint32_t)kTr2T;
            if (!PIPE) fetch(tl);
            if (abl == 3) {  // ablation: consume the loads, nothing else
                uint32_t z = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) z ^= V[j].x ^ V[j].y ^ V[j].z ^ V[j].w;
                if (z == 0x12345678u) out[0] = 1;
                if (PIPE && tl + 1 < tend) fetch(tl + 1);
                continue;
            }
            __syncthreads();  // previous tile fully turned
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = (uint32_t)j * 256 + threadIdx.x, row = i >> 3, q4 = (i & 7) * 4;
                const uint32_t sw = tr2_swz(row);
                uint32_t *r = tile + row * (kTr2C / 4);
                r[(q4 + 0) ^ sw] = V[j].x;
                r[(q4 + 1) ^ sw] = V[j].y;
                r[(q4 + 2) ^ sw] = V[j].z;
                r[(q4 + 3) ^ sw] = V[j].w;
            }
            __syncthreads();
            if (PIPE && tl + 1 < tend) fetch(tl + 1);
            if (abl == 2) continue;
            // turn: a work item = 4 channels x G blocks of 16 time steps; G = 1 (bytes out), 2 (4-bit) or
            // 4 (2-bit), so that an item always emits 16 bytes per channel (narrow stores cost several
            // times a 16-byte store per byte on this part)
            constexpr int G = PK == 0 ? 1 : PK == 4 ? 2 : 4;
            constexpr uint32_t NTB = 16 / G;  // items along the tile's 256 time steps
#pragma unroll 1
            for (int uu = 0; uu < kUU; ++uu) {
                const uint32_t id = threadIdx.x + 256u * uu, tb = id % NTB, cg = id / NTB;
                if (cg >= (uint32_t)(kTr2C / 4) || cg * 4 >= cw || tb * G * 16 >= th) continue;
                u32x4 pk[4] = {};  // packed output of the item's 4 channels, filled by shifting pieces in
#pragma unroll 1
                for (int g = 0; g < G; ++g) {  // (rolled: the transposition's registers are reused per block)
                    const uint32_t tbb = tb * G + g;  // 16-step block inside the tile; rows past th hold zeros
                    uint32_t d[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        d[i] = tile[(tbb * 16 + i) * (kTr2C / 4) + (cg ^ (tbb << 1) ^ (uint32_t)(i & 3))];
                    u32x4 o[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {  // four rows -> dword m of each channel's 16 bytes
                        const uint32_t a = d[tr2_row<PK>(m, 0)], b = d[tr2_row<PK>(m, 1)], c = d[tr2_row<PK>(m, 2)],
                                       e = d[tr2_row<PK>(m, 3)];
                        const uint32_t t0_ = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
                        const uint32_t t1_ = __builtin_amdgcn_perm(e, c, 0x05010400u);  // c0 e0 c1 e1
                        const uint32_t t2_ = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
                        const uint32_t t3_ = __builtin_amdgcn_perm(e, c, 0x07030602u);
                        o[0][m] = __builtin_amdgcn_perm(t1_, t0_, 0x05040100u);
                        o[1][m] = __builtin_amdgcn_perm(t1_, t0_, 0x07060302u);
                        o[2][m] = __builtin_amdgcn_perm(t3_, t2_, 0x05040100u);
                        o[3][m] = __builtin_amdgcn_perm(t3_, t2_, 0x07060302u);
                    }
                    if (PK == 0) {
                        const bool whole = tbb * 16 + 16 <= th;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t c = cg * 4 + k;
                            if (c >= cw) break;
                            uint8_t *dst = out + out_off[c0 + c] + t0 + tbb * 16;
                            if (abl == 1) {
                                if ((o[k].x ^ o[k].y) == 0x12345678u && o[k].z == 77u) dst[0] = 1;
                            } else if (whole) {
                                __builtin_nontemporal_store(o[k], reinterpret_cast<u32x4_u *>(dst));
                            } else {
                                tr2_store_partial(dst, o[k], th - tbb * 16);
                            }
                        }
                    } else {  // rows past th were loaded as zeros: a cut piece is zero-padded
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint32_t p0, p1;
                            pack_piece<(PK ? PK : 4)>(o[k], p0, p1);
                            const u32x4 q = pk[k];
                            if (PK == 4) {
                                const u32x4 r = {q.z, q.w, p0, p1};
                                pk[k] = r;
                            } else {
                                const u32x4 r = {q.y, q.z, q.w, p0};
                                pk[k] = r;
                            }
                        }
                    }
                }
                if (PK != 0) {  // 16 bytes per channel: G pieces; channel regions are padded to 16 bytes
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t c = cg * 4 + k;
                        if (c >= cw) break;
                        const u32x4 pv = pk[k];
                        // blk_stride != 0: CHUNK-BLOCKED layout -- the 16384-sample chunk j of a channel sits at
                        // out_off[c] + j * blk_stride, so that one tile's stores stay inside one small region
                        const uint64_t piece = (t0 >> 4) + (uint64_t)tb * G;  // piece index in the channel
                        uint8_t *dst = out + obase[k] +
                                       (blk_stride ? (piece >> 10) * blk_stride + (piece & 1023u) * (PK == 4 ? 8 : 4)
                                                   : piece * (PK == 4 ? 8 : 4));
                        if (abl == 1) {
                            if ((pv.x ^ pv.y) == 0x12345678u && pv.z == 77u) dst[0] = 1;
                        } else {
                            // plain, not non-temporal: the workgroup's consecutive tiles extend the same lines of
                            // this channel, and the XCD's L2 merges them into whole lines before they go out --
                            // each visit of a channel's stream costs a DRAM row activation whatever it carries
                            if (abl == 4)
                                __builtin_nontemporal_store(pv, reinterpret_cast<u32x4_u *>(dst));
                            else
                                *reinterpret_cast<u32x4_u *>(dst) = pv;
                        }
                    }
                }
            }
        }
    }
}

// k_deinterleave_p<2>: time-major bytes -> 2-bit packed pieces (what k_deinterleave2<2> produced, byte for byte), with
// the PACKING DONE BEFORE THE TURN.  A piece is 16 consecutive time steps of one channel, 2 bits each, little-endian:
// its byte r holds steps 4r .. 4r+3.  In the time-major matrix those four steps are the same byte column of four
// consecutive rows, so `V0 | V1 << 2 | V2 << 4 | V3 << 6` of four rows' 16-byte vectors IS byte r of the pieces of 16
// channels -- three VALU instructions per dword for 16 samples, before anything is transposed.  What remains to be
// turned is a quarter of the data, so the LDS tile can hold FOUR times the time span: a workgroup's whole visit of
// `tpw` (4) tiles = 1024 time steps x 128 channels is packed into 32 KiB and turned at once, and every channel then
// receives its 256 output bytes of the visit in ONE run -- 16 lanes x 16 bytes, two whole lines -- where the byte-wise
// turn wrote 64-byte runs four times over and left the merging to the L2.
// Same grid, tiles, output addressing (plain and chunk-blocked) and zero padding as k_deinterleave2.
// LDS rows are 32 dwords (128 channels of one packed row); 4 more are skipped after every 16 rows, so that the 16 time
// quarters x 4 channel groups a wave reads at once sit in 64 different banks.
// (4-bit pieces -- S = 5..16 -- the same way: two rows per packed byte, `V0 | V1 << 4`, so a visit is 512 time steps;
// its item is 32 steps = two 8-byte pieces per channel, which the SAME four transposes produce: piece p = dwords 2p, 2p+1.)
template <int PK>
struct P2 {
    static constexpr int kRowsPerByte = 8 / PK;                        // time steps per packed row: 4 (2-bit), 2 (4-bit)
    static constexpr int kTpw = PK == 2 ? 4 : 2;                       // tiles of kTr2T rows per LDS tile (a workgroup's visit)
    static constexpr int kRows = kTpw * kTr2T / kRowsPerByte;          // packed rows in LDS: 256
    static constexpr int kDw = kRows * 32 + (kRows / 16) * 4;          // dwords of the LDS tile
    static constexpr int kItemSteps = 16 * kRowsPerByte;               // time steps of a work item: 64 / 32
    static constexpr int kGroups = kTr2T / kRowsPerByte / 32;          // packed rows per thread and tile: 2 / 4
};
constexpr int kP2Tpw = P2<2>::kTpw;
__device__ __forceinline__ uint32_t p2_row_dw(uint32_t rg) { return rg * 32u + (rg >> 4) * 4u; }

template <int PK>
__global__ __launch_bounds__(256) void k_deinterleave_p(const uint8_t *__restrict__ in, uint64_t T, uint32_t C, uint32_t tpw,
                                                        uint8_t *__restrict__ out, const uint64_t *out_off,
                                                        uint32_t abl = 0, uint64_t blk_stride = 0, uint32_t cached_stores = 0)
{
    typedef P2<PK> G;
    constexpr int RP = G::kRowsPerByte, NG = G::kGroups;
    __shared__ __attribute__((aligned(16))) uint32_t tile[G::kDw];
    const uint32_t nstrip = (C + kTr2C - 1) / kTr2C;
    const uint32_t c0 = (blockIdx.x % nstrip) * kTr2C;
    const uint32_t cw = C - c0 < (uint32_t)kTr2C ? C - c0 : (uint32_t)kTr2C;
    const uint64_t ntiles = (T + kTr2T - 1) / kTr2T;
    const uint64_t gx = gridDim.x / nstrip;
    tpw = tpw < (uint32_t)G::kTpw ? tpw : (uint32_t)G::kTpw;  // (the host passes kTpw; the tuning knob may pass less)
    // turn: two items per thread, item = (4 channels cg, 16 packed rows tq of the visit); lanes = tq fastest
    const uint32_t tq = threadIdx.x & 15u;
    uint64_t obase[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = (((threadIdx.x >> 4) + 16u * (uint32_t)u) << 2) + (uint32_t)k;
            obase[u][k] = c < cw ? out_off[c0 + c] : 0;
        }
    // loads: thread = (16-byte column chunk q, packed row rgl of a group); NG packed rows per thread and tile
    const uint32_t q = (threadIdx.x & 7u) * 16u, rgl = threadIdx.x >> 3;
    u32x4 V[8];  // the tile in flight: always the NEXT one to be packed, also across visits (its loads run under the turn)
    const uint64_t first = (uint64_t)(blockIdx.x / nstrip) * tpw;
    for (uint64_t tile0 = first; tile0 < ntiles; tile0 += gx * tpw) {
        const uint64_t tend = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
        const uint64_t t0v = tile0 * kTr2T;                       // first time step of the visit
        const uint64_t left = T - t0v;
        const uint32_t thv = left < (uint64_t)(tend - tile0) * kTr2T ? (uint32_t)left : (uint32_t)(tend - tile0) * kTr2T;
        auto fetch = [&](uint64_t tl) {
            const uint64_t t0 = tl * kTr2T;
            const uint32_t th = T - t0 < (uint64_t)kTr2T ? (uint32_t)(T - t0) : (uint32_t)kTr2T;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t row = ((uint32_t)(j / RP) * 32u + rgl) * (uint32_t)RP + (uint32_t)(j % RP);
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < th) {
                    const uint8_t *src = in + (t0 + row) * C + c0 + q;
                    if (q + 16 <= cw) {
                        v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(src));
                    } else if (q < cw) {
                        v = tr2_load_partial(src, cw - q);
                    }
                }
                V[j] = v;
            }
        };
        if (tile0 == first) fetch(tile0);
        __syncthreads();  // the previous visit is fully turned
        for (uint64_t tl = tile0; tl < tend; ++tl) {
            const uint32_t rg0 = (uint32_t)(tl - tile0) * (kTr2T / RP);  // first packed row of this tile in LDS
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                constexpr uint32_t lim = (1u << PK) - 1u, hi = 0x01010101u * (0xFFu & ~lim);
                u32x4 r[RP];
                u32x4 any = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    r[i] = V[RP * g + i];
                    any |= r[i];
                }
                if ((any.x | any.y | any.z | any.w) & hi) {  // rare: a count above the field's range
#pragma unroll
                    for (int i = 0; i < RP; ++i)
#pragma unroll
                        for (int w = 0; w < 4; ++w) r[i][w] = clip_bytes(r[i][w], lim);
                }
                u32x4 pv = r[0];
#pragma unroll
                for (int i = 1; i < RP; ++i) pv |= r[i] << (uint32_t)(PK * i);
                *reinterpret_cast<u32x4 *>(tile + p2_row_dw(rg0 + (uint32_t)g * 32u + rgl) + (threadIdx.x & 7u) * 4u) = pv;
            }
            if (tl + 1 < tend)
                fetch(tl + 1);
            else if (tile0 + gx * tpw < ntiles)
                fetch(tile0 + gx * tpw);
        }
        __syncthreads();
        if (abl == 2 || abl == 3) continue;
#pragma unroll  // (unrolled: obase[u] must stay in registers -- a scratch reload here waits with vmcnt(0), i.e. for the
        // next tile's loads as well)
        for (int u = 0; u < 2; ++u) {
            const uint32_t cg = (threadIdx.x >> 4) + 16u * (uint32_t)u;
            if (cg * 4 >= cw || tq * G::kItemSteps >= thv) continue;
            uint32_t d[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = tile[p2_row_dw(tq * 16u + (uint32_t)i) + cg];
            u32x4 pk[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {  // byte k of d[4m .. 4m+3] -> dword m of channel k's 16 output bytes
                const uint32_t a = d[4 * m], b = d[4 * m + 1], c = d[4 * m + 2], e = d[4 * m + 3];
                const uint32_t t0_ = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
                const uint32_t t1_ = __builtin_amdgcn_perm(e, c, 0x05010400u);  // c0 e0 c1 e1
                const uint32_t t2_ = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
                const uint32_t t3_ = __builtin_amdgcn_perm(e, c, 0x07030602u);
                pk[0][m] = __builtin_amdgcn_perm(t1_, t0_, 0x05040100u);
                pk[1][m] = __builtin_amdgcn_perm(t1_, t0_, 0x07060302u);
                pk[2][m] = __builtin_amdgcn_perm(t3_, t2_, 0x05040100u);
                pk[3][m] = __builtin_amdgcn_perm(t3_, t2_, 0x07060302u);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ch = cg * 4 + k;
                if (ch >= cw) break;
                constexpr uint32_t kPieceBytes = PK == 4 ? 8 : 4, kItemPieces = 16 / kPieceBytes;
                const uint64_t piece = (t0v >> 4) + (uint64_t)tq * kItemPieces;  // first of the item's pieces
                uint8_t *dst = out + obase[u][k] +
                               (blk_stride ? (piece >> 10) * blk_stride + (piece & 1023u) * kPieceBytes : piece * kPieceBytes);
                if (abl == 1) {
                    if ((pk[k].x ^ pk[k].y) == 0x12345678u && pk[k].z == 77u) dst[0] = 1;
                } else if (abl == 4 || cached_stores) {
                    // plain: an intermediate that fits the Infinity Cache is re-read from there by the encoder
                    // (cached_stores, set by the host for blocks up to 192 MiB of pieces); also the A/B knob
                    *reinterpret_cast<u32x4_u *>(dst) = pk[k];
                } else {
                    // non-temporal: a wave instruction writes whole lines here (16 lanes x 16 bytes per channel), nothing
                    // is left for the L2 to merge -- 1024 ch x 1e7 steps 2.38 (plain) / 2.30 ms (nt); k_deinterleave2's
                    // 64-byte runs were the other way round (2.59 / 3.03 ms)
                    __builtin_nontemporal_store(pk[k], reinterpret_cast<u32x4_u *>(dst));
                }
            }
        }
    }
}

// k_interleave: the inverse of k_deinterleave2 (channel-major -> time-major [T][C]) with the same
// tile, swizzle and byte transposes run the other way: 16-byte reads of 16 time steps of each of
// 4 channels, transposed in registers into 16 dwords (4 channels of one time step each) for
// LDS, then 128 contiguous bytes per time step written with 16-byte stores.
__global__ __launch_bounds__(256) void k_interleave(const uint8_t *__restrict__ in, const uint64_t *in_off,
                                                    uint64_t T, uint32_t C, uint32_t tpw,
                                                    uint8_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTr2T * (kTr2C / 4)];
    // 1-D grid, channel strip fastest: consecutive workgroups -- dealt round-robin over the XCDs, running at
    // the same time -- take the 128-byte strips of the SAME rows, so every 1 KiB-ish row of the time-major
    // matrix is consumed while its DRAM page is open.  (Strip-major order walks one strip through all of time
    // first: each page is then opened once per strip, and the kernel runs at 2.6 TB/s instead of 5+.)
    const uint32_t nstrip = (C + kTr2C - 1) / kTr2C;
    const uint32_t c0 = (blockIdx.x % nstrip) * kTr2C;
    const uint32_t cw = C - c0 < (uint32_t)kTr2C ? C - c0 : (uint32_t)kTr2C;
    const uint64_t ntiles = (T + kTr2T - 1) / kTr2T;
    const uint64_t gx = gridDim.x / nstrip;
    for (uint64_t tile0 = (uint64_t)(blockIdx.x / nstrip) * tpw; tile0 < ntiles; tile0 += gx * tpw) {
        const uint64_t tend = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
        for (uint64_t tl = tile0; tl < tend; ++tl) {
            const uint64_t t0 = tl * kTr2T;
            const uint32_t th = T - t0 < (uint64_t)kTr2T ? (uint32_t)(T - t0) : (uint32_t)kTr2T;
            __syncthreads();  // previous tile fully written out
#pragma unroll 1
            for (int uu = 0; uu < 2; ++uu) {
                const uint32_t id = threadIdx.x + 256u * uu, tb = id & 15u, cg = id >> 4;
                if (cg * 4 >= cw || tb * 16 >= th) continue;
                const uint32_t nt = th - tb * 16 < 16u ? th - tb * 16 : 16u;
                u32x4 x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t c = cg * 4 + k;
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    x[k] = z;
                    if (c < cw) {
                        const uint8_t *src = in + in_off[c0 + c] + t0 + tb * 16;
                        x[k] = nt == 16 ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(src))
                                        : tr2_load_partial(src, nt);
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {  // times 4m..4m+3 of the 4 channels -> one dword per time step
                    const uint32_t a = x[0][m], b = x[1][m], c = x[2][m], e = x[3][m];
                    const uint32_t t0_ = __builtin_amdgcn_perm(b, a, 0x05010400u);
                    const uint32_t t1_ = __builtin_amdgcn_perm(e, c, 0x05010400u);
                    const uint32_t t2_ = __builtin_amdgcn_perm(b, a, 0x07030602u);
                    const uint32_t t3_ = __builtin_amdgcn_perm(e, c, 0x07030602u);
                    const uint32_t d0 = __builtin_amdgcn_perm(t1_, t0_, 0x05040100u);
                    const uint32_t d1 = __builtin_amdgcn_perm(t1_, t0_, 0x07060302u);
                    const uint32_t d2 = __builtin_amdgcn_perm(t3_, t2_, 0x05040100u);
                    const uint32_t d3 = __builtin_amdgcn_perm(t3_, t2_, 0x07060302u);
                    const uint32_t r0 = tb * 16 + 4 * m, col = cg ^ (tb << 1);
                    tile[(r0 + 0) * (kTr2C / 4) + (col ^ 0u)] = d0;
                    tile[(r0 + 1) * (kTr2C / 4) + (col ^ 1u)] = d1;
                    tile[(r0 + 2) * (kTr2C / 4) + (col ^ 2u)] = d2;
                    tile[(r0 + 3) * (kTr2C / 4) + (col ^ 3u)] = d3;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = (uint32_t)j * 256 + threadIdx.x, row = i >> 3, q = (i & 7) * 16, q4 = (i & 7) * 4;
                if (row >= th || q >= cw) continue;
                const uint32_t sw = tr2_swz(row);
                const uint32_t *r = tile + row * (kTr2C / 4);
                const u32x4 v = {r[(q4 + 0) ^ sw], r[(q4 + 1) ^ sw], r[(q4 + 2) ^ sw], r[(q4 + 3) ^ sw]};
                uint8_t *dst = out + (t0 + row) * C + c0 + q;
                if (q + 16 <= cw)
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_u *>(dst));
                else
                    tr2_store_partial(dst, v, cw - q);
            }
        }
    }
}

}  // namespace mh
