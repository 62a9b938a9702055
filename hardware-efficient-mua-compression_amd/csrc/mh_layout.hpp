// mh_layout.hpp -- data-layout kernels either side of the codec.
//
//   k_rebin2<SAT>    per-channel re-binning (ref: Compressing data/functions_1.py:11-24 and the
//                    MATLAB histogram2 binning, Data/Load_and_bin_Sabes_store_as_mat_file.m:50-54):
//                    a workgroup stages a contiguous 32 KiB span of one channel in LDS with
//                    16-byte loads, then each thread sums its r bytes with v_sad_u8 on dwords.
//   k_deinterleave   time-major interleaved samples |CH1|CH2|...|CHN| per time step (the FPGA's
//                    compression-phase input order, ref: FPGA implementation/README.md:31) ->
//                    the channel-major layout the codec reads.  256(t) x 64(c) byte tiles
//                    through LDS: 64-byte row reads, 256-byte contiguous writes per channel.
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

constexpr int kTrT = 256, kTrC = 64, kTrPitch = kTrC + 4;  // LDS row pitch (bytes)

// in: [T][C] bytes (time-major).  out channel c: bytes out + out_off[c] + t.
__global__ __launch_bounds__(256) void k_deinterleave(const uint8_t *__restrict__ in, uint64_t T, uint32_t C,
                                                      uint8_t *__restrict__ out, const uint64_t *out_off)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTrT * kTrPitch];
    const uint32_t c0 = blockIdx.y * kTrC;
    const uint32_t cw = C - c0 < (uint32_t)kTrC ? C - c0 : (uint32_t)kTrC;
    for (uint64_t t0 = (uint64_t)blockIdx.x * kTrT; t0 < T; t0 += (uint64_t)gridDim.x * kTrT) {
        const uint32_t th = T - t0 < (uint64_t)kTrT ? (uint32_t)(T - t0) : (uint32_t)kTrT;
        __syncthreads();
        // load: 256 rows x 64 B; thread -> (row = i / 4, 16-byte quarter = i % 4), 4 passes
        for (uint32_t i = threadIdx.x; i < (uint32_t)kTrT * 4; i += 256) {
            const uint32_t row = i >> 2, q = (i & 3) * 16;
            if (row < th) {
                const uint8_t *src = in + (t0 + row) * C + c0 + q;
                uint8_t *dst = tile + row * kTrPitch + q;
                if (q + 16 <= cw) {
                    const u32x4 v = *reinterpret_cast<const u32x4_u *>(src);
                    reinterpret_cast<uint32_t *>(dst)[0] = v.x;
                    reinterpret_cast<uint32_t *>(dst)[1] = v.y;
                    reinterpret_cast<uint32_t *>(dst)[2] = v.z;
                    reinterpret_cast<uint32_t *>(dst)[3] = v.w;
                } else {
                    for (uint32_t k = q; k < cw; ++k) tile[row * kTrPitch + k] = in[(t0 + row) * C + c0 + k];
                }
            }
        }
        __syncthreads();
        // store: thread -> (channel = tid / 4, 64-sample quarter = tid % 4): 4 x 16-byte stores
        const uint32_t c = threadIdx.x >> 2, tq = (threadIdx.x & 3) * 64;
        if (c < cw) {
            uint8_t *dst = out + out_off[c0 + c] + t0 + tq;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t tb = tq + v * 16;
                if (tb + 16 <= th) {
                    u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        o[k >> 2] |= (uint32_t)tile[(tb + k) * kTrPitch + c] << (8 * (k & 3));
                    *reinterpret_cast<u32x4_u *>(dst + v * 16) = o;
                } else {
                    for (uint32_t k = tb; k < th; ++k) dst[k - tq] = tile[k * kTrPitch + c];
                }
            }
        }
    }
}

}  // namespace mh
