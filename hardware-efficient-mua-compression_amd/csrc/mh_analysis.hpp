// mh_analysis.hpp -- the result consumers' reductions on the GPU,
// bit-exact with the NumPy statements of the reference:
//   k_power_draws   Analyse results/max_nb_channels_p_value_power_budget.py:100-105: for every random
//                   channel subset, np.sum(BRs[subset]) and the implant power built from it
//   k_reduce_rows   Analyse results/integrate_BR_and_BDP_results_into_excel.py:118-119:
//                   np.mean / np.max of a design point's per-channel bit rates
// float64 sums follow NumPy's pairwise summation (numpy/core/src/umath/loops_utils.h.src,
// pairwise_sum_DOUBLE: blocks of <= 128 with eight running sums, halves split at a multiple of 8),
// written with plain operators under `#pragma clang fp contract(off)` in every function: without it
// hipcc fuses a*b+c into one FMA and the last bits differ (HIP's __dadd_rn / __dmul_rn do not help --
// they are inline operators that carry the header's own contraction setting).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mh {

// value of element i of the vector being summed: a plain vector, or a gather table[idx[i * stride]]
struct VecRef {
    const double *v;
    const int32_t *idx;  // NULL = plain vector
    uint64_t stride;
    __device__ __forceinline__ double at(uint64_t i) const { return idx ? v[idx[i * stride]] : v[i]; }
};

__device__ inline double pairwise_block(const VecRef &a, uint64_t lo, uint64_t n)
{
#pragma clang fp contract(off)
    if (n < 8) {
        double res = 0.;
        for (uint64_t i = 0; i < n; ++i) res = res + a.at(lo + i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a.at(lo + j);
    uint64_t i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a.at(lo + i + j);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a.at(lo + i);
    return res;
}

// NumPy's recursion, iteratively: ranges above 128 elements split into a first half rounded down to
// a multiple of 8 and the rest; the stack holds pending right halves and partial sums.
__device__ inline double pairwise_sum(const VecRef &a, uint64_t n)
{
#pragma clang fp contract(off)
    // depth <= log2(n / 128) + 1; 48 levels cover any 64-bit n
    uint64_t lo_st[48], n_st[48];
    double acc_st[48];
    uint8_t state[48];  // 0 = left half pending, 1 = right half pending (acc holds the left sum)
    int sp = 0;
    uint64_t lo = 0;
    double ret = 0.;
    for (;;) {
        if (n <= 128) {
            ret = pairwise_block(a, lo, n);
            // unwind
            for (;;) {
                if (sp == 0) return ret;
                if (state[sp - 1] == 0) {  // that was a left half: now its right half
                    acc_st[sp - 1] = ret;
                    state[sp - 1] = 1;
                    const uint64_t N = n_st[sp - 1];
                    uint64_t n2 = N / 2;
                    n2 -= n2 % 8;
                    lo = lo_st[sp - 1] + n2;
                    n = N - n2;
                    break;
                }
                ret = acc_st[sp - 1] + ret;
                --sp;
            }
            continue;
        }
        lo_st[sp] = lo;
        n_st[sp] = n;
        state[sp] = 0;
        ++sp;
        uint64_t n2 = n / 2;
        n2 -= n2 % 8;
        n = n2;  // descend into the left half (same lo)
    }
}

// One thread per draw: x[d * x_stride] += comm_energy * sum(BR[idx[:, d]]) + per_channels + static_power
// (:104-105; per_channels = nb_channels * (ADC_power + chan_processing_power), a host float64).
// idx is [Z][n_draws] (draw-minor, so that the threads of a wave read consecutive words).
__global__ __launch_bounds__(256) void k_power_draws(const double *br, const int32_t *idx, uint32_t Z, uint64_t n_draws,
                                                     double comm_energy, double per_channels, double static_power,
                                                     double *x, uint64_t x_stride)
{
#pragma clang fp contract(off)
    const uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_draws) return;
    VecRef a{br, idx + d, n_draws};
    const double s = pairwise_sum(a, Z);
    const double temp = comm_energy * s + per_channels + static_power;  // left to right, as Python evaluates :104
    x[d * x_stride] = x[d * x_stride] + temp;
}

// One thread per row r = vals[row_off[r] .. row_off[r+1]): pairwise sum and np.max (NaN propagates).
__global__ __launch_bounds__(256) void k_reduce_rows(const double *vals, const uint64_t *row_off, uint64_t n_rows,
                                                     double *sum, double *mx)
{
#pragma clang fp contract(off)
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const uint64_t lo = row_off[r], n = row_off[r + 1] - lo;
    VecRef a{vals + lo, nullptr, 1};
    sum[r] = pairwise_sum(a, n);
    double m = n ? vals[lo] : __longlong_as_double(0x7FF8000000000000ll);
    bool nan = m != m;
    for (uint64_t i = 1; i < n; ++i) {
        const double v = vals[lo + i];
        nan |= v != v;
        m = v > m ? v : m;
    }
    mx[r] = nan ? __longlong_as_double(0x7FF8000000000000ll) : m;
}

}  // namespace mh
