/* examples/abi_roundtrip.c -- libmuahuff from plain C: no Python, no torch.
 *
 * The drop-in boundary of this project is the C ABI in include/muahuff.h; this program is the
 * smallest complete client of it.  Device memory comes straight from the HIP runtime; the
 * input is generated on the GPU (mh_synth_poisson), then measured, encoded, decoded, and the
 * decoded window is compared on the host with min(x, S-1) -- the clip the reference applies
 * before it histograms a channel (Compressing data/get_BR_with_approx_sort.py:164).  It then
 * walks the rest of the device entry points: mh_compact + decode from the dense stream,
 * mh_encode_preset, mh_interleave / mh_deinterleave, mh_rebin.
 *
 * Build (done by __graft_entry__.build()):
 *   gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/abi_roundtrip.c \
 *       -o examples/abi_roundtrip -L<pkg dir> -lmuahuff -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,...
 * Run: examples/abi_roundtrip [channels] [bins]      (prints "OK ..." and exits 0 on success)
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "muahuff.h"

#define HIP(x)                                                                          \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define MH(x)                                                                  \
    do {                                                                       \
        int rc_ = (x);                                                         \
        if (rc_ != MH_OK) {                                                    \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, mh_last_error()); \
            return 3;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv)
{
    const uint32_t C = argc > 1 ? (uint32_t)atoi(argv[1]) : 24;
    const uint64_t T = argc > 2 ? (uint64_t)atoll(argv[2]) : 100003;
    const uint32_t S = 3, h = 6;
    const uint8_t sclv[3] = {1, 2, 2}; /* the chosen system: codewords 0, 10, 11 */

    char name[128], arch[64];
    int cus = 0;
    uint64_t hbm = 0;
    MH(mh_device_info(0, &cus, &hbm, name, sizeof name, arch, sizeof arch));

    /* channel-major layout, every channel on a 16-byte boundary */
    uint64_t *off = malloc(C * sizeof *off), *len = malloc(C * sizeof *len);
    uint64_t total = 0;
    for (uint32_t c = 0; c < C; ++c) {
        off[c] = total;
        len[c] = T - (c % 5) * 7; /* ragged */
        total += (len[c] + 15) & ~(uint64_t)15;
    }
    /* synthetic Poisson-like rates 0.2 .. 2.0 counts/bin as 16-bit CDF thresholds (15 per channel) */
    uint32_t *thr = malloc((size_t)C * 15 * sizeof *thr);
    for (uint32_t c = 0; c < C; ++c) {
        const double lam = 0.2 + 1.8 * c / (C > 1 ? C - 1 : 1);
        double em = 1.0, term = 1.0, cdf = 0.0; /* exp(-lam) by its series: no libm needed */
        for (int k = 1; k < 60; ++k) {
            term *= -lam / k;
            em += term;
        }
        double p = em;
        for (int s = 0; s < 15; ++s) {
            cdf += p;
            thr[c * 15 + s] = (uint32_t)(cdf >= 1.0 ? 65536 : cdf * 65536.0);
            p *= lam / (s + 1);
        }
    }

    uint8_t *d_data, *d_out, *d_peak, *d_enc, *d_skip;
    uint64_t *d_off, *d_len, *d_bits, *d_segw, *d_chbits, *d_post;
    uint32_t *d_thr, *d_pay;
    HIP(hipMalloc((void **)&d_data, total + 16));
    HIP(hipMalloc((void **)&d_out, total + 16));
    HIP(hipMemset(d_data, 0, total + 16));
    HIP(hipMemset(d_out, 0xEE, total + 16));
    HIP(hipMalloc((void **)&d_off, C * 8));
    HIP(hipMalloc((void **)&d_len, C * 8));
    HIP(hipMalloc((void **)&d_thr, (size_t)C * 15 * 4));
    HIP(hipMemcpy(d_off, off, C * 8, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_len, len, C * 8, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_thr, thr, (size_t)C * 15 * 4, hipMemcpyHostToDevice));
    MH(mh_synth_poisson(d_data, d_off, d_len, C, T, d_thr, 7, NULL));

    mh_plan *plan = NULL;
    MH(mh_plan_create(&plan, off, len, C, S, h, MH_MODE_APPROX, MH_WIN_AFTER_CAL, sclv, 1, 2));
    mh_plan_info_t info;
    MH(mh_plan_info(plan, &info));

    HIP(hipMalloc((void **)&d_peak, C));
    HIP(hipMalloc((void **)&d_enc, C));
    HIP(hipMalloc((void **)&d_skip, C));
    HIP(hipMalloc((void **)&d_bits, C * 8));
    HIP(hipMalloc((void **)&d_chbits, C * 8));
    HIP(hipMalloc((void **)&d_post, (size_t)C * S * 8));
    HIP(hipMalloc((void **)&d_segw, (info.n_segments + 1) * 8));
    HIP(hipMalloc((void **)&d_pay, info.payload_cap_words * 4));

    MH(mh_measure(plan, d_data, NULL, NULL, NULL, NULL, d_post, d_bits, NULL, NULL));
    MH(mh_encode(plan, d_data, d_pay, info.payload_cap_words, d_segw, d_chbits, d_peak, d_enc, d_skip, NULL));
    MH(mh_decode(plan, d_pay, info.payload_cap_words, NULL, d_peak, d_enc, d_out, NULL));
    HIP(hipDeviceSynchronize());

    uint8_t *x = malloc(total + 16), *y = malloc(total + 16);
    uint64_t *bits = malloc(C * 8), *chbits = malloc(C * 8), *post = malloc((size_t)C * S * 8);
    HIP(hipMemcpy(x, d_data, total, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(y, d_out, total, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(bits, d_bits, C * 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(chbits, d_chbits, C * 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(post, d_post, (size_t)C * S * 8, hipMemcpyDeviceToHost));

    uint64_t bad = 0, sum_bits = 0;
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t c0 = len[c] < 64 ? len[c] : 64; /* calibration cutoff min(2^h, T) */
        for (uint64_t t = 0; t < len[c]; ++t) {
            const uint8_t want = t < c0 ? 0xEE : (x[off[c] + t] < S - 1 ? x[off[c] + t] : S - 1);
            bad += y[off[c] + t] != want;
        }
        /* payload bits == code lengths . rank-ordered post histogram (the reference's numerator) */
        uint64_t dot = 0;
        for (uint32_t r = 0; r < S; ++r) dot += sclv[r] * post[(size_t)c * S + r];
        bad += dot != bits[c];
        bad += chbits[c] != bits[c];
        sum_bits += bits[c];
    }
    /* ---- dense form: mh_compact, then decode straight from the packed stream ---------------- */
    uint64_t *d_doff, *d_total, total_words = 0;
    uint32_t *d_dense;
    HIP(hipMalloc((void **)&d_doff, (info.n_segments + 1) * 8));
    HIP(hipMalloc((void **)&d_total, 8));
    HIP(hipMalloc((void **)&d_dense, info.payload_cap_words * 4));
    MH(mh_compact(plan, d_pay, d_segw, d_dense, info.payload_cap_words, d_doff, d_total, NULL));
    HIP(hipMemset(d_out, 0xEE, total + 16));
    MH(mh_decode(plan, d_dense, info.payload_cap_words, d_doff, d_peak, d_enc, d_out, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(&total_words, d_total, 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(y, d_out, total, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t c0 = len[c] < 64 ? len[c] : 64;
        for (uint64_t t = c0; t < len[c]; ++t)
            bad += y[off[c] + t] != (x[off[c] + t] < S - 1 ? x[off[c] + t] : S - 1);
    }
    bad += total_words * 32 < sum_bits; /* the container holds at least the code bits */

    /* ---- calibrate-then-stream: encode again with the (peak, encoder) words found above ----- */
    MH(mh_encode_preset(plan, d_data, d_peak, d_enc, d_pay, info.payload_cap_words, d_segw, d_chbits, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(chbits, d_chbits, C * 8, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < C; ++c) bad += chbits[c] != bits[c];

    /* ---- layout helpers: time-major <-> channel-major, re-binning ---------------------------- */
    {
        const uint64_t Tm = 1000;           /* first 1000 bins of every channel as a [Tm][C] block */
        uint8_t *d_tm, *d_cm, *tm = malloc(Tm * C), *cm = malloc((size_t)C * 1008);
        uint64_t *coff = malloc(C * 8), *d_coff;
        for (uint32_t c = 0; c < C; ++c) coff[c] = (uint64_t)c * 1008;
        HIP(hipMalloc((void **)&d_tm, Tm * C));
        HIP(hipMalloc((void **)&d_cm, (size_t)C * 1008));
        HIP(hipMalloc((void **)&d_coff, C * 8));
        HIP(hipMemcpy(d_coff, coff, C * 8, hipMemcpyHostToDevice));
        if (T >= Tm + 40) {
            MH(mh_interleave(d_data, d_off, Tm, C, d_tm, NULL));       /* channel-major -> |CH1|..|CHN| */
            MH(mh_deinterleave(d_tm, Tm, C, d_cm, d_coff, NULL));       /* and back into a new layout   */
            HIP(hipDeviceSynchronize());
            HIP(hipMemcpy(tm, d_tm, Tm * C, hipMemcpyDeviceToHost));
            HIP(hipMemcpy(cm, d_cm, (size_t)C * 1008, hipMemcpyDeviceToHost));
            for (uint32_t c = 0; c < C; ++c)
                for (uint64_t t = 0; t < Tm; ++t) {
                    bad += tm[t * C + c] != x[off[c] + t];
                    bad += cm[coff[c] + t] != x[off[c] + t];
                }
            /* 5 ms bins from the 1 ms counts, saturating like MATLAB uint8() */
            uint8_t *d_rb, *rb = malloc((size_t)C * 200);
            uint64_t *roff = malloc(C * 8), *rlen = malloc(C * 8), *d_roff, *d_rlen;
            for (uint32_t c = 0; c < C; ++c) { roff[c] = (uint64_t)c * 200; rlen[c] = Tm; }
            HIP(hipMalloc((void **)&d_rb, (size_t)C * 200));
            HIP(hipMalloc((void **)&d_roff, C * 8));
            HIP(hipMalloc((void **)&d_rlen, C * 8));
            HIP(hipMemcpy(d_roff, roff, C * 8, hipMemcpyHostToDevice));
            HIP(hipMemcpy(d_rlen, rlen, C * 8, hipMemcpyHostToDevice));
            MH(mh_rebin(d_cm, d_coff, d_rlen, C, Tm, 5, 1, d_rb, d_roff, NULL));
            HIP(hipDeviceSynchronize());
            HIP(hipMemcpy(rb, d_rb, (size_t)C * 200, hipMemcpyDeviceToHost));
            for (uint32_t c = 0; c < C; ++c)
                for (uint64_t b = 0; b < 200; ++b) {
                    unsigned sum = 0;
                    for (int k = 0; k < 5; ++k) sum += x[off[c] + b * 5 + k];
                    bad += rb[roff[c] + b] != (sum > 255 ? 255 : sum);
                }
        }
    }
    MH(mh_plan_destroy(plan));
    if (bad) {
        fprintf(stderr, "MISMATCH: %llu differences\n", (unsigned long long)bad);
        return 1;
    }
    printf("OK %s (%s, %d CUs): %u channels, %llu samples, %.4f bits/sample, decode == clip(x) (slots and dense), "
           "encoded bits == SCLV . histogram, preset encode, layout and re-binning helpers agree with the host\n",
           name, arch, cus, C, (unsigned long long)info.window_samples, info.window_samples ? (double)sum_bits / (double)info.window_samples : 0.0);
    return 0;
}
