// muahuff.hip -- C ABI (include/muahuff.h) over the gfx950 kernels in mh_kernels.hpp.
// Host side: argument checking, the segment/tile planner, table upload, kernel launches.
// There is no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mh_codec2.hpp"
#include "mh_layout.hpp"
#include "muahuff.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define MH_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(MH_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

uint32_t bitrev(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

int check_row(const uint8_t *row, int S, uint32_t *maxlen)
{
    uint32_t m = 0;
    for (int r = 0; r < S; ++r) {
        if (row[r] == 0 || row[r] > 9) return MH_ERR_SCLV;
        if (r && row[r] < row[r - 1]) return MH_ERR_SCLV;
        if (row[r] > m) m = row[r];
    }
    uint32_t kraft = 0;
    for (int r = 0; r < S; ++r) kraft += 1u << (m - row[r]);
    if (kraft != (1u << m)) return MH_ERR_SCLV;
    *maxlen = m;
    return MH_OK;
}

constexpr uint32_t kHistTile = 256 * 16 * 32;  // bytes of one histogram tile (128 KiB)
constexpr uint64_t kCalDirect = 4096;          // longest calibration window k_calibrate scans itself

template <typename T>
int upload(T **dst, const std::vector<T> &src)
{
    const size_t bytes = (src.size() ? src.size() : 1) * sizeof(T);
    MH_HIP(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (src.size()) MH_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return MH_OK;
}

template <typename T>
int alloc(T **dst, size_t n)
{
    MH_HIP(hipMalloc(reinterpret_cast<void **>(dst), (n ? n : 1) * sizeof(T)));
    return MH_OK;
}

}  // namespace

struct mh_plan {
    int device = 0;
    mh_plan_info_t info{};
    uint64_t max_T = 0;
    std::vector<uint32_t> seg_ch;
    std::vector<uint64_t> seg_first, seg_n, seg_off;
    uint64_t n_tiles = 0;
    // device tables
    uint64_t *d_ch_off = nullptr, *d_ch_len = nullptr, *d_w0 = nullptr, *d_w1 = nullptr;
    uint8_t *d_skip = nullptr, *d_sclv = nullptr;
    uint32_t *d_codes = nullptr;
    uint32_t *d_seg_ch = nullptr;
    uint64_t *d_seg_first = nullptr, *d_seg_n = nullptr, *d_seg_off = nullptr;
    uint32_t *d_tile_ch = nullptr, *d_tile_n = nullptr;
    uint64_t *d_tile_start = nullptr;
    // device scratch
    unsigned long long *d_hist = nullptr;
    uint8_t *d_peak = nullptr, *d_enc = nullptr, *d_dtab = nullptr, *d_dlen = nullptr;
    uint2 *d_lut = nullptr;
    // workgroup tasks: up to 4 consecutive segments of one channel
    uint32_t *d_task_seg0 = nullptr;
    uint8_t *d_task_n = nullptr;
    uint32_t n_tasks = 0;
    uint32_t W = 0;  // decode table index bits
    uint32_t dec_K = 4;  // symbols per decode-table lookup
    uint32_t dec_NR = 32; // staging registers per lane of the hybrid decoder
    uint64_t *d_scan = nullptr;  // block sums of mh_compact's segment scan
    // calibration windows above kCalDirect samples: tiles for the window-histogram kernel
    uint32_t *d_cal_tile_ch = nullptr, *d_cal_tile_n = nullptr;
    uint64_t *d_cal_tile_start = nullptr;
    unsigned long long *d_calhist = nullptr;
    uint64_t n_cal_tiles = 0;
    uint2 *d_dtab2 = nullptr;  // 4-symbol decode tables (dec_K == 4 plans only)
};

struct mh_sweep {
    uint32_t C = 0, nh = 0, ni = 0;
    std::vector<uint64_t> bounds;  // C * (ni + 1), sorted per channel
    uint64_t n_tiles = 0, n_slots = 0;
    uint64_t *d_ch_off = nullptr, *d_tile_start = nullptr, *d_slot_len = nullptr;
    uint32_t *d_tile_ch = nullptr, *d_tile_n = nullptr, *d_tile_slot = nullptr;
    unsigned long long *d_scratch = nullptr;
};

// device operations run on the plan's device: its tables live there
static int check_device(const mh_plan *p, const char *who)
{
    int d = -1;
    MH_HIP(hipGetDevice(&d));
    if (d != p->device)
        return fail(MH_ERR_ARG, "%s: plan was created on device %d, the current device is %d", who, p->device, d);
    return MH_OK;
}

static int launch_calibrate(mh_plan *p, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
                            uint8_t *peak, uint8_t *enc, hipStream_t st, unsigned long long *zero_hist,
                            unsigned long long *zero_bits, uint8_t *skip_dst)
{
    mh::CalArgs a;
    a.zero_hist = zero_hist;
    a.zero_bits = zero_bits;
    a.skip_src = p->d_skip;
    a.skip_dst = skip_dst;
    a.data = data;
    a.ch_off = p->d_ch_off;
    a.ch_len = p->d_ch_len;
    a.sclv = p->d_sclv;
    a.codes = p->d_codes;
    a.C = p->info.C;
    a.S = p->info.S;
    a.h = p->info.h;
    a.mode = p->info.mode;
    a.K = p->info.K;
    a.cutoff = cutoff;
    a.cal_sorted = cal_hist;
    a.peak = peak;
    a.enc = enc;
    a.lut = p->d_lut;
    a.pre_hist = nullptr;
    if (p->n_cal_tiles) {  // long calibration windows (2^h > kCalDirect): tiled histogram first
        MH_HIP(hipMemsetAsync(p->d_calhist, 0, (size_t)a.C * mh::kHistStride * sizeof(unsigned long long), st));
        mh::HistArgs ha;
        ha.data = data;
        ha.ch_off = p->d_ch_off;
        ha.tile_ch = p->d_cal_tile_ch;
        ha.tile_start = p->d_cal_tile_start;
        ha.tile_n = p->d_cal_tile_n;
        ha.hist = p->d_calhist;
        ha.tile_slot = nullptr;
        hipLaunchKernelGGL(mh::k_hist2<4>, dim3((unsigned)p->n_cal_tiles), dim3(256), 0, st, ha, p->info.S);
        a.pre_hist = p->d_calhist;
    }
    hipLaunchKernelGGL(mh::k_calibrate, dim3((a.C + 3) / 4), dim3(256), 0, st, a);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

template <int NS>
static void launch_hist(const mh::HistArgs &a, uint64_t n_tiles, hipStream_t st)
{
    hipLaunchKernelGGL(mh::k_hist<NS>, dim3((unsigned)n_tiles), dim3(256), 0, st, a);
}

int g_ablate = 0;  // debug only (mhdbg_set_ablation); 0 in production

// The launch helpers double as "prepare" helpers: with this thread-local flag set they only
// raise the kernel's dynamic-LDS limit (hipFuncSetAttribute) and do not launch.  mh_plan_create
// runs them once that way, so mh_encode / mh_decode issue nothing but stream work and stay
// capturable into a hipGraph.
thread_local bool g_prepare_only = false;
static inline bool st_prepare_only_flag() { return g_prepare_only; }

template <int LC, int PB, int ABL = 0>
static int launch_encode2(const mh::Enc2Args &a, hipStream_t st)
{
    const size_t lds = ((size_t)mh::kEncSharedDw + 4 * (size_t)mh::enc2_wave_dwords(a.e.stage_dw)) * sizeof(uint32_t);
    auto kern = mh::k_encode2<LC, PB, ABL>;
    if (!st_prepare_only_flag()) {
        hipLaunchKernelGGL(kern, dim3(a.t.ntask), dim3(256), lds, st, a);
    } else if (lds > 64 * 1024) {  // plan creation: raise the dynamic-LDS limit once, outside any capture
        MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        return MH_OK;
    } else {
        return MH_OK;
    }
    MH_HIP(hipGetLastError());
    return MH_OK;
}

template <int K, int M, int NR, int RL, bool HY>
static int launch_decode2(const mh::Dec2Args &a, hipStream_t st)
{
    const size_t lds = ((size_t)mh::dec2_shared_dwords(a.W, K) + 4 * (size_t)NR * 64) * sizeof(uint32_t);
    auto kern = mh::k_decode2<K, M, NR, RL, HY>;
    if (!st_prepare_only_flag()) {
        hipLaunchKernelGGL(kern, dim3(a.t.ntask), dim3(256), lds, st, a);
    } else if (lds > 64 * 1024) {
        MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        return MH_OK;
    } else {
        return MH_OK;
    }
    MH_HIP(hipGetLastError());
    return MH_OK;
}

// lane-private LDS staging of the encoder: worst case 8*maxlen dwords per lane, capped at 32
// (= 4 bits per sample averaged over a 256-sample sub-stream; a clipped spike-count channel at
// S <= 10 stays well below that).  Chunks that outgrow it take the two-pass global slow path.
static inline uint32_t enc_stage_dw(uint32_t maxlen) { return 8 * maxlen < 32 ? 8 * maxlen : 32; }

static int dispatch_encode(const mh_plan *p, const mh::Enc2Args &a2, hipStream_t st)
{
    const uint32_t L = p->info.maxlen;
    const bool pb3 = p->info.S <= 8;  // 3-bit pair packing when every symbol fits 3 bits
    if (L <= 2 && pb3 && g_ablate) {  // debug ablations of the S<=3 kernel
        switch (g_ablate) {
        case 1: return launch_encode2<0, 3, 1>(a2, st);
        case 2: return launch_encode2<0, 3, 2>(a2, st);
        case 3: return launch_encode2<0, 3, 3>(a2, st);
        default: return launch_encode2<0, 3, 4>(a2, st);
        }
    }
    if (L <= 2) return pb3 ? launch_encode2<0, 3>(a2, st) : launch_encode2<0, 4>(a2, st);
    if (L <= 4) return pb3 ? launch_encode2<1, 3>(a2, st) : launch_encode2<1, 4>(a2, st);
    if (L <= 8) return pb3 ? launch_encode2<2, 3>(a2, st) : launch_encode2<2, 4>(a2, st);
    return launch_encode2<3, 4>(a2, st);
}

static int dispatch_decode(const mh_plan *p, const mh::Dec2Args &a2, hipStream_t st)
{
    const uint32_t L = p->info.maxlen;
    // window maintenance (decode_staged_chunk): 1 = reload, 0 = branchy top-up, 2 = select top-up.
    // MH_DEC_RELOAD overrides for tuning; defaults are the measured best (profiles/README.md).
    static const int force = [] { const char *e = getenv("MH_DEC_RELOAD"); return e ? atoi(e) : -1; }();
    if (L <= 2) return launch_decode2<4, 4, 17, 1, false>(a2, st);  // worst-case chunk = 1027 words: never oversize
    // measured (profiles/README.md): select top-up for maxlen 3 and the hybrid table, branchy top-up between
    const int rl = force >= 0 ? force : (L == 3 || a2.W < 2 * L) ? 2 : 0;
    if (L == 3)
        return rl == 1 ? launch_decode2<2, 2, 25, 1, false>(a2, st)
             : rl == 2 ? launch_decode2<2, 2, 25, 2, false>(a2, st) : launch_decode2<2, 2, 25, 0, false>(a2, st);
    if (a2.W >= 2 * L)
        return rl == 1 ? launch_decode2<2, 2, 32, 1, false>(a2, st)
             : rl == 2 ? launch_decode2<2, 2, 32, 2, false>(a2, st) : launch_decode2<2, 2, 32, 0, false>(a2, st);
    // hybrid pair table: W < 2 * maxlen index bits, one-symbol entries flagged
    if (p->dec_NR == 31)
        return rl == 2 ? launch_decode2<2, 2, 31, 2, true>(a2, st) : launch_decode2<2, 2, 31, 0, true>(a2, st);
    return launch_decode2<2, 2, 32, 0, true>(a2, st);
}

// raise the dynamic-LDS limits of the kernels this plan will launch (once, at plan creation)
static int prepare_kernels(const mh_plan *p)
{
    mh::Enc2Args e{};
    e.e.stage_dw = enc_stage_dw(p->info.maxlen);
    mh::Dec2Args d{};
    d.W = p->W;
    g_prepare_only = true;
    int rc = dispatch_encode(p, e, nullptr);
    if (rc == MH_OK) rc = dispatch_decode(p, d, nullptr);
    g_prepare_only = false;
    return rc;
}

extern "C" {

int mh_version(void) { return MH_VERSION; }

/* debug hook, not part of the public ABI: selects a timing-only ablation of k_encode2 */
void mhdbg_set_ablation(int level) { g_ablate = level; }

const char *mh_last_error(void) { return g_err; }

int mh_device_info(int device, int *cu_count, uint64_t *hbm_bytes, char *name, int name_cap,
                   char *arch, int arch_cap)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MH_ERR_NO_DEVICE, "no HIP device visible (libmuahuff has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MH_ERR_ARG, "device %d out of range (%d)", device, n);
    hipDeviceProp_t p;
    MH_HIP(hipGetDeviceProperties(&p, device));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", p.name);
    if (arch && arch_cap > 0) snprintf(arch, (size_t)arch_cap, "%s", p.gcnArchName);
    return MH_OK;
}

int mh_codebook(const uint8_t *sclv_row, int S, uint16_t *code, uint8_t *len)
{
    if (!sclv_row || !code || !len || S < 2 || S > MH_LUT_SYMS) return fail(MH_ERR_ARG, "mh_codebook: bad argument");
    uint32_t m;
    if (check_row(sclv_row, S, &m) != MH_OK)
        return fail(MH_ERR_SCLV, "SCLV row is not a non-decreasing complete prefix-code length vector");
    uint32_t c = 0;
    for (int r = 0; r < S; ++r) {
        if (r) c = (c + 1) << (sclv_row[r] - sclv_row[r - 1]);
        code[r] = (uint16_t)c;
        len[r] = sclv_row[r];
    }
    return MH_OK;
}

int mh_approx_sort_perm(int S, int peak, uint8_t *idx)
{
    if (!idx || S < 2 || S > MH_LUT_SYMS || peak < 0 || peak >= S)
        return fail(MH_ERR_ARG, "mh_approx_sort_perm: bad argument");
    for (int k = 0; k < S; ++k) idx[k] = (uint8_t)mh::symbol_of_rank(MH_MODE_APPROX, S, peak, k);
    return MH_OK;
}

int mh_plan_destroy(mh_plan *p)
{
    if (!p) return MH_OK;
    void *ptrs[] = {p->d_ch_off, p->d_ch_len, p->d_w0, p->d_w1, p->d_skip, p->d_sclv, p->d_codes,
                    p->d_seg_ch, p->d_seg_first, p->d_seg_n, p->d_seg_off, p->d_tile_ch,
                    p->d_tile_n, p->d_tile_start, p->d_hist, p->d_peak, p->d_enc, p->d_dtab,
                    p->d_dlen, p->d_lut, p->d_task_seg0, p->d_task_n, p->d_dtab2, p->d_scan,
                    p->d_cal_tile_ch, p->d_cal_tile_n, p->d_cal_tile_start, p->d_calhist};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
    return MH_OK;
}

static int plan_build(mh_plan *p, const uint64_t *ch_off, const uint64_t *ch_len,
                      const uint8_t *sclv)
{
    const mh_plan_info_t &I = p->info;
    const uint32_t C = I.C, S = I.S, K = I.K;
    // windows: c = min(2^h, T) (functions_1.py:59-64), e = c + T/2 (get_BR_with_approx_sort.py:180)
    std::vector<uint64_t> w0(C), w1(C), off(ch_off, ch_off + C), len(ch_len, ch_len + C);
    std::vector<uint8_t> skip(C, 0);
    const uint64_t lim = (uint64_t)1 << I.h;
    uint64_t total = 0, nskip = 0;
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t T = ch_len[c];
        if (T > p->max_T) p->max_T = T;
        const uint64_t cut = T < lim ? T : lim, e = cut + T / 2;
        switch (I.window) {
        case MH_WIN_REF_HALF:
            if (e > T) {  // :183-185 skipped, shows up as NaN in the reference
                skip[c] = 1;
                w0[c] = w1[c] = cut;
                ++nskip;
            } else {
                w0[c] = cut;
                w1[c] = e;
            }
            break;
        case MH_WIN_REF_HALF_TRUNC: w0[c] = cut; w1[c] = e > T ? T : e; break;
        case MH_WIN_AFTER_CAL: w0[c] = cut; w1[c] = T; break;
        default: w0[c] = 0; w1[c] = T; break;
        }
        total += w1[c] - w0[c];
    }
    p->info.window_samples = total;
    p->info.n_skipped = nskip;
    // segments: seg_chunks chunks each, slot sized for the longest code of any encoder
    const uint64_t seg_samples = (uint64_t)I.seg_chunks * MH_CHUNK, L = I.maxlen;
    uint64_t slot = 0;
    std::vector<uint32_t> tile_ch, tile_n, task_seg0;
    std::vector<uint8_t> task_n;
    std::vector<uint64_t> tile_start;
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t n = w1[c] - w0[c];
        const size_t seg_begin = p->seg_ch.size();
        for (uint64_t first = 0; first < n; first += seg_samples) {
            const uint64_t m = n - first < seg_samples ? n - first : seg_samples;
            const uint64_t full = m / MH_CHUNK, rem = m % MH_CHUNK;
            uint64_t words = (full + (rem ? 1 : 0)) * MH_HDR_WORDS + full * ((MH_CHUNK * L + 31) / 32);
            if (rem) words += (rem * L + 31) / 32;
            words = (words + 31) & ~(uint64_t)31;  // slots start on 128-byte lines
            p->seg_ch.push_back(c);
            p->seg_first.push_back(first);
            p->seg_n.push_back(m);
            p->seg_off.push_back(slot);
            slot += words;
        }
        for (size_t s0 = seg_begin; s0 < p->seg_ch.size(); s0 += 4) {
            task_seg0.push_back((uint32_t)s0);
            task_n.push_back((uint8_t)(p->seg_ch.size() - s0 < 4 ? p->seg_ch.size() - s0 : 4));
        }
        for (uint64_t first = 0; first < n; first += kHistTile) {
            tile_ch.push_back(c);
            tile_start.push_back(w0[c] + first);
            tile_n.push_back((uint32_t)(n - first < kHistTile ? n - first : kHistTile));
        }
    }
    // calibration: one wave per channel reads the window directly up to kCalDirect samples (the
    // reference's range is 2^2..2^10); longer windows go through the tiled histogram kernel
    std::vector<uint32_t> cal_tile_ch, cal_tile_n;
    std::vector<uint64_t> cal_tile_start;
    if (lim > kCalDirect)
        for (uint32_t c = 0; c < C; ++c) {
            const uint64_t n = len[c] < lim ? len[c] : lim;
            for (uint64_t first = 0; first < n; first += kHistTile) {
                cal_tile_ch.push_back(c);
                cal_tile_start.push_back(first);
                cal_tile_n.push_back((uint32_t)(n - first < kHistTile ? n - first : kHistTile));
            }
        }
    p->n_cal_tiles = cal_tile_ch.size();
    p->info.n_segments = p->seg_ch.size();
    p->info.payload_cap_words = slot + 4;  // decode reads <= 3 words past the last chunk
    p->n_tiles = tile_ch.size();
    p->n_tasks = (uint32_t)task_seg0.size();
    // decode table: K symbols per lookup, W index bits.  maxlen <= 5: W = K * maxlen (<= 10), every
    // entry holds K whole codewords; longer codes: hybrid pair table of 10 index bits and 31
    // staging registers, which keeps 4 workgroups per CU (tables + staging <= 40 KiB of LDS).
    // MH_DEC_W / MH_DEC_NR: tuning overrides (index-bit cap 8..12, staging registers 31|32).
    p->dec_K = I.maxlen <= 2 ? 4 : 2;
    p->W = p->dec_K * I.maxlen;
    p->dec_NR = 32;
    if (p->dec_K == 2) {
        static const int w_env = [] { const char *e = getenv("MH_DEC_W"); return e ? atoi(e) : 0; }();
        static const int nr_env = [] { const char *e = getenv("MH_DEC_NR"); return e ? atoi(e) : 0; }();
        uint32_t cap = w_env >= 8 && w_env <= 12 ? (uint32_t)w_env : 10u;
        if (cap < I.maxlen) cap = I.maxlen;  // a flagged entry still holds its first codeword
        if (p->W > cap) p->W = cap;
        if (p->W < 2 * I.maxlen) p->dec_NR = nr_env == 32 ? 32 : 31;
    }
    // codebooks by rank: bit-reversed code (first code bit at bit 0) | len << 16
    std::vector<uint32_t> codes((size_t)K * 16, 0);
    for (uint32_t k = 0; k < K; ++k) {
        uint16_t code[16];
        uint8_t ln[16];
        int rc = mh_codebook(sclv + (size_t)k * S, (int)S, code, ln);
        if (rc != MH_OK) return rc;
        for (uint32_t r = 0; r < S; ++r) codes[k * 16 + r] = bitrev(code[r], ln[r]) | ((uint32_t)ln[r] << 16);
    }
    std::vector<uint8_t> sc(sclv, sclv + (size_t)K * S);
    int rc;
    if ((rc = upload(&p->d_ch_off, off)) || (rc = upload(&p->d_ch_len, len)) ||
        (rc = upload(&p->d_w0, w0)) || (rc = upload(&p->d_w1, w1)) || (rc = upload(&p->d_skip, skip)) ||
        (rc = upload(&p->d_sclv, sc)) || (rc = upload(&p->d_codes, codes)) ||
        (rc = upload(&p->d_seg_ch, p->seg_ch)) || (rc = upload(&p->d_seg_first, p->seg_first)) ||
        (rc = upload(&p->d_seg_n, p->seg_n)) || (rc = upload(&p->d_seg_off, p->seg_off)) ||
        (rc = upload(&p->d_tile_ch, tile_ch)) || (rc = upload(&p->d_tile_n, tile_n)) ||
        (rc = upload(&p->d_tile_start, tile_start)) ||
        (rc = alloc(&p->d_hist, (size_t)C * mh::kHistStride)) || (rc = alloc(&p->d_peak, C)) ||
        (rc = alloc(&p->d_enc, C)) || (rc = alloc(&p->d_dtab, (size_t)C * mh::kDtab)) ||
        (rc = alloc(&p->d_dlen, C)) || (rc = alloc(&p->d_lut, (size_t)C * mh::kLut)) ||
        (rc = upload(&p->d_task_seg0, task_seg0)) || (rc = upload(&p->d_task_n, task_n)) ||
        (p->dec_K == 4 && (rc = alloc(&p->d_dtab2, (size_t)C << p->W))) ||
        (rc = alloc(&p->d_scan, p->seg_ch.size() / mh::kScanBlock + 2)) ||
        (p->n_cal_tiles && ((rc = upload(&p->d_cal_tile_ch, cal_tile_ch)) || (rc = upload(&p->d_cal_tile_n, cal_tile_n)) ||
                            (rc = upload(&p->d_cal_tile_start, cal_tile_start)) ||
                            (rc = alloc(&p->d_calhist, (size_t)C * mh::kHistStride)))))
        return rc;
    return MH_OK;
}

int mh_plan_create(mh_plan **plan, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                   uint32_t S, uint32_t h, uint32_t mode, uint32_t window, const uint8_t *sclv,
                   uint32_t K, uint32_t seg_chunks)
{
    if (!plan || !ch_off || !ch_len || !sclv) return fail(MH_ERR_ARG, "mh_plan_create: NULL argument");
    *plan = nullptr;
    if (C == 0) return fail(MH_ERR_ARG, "mh_plan_create: C == 0");
    if (S < 2 || S > MH_LUT_SYMS) return fail(MH_ERR_ARG, "S=%u outside 2..10", S);
    if (h > 30) return fail(MH_ERR_ARG, "h=%u outside 0..30", h);
    if (mode > MH_MODE_APPROX) return fail(MH_ERR_ARG, "mode=%u unknown", mode);
    if (window > MH_WIN_FULL) return fail(MH_ERR_ARG, "window=%u unknown", window);
    if (K == 0 || K > 255) return fail(MH_ERR_ARG, "K=%u outside 1..255", K);
    if (seg_chunks == 0) seg_chunks = 2;  // measured sweet spot on MI355X (profiles/)
    uint32_t maxlen = 0;
    for (uint32_t k = 0; k < K; ++k) {
        uint32_t m;
        if (check_row(sclv + (size_t)k * S, (int)S, &m) != MH_OK)
            return fail(MH_ERR_SCLV, "SCLV row %u is not a non-decreasing complete code-length vector", k);
        if (m > maxlen) maxlen = m;
    }
    for (uint32_t c = 0; c < C; ++c)
        if (ch_len[c] == 0)
            return fail(MH_ERR_EMPTY_CHANNEL, "channel %u has no bins (the reference raises IndexError)", c);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MH_ERR_NO_DEVICE, "no HIP device visible (libmuahuff has no CPU fallback)");
    mh_plan *p = new (std::nothrow) mh_plan;
    if (!p) return fail(MH_ERR_ARG, "out of host memory");
    MH_HIP(hipGetDevice(&p->device));
    p->info.C = C;
    p->info.S = S;
    p->info.h = h;
    p->info.mode = mode;
    p->info.window = window;
    p->info.K = K;
    p->info.seg_chunks = seg_chunks;
    p->info.maxlen = maxlen;
    int rc = plan_build(p, ch_off, ch_len, sclv);
    if (rc == MH_OK) rc = prepare_kernels(p);
    if (rc != MH_OK) {
        mh_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return MH_OK;
}

int mh_plan_info(const mh_plan *plan, mh_plan_info_t *info)
{
    if (!plan || !info) return fail(MH_ERR_ARG, "mh_plan_info: NULL argument");
    *info = plan->info;
    return MH_OK;
}

int mh_plan_segments(const mh_plan *plan, uint32_t *seg_ch, uint64_t *seg_first, uint64_t *seg_n,
                     uint64_t *seg_off)
{
    if (!plan) return fail(MH_ERR_ARG, "mh_plan_segments: NULL plan");
    const size_t n = plan->seg_ch.size();
    if (seg_ch && n) memcpy(seg_ch, plan->seg_ch.data(), n * sizeof(uint32_t));
    if (seg_first && n) memcpy(seg_first, plan->seg_first.data(), n * sizeof(uint64_t));
    if (seg_n && n) memcpy(seg_n, plan->seg_n.data(), n * sizeof(uint64_t));
    if (seg_off && n) memcpy(seg_off, plan->seg_off.data(), n * sizeof(uint64_t));
    return MH_OK;
}

int mh_measure(mh_plan *p, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
               uint8_t *peak, uint8_t *enc, uint64_t *post_hist, uint64_t *bits,
               uint8_t *skipped, void *stream)
{
    if (!p || !data) return fail(MH_ERR_ARG, "mh_measure: NULL argument");
    if (int rc_ = check_device(p, "mh_measure")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    uint8_t *pk = peak ? peak : p->d_peak, *en = enc ? enc : p->d_enc;
    int rc = launch_calibrate(p, data, cutoff, cal_hist, pk, en, st, p->d_hist, nullptr, nullptr);
    if (rc) return rc;
    if (p->n_tiles) {
        mh::HistArgs a;
        a.data = data;
        a.ch_off = p->d_ch_off;
        a.tile_ch = p->d_tile_ch;
        a.tile_start = p->d_tile_start;
        a.tile_n = p->d_tile_n;
        a.hist = p->d_hist;
        a.tile_slot = nullptr;
        const unsigned nt = (unsigned)p->n_tiles;
        if (p->info.S == 2)
            launch_hist<1>(a, p->n_tiles, st);  // byte-compare kernel: already at the read floor
        else if (p->info.S == 3)
            launch_hist<2>(a, p->n_tiles, st);
        else if (p->info.S <= 8)  // pair-LUT histogram, 3-bit pair packing
            hipLaunchKernelGGL(mh::k_hist2<3>, dim3(nt), dim3(256), 0, st, a, p->info.S);
        else                      // S = 9, 10: 4-bit packing, xor-swizzled
            hipLaunchKernelGGL(mh::k_hist2<4>, dim3(nt), dim3(256), 0, st, a, p->info.S);
        MH_HIP(hipGetLastError());
    }
    mh::FinArgs f;
    f.hist = p->d_hist;
    f.w0 = p->d_w0;
    f.w1 = p->d_w1;
    f.skipflag = p->d_skip;
    f.peak = pk;
    f.enc = en;
    f.sclv = p->d_sclv;
    f.C = p->info.C;
    f.S = p->info.S;
    f.mode = p->info.mode;
    f.post = post_hist;
    f.bits = bits;
    f.skipped = skipped;
    hipLaunchKernelGGL(mh::k_finalize, dim3((f.C + 255) / 256), dim3(256), 0, st, f);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

static int encode_common(mh_plan *p, const uint8_t *data, uint32_t *payload, uint64_t *seg_words,
                         uint64_t *ch_bits, hipStream_t st)
{
    if (p->info.n_segments == 0) return MH_OK;
    mh::EncArgs a;
    a.data = data;
    a.ch_off = p->d_ch_off;
    a.w0 = p->d_w0;
    a.seg_ch = p->d_seg_ch;
    a.seg_first = p->d_seg_first;
    a.seg_n = p->d_seg_n;
    a.seg_off = p->d_seg_off;
    a.lut = p->d_lut;
    a.payload = payload;
    a.seg_words = seg_words;
    a.ch_bits = reinterpret_cast<unsigned long long *>(ch_bits);
    a.nseg = (uint32_t)p->info.n_segments;
    a.stage_dw = enc_stage_dw(p->info.maxlen);
    mh::Enc2Args a2;
    a2.e = a;
    a2.t.task_seg0 = p->d_task_seg0;
    a2.t.task_n = p->d_task_n;
    a2.t.ntask = p->n_tasks;
    return dispatch_encode(p, a2, st);
}

int mh_encode(mh_plan *p, const uint8_t *data, uint32_t *payload, uint64_t payload_cap_words,
              uint64_t *seg_words, uint64_t *ch_bits, uint8_t *peak, uint8_t *enc,
              uint8_t *skipped, void *stream)
{
    if (!p || !data || !payload || !seg_words || !ch_bits)
        return fail(MH_ERR_ARG, "mh_encode: NULL argument");
    if (int rc_ = check_device(p, "mh_encode")) return rc_;
    if (payload_cap_words < p->info.payload_cap_words)
        return fail(MH_ERR_CAPACITY, "payload buffer holds %llu words, plan needs %llu",
                    (unsigned long long)payload_cap_words,
                    (unsigned long long)p->info.payload_cap_words);
    hipStream_t st = (hipStream_t)stream;
    uint8_t *pk = peak ? peak : p->d_peak, *en = enc ? enc : p->d_enc;
    int rc = launch_calibrate(p, data, nullptr, nullptr, pk, en, st, nullptr,
                              reinterpret_cast<unsigned long long *>(ch_bits), skipped);
    if (rc) return rc;
    return encode_common(p, data, payload, seg_words, ch_bits, st);
}

int mh_encode_preset(mh_plan *p, const uint8_t *data, const uint8_t *peak, const uint8_t *enc,
                     uint32_t *payload, uint64_t payload_cap_words, uint64_t *seg_words,
                     uint64_t *ch_bits, void *stream)
{
    if (!p || !data || !peak || !enc || !payload || !seg_words || !ch_bits)
        return fail(MH_ERR_ARG, "mh_encode_preset: NULL argument");
    if (int rc_ = check_device(p, "mh_encode_preset")) return rc_;
    if (payload_cap_words < p->info.payload_cap_words)
        return fail(MH_ERR_CAPACITY, "payload buffer holds %llu words, plan needs %llu",
                    (unsigned long long)payload_cap_words,
                    (unsigned long long)p->info.payload_cap_words);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mh::k_lut_preset, dim3((p->info.C + 15) / 16), dim3(256), 0, st, peak, enc,
                       (const uint32_t *)p->d_codes, p->info.C, p->info.S, p->info.mode, p->info.K, p->d_lut,
                       reinterpret_cast<unsigned long long *>(ch_bits), (uint8_t *)nullptr, (uint8_t *)nullptr);
    MH_HIP(hipGetLastError());
    return encode_common(p, data, payload, seg_words, ch_bits, st);
}

int mh_decode(mh_plan *p, const uint32_t *payload, const uint64_t *seg_off, const uint8_t *peak,
              const uint8_t *enc, uint8_t *out, void *stream)
{
    if (!p || !payload || !peak || !enc || !out) return fail(MH_ERR_ARG, "mh_decode: NULL argument");
    if (int rc_ = check_device(p, "mh_decode")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    if (p->info.n_segments == 0) return MH_OK;
    mh::DecArgs a;
    a.payload = payload;
    a.ch_off = p->d_ch_off;
    a.w0 = p->d_w0;
    a.seg_ch = p->d_seg_ch;
    a.seg_first = p->d_seg_first;
    a.seg_n = p->d_seg_n;
    a.seg_off = seg_off ? seg_off : p->d_seg_off;
    a.dtab = p->d_dtab;
    a.dlen = p->d_dlen;
    a.out = out;
    a.nseg = (uint32_t)p->info.n_segments;
    mh::Dtab2Args t2;
    t2.peak = peak;
    t2.enc = enc;
    t2.sclv = p->d_sclv;
    t2.codes = p->d_codes;
    t2.C = p->info.C;
    t2.S = p->info.S;
    t2.mode = p->info.mode;
    t2.W = p->W;
    t2.K = p->dec_K;
    t2.dtab2 = p->d_dtab2;
    t2.dtab = p->d_dtab;
    t2.dlen = p->d_dlen;
    hipLaunchKernelGGL(mh::k_build_dtab2, dim3(t2.C), dim3(256), 0, st, t2);
    MH_HIP(hipGetLastError());
    mh::Dec2Args a2;
    a2.d = a;
    a2.t.task_seg0 = p->d_task_seg0;
    a2.t.task_n = p->d_task_n;
    a2.t.ntask = p->n_tasks;
    a2.dtab2 = p->d_dtab2;
    a2.W = p->W;
    return dispatch_decode(p, a2, st);
}

int mh_compact(mh_plan *p, const uint32_t *payload, const uint64_t *seg_words, uint32_t *dense,
               uint64_t dense_cap_words, uint64_t *dense_off, uint64_t *total_words, void *stream)
{
    if (!p || !payload || !seg_words || !dense || !dense_off || !total_words)
        return fail(MH_ERR_ARG, "mh_compact: NULL argument");
    if (int rc_ = check_device(p, "mh_compact")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t nseg = p->info.n_segments;
    const uint64_t nblocks = (nseg + mh::kScanBlock - 1) / mh::kScanBlock;
    if (nblocks <= 1) {
        hipLaunchKernelGGL(mh::k_scan_small, dim3(1), dim3(256), 0, st, seg_words, nseg, dense_off, total_words);
    } else {
        hipLaunchKernelGGL(mh::k_scan_block_sums, dim3((unsigned)nblocks), dim3(256), 0, st, seg_words, nseg, p->d_scan);
        hipLaunchKernelGGL(mh::k_scan_top, dim3(1), dim3(1024), 0, st, p->d_scan, nblocks, total_words);
        hipLaunchKernelGGL(mh::k_scan_apply, dim3((unsigned)nblocks), dim3(256), 0, st, seg_words, nseg,
                           (const uint64_t *)p->d_scan, dense_off);
    }
    MH_HIP(hipGetLastError());
    if (nseg) {
        const uint64_t nwg = (nseg + mh::kCompactSegs - 1) / mh::kCompactSegs;
        hipLaunchKernelGGL(mh::k_compact, dim3((unsigned)nwg), dim3(256), 0, st, payload,
                           (const uint64_t *)p->d_seg_off, seg_words, (const uint64_t *)dense_off,
                           dense, dense_cap_words, nseg);
        MH_HIP(hipGetLastError());
    }
    return MH_OK;
}

int mh_synth_poisson(uint8_t *data, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                     uint64_t max_len, const uint32_t *thr, uint64_t seed, void *stream)
{
    if (!data || !ch_off || !ch_len || !thr || C == 0) return fail(MH_ERR_ARG, "mh_synth_poisson: bad argument");
    uint64_t bx = (max_len + 256 * 16 - 1) / (256 * 16);
    if (bx == 0) bx = 1;
    if (bx > 4096) bx = 4096;
    const uint32_t by = C > 65535 ? 65535 : C;
    hipLaunchKernelGGL(mh::k_synth, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream, data,
                       ch_off, ch_len, C, thr, seed);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_rebin(const uint8_t *data, const uint64_t *in_off, const uint64_t *in_len, uint32_t C,
             uint64_t max_len, uint32_t r, int saturate, void *out, const uint64_t *out_off,
             void *stream)
{
    if (!data || !in_off || !in_len || !out || !out_off || C == 0 || r == 0)
        return fail(MH_ERR_ARG, "mh_rebin: bad argument");
    if (r > 4096) return fail(MH_ERR_ARG, "mh_rebin: r=%u above 4096", r);
    const uint32_t by = C > 65535 ? 65535 : C;
    if (r >= 4) {
        // unit = g bins = u whole dwords; tile = upt units (<= 32 KiB, whole passes of 256 threads
        // when there are that many); a workgroup walks 4 consecutive tiles of a channel
        const uint32_t g = r % 4 == 0 ? 1u : r % 2 == 0 ? 2u : 4u;
        const uint32_t u = g * r / 4;
        uint32_t upt = mh::kRebinTileBytes / 4 / u;
        if (upt >= 256) upt -= upt % 256;
        const uint32_t tpw = 4;
        const uint64_t tile_bytes = (uint64_t)upt * u * 4;
        uint64_t bx = ((max_len + tile_bytes - 1) / tile_bytes + tpw - 1) / tpw;
        if (bx == 0) bx = 1;
        if (bx > 65535) bx = 65535;
#define MH_REBIN3(SAT_, R_)                                                                              \
    hipLaunchKernelGGL((mh::k_rebin3<SAT_, R_>), dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream, \
                       data, in_off, in_len, C, r, g, u, upt, tpw, out, out_off)
#define MH_REBIN3_R(R_)               \
    do {                              \
        if (saturate)                 \
            MH_REBIN3(true, R_);      \
        else                          \
            MH_REBIN3(false, R_);     \
    } while (0)
        switch (r) {  // the reference's bin periods get straight-line kernels
        case 5: MH_REBIN3_R(5); break;
        case 10: MH_REBIN3_R(10); break;
        case 20: MH_REBIN3_R(20); break;
        case 50: MH_REBIN3_R(50); break;
        case 100: MH_REBIN3_R(100); break;
        default: MH_REBIN3_R(0); break;
        }
#undef MH_REBIN3_R
#undef MH_REBIN3
        MH_HIP(hipGetLastError());
        return MH_OK;
    }
    const uint64_t bins_per_tile = mh::kRebinTileBytes / r;
    uint64_t bx = ((max_len + r - 1) / r + bins_per_tile - 1) / bins_per_tile;
    if (bx == 0) bx = 1;
    if (bx > 4096) bx = 4096;
    if (saturate)
        hipLaunchKernelGGL(mh::k_rebin2<true>, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream,
                           data, in_off, in_len, C, r, out, out_off);
    else
        hipLaunchKernelGGL(mh::k_rebin2<false>, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream,
                           data, in_off, in_len, C, r, out, out_off);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_deinterleave(const uint8_t *in, uint64_t T, uint32_t C, uint8_t *out, const uint64_t *out_off,
                    void *stream)
{
    if (!in || !out || !out_off || C == 0) return fail(MH_ERR_ARG, "mh_deinterleave: bad argument");
    if (T == 0) return MH_OK;
    const uint32_t tpw = 4;
    uint64_t bx = ((T + mh::kTr2T - 1) / mh::kTr2T + tpw - 1) / tpw;
    if (bx > 0x7FFFFFFFull) bx = 0x7FFFFFFFull;
    const uint32_t by = (C + mh::kTr2C - 1) / mh::kTr2C;
    if (by > 65535) return fail(MH_ERR_ARG, "mh_deinterleave: C=%u too large", C);
    hipLaunchKernelGGL(mh::k_deinterleave2, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream, in, T, C, tpw,
                       out, out_off);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_interleave(const uint8_t *in, const uint64_t *in_off, uint64_t T, uint32_t C, uint8_t *out, void *stream)
{
    if (!in || !in_off || !out || C == 0) return fail(MH_ERR_ARG, "mh_interleave: bad argument");
    if (T == 0) return MH_OK;
    const uint32_t tpw = 4;
    uint64_t bx = ((T + mh::kTr2T - 1) / mh::kTr2T + tpw - 1) / tpw;
    if (bx > 0x7FFFFFFFull) bx = 0x7FFFFFFFull;
    const uint32_t by = (C + mh::kTr2C - 1) / mh::kTr2C;
    if (by > 65535) return fail(MH_ERR_ARG, "mh_interleave: C=%u too large", C);
    hipLaunchKernelGGL(mh::k_interleave, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream, in, in_off, T, C,
                       tpw, out);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_sweep_destroy(mh_sweep *w)
{
    if (!w) return MH_OK;
    void *ptrs[] = {w->d_ch_off, w->d_tile_start, w->d_slot_len, w->d_tile_ch, w->d_tile_n, w->d_tile_slot,
                    w->d_scratch};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete w;
    return MH_OK;
}

int mh_sweep_create(mh_sweep **sweep, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                    const uint32_t *hist_bits, uint32_t nh)
{
    if (!sweep || !ch_off || !ch_len || !hist_bits) return fail(MH_ERR_ARG, "mh_sweep_create: NULL argument");
    *sweep = nullptr;
    if (C == 0 || nh == 0 || nh > 16) return fail(MH_ERR_ARG, "mh_sweep_create: C=%u nh=%u", C, nh);
    for (uint32_t i = 0; i < nh; ++i)
        if (hist_bits[i] > 30) return fail(MH_ERR_ARG, "hist_bits[%u]=%u outside 0..30", i, hist_bits[i]);
    for (uint32_t c = 0; c < C; ++c)
        if (ch_len[c] == 0)
            return fail(MH_ERR_EMPTY_CHANNEL, "channel %u has no bins (the reference raises IndexError)", c);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MH_ERR_NO_DEVICE, "no HIP device visible (libmuahuff has no CPU fallback)");
    mh_sweep *w = new (std::nothrow) mh_sweep;
    if (!w) return fail(MH_ERR_ARG, "out of host memory");
    w->C = C;
    w->nh = nh;
    w->ni = 2 * nh + 1;
    const uint32_t np = w->ni + 1;
    w->bounds.resize((size_t)C * np);
    std::vector<uint32_t> tile_ch, tile_n, tile_slot;
    std::vector<uint64_t> tile_start, slot_len((size_t)C * w->ni), off(ch_off, ch_off + C);
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t T = ch_len[c];
        uint64_t *b = &w->bounds[(size_t)c * np];
        uint32_t k = 0;
        b[k++] = 0;
        for (uint32_t i = 0; i < nh; ++i) {
            const uint64_t lim = (uint64_t)1 << hist_bits[i];
            const uint64_t cut = T < lim ? T : lim;  // functions_1.py:59-64
            const uint64_t e = cut + T / 2;          // get_BR_with_approx_sort.py:180
            b[k++] = cut;
            b[k++] = e > T ? T : e;
        }
        b[k++] = T;
        for (uint32_t i = 1; i < np; ++i)  // insertion sort, np <= 34
            for (uint32_t j = i; j > 0 && b[j] < b[j - 1]; --j) {
                const uint64_t t = b[j];
                b[j] = b[j - 1];
                b[j - 1] = t;
            }
        for (uint32_t j = 0; j < w->ni; ++j) {
            const uint64_t lo = b[j], hi = b[j + 1], slot = (uint64_t)c * w->ni + j;
            slot_len[slot] = hi - lo;
            for (uint64_t first = lo; first < hi; first += kHistTile) {
                tile_ch.push_back(c);
                tile_slot.push_back((uint32_t)slot);
                tile_start.push_back(first);
                tile_n.push_back((uint32_t)(hi - first < kHistTile ? hi - first : kHistTile));
            }
        }
    }
    w->n_tiles = tile_ch.size();
    w->n_slots = (uint64_t)C * w->ni;
    int rc;
    if ((rc = upload(&w->d_ch_off, off)) || (rc = upload(&w->d_tile_ch, tile_ch)) ||
        (rc = upload(&w->d_tile_n, tile_n)) || (rc = upload(&w->d_tile_slot, tile_slot)) ||
        (rc = upload(&w->d_tile_start, tile_start)) || (rc = upload(&w->d_slot_len, slot_len)) ||
        (rc = alloc(&w->d_scratch, (size_t)w->n_slots * mh::kHistStride))) {
        mh_sweep_destroy(w);
        return rc;
    }
    *sweep = w;
    return MH_OK;
}

int mh_sweep_info(const mh_sweep *w, uint32_t *n_intervals, uint64_t *bounds)
{
    if (!w) return fail(MH_ERR_ARG, "mh_sweep_info: NULL sweep");
    if (n_intervals) *n_intervals = w->ni;
    if (bounds) memcpy(bounds, w->bounds.data(), w->bounds.size() * sizeof(uint64_t));
    return MH_OK;
}

int mh_sweep_run(mh_sweep *w, const uint8_t *data, uint64_t *hist, void *stream)
{
    if (!w || !data || !hist) return fail(MH_ERR_ARG, "mh_sweep_run: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    MH_HIP(hipMemsetAsync(w->d_scratch, 0, (size_t)w->n_slots * mh::kHistStride * sizeof(unsigned long long), st));
    if (w->n_tiles) {
        mh::HistArgs a;
        a.data = data;
        a.ch_off = w->d_ch_off;
        a.tile_ch = w->d_tile_ch;
        a.tile_start = w->d_tile_start;
        a.tile_n = w->d_tile_n;
        a.hist = w->d_scratch;
        a.tile_slot = w->d_tile_slot;
        hipLaunchKernelGGL(mh::k_hist2<4>, dim3((unsigned)w->n_tiles), dim3(256), 0, st, a, (uint32_t)MH_SWEEP_BINS);
        MH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(mh::k_sweep_finalize, dim3((unsigned)((w->n_slots + 255) / 256)), dim3(256), 0, st,
                       (const unsigned long long *)w->d_scratch, (const uint64_t *)w->d_slot_len,
                       (uint32_t)w->n_slots, hist);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

}  // extern "C"
