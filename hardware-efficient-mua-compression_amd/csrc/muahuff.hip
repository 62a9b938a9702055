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

#include "mh_analysis.hpp"
#include "mh_codec2.hpp"
#include "mh_layout.hpp"
#include "mh_planner.hpp"
// The library is built with -fvisibility=hidden: the C ABI of include/muahuff.h is ALL it exports
// (tests/test_host.py compares the dynamic symbol table with the header's prototypes).
#pragma GCC visibility push(default)
#include "muahuff.h"
#pragma GCC visibility pop

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

// Tuning knobs exist only in -DMH_TUNING builds (A/B runs, tools/); the production library reads
// no environment variable and exports no debug hook.
mh::PlanTuning plan_tuning()
{
    mh::PlanTuning t;
#ifdef MH_TUNING
    if (const char *e = getenv("MH_DEC_W")) t.dec_w_cap = atoi(e);
    if (const char *e = getenv("MH_DEC_NR")) t.dec_nr = atoi(e);
    if (const char *e = getenv("MH_WAVE_TASKS")) t.wave_tasks = atoi(e);
#endif
    return t;
}

}  // namespace

struct mh_plan {
    int device = 0;
    mh::PlanHost h;  // everything the planner computed (host copies)
    // device tables
    uint64_t *d_ch_off = nullptr, *d_ch_len = nullptr, *d_w0 = nullptr, *d_w1 = nullptr;
    uint8_t *d_skip = nullptr, *d_sclv = nullptr;
    uint32_t *d_sclv16 = nullptr;
    uint32_t *d_tile_cnt = nullptr, *d_tile_done = nullptr;  // fused measure: tiles per channel, arrival tickets  // the K rows padded to 16 bytes: one vector load per lane in the in-wave calibration
    uint32_t *d_codes = nullptr;
    uint32_t *d_seg_ch = nullptr;
    uint64_t *d_seg_first = nullptr, *d_seg_n = nullptr, *d_seg_off = nullptr;
    uint32_t *d_tile_ch = nullptr, *d_tile_n = nullptr;
    uint64_t *d_tile_start = nullptr;
    // device scratch
    unsigned long long *d_hist = nullptr;
    uint8_t *d_peak = nullptr, *d_enc = nullptr;
    uint2 *d_lut = nullptr;
    // shared-table kernels: workgroup tasks (first segment, count); per-wave-table kernels: the
    // segment of every wave task
    mh::WgTask *d_wg_tasks = nullptr;
    mh::WaveTask *d_wave_tasks = nullptr;
    uint64_t *d_scan = nullptr;  // block sums of mh_compact's segment scan
    // calibration windows above kCalDirect samples: tiles for the window-histogram kernel
    uint32_t *d_cal_tile_ch = nullptr, *d_cal_tile_n = nullptr;
    uint64_t *d_cal_tile_start = nullptr;
    unsigned long long *d_calhist = nullptr;
    unsigned long long *d_acc = nullptr;  // wave-task encoder: per-channel {bits << 24 | finished records} (zero between launches)
    uint32_t *d_err = nullptr;  // decode status word (mh_decode_status): non-zero once a decode abandoned a segment
};

struct mh_sweep {
    int device = 0;
    uint32_t C = 0, nh = 0, ni = 0;
    std::vector<uint64_t> bounds;  // C * (ni + 1), sorted per channel
    uint64_t n_tiles = 0, n_slots = 0;
    uint64_t *d_ch_off = nullptr, *d_tile_start = nullptr, *d_slot_len = nullptr;
    uint32_t *d_tile_ch = nullptr, *d_tile_n = nullptr, *d_tile_slot = nullptr;
    unsigned long long *d_scratch = nullptr;
};

// device operations run on the plan's device: its tables live there
static int check_device(int device, const char *who)
{
    int d = -1;
    MH_HIP(hipGetDevice(&d));
    if (d != device)
        return fail(MH_ERR_ARG, "%s: the plan was created on device %d, the current device is %d", who, device, d);
    return MH_OK;
}

static mh::CalArgs calibrate_args(const mh_plan *p, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
                                  uint8_t *peak, uint8_t *enc, unsigned long long *zero_hist,
                                  unsigned long long *zero_bits, uint8_t *skip_dst);

static int launch_calibrate(mh_plan *p, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
                            uint8_t *peak, uint8_t *enc, hipStream_t st, unsigned long long *zero_hist,
                            unsigned long long *zero_bits, uint8_t *skip_dst)
{
    const mh_plan_info_t &I = p->h.info;
    mh::CalArgs a = calibrate_args(p, data, cutoff, cal_hist, peak, enc, zero_hist, zero_bits, skip_dst);
    if (!p->h.cal_tile_ch.empty()) {  // long calibration windows (2^h > kCalDirect): tiled histogram first
        MH_HIP(hipMemsetAsync(p->d_calhist, 0, (size_t)a.C * mh::kHistStride * sizeof(unsigned long long), st));
        mh::HistArgs ha{};
        ha.data = data;
        ha.ch_off = p->d_ch_off;
        ha.tile_ch = p->d_cal_tile_ch;
        ha.tile_start = p->d_cal_tile_start;
        ha.tile_n = p->d_cal_tile_n;
        ha.hist = p->d_calhist;
        ha.tile_slot = nullptr;
        hipLaunchKernelGGL(mh::k_hist2<4>, dim3((unsigned)p->h.cal_tile_ch.size()), dim3(256), 0, st, ha, I.S);
        a.pre_hist = p->d_calhist;
    }
    hipLaunchKernelGGL(mh::k_calibrate, dim3((a.C + 3) / 4), dim3(256), 0, st, a);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

static mh::CalArgs calibrate_args(const mh_plan *p, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
                                  uint8_t *peak, uint8_t *enc, unsigned long long *zero_hist,
                                  unsigned long long *zero_bits, uint8_t *skip_dst)
{
    const mh_plan_info_t &I = p->h.info;
    mh::CalArgs a;
    a.zero_hist = zero_hist;
    a.zero_bits = zero_bits;
    a.skip_src = p->d_skip;
    a.skip_dst = skip_dst;
    a.data = data;
    a.ch_off = p->d_ch_off;
    a.ch_len = p->d_ch_len;
    a.sclv = p->d_sclv;
    a.sclv16 = p->d_sclv16;
    a.codes = p->d_codes;
    a.C = I.C;
    a.S = I.S;
    a.h = I.h;
    a.mode = I.mode;
    a.K = I.K;
    a.cutoff = cutoff;
    a.cal_sorted = cal_hist;
    a.peak = peak;
    a.enc = enc;
    a.lut = p->d_lut;
    a.pre_hist = nullptr;
    return a;
}

template <int NS>
static void launch_hist(const mh::HistArgs &a, uint64_t n_tiles, hipStream_t st)
{
    hipLaunchKernelGGL(mh::k_hist<NS>, dim3((unsigned)n_tiles), dim3(256), 0, st, a);
}

#ifdef MH_TUNING
static int g_ablate = 0;  // timing-only ablations of the S <= 3 encoder (mhdbg_set_ablation)
#endif

// The launch helpers double as "prepare" helpers: with this thread-local flag set they only
// raise the kernel's dynamic-LDS limit (hipFuncSetAttribute), check that the kernel has no static
// LDS (the decoders address their table by raw LDS offset) and do not launch.  mh_plan_create
// runs them once that way, so mh_encode / mh_decode issue nothing but stream work and stay
// capturable into a hipGraph.
static thread_local bool g_prepare_only = false;

static int prepare_kernel(const void *kern, size_t lds, bool needs_lds_base_0)
{
    if (needs_lds_base_0) {
        hipFuncAttributes fa;
        MH_HIP(hipFuncGetAttributes(&fa, kern));
        if (fa.sharedSizeBytes != 0)
            return fail(MH_ERR_HIP, "decode kernel has %zu bytes of static LDS: its table would not sit at LDS offset 0",
                        (size_t)fa.sharedSizeBytes);
    }
    if (lds > 64 * 1024)
        MH_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return MH_OK;
}

#ifndef MH_DEC_K4_LDS_FLOOR
#define MH_DEC_K4_LDS_FLOOR (41 * 1024)
#endif
constexpr size_t kDecK4LdsFloor = MH_DEC_K4_LDS_FLOOR;  // 3 workgroups per CU (see launch_decode2, launch_encode2)

template <int LC, int PB, int ABL = 0, int PK = 0>
static int launch_encode2(const mh::Enc2Args &a, hipStream_t st)
{
    size_t lds = 4 * (size_t)mh::enc2_wave_dwords(a.e.stage_dw) * sizeof(uint32_t);  // + the static tables
    // The short-code byte-input encoder (S <= 3), like the S <= 3 decoder, is bound by the memory system and not by its
    // arithmetic, and runs FASTER with 3 workgroups per CU than with the 4 its registers allow: 1024 ch x 1e7 bins
    // 2.02-2.16 -> 1.975 ms with placement-probed payload buffers, and no longer sensitive to where the input sits
    // (profiles/r03_occupancy_ab.txt).  The longer-code encoders are compute-bound before their stores and lose
    // (S = 8: +4 %), S = 4..6 are indifferent (-0.8 %): only LC = 0 is capped, through the LDS request.
    // Only where the launch has many rounds of workgroups: with 2640 tasks (96 ch x 3.6e6 bins) a quarter fewer slots cost
    // a whole extra round (77.7 -> 80.2 us).
    if (LC == 0 && PK == 0 && ABL == 0 && a.t.ntask >= 16384 && lds < kDecK4LdsFloor) lds = kDecK4LdsFloor;
#ifdef MH_TUNING  // occupancy cap through the LDS request (A/B runs)
    if (const char *e = getenv("MH_ENC_LDS_MIN")) {  // replaces the floor above
        const size_t need = 4 * (size_t)mh::enc2_wave_dwords(a.e.stage_dw) * sizeof(uint32_t);
        lds = (size_t)atoi(e) > need ? (size_t)atoi(e) : need;
    }
#endif
    auto kern = mh::k_encode2<LC, PB, ABL, PK>;
    if (g_prepare_only) return prepare_kernel(reinterpret_cast<const void *>(kern), lds, false);
    hipLaunchKernelGGL(kern, dim3(a.t.ntask), dim3(256), lds, st, a);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

template <int LC, int PB, int PK = 0>
static int launch_encode2w(const mh::Enc2Args &a, hipStream_t st)
{
    const size_t lds = 4 * (size_t)mh::enc2w_wave_dwords<PB, PK>(a.e.stage_dw) * sizeof(uint32_t);
    auto kern = mh::k_encode2w<LC, PB, PK>;
    if (g_prepare_only) return prepare_kernel(reinterpret_cast<const void *>(kern), lds, false);
    hipLaunchKernelGGL(kern, dim3((a.t.ntask + 3) / 4), dim3(256), lds, st, a);
    MH_HIP(hipGetLastError());
    return MH_OK;
}



template <int K, int M, int NR, int RL, bool HY, bool DUAL = false>
static int launch_decode2(const mh::Dec2Args &a, bool wave_tasks, hipStream_t st)
{
    if (wave_tasks) {
        const size_t lds = 4 * ((size_t)mh::dec2_shared_dwords(a.W, K) + mh::dec2_stage_dwords(NR)) * sizeof(uint32_t);
        auto kern = mh::k_decode2w<K, M, NR, RL, HY, DUAL>;
        if (g_prepare_only) return prepare_kernel(reinterpret_cast<const void *>(kern), lds, false);
        hipLaunchKernelGGL(kern, dim3((a.t.ntask + 3) / 4), dim3(256), lds, st, a);
    } else {
        const size_t lds_need = ((size_t)mh::dec2_shared_dwords(a.W, K) + 4 * (size_t)mh::dec2_stage_dwords(NR)) * sizeof(uint32_t);
        size_t lds = lds_need;
        // The four-symbol decoder (S <= 3) is bound by its 1-KiB row stores, not by its arithmetic, and the part
        // writes FASTER with fewer waves streaming at once: 3 workgroups per CU instead of the 4 its registers
        // allow, enforced through the LDS request (160 KiB / 41 KiB = 3): 1024 ch x 1e7 bins decode 2.25 -> 2.04 ms
        // on one box; 2 per CU: 2.40 ms (profiles/r03_occupancy_ab.txt).  The pair-table decoders (S >= 4) are
        // bound by their dependent lookup chain and lose with fewer waves (S = 5: 2.27 -> 2.38 ms).
        if (K == 4 && lds < kDecK4LdsFloor) lds = kDecK4LdsFloor;
#ifdef MH_TUNING
        if (const char *e = getenv("MH_DEC_LDS_MIN")) lds = (size_t)atoi(e) > lds_need ? (size_t)atoi(e) : lds_need;
#endif
        auto kern = mh::k_decode2<K, M, NR, RL, HY, DUAL>;
        if (g_prepare_only) return prepare_kernel(reinterpret_cast<const void *>(kern), lds, K == 2);
        hipLaunchKernelGGL(kern, dim3(a.t.ntask), dim3(256), lds, st, a);
    }
    MH_HIP(hipGetLastError());
    return MH_OK;
}

// lane-private LDS staging of the encoder: 16 dwords per lane for codes of at most 2 bits (the worst case of a
// 256-sample sub-stream), else 32 (= 4 bits per sample on average: the worst case up to 4-bit codes; a clipped
// spike-count channel at S <= 10 stays well below that, and chunks that outgrow it take the two-pass global slow
// path).  Always 4 * stage_ne(LC) rows: the staging rows are permuted so that the merge gathers consecutive dwords
// (MH_STAGE_AT), which needs the row count the kernel class was compiled for.
static inline uint32_t enc_stage_dw(uint32_t maxlen) { return maxlen <= 2 ? 16 : 32; }

// packed input (time-major path): the same kernels reading 4-bit resp. 2-bit pieces
template <int PK>
static int dispatch_encode_packed(const mh_plan *p, const mh::Enc2Args &a2, hipStream_t st)
{
    const uint32_t L = p->h.info.maxlen;
    if (PK == 2) {  // S <= 4 only: maxlen <= 3; the table is the four-symbol one (PB unused)
        if (p->h.use_wave_tasks) return L <= 2 ? launch_encode2w<0, 4, 2>(a2, st) : launch_encode2w<1, 4, 2>(a2, st);
        return L <= 2 ? launch_encode2<0, 4, 0, 2>(a2, st) : launch_encode2<1, 4, 0, 2>(a2, st);
    }
    // 4-bit pieces: a byte of the stream is the PB = 4 pair index
    if (p->h.use_wave_tasks) {
        if (L <= 2) return launch_encode2w<0, 4, 4>(a2, st);
        if (L <= 4) return launch_encode2w<1, 4, 4>(a2, st);
        if (L <= 8) return launch_encode2w<2, 4, 4>(a2, st);
        return launch_encode2w<3, 4, 4>(a2, st);
    }
    if (L <= 2) return launch_encode2<0, 4, 0, 4>(a2, st);
    if (L <= 4) return launch_encode2<1, 4, 0, 4>(a2, st);
    if (L <= 8) return launch_encode2<2, 4, 0, 4>(a2, st);
    return launch_encode2<3, 4, 0, 4>(a2, st);
}

static int dispatch_encode(const mh_plan *p, const mh::Enc2Args &a2, hipStream_t st)
{
    const uint32_t L = p->h.info.maxlen;
    const bool pb3 = p->h.info.S <= 8;  // 3-bit pair packing when every symbol fits 3 bits
    if (p->h.input_bits == 4) return dispatch_encode_packed<4>(p, a2, st);
    if (p->h.input_bits == 2) return dispatch_encode_packed<2>(p, a2, st);
    if (p->h.use_wave_tasks) {
        if (L <= 2) return pb3 ? launch_encode2w<0, 3>(a2, st) : launch_encode2w<0, 4>(a2, st);
        if (L <= 4) return pb3 ? launch_encode2w<1, 3>(a2, st) : launch_encode2w<1, 4>(a2, st);
        if (L <= 8) return pb3 ? launch_encode2w<2, 3>(a2, st) : launch_encode2w<2, 4>(a2, st);
        return launch_encode2w<3, 4>(a2, st);
    }
#ifdef MH_TUNING
    if (L <= 2 && pb3 && g_ablate) {
        switch (g_ablate) {
        case 1: return launch_encode2<0, 3, 1>(a2, st);
        case 2: return launch_encode2<0, 3, 2>(a2, st);
        case 3: return launch_encode2<0, 3, 3>(a2, st);
        case 4: return launch_encode2<0, 3, 4>(a2, st);
        case 5: return launch_encode2<0, 3, 5>(a2, st);
        case 6: return launch_encode2<0, 3, 6>(a2, st);
        case 7: return launch_encode2<0, 3, 7>(a2, st);
        case 11: return launch_encode2<0, 3, 11>(a2, st);
        case 12: return launch_encode2<0, 3, 12>(a2, st);
        case 13: return launch_encode2<0, 3, 13>(a2, st);
        case 14: return launch_encode2<0, 3, 14>(a2, st);
        case 15: return launch_encode2<0, 3, 15>(a2, st);
        default: return launch_encode2<0, 3, 8>(a2, st);
        }
    }
    if (L > 2 && L <= 4 && pb3 && g_ablate) {  // the same for the S = 4..6 kernel (levels 1, 2, 4, 8)
        switch (g_ablate) {
        case 1: return launch_encode2<1, 3, 1>(a2, st);
        case 2: return launch_encode2<1, 3, 2>(a2, st);
        case 4: return launch_encode2<1, 3, 4>(a2, st);
        default: return launch_encode2<1, 3, 8>(a2, st);
        }
    }
    if (L > 4 && L <= 8 && g_ablate) {  // S = 7, 8 (3-bit pairs) and S = 9 (4-bit pairs): levels 1, 2, 4
        if (pb3) {
            switch (g_ablate) {
            case 1: return launch_encode2<2, 3, 1>(a2, st);
            case 2: return launch_encode2<2, 3, 2>(a2, st);
            default: return launch_encode2<2, 3, 4>(a2, st);
            }
        }
        switch (g_ablate) {
        case 1: return launch_encode2<2, 4, 1>(a2, st);
        case 2: return launch_encode2<2, 4, 2>(a2, st);
        default: return launch_encode2<2, 4, 4>(a2, st);
        }
    }
    if (L > 8 && g_ablate) {  // and for the S = 10 kernel (levels 1, 2, 4)
        switch (g_ablate) {
        case 1: return launch_encode2<3, 4, 1>(a2, st);
        case 2: return launch_encode2<3, 4, 2>(a2, st);
        default: return launch_encode2<3, 4, 4>(a2, st);
        }
    }
#endif
    if (L <= 2) return pb3 ? launch_encode2<0, 3>(a2, st) : launch_encode2<0, 4>(a2, st);
    if (L <= 4) return pb3 ? launch_encode2<1, 3>(a2, st) : launch_encode2<1, 4>(a2, st);
    if (L <= 8) return pb3 ? launch_encode2<2, 3>(a2, st) : launch_encode2<2, 4>(a2, st);
    return launch_encode2<3, 4>(a2, st);
}

static int dispatch_decode(const mh_plan *p, const mh::Dec2Args &a2, hipStream_t st)
{
#ifdef MH_TUNING
    {
        static int last = -1;
        const char *e = getenv("MH_DEC_ABL");
        const int want = e ? atoi(e) : 0;
        if (want != last) {
            (void)hipMemcpyToSymbol(HIP_SYMBOL(mh::d_dec_abl), &want, sizeof(int));
            last = want;
        }
    }
#endif
    const uint32_t L = p->h.info.maxlen;
    const bool wt = p->h.use_wave_tasks;
    // window maintenance (decode_staged_chunk): 1 = reload, 0 = branchy top-up, 2 = select top-up;
    // the choices are the measured best per variant (profiles/README.md)
    if (L <= 2) return launch_decode2<4, 4, 17, 1, false>(a2, wt, st);  // worst-case chunk = 1027 words: never oversize
    if (L == 3) return launch_decode2<2, 2, 25, 2, false>(a2, wt, st);
    if (a2.W >= 2 * L) return launch_decode2<2, 2, 32, 0, false>(a2, wt, st);
#ifndef MH_DEC_K1
#define MH_DEC_K1 1  // A/B builds: 0 = never, 1 = wave-task plans, 2 = every plan whose pair table would be hybrid
#endif
#ifndef MH_DEC_K1_NR
#define MH_DEC_K1_NR 36
#endif
    // Long codes on SHORT channels (wave tasks, where every wave builds its own tables): the one-symbol decoder -- a
    // 2^maxlen-byte table instead of a 256-entry hybrid pair table whose flagged entries make almost every lookup of
    // the wave take the slow path (8 index bits) -- with the two chunks of a segment side by side (decode_staged_pair1).
    // 10 000 x 20 000: decode S=8 82 -> 74 us, S=10 92 -> 77 us; 2400 x 72 000: 70 -> 68 us.  On long channels (shared
    // 1024-entry tables) it loses: S=8 2.35 -> 2.82 ms -- twice the LDS lookups, and those decoders are bound by LDS
    // bank-conflict throughput, not by the latency of the chain (profiles/r03_k1_pair_decoding_ab.txt).
    if ((MH_DEC_K1 >= 2 || (MH_DEC_K1 == 1 && wt)) && a2.W < 2 * L) return launch_decode2<1, 2, MH_DEC_K1_NR, 2, false, true>(a2, wt, st);
    // hybrid pair table: W < 2 * maxlen index bits, one-symbol entries flagged
    if (p->h.dec_NR == 31) return launch_decode2<2, 2, 31, 2, true>(a2, wt, st);
    return launch_decode2<2, 2, 32, 0, true>(a2, wt, st);
}

// raise the dynamic-LDS limits of the kernels this plan will launch (once, at plan creation)
static int prepare_kernels(const mh_plan *p)
{
    mh::Enc2Args e{};
    e.e.stage_dw = enc_stage_dw(p->h.info.maxlen);
    mh::Dec2Args d{};
    d.W = p->h.W;
    g_prepare_only = true;
    int rc = dispatch_encode(p, e, nullptr);
    if (rc == MH_OK) rc = dispatch_decode(p, d, nullptr);
    g_prepare_only = false;
    return rc;
}

#pragma GCC visibility push(default)
extern "C" {

int mh_version(void) { return MH_VERSION; }

#ifdef MH_TUNING
/* tuning builds only: selects a timing-only ablation of k_encode2 */
void mhdbg_set_ablation(int level) { g_ablate = level; }
#endif

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
    if (mh::check_sclv_row(sclv_row, S, &m) != MH_OK)
        return fail(MH_ERR_SCLV, "SCLV row is not a non-decreasing complete prefix-code length vector");
    mh::canonical_codes(sclv_row, S, code, len);
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
    void *ptrs[] = {p->d_ch_off, p->d_ch_len, p->d_w0, p->d_w1, p->d_skip, p->d_sclv, p->d_sclv16, p->d_tile_cnt,
                    p->d_tile_done, p->d_codes,
                    p->d_seg_ch, p->d_seg_first, p->d_seg_n, p->d_seg_off, p->d_tile_ch,
                    p->d_tile_n, p->d_tile_start, p->d_hist, p->d_peak, p->d_enc,
                    p->d_lut, p->d_wg_tasks, p->d_wave_tasks, p->d_scan,
                    p->d_cal_tile_ch, p->d_cal_tile_n, p->d_cal_tile_start, p->d_calhist, p->d_err, p->d_acc};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
    return MH_OK;
}

// device side of plan creation: upload the planner's tables, allocate the per-channel scratch
static int plan_upload(mh_plan *p)
{
    const mh::PlanHost &H = p->h;
    const uint32_t C = H.info.C;
    const bool cal = !H.cal_tile_ch.empty();
    int rc;
    std::vector<uint32_t> rows16((size_t)H.info.K * 4, 0u);
    for (uint32_t k = 0; k < H.info.K; ++k)
        for (uint32_t r = 0; r < H.info.S; ++r)
            rows16[(size_t)k * 4 + (r >> 2)] |= (uint32_t)H.sclv[(size_t)k * H.info.S + r] << (8 * (r & 3));
    if ((rc = upload(&p->d_sclv16, rows16)) || (rc = upload(&p->d_tile_cnt, H.tile_cnt)) || (rc = alloc(&p->d_tile_done, C))) return rc;
    if ((rc = upload(&p->d_ch_off, H.ch_off)) || (rc = upload(&p->d_ch_len, H.ch_len)) ||
        (rc = upload(&p->d_w0, H.w0)) || (rc = upload(&p->d_w1, H.w1)) || (rc = upload(&p->d_skip, H.skip)) ||
        (rc = upload(&p->d_sclv, H.sclv)) || (rc = upload(&p->d_codes, H.codes)) ||
        (rc = upload(&p->d_seg_ch, H.seg_ch)) || (rc = upload(&p->d_seg_first, H.seg_first)) ||
        (rc = upload(&p->d_seg_n, H.seg_n)) || (rc = upload(&p->d_seg_off, H.seg_off)) ||
        (rc = upload(&p->d_tile_ch, H.tile_ch)) || (rc = upload(&p->d_tile_n, H.tile_n)) ||
        (rc = upload(&p->d_tile_start, H.tile_start)) ||
        (rc = alloc(&p->d_hist, (size_t)C * mh::kHistStride)) || (rc = alloc(&p->d_peak, C)) ||
        (rc = alloc(&p->d_enc, C)) || (rc = alloc(&p->d_lut, (size_t)C * mh::kLut)) ||
        (rc = upload(&p->d_wg_tasks, H.wg_tasks)) ||
        (H.use_wave_tasks && (rc = upload(&p->d_wave_tasks, H.wave_tasks))) ||
        (rc = alloc(&p->d_scan, H.seg_ch.size() / mh::kScanBlock + 2)) || (rc = upload(&p->d_err, std::vector<uint32_t>(1, 0u))) ||
        (H.use_wave_tasks && (rc = upload(&p->d_acc, std::vector<unsigned long long>(C, 0ull)))) ||
        (cal && ((rc = upload(&p->d_cal_tile_ch, H.cal_tile_ch)) || (rc = upload(&p->d_cal_tile_n, H.cal_tile_n)) ||
                 (rc = upload(&p->d_cal_tile_start, H.cal_tile_start)) ||
                 (rc = alloc(&p->d_calhist, (size_t)C * mh::kHistStride)))))
        return rc;
    return MH_OK;
}

static int plan_args(const uint64_t *ch_len, uint32_t C, uint32_t S, uint32_t h, uint32_t mode, uint32_t window,
                     const uint8_t *sclv, uint32_t K, uint32_t seg_chunks, mh_plan_info_t *I)
{
    const char *msg = "";
    uint32_t arg = 0, maxlen = 0;
    const int rc = mh::plan_check_args(ch_len, C, S, h, mode, window, sclv, K, &maxlen, &msg, &arg);
    if (rc != MH_OK) return fail(rc, msg, arg);
    if (seg_chunks > mh::kMaxSegChunks)  // a wave task counts its samples in 32 bits
        return fail(MH_ERR_ARG, "seg_chunks=%u above %u", seg_chunks, mh::kMaxSegChunks);
    *I = mh_plan_info_t{};
    I->C = C;
    I->S = S;
    I->h = h;
    I->mode = mode;
    I->window = window;
    I->K = K;
    I->seg_chunks = seg_chunks;  // 0 = the planner chooses
    I->maxlen = maxlen;
    return MH_OK;
}

int mh_plan_query(const uint64_t *ch_len, uint32_t C, uint32_t S, uint32_t h, uint32_t mode, uint32_t window,
                  const uint8_t *sclv, uint32_t K, uint32_t seg_chunks, mh_plan_info_t *info, uint32_t *seg_ch,
                  uint64_t *seg_first, uint64_t *seg_n, uint64_t *seg_off, uint64_t seg_cap)
{
    if (!ch_len || !sclv || !info) return fail(MH_ERR_ARG, "mh_plan_query: NULL argument");
    mh::PlanHost H;
    if (int rc = plan_args(ch_len, C, S, h, mode, window, sclv, K, seg_chunks, &H.info)) return rc;
    std::vector<uint64_t> off(C, 0);  // offsets do not enter the directory
    mh::plan_host_build(H, off.data(), ch_len, sclv, plan_tuning());
    *info = H.info;
    const size_t n = H.seg_ch.size() < seg_cap ? H.seg_ch.size() : (size_t)seg_cap;
    if (seg_ch && n) memcpy(seg_ch, H.seg_ch.data(), n * sizeof(uint32_t));
    if (seg_first && n) memcpy(seg_first, H.seg_first.data(), n * sizeof(uint64_t));
    if (seg_n && n) memcpy(seg_n, H.seg_n.data(), n * sizeof(uint64_t));
    if (seg_off && n) memcpy(seg_off, H.seg_off.data(), n * sizeof(uint64_t));
    return MH_OK;
}

int mh_plan_create(mh_plan **plan, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                   uint32_t S, uint32_t h, uint32_t mode, uint32_t window, const uint8_t *sclv,
                   uint32_t K, uint32_t seg_chunks)
{
    return mh_plan_create_packed(plan, ch_off, ch_len, C, S, h, mode, window, sclv, K, seg_chunks, 8, 0);
}

int mh_plan_create_packed(mh_plan **plan, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                          uint32_t S, uint32_t h, uint32_t mode, uint32_t window, const uint8_t *sclv,
                          uint32_t K, uint32_t seg_chunks, uint32_t input_bits, uint64_t chunk_stride)
{
    if (!plan || !ch_off || !ch_len || !sclv) return fail(MH_ERR_ARG, "mh_plan_create: NULL argument");
    *plan = nullptr;
    if (input_bits != 8 && input_bits != 4 && input_bits != 2)
        return fail(MH_ERR_ARG, "input_bits=%u (8, 4 or 2)", input_bits);
    if (input_bits != 8 && (window & ~MH_WIN_REV2_SEGMENTS) != MH_WIN_FULL)
        return fail(MH_ERR_ARG, "packed input needs the whole-channel window (MH_WIN_FULL)");
    if (input_bits == 2 && S > 4) return fail(MH_ERR_ARG, "2-bit input holds symbols 0..3: S=%u is above 4", S);
    if (chunk_stride && (input_bits == 8 || chunk_stride % 16 || chunk_stride < (uint64_t)MH_CHUNK * input_bits / 8))
        return fail(MH_ERR_ARG, "chunk_stride=%llu: packed input only, a multiple of 16, at least one chunk",
                    (unsigned long long)chunk_stride);
    mh_plan_info_t I;
    if (int rc = plan_args(ch_len, C, S, h, mode, window, sclv, K, seg_chunks, &I)) return rc;
    int ndev = 0, dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MH_ERR_NO_DEVICE, "no HIP device visible (libmuahuff has no CPU fallback)");
    MH_HIP(hipGetDevice(&dev));
    mh_plan *p = new (std::nothrow) mh_plan;
    if (!p) return fail(MH_ERR_ARG, "out of host memory");
    p->device = dev;
    p->h.info = I;
    p->h.input_bits = input_bits;
    p->h.chunk_stride = chunk_stride;
    mh::plan_host_build(p->h, ch_off, ch_len, sclv, plan_tuning());
    if (p->h.seg_ch.size() > 0xFFFFFFF0ull) {  // segment and task indices are 32-bit on the device
        const size_t nseg = p->h.seg_ch.size();
        mh_plan_destroy(p);
        return fail(MH_ERR_ARG, "mh_plan_create: %zu segments exceed the 32-bit directory", nseg);
    }
    int rc = plan_upload(p);
    if (rc == MH_OK && (hipMemset(p->d_hist, 0, (size_t)I.C * mh::kHistStride * sizeof(unsigned long long)) != hipSuccess ||
                        hipMemset(p->d_tile_done, 0, (size_t)I.C * sizeof(uint32_t)) != hipSuccess))
        rc = fail(MH_ERR_HIP, "mh_plan_create: clearing the measure scratch failed");
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
    *info = plan->h.info;
    return MH_OK;
}

int mh_plan_segments(const mh_plan *plan, uint32_t *seg_ch, uint64_t *seg_first, uint64_t *seg_n,
                     uint64_t *seg_off)
{
    if (!plan) return fail(MH_ERR_ARG, "mh_plan_segments: NULL plan");
    const size_t n = plan->h.seg_ch.size();
    if (seg_ch && n) memcpy(seg_ch, plan->h.seg_ch.data(), n * sizeof(uint32_t));
    if (seg_first && n) memcpy(seg_first, plan->h.seg_first.data(), n * sizeof(uint64_t));
    if (seg_n && n) memcpy(seg_n, plan->h.seg_n.data(), n * sizeof(uint64_t));
    if (seg_off && n) memcpy(seg_off, plan->h.seg_off.data(), n * sizeof(uint64_t));
    return MH_OK;
}

int mh_measure(mh_plan *p, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
               uint8_t *peak, uint8_t *enc, uint64_t *post_hist, uint64_t *bits,
               uint8_t *skipped, void *stream)
{
    if (!p || !data) return fail(MH_ERR_ARG, "mh_measure: NULL argument");
    if (p->h.input_bits != 8) return fail(MH_ERR_ARG, "mh_measure: this plan reads packed pieces (mh_encode_preset only)");
    if (int rc_ = check_device(p->device, "mh_measure")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    uint8_t *pk = peak ? peak : p->d_peak, *en = enc ? enc : p->d_enc;
    mh::FinArgs f;
    f.hist = p->d_hist;
    f.w0 = p->d_w0;
    f.w1 = p->d_w1;
    f.skipflag = p->d_skip;
    f.peak = pk;
    f.enc = en;
    f.sclv = p->d_sclv;
    f.sclv16 = p->d_sclv16;
    f.C = p->h.info.C;
    f.S = p->h.info.S;
    f.mode = p->h.info.mode;
    f.post = post_hist;
    f.bits = bits;
    f.skipped = skipped;
    // Launch-bound shapes (mh_planner.hpp: kFusedMeasureChannels): ONE launch.
    // The workgroup that adds a channel's last tile to the histogram calibrates and prices the channel
    // (measure_tail); the histogram scratch and the tickets are left zero for the next call.
    const bool fused = p->h.measure_fused;
    if (!fused) {
        int rc = launch_calibrate(p, data, cutoff, cal_hist, pk, en, st, p->d_hist, nullptr, nullptr);
        if (rc) return rc;
    }
    if (!p->h.tile_ch.empty()) {
        mh::HistArgs a{};
        a.data = data;
        a.ch_off = p->d_ch_off;
        a.tile_ch = p->d_tile_ch;
        a.tile_start = p->d_tile_start;
        a.tile_n = p->d_tile_n;
        a.hist = p->d_hist;
        a.tile_slot = nullptr;
        if (fused) {
            a.tile_cnt = p->d_tile_cnt;
            a.tile_done = p->d_tile_done;
            a.cal = calibrate_args(p, data, cutoff, cal_hist, pk, en, nullptr, nullptr, nullptr);
            a.fin = f;
        }
        const uint64_t n_tiles = p->h.tile_ch.size();
        const unsigned nt = (unsigned)n_tiles;
        if (p->h.info.S == 2)
            launch_hist<1>(a, n_tiles, st);  // byte-compare kernel: already at the read floor
        else if (p->h.info.S == 3)
            launch_hist<2>(a, n_tiles, st);
        else if (p->h.info.S <= 8)  // pair-LUT histogram, 3-bit pair packing
            hipLaunchKernelGGL(mh::k_hist2<3>, dim3(nt), dim3(256), 0, st, a, p->h.info.S);
        else                      // S = 9, 10: 4-bit packing, xor-swizzled
            hipLaunchKernelGGL(mh::k_hist2<4>, dim3(nt), dim3(256), 0, st, a, p->h.info.S);
        MH_HIP(hipGetLastError());
    }
    if (fused) return MH_OK;
    hipLaunchKernelGGL(mh::k_finalize, dim3((f.C + 255) / 256), dim3(256), 0, st, f);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

static mh::TaskArgs task_args(const mh_plan *p)
{
    mh::TaskArgs t{};
    if (p->h.use_wave_tasks) {  // one wave per segment, longest first
        t.wt = p->d_wave_tasks;
        t.ntask = (uint32_t)p->h.wave_tasks.size();
    } else {                    // one workgroup per <= 4 consecutive segments of a channel
        t.wg = p->d_wg_tasks;
        t.ntask = (uint32_t)p->h.wg_tasks.size();
        t.seg_samples = p->h.info.seg_chunks * MH_CHUNK;
        t.seg_src_stride = p->h.seg_src_stride;
        t.slot_full = p->h.slot_full;
    }
    return t;
}

// cal_mode: see EncArgs.  Modes 1 and 2 exist for wave-task plans only.
static int encode_common(mh_plan *p, const uint8_t *data, uint32_t *payload, uint64_t *seg_words,
                         uint64_t *ch_bits, uint32_t cal_mode, const uint8_t *peak_in, const uint8_t *enc_in,
                         uint8_t *peak_out, uint8_t *enc_out, uint8_t *skip_out, hipStream_t st)
{
    if (cal_mode == 0 && p->h.info.n_segments == 0) return MH_OK;
    mh::EncArgs a;
    a.cal_mode = cal_mode;
    a.chunk_stride = p->h.chunk_stride;
    a.S = p->h.info.S;
    a.mode = p->h.info.mode;
    a.K = p->h.info.K;
    a.sclv = p->d_sclv;
    a.sclv16 = p->d_sclv16;
    a.codes = p->d_codes;
    a.peak_in = peak_in;
    a.enc_in = enc_in;
    a.peak_out = peak_out;
    a.enc_out = enc_out;
    a.skip_out = skip_out;
    a.acc = p->d_acc;
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
    a.nseg = (uint32_t)p->h.info.n_segments;
    a.stage_dw = enc_stage_dw(p->h.info.maxlen);
    mh::Enc2Args a2;
    a2.e = a;
    a2.t = task_args(p);
    return dispatch_encode(p, a2, st);
}

int mh_encode(mh_plan *p, const uint8_t *data, uint32_t *payload, uint64_t payload_cap_words,
              uint64_t *seg_words, uint64_t *ch_bits, uint8_t *peak, uint8_t *enc,
              uint8_t *skipped, void *stream)
{
    if (!p || !data || !payload || !seg_words || !ch_bits)
        return fail(MH_ERR_ARG, "mh_encode: NULL argument");
    if (p->h.input_bits != 8) return fail(MH_ERR_ARG, "mh_encode: this plan reads packed pieces (mh_encode_preset only)");
    if (int rc_ = check_device(p->device, "mh_encode")) return rc_;
    if (payload_cap_words < p->h.info.payload_cap_words)
        return fail(MH_ERR_CAPACITY, "payload buffer holds %llu words, plan needs %llu",
                    (unsigned long long)payload_cap_words,
                    (unsigned long long)p->h.info.payload_cap_words);
    hipStream_t st = (hipStream_t)stream;
    if (p->h.fused_calibration && p->h.tickets_fit)  // short channels: every wave calibrates its own channel, one launch in all
        return encode_common(p, data, payload, seg_words, ch_bits, 1u, nullptr, nullptr, peak, enc, skipped, st);
    uint8_t *pk = peak ? peak : p->d_peak, *en = enc ? enc : p->d_enc;
    int rc = launch_calibrate(p, data, nullptr, nullptr, pk, en, st, nullptr,
                              reinterpret_cast<unsigned long long *>(ch_bits), skipped);
    if (rc) return rc;
    return encode_common(p, data, payload, seg_words, ch_bits, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, st);
}

int mh_encode_preset(mh_plan *p, const uint8_t *data, const uint8_t *peak, const uint8_t *enc,
                     uint32_t *payload, uint64_t payload_cap_words, uint64_t *seg_words,
                     uint64_t *ch_bits, void *stream)
{
    if (!p || !data || !peak || !enc || !payload || !seg_words || !ch_bits)
        return fail(MH_ERR_ARG, "mh_encode_preset: NULL argument");
    if (int rc_ = check_device(p->device, "mh_encode_preset")) return rc_;
    if (payload_cap_words < p->h.info.payload_cap_words)
        return fail(MH_ERR_CAPACITY, "payload buffer holds %llu words, plan needs %llu",
                    (unsigned long long)payload_cap_words,
                    (unsigned long long)p->h.info.payload_cap_words);
    hipStream_t st = (hipStream_t)stream;
    if (p->h.use_wave_tasks && p->h.tickets_fit)  // the waves build their tables from the preset word themselves: one launch
        return encode_common(p, data, payload, seg_words, ch_bits, 2u, peak, enc, nullptr, nullptr, nullptr, st);
    hipLaunchKernelGGL(mh::k_lut_preset, dim3((p->h.info.C + 15) / 16), dim3(256), 0, st, peak, enc,
                       (const uint32_t *)p->d_codes, p->h.info.C, p->h.info.S, p->h.info.mode, p->h.info.K, p->d_lut,
                       reinterpret_cast<unsigned long long *>(ch_bits), (uint8_t *)nullptr, (uint8_t *)nullptr);
    MH_HIP(hipGetLastError());
    return encode_common(p, data, payload, seg_words, ch_bits, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, st);
}

int mh_decode(mh_plan *p, const uint32_t *payload, uint64_t payload_words, const uint64_t *seg_off,
              const uint8_t *peak, const uint8_t *enc, uint8_t *out, void *stream)
{
    if (!p || !payload || !peak || !enc || !out) return fail(MH_ERR_ARG, "mh_decode: NULL argument");
    if (p->h.input_bits != 8)  // a packed plan's offsets describe the packed buffer, not a byte layout to decode into
        return fail(MH_ERR_ARG, "mh_decode: this plan reads packed pieces (mh_encode_preset only); decode with a byte-layout plan");
    if (int rc_ = check_device(p->device, "mh_decode")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    if (p->h.info.n_segments == 0) return MH_OK;
    mh::DecArgs a;
    a.payload = payload;
    a.ch_off = p->d_ch_off;
    a.w0 = p->d_w0;
    a.seg_ch = p->d_seg_ch;
    a.seg_first = p->d_seg_first;
    a.seg_n = p->d_seg_n;
    a.seg_off = seg_off ? seg_off : p->d_seg_off;
    a.out = out;
    a.nseg = (uint32_t)p->h.info.n_segments;
    a.payload_words = payload_words;
    a.err = p->d_err;
    a.epoch = 1u;  // the status word is a sticky flag (mh_decode_status reads and clears it)
    mh::Dec2Args a2;
    a2.d = a;
    a2.t = task_args(p);
    a2.W = p->h.W;
    a2.peak = peak;
    a2.enc = enc;
    a2.codes = p->d_codes;
    a2.S = p->h.info.S;
    a2.mode = p->h.info.mode;
    a2.nK = p->h.info.K;
    a2.plan_slots = seg_off ? 0u : 1u;
    // both kernel families build their tables themselves from (peak, enc): one launch
    return dispatch_decode(p, a2, st);
}

int mh_decode_status(mh_plan *p, uint32_t *flags, void *stream)
{
    if (!p || !flags) return fail(MH_ERR_ARG, "mh_decode_status: NULL argument");
    if (int rc_ = check_device(p->device, "mh_decode_status")) return rc_;
    uint32_t seen = 0;
    MH_HIP(hipMemcpyAsync(&seen, p->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    MH_HIP(hipStreamSynchronize((hipStream_t)stream));
    // sticky: any decode since the last status call that abandoned a segment has raised the word -- direct
    // calls and replays of a captured decode alike (a captured kernel carries no per-call state)
    *flags = seen ? 1u : 0u;
    if (seen) {
        MH_HIP(hipMemsetAsync(p->d_err, 0, sizeof(uint32_t), (hipStream_t)stream));
        MH_HIP(hipStreamSynchronize((hipStream_t)stream));
    }
    return MH_OK;
}

int mh_validate_stream(const uint64_t *ch_len, uint32_t C, uint32_t S, uint32_t h, uint32_t mode, uint32_t window,
                       const uint8_t *sclv, uint32_t K, uint32_t seg_chunks, const uint32_t *payload,
                       uint64_t payload_words, const uint64_t *seg_words, uint64_t n_segments,
                       const uint8_t *peak, const uint8_t *enc)
{
    if (!ch_len || !sclv || !payload || !seg_words || !peak || !enc)
        return fail(MH_ERR_ARG, "mh_validate_stream: NULL argument");
    if (seg_chunks == 0) return fail(MH_ERR_ARG, "mh_validate_stream: a stored stream names its seg_chunks");
    mh::PlanHost H;
    if (int rc = plan_args(ch_len, C, S, h, mode, window, sclv, K, seg_chunks, &H.info)) return rc;
    std::vector<uint64_t> off(C, 0);
    mh::plan_host_build(H, off.data(), ch_len, sclv);
    if (H.seg_ch.size() != n_segments)
        return fail(MH_ERR_STREAM, "directory has %llu segments, the layout implies %zu",
                    (unsigned long long)n_segments, H.seg_ch.size());
    for (uint32_t c = 0; c < C; ++c)
        if (peak[c] >= S || enc[c] >= K)
            return fail(MH_ERR_STREAM, "channel %u: (peak %u, encoder %u) outside (S=%u, K=%u)", c, peak[c], enc[c], S, K);
    uint64_t pos = 0;
    for (size_t s = 0; s < H.seg_ch.size(); ++s) {
        const uint64_t end = pos + seg_words[s];
        if (end < pos || end > payload_words)
            return fail(MH_ERR_STREAM, "segment %zu ends at word %llu, the payload has %llu", s,
                        (unsigned long long)end, (unsigned long long)payload_words);
        const uint8_t *row = sclv + (size_t)enc[H.seg_ch[s]] * S;
        const uint64_t maxlen = row[S - 1];
        uint64_t left = H.seg_n[s];
        while (left) {
            const uint64_t m = left < MH_CHUNK ? left : MH_CHUNK;
            if (pos >= end) return fail(MH_ERR_STREAM, "segment %zu is shorter than its chunk headers say", s);
            const uint32_t w0 = payload[pos];
            const uint32_t mn = w0 & 0xFFFu, w = (w0 >> 12) & 15u;
            if (w > 12) return fail(MH_ERR_STREAM, "segment %zu: chunk header field width %u above 12", s, w);
            const uint64_t hw = (16u + 64u * w + 31u) >> 5;
            if (pos + hw > end) return fail(MH_ERR_STREAM, "segment %zu: chunk header runs past the segment", s);
            uint64_t sum = 0, longest = 0;
            for (uint32_t l = 0; l < 64; ++l) {
                uint64_t f = 0;
                if (w) {
                    const uint32_t fb = 16u + l * w;
                    uint64_t v = payload[pos + (fb >> 5)];
                    if ((fb & 31) + w > 32) v |= (uint64_t)payload[pos + (fb >> 5) + 1] << 32;
                    f = (v >> (fb & 31)) & ((1u << w) - 1u);
                }
                sum += mn + f;
                if (mn + f > longest) longest = mn + f;
            }
            // every codeword has 1..maxlen bits; a sub-stream holds <= 256 samples
            if (longest > 256 * maxlen || sum > m * maxlen || sum < m)
                return fail(MH_ERR_STREAM, "segment %zu: sub-stream lengths impossible for this code", s);
            pos += hw + ((sum + 31) >> 5);
            left -= m;
        }
        if (pos != end) return fail(MH_ERR_STREAM, "segment %zu: chunk sizes do not add up to its %llu words", s,
                                    (unsigned long long)seg_words[s]);
    }
    if (pos != payload_words)
        return fail(MH_ERR_STREAM, "payload has %llu words, the directory accounts for %llu",
                    (unsigned long long)payload_words, (unsigned long long)pos);
    return MH_OK;
}

int mh_compact(mh_plan *p, const uint32_t *payload, const uint64_t *seg_words, uint32_t *dense,
               uint64_t dense_cap_words, uint64_t *dense_off, uint64_t *total_words, void *stream)
{
    if (!p || !payload || !seg_words || !dense || !dense_off || !total_words)
        return fail(MH_ERR_ARG, "mh_compact: NULL argument");
    if (int rc_ = check_device(p->device, "mh_compact")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t nseg = p->h.info.n_segments;
    const uint64_t nblocks = (nseg + mh::kScanBlock - 1) / mh::kScanBlock;
    if (nblocks <= 1) {  // one launch: every wave scans for itself
        const uint64_t nwg = nseg ? (nseg + mh::kCompactSegs - 1) / mh::kCompactSegs : 1;
        hipLaunchKernelGGL(mh::k_compact<true>, dim3((unsigned)nwg), dim3(256), 0, st, payload,
                           (const uint64_t *)p->d_seg_off, seg_words, dense_off, dense, dense_cap_words, nseg, total_words);
        MH_HIP(hipGetLastError());
        return MH_OK;
    }
    hipLaunchKernelGGL(mh::k_scan_block_sums, dim3((unsigned)nblocks), dim3(256), 0, st, seg_words, nseg, p->d_scan);
    hipLaunchKernelGGL(mh::k_scan_top, dim3(1), dim3(1024), 0, st, p->d_scan, nblocks, total_words);
    hipLaunchKernelGGL(mh::k_scan_apply, dim3((unsigned)nblocks), dim3(256), 0, st, seg_words, nseg,
                       (const uint64_t *)p->d_scan, dense_off);
    MH_HIP(hipGetLastError());
    const uint64_t nwg = (nseg + mh::kCompactSegs - 1) / mh::kCompactSegs;
    hipLaunchKernelGGL(mh::k_compact<false>, dim3((unsigned)nwg), dim3(256), 0, st, payload,
                       (const uint64_t *)p->d_seg_off, seg_words, dense_off, dense, dense_cap_words, nseg, total_words);
    MH_HIP(hipGetLastError());
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

static uint32_t layout_ablation()
{
#ifdef MH_TUNING
    if (const char *e = getenv("MH_LAYOUT_ABL")) return (uint32_t)atoi(e);
#endif
    return 0;
}

int mh_deinterleave(const uint8_t *in, uint64_t T, uint32_t C, uint8_t *out, const uint64_t *out_off,
                    void *stream)
{
    if (!in || !out || !out_off || C == 0) return fail(MH_ERR_ARG, "mh_deinterleave: bad argument");
    if (T == 0) return MH_OK;
    const uint32_t tpw = 4;
    uint64_t bx = ((T + mh::kTr2T - 1) / mh::kTr2T + tpw - 1) / tpw;
    const uint32_t by = (C + mh::kTr2C - 1) / mh::kTr2C;
    if (bx > 0x7FFFFFFFull / by) bx = 0x7FFFFFFFull / by;  // one grid dimension: strip fastest
    if (by > 65535) return fail(MH_ERR_ARG, "mh_deinterleave: C=%u too large", C);
    hipLaunchKernelGGL(mh::k_deinterleave2<0>, dim3((unsigned)(bx * by)), dim3(256), 0, (hipStream_t)stream, in, T, C, tpw,
                       out, out_off, layout_ablation());
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_deinterleave_packed(const uint8_t *in, uint64_t T, uint32_t C, uint32_t bits, uint8_t *out,
                           const uint64_t *out_off, uint64_t chunk_stride, void *stream)
{
    if (!in || !out || !out_off || C == 0) return fail(MH_ERR_ARG, "mh_deinterleave_packed: bad argument");
    if (bits != 4 && bits != 2) return fail(MH_ERR_ARG, "mh_deinterleave_packed: bits=%u (4 or 2)", bits);
    if (chunk_stride && (chunk_stride % 16 || chunk_stride < (uint64_t)MH_CHUNK * bits / 8))
        return fail(MH_ERR_ARG, "mh_deinterleave_packed: chunk_stride=%llu", (unsigned long long)chunk_stride);
    if (T == 0) return MH_OK;
    const uint32_t tpw_max = bits == 2 ? (uint32_t)mh::P2<2>::kTpw : (uint32_t)mh::P2<4>::kTpw;  // tiles of a visit held in LDS
    uint32_t tpw = tpw_max;
#ifdef MH_TUNING
    if (const char *e = getenv("MH_LAYOUT_TPW")) tpw = (uint32_t)atoi(e);
    if (tpw < 1 || tpw > tpw_max) tpw = tpw_max;
#endif
    uint64_t bx = ((T + mh::kTr2T - 1) / mh::kTr2T + tpw - 1) / tpw;
    const uint32_t by = (C + mh::kTr2C - 1) / mh::kTr2C;
    if (bx > 0x7FFFFFFFull / by) bx = 0x7FFFFFFFull / by;  // one grid dimension: strip fastest
    if (by > 65535) return fail(MH_ERR_ARG, "mh_deinterleave_packed: C=%u too large", C);
    // pieces that fit the Infinity Cache (256 MiB) stay cacheable for the encoder that reads them next
    uint32_t cached = (double)T * C * bits / 8.0 <= 192.0 * 1048576.0 ? 1u : 0u;
#ifdef MH_TUNING
    if (const char *e = getenv("MH_LAYOUT_CACHED")) cached = (uint32_t)atoi(e);
#endif
    if (bits == 4)
        hipLaunchKernelGGL(mh::k_deinterleave_p<4>, dim3((unsigned)(bx * by)), dim3(256), 0, (hipStream_t)stream, in, T, C, tpw,
                           out, out_off, layout_ablation(), chunk_stride, cached);
    else
        hipLaunchKernelGGL(mh::k_deinterleave_p<2>, dim3((unsigned)(bx * by)), dim3(256), 0, (hipStream_t)stream, in, T, C, tpw,
                           out, out_off, layout_ablation(), chunk_stride, cached);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_interleave(const uint8_t *in, const uint64_t *in_off, uint64_t T, uint32_t C, uint8_t *out, void *stream)
{
    if (!in || !in_off || !out || C == 0) return fail(MH_ERR_ARG, "mh_interleave: bad argument");
    if (T == 0) return MH_OK;
    const uint32_t tpw = 4;
    uint64_t bx = ((T + mh::kTr2T - 1) / mh::kTr2T + tpw - 1) / tpw;
    const uint32_t by = (C + mh::kTr2C - 1) / mh::kTr2C;
    if (bx > 0x7FFFFFFFull / by) bx = 0x7FFFFFFFull / by;  // one grid dimension: strip fastest
    if (by > 65535) return fail(MH_ERR_ARG, "mh_interleave: C=%u too large", C);
    hipLaunchKernelGGL(mh::k_interleave, dim3((unsigned)(bx * by)), dim3(256), 0, (hipStream_t)stream, in, in_off, T, C,
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
    int dev = 0;
    MH_HIP(hipGetDevice(&dev));
    mh_sweep *w = new (std::nothrow) mh_sweep;
    if (!w) return fail(MH_ERR_ARG, "out of host memory");
    w->device = dev;
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
            for (uint64_t first = lo; first < hi; first += mh::kHistTileBytes) {
                tile_ch.push_back(c);
                tile_slot.push_back((uint32_t)slot);
                tile_start.push_back(first);
                tile_n.push_back((uint32_t)(hi - first < mh::kHistTileBytes ? hi - first : mh::kHistTileBytes));
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
    if (int rc_ = check_device(w->device, "mh_sweep_run")) return rc_;
    hipStream_t st = (hipStream_t)stream;
    MH_HIP(hipMemsetAsync(w->d_scratch, 0, (size_t)w->n_slots * mh::kHistStride * sizeof(unsigned long long), st));
    if (w->n_tiles) {
        mh::HistArgs a{};
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

int mh_power_draws(const double *br, uint32_t n_br, const int32_t *idx, uint32_t Z, uint64_t n_draws,
                   double comm_energy, double per_channels, double static_power, double *x, uint64_t x_stride,
                   void *stream)
{
    if (!br || !idx || !x || n_br == 0 || x_stride == 0) return fail(MH_ERR_ARG, "mh_power_draws: bad argument");
    if (n_draws == 0) return MH_OK;
    if (n_draws > 0x7FFFFFFFull * 256) return fail(MH_ERR_ARG, "mh_power_draws: too many draws for one launch");
    hipLaunchKernelGGL(mh::k_power_draws, dim3((unsigned)((n_draws + 255) / 256)), dim3(256), 0, (hipStream_t)stream, br,
                       idx, Z, n_draws, comm_energy, per_channels, static_power, x, x_stride);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

int mh_reduce_rows(const double *vals, const uint64_t *row_off, uint64_t n_rows, double *sum, double *mx, void *stream)
{
    if (!vals || !row_off || !sum || !mx) return fail(MH_ERR_ARG, "mh_reduce_rows: NULL argument");
    if (n_rows == 0) return MH_OK;
    if (n_rows > 0x7FFFFFFFull * 256) return fail(MH_ERR_ARG, "mh_reduce_rows: too many rows for one launch");
    hipLaunchKernelGGL(mh::k_reduce_rows, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vals,
                       row_off, n_rows, sum, mx);
    MH_HIP(hipGetLastError());
    return MH_OK;
}

}  // extern "C"
#pragma GCC visibility pop
