// mh_planner.hpp -- the host-side planner of libmuahuff: windows, segment directory, slots,
// workgroup tasks, histogram tiles, codebooks, decoder geometry.  Pure C++ (no HIP, no device):
// muahuff.hip uploads what it computes; tests/planner_check.cpp compiles the same header with
// -fsanitize=address,undefined and checks it against the CPU oracle's directory.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "muahuff.h"

namespace mh {

constexpr uint32_t kHistTileBytes = 256 * 16 * 32;  // bytes of one histogram tile (128 KiB)
constexpr uint64_t kCalDirect = 4096;                // longest calibration window k_calibrate scans itself
// mh_measure as ONE launch (the workgroup of a channel's last tile finishes the channel) pays two atomic round
// trips at the end of every workgroup and a serial tail per channel; it wins while the call is bound by its
// launches: 2400 x 72 000 37 against 41 us, 96 x 72 000 11 against 16 -- but 10 000 x 20 000 56 against 47 us and
// 1024 x 1e7 0.95 against 0.80 ms.  Hence the two limits.
constexpr uint32_t kFusedMeasureChannels = 4096;
constexpr uint64_t kFusedMeasureTiles = 16384;
// seg_chunks = 0: two chunks per segment, one when that would leave fewer than this many segments (not enough
// waves to fill the part).  encode + decode in us, one / two chunks per segment (tools/small_shape_probe.py):
// 300 x 72 000 (900 two-chunk segments) 29 / 35, 600 x 72 000 (1800) 37 / 42, 1344 x 72 000 (4032) 60 / 58,
// 2400 x 72 000 (7200) 91 / 91, 10 000 x 20 000 (10 000) 130 / 120, 96 x 3.6e6 (10 560) 209 / 169.
constexpr uint64_t kAutoSegLimit = 3072;

inline uint32_t bitrev(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// A row is a sorted codeword-length vector of a complete prefix code (Kraft sum exactly 1,
// Compressing data/Produce SCLVs/produce_all_SCLVs_given_S.py:87-98), lengths 1..9.
inline int check_sclv_row(const uint8_t *row, int S, uint32_t *maxlen)
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

// canonical codewords, MSB-first values, rank 0 shortest ([1,2,2] -> 0,10,11 == test_chosen_system.py:26)
inline void canonical_codes(const uint8_t *row, int S, uint16_t *code, uint8_t *len)
{
    uint32_t c = 0;
    for (int r = 0; r < S; ++r) {
        if (r) c = (c + 1) << (row[r] - row[r - 1]);
        code[r] = (uint16_t)c;
        len[r] = row[r];
    }
}

// words a segment of n samples may need at most (what slots are sized by): every chunk a full
// 32-word header bound plus maxlen bits per sample, rounded up to a 128-byte line
inline uint64_t slot_words(uint64_t n, uint32_t maxlen)
{
    const uint64_t full = n / MH_CHUNK, rem = n % MH_CHUNK;
    uint64_t words = (full + (rem ? 1 : 0)) * MH_HDR_WORDS + full * (((uint64_t)MH_CHUNK * maxlen + 31) / 32);
    if (rem) words += (rem * maxlen + 31) / 32;
    return (words + 31) & ~(uint64_t)31;
}

// One wave task of the per-wave-table kernels: everything the wave needs to start, in one 48-byte
// record (one scalar load instead of the task -> segment -> channel chain of dependent loads).
// A channel without segments (window empty or skipped) still gets one record with n = 0, so that
// the encoder -- which calibrates inside the wave -- reports its (peak, encoder) word too.
struct WaveTask {
    uint64_t src_off;   // bytes from the data pointer to the segment's first sample
    uint64_t dst_off;   // the segment's slot, words from the payload pointer
    uint64_t cal_off;   // bytes from the data pointer to the channel's first bin (calibration window)
    uint32_t n;         // samples (0: calibrate-only record)
    uint32_t ch;        // channel
    uint32_t seg;       // directory entry (unused when n == 0)
    uint32_t cal_n;     // calibration window min(2^h, T), when it is short enough to scan in the wave
    uint32_t nseg_ch;   // records of this channel (the last wave to finish publishes the channel's bit total)
    uint32_t flags;     // bit 0: first record of the channel (publishes peak / encoder / skipped); bit 1: skipped
};

// One workgroup task of the shared-table kernels in ONE 32-byte record (one scalar load): up to 4 consecutive
// segments of one channel.  All but the last segment of a channel are full (seg_chunks chunks), so a wave finds
// its segment by arithmetic: source = src_off + wave * seg_src_stride, slot = dst_off + wave * slot_full,
// samples = wave + 1 < nseg ? seg_samples : n_last.  It replaces the task -> segment -> channel -> offsets chain
// of dependent loads from directories that every workgroup reads exactly once (each link an HBM miss under a
// streaming load: tools/occ_probe.hip measures ~0.02-0.03 ms of the 1024 x 1e7 encode per link).
struct WgTask {
    uint64_t src_off;  // bytes from the data pointer to the first segment's first sample
    uint64_t dst_off;  // the first segment's slot, words from the payload pointer
    uint32_t n_last;   // samples of the task's last segment (the others hold seg_chunks whole chunks)
    uint32_t ch;       // channel
    uint32_t seg0;     // directory entry of the first segment
    uint32_t nseg;     // segments (1..4)
};

struct PlanHost {
    mh_plan_info_t info{};
    uint32_t input_bits = 8;  // 8: one byte per sample; 4 / 2: packed pieces (mh_deinterleave_packed), whole-channel windows only
    uint64_t chunk_stride = 0;  // packed input only: bytes between consecutive chunks of a channel (0 = contiguous)
    uint64_t max_T = 0;
    std::vector<uint64_t> ch_off, ch_len, w0, w1;
    std::vector<uint8_t> skip, sclv;
    std::vector<uint32_t> codes;  // K*16: bit-reversed code | len << 16, by rank
    // segment directory
    std::vector<uint32_t> seg_ch;
    std::vector<uint64_t> seg_first, seg_n, seg_off;
    // workgroup tasks of the shared-table kernels: <= 4 consecutive segments of one channel
    std::vector<uint32_t> task_seg0;
    std::vector<uint8_t> task_n;
    std::vector<WgTask> wg_tasks;   // the same tasks as self-contained records (what the kernels read)
    uint64_t seg_src_stride = 0;    // bytes between the sources of consecutive full segments of a channel
    uint64_t slot_full = 0;         // slot words of a full segment
    // wave tasks of the per-wave-table kernels: every segment once, longest first
    std::vector<WaveTask> wave_tasks;
    bool use_wave_tasks = false;
    bool measure_fused = false;   // mh_measure in one launch (see kFusedMeasureChannels)
    bool fused_calibration = false;  // wave tasks calibrate in the wave (2^h <= kCalDirect): encode is ONE launch
    bool tickets_fit = false;        // every channel: records < 2^24 and 9 bits/sample * T < 2^40 (the packed total word)
    // window-histogram tiles, calibration tiles (windows above kCalDirect samples)
    std::vector<uint32_t> tile_ch, tile_n, tile_cnt, cal_tile_ch, cal_tile_n;  // tile_cnt: tiles per channel
    std::vector<uint64_t> tile_start, cal_tile_start;
    // decoder geometry
    uint32_t W = 0, dec_K = 4, dec_NR = 32;
};

struct PlanTuning {  // MH_TUNING builds only; the defaults are the measured best (profiles/README.md)
    int dec_w_cap = 0;    // index bits of the hybrid pair table (0: 10, and 8 for wave-task plans)
    int dec_nr = 31;      // staging registers per lane of the hybrid decoder
    int wave_tasks = -1;  // -1 = planner's rule, 0 / 1 = force
};

// Head segment of a window [w0, w1) (container format revision 3, include/muahuff.h): samples up to the next
// multiple of MH_HEAD_ALIGN when the window is long enough for the alignment of its rows to matter.
inline uint64_t head_samples(uint64_t w0, uint64_t w1, uint32_t window)
{
    if ((window & MH_WIN_REV2_SEGMENTS) || w1 - w0 < MH_HEAD_MIN_WINDOW || w0 % MH_HEAD_ALIGN == 0) return 0;
    return MH_HEAD_ALIGN - w0 % MH_HEAD_ALIGN;
}

// windows of one channel: c = min(2^h, T) (functions_1.py:59-64), e = c + T/2 (get_BR_with_approx_sort.py:180)
inline void channel_window(uint64_t T, uint32_t h, uint32_t window, uint64_t *w0, uint64_t *w1, uint8_t *skip)
{
    const uint64_t lim = (uint64_t)1 << h;
    const uint64_t cut = T < lim ? T : lim, e = cut + T / 2;
    *skip = 0;
    switch (window) {
    case MH_WIN_REF_HALF:
        if (e > T) {  // :183-185 skipped, shows up as NaN in the reference
            *skip = 1;
            *w0 = *w1 = cut;
        } else {
            *w0 = cut;
            *w1 = e;
        }
        break;
    case MH_WIN_REF_HALF_TRUNC: *w0 = cut; *w1 = e > T ? T : e; break;
    case MH_WIN_AFTER_CAL: *w0 = cut; *w1 = T; break;
    default: *w0 = 0; *w1 = T; break;
    }
}

// Argument checks shared by mh_plan_create and mh_plan_query; returns MH_OK or the error code and
// a message (static strings with at most one %u).
constexpr uint32_t kMaxSegChunks = 0xFFFFFFFFu / MH_CHUNK;  // segment sample counts fit 32 bits (WaveTask.n)

inline int plan_check_args(const uint64_t *ch_len, uint32_t C, uint32_t S, uint32_t h, uint32_t mode, uint32_t window,
                           const uint8_t *sclv, uint32_t K, uint32_t *maxlen_out, const char **msg, uint32_t *msg_arg)
{
    *msg_arg = 0;
    if (C == 0) { *msg = "C == 0"; return MH_ERR_ARG; }
    if (S < 2 || S > 10) { *msg = "S=%u outside 2..10"; *msg_arg = S; return MH_ERR_ARG; }
    if (h > 30) { *msg = "h=%u outside 0..30"; *msg_arg = h; return MH_ERR_ARG; }
    if (mode > MH_MODE_APPROX) { *msg = "mode=%u unknown"; *msg_arg = mode; return MH_ERR_ARG; }
    if ((window & ~MH_WIN_REV2_SEGMENTS) > MH_WIN_FULL) { *msg = "window=%u unknown"; *msg_arg = window; return MH_ERR_ARG; }
    if (K == 0 || K > 255) { *msg = "K=%u outside 1..255"; *msg_arg = K; return MH_ERR_ARG; }
    uint32_t maxlen = 0;
    for (uint32_t k = 0; k < K; ++k) {
        uint32_t m;
        if (check_sclv_row(sclv + (size_t)k * S, (int)S, &m) != MH_OK) {
            *msg = "SCLV row %u is not a non-decreasing complete code-length vector";
            *msg_arg = k;
            return MH_ERR_SCLV;
        }
        if (m > maxlen) maxlen = m;
    }
    for (uint32_t c = 0; c < C; ++c)
        if (ch_len[c] == 0) {
            *msg = "channel %u has no bins (the reference raises IndexError)";
            *msg_arg = c;
            return MH_ERR_EMPTY_CHANNEL;
        }
    *maxlen_out = maxlen;
    return MH_OK;
}

// Everything mh_plan_create uploads, computed on the host.  `info` must hold C, S, h, mode, window,
// K, seg_chunks (0 = choose) and maxlen; arguments are already checked.
inline void plan_host_build(PlanHost &p, const uint64_t *ch_off, const uint64_t *ch_len, const uint8_t *sclv,
                            const PlanTuning &tune = PlanTuning())
{
    mh_plan_info_t &I = p.info;
    const uint32_t C = I.C, S = I.S, K = I.K;
    p.ch_off.assign(ch_off, ch_off + C);
    p.ch_len.assign(ch_len, ch_len + C);
    p.w0.resize(C);
    p.w1.resize(C);
    p.skip.assign(C, 0);
    uint64_t total = 0, nskip = 0;
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t T = ch_len[c];
        if (T > p.max_T) p.max_T = T;
        channel_window(T, I.h, I.window & ~MH_WIN_REV2_SEGMENTS, &p.w0[c], &p.w1[c], &p.skip[c]);
        nskip += p.skip[c];
        total += p.w1[c] - p.w0[c];
    }
    I.window_samples = total;
    I.n_skipped = nskip;
    // segment length: the caller's, or two chunks -- one when that leaves the chip short of work
    // (short recordings: finer tasks fill the last round of workgroups better)
    if (I.seg_chunks == 0) {
        uint64_t nseg2 = 0;
        for (uint32_t c = 0; c < C; ++c) nseg2 += (p.w1[c] - p.w0[c] + 2 * MH_CHUNK - 1) / (2 * MH_CHUNK);
        I.seg_chunks = nseg2 < kAutoSegLimit ? 1 : 2;
    }
    const uint64_t seg_samples = (uint64_t)I.seg_chunks * MH_CHUNK;
    uint64_t slot = 0, padded_waves = 0;
    // byte offset of window sample `first` (packed input: a multiple of 16) from the channel's first byte
    auto src_bytes = [&](uint64_t first) {
        return p.input_bits == 8 ? first
               : p.chunk_stride  ? (first / MH_CHUNK) * p.chunk_stride
                                 : (first >> 4) * (p.input_bits == 4 ? 8u : 4u);
    };
    p.seg_src_stride = src_bytes(seg_samples);
    p.slot_full = slot_words(seg_samples, I.maxlen);
    for (uint32_t c = 0; c < C; ++c) {
        const uint64_t n = p.w1[c] - p.w0[c];
        size_t seg_begin = p.seg_ch.size();
        auto add_segment = [&](uint64_t first, uint64_t m) {
            p.seg_ch.push_back(c);
            p.seg_first.push_back(first);
            p.seg_n.push_back(m);
            p.seg_off.push_back(slot);
            slot += slot_words(m, I.maxlen);
        };
        const uint64_t head = head_samples(p.w0[c], p.w1[c], I.window);
        if (head) add_segment(0, head);
        for (uint64_t first = head; first < n; first += seg_samples)
            add_segment(first, n - first < seg_samples ? n - first : seg_samples);
        if (head) {  // the head segment is a task of its own: the regular ones stay an arithmetic sequence
            WgTask t{};
            t.src_off = ch_off[c] + src_bytes(p.w0[c]);
            t.dst_off = p.seg_off[seg_begin];
            t.n_last = (uint32_t)head;
            t.ch = c;
            t.seg0 = (uint32_t)seg_begin;
            t.nseg = 1;
            p.task_seg0.push_back((uint32_t)seg_begin);
            p.task_n.push_back(1);
            p.wg_tasks.push_back(t);
            padded_waves += 1;  // (its three idle waves leave at once: not what the wave-task rule below is about)
            ++seg_begin;
        }
        for (size_t s0 = seg_begin; s0 < p.seg_ch.size(); s0 += 4) {
            const uint32_t cnt = (uint32_t)(p.seg_ch.size() - s0 < 4 ? p.seg_ch.size() - s0 : 4);
            p.task_seg0.push_back((uint32_t)s0);
            p.task_n.push_back((uint8_t)cnt);
            WgTask t{};
            t.src_off = ch_off[c] + src_bytes(p.w0[c] + p.seg_first[s0]);
            t.dst_off = p.seg_off[s0];
            t.n_last = (uint32_t)p.seg_n[s0 + cnt - 1];
            t.ch = c;
            t.seg0 = (uint32_t)s0;
            t.nseg = cnt;
            p.wg_tasks.push_back(t);
            padded_waves += 4;
        }
        for (uint64_t first = 0; first < n; first += kHistTileBytes) {
            p.tile_ch.push_back(c);
            p.tile_start.push_back(p.w0[c] + first);
            p.tile_n.push_back((uint32_t)(n - first < kHistTileBytes ? n - first : kHistTileBytes));
        }
        // every channel has at least one tile (an empty one for an empty window): the workgroup of a channel's
        // last tile finishes the channel when mh_measure runs as one launch
        if (n == 0) {
            p.tile_ch.push_back(c);
            p.tile_start.push_back(p.w0[c]);
            p.tile_n.push_back(0);
        }
        p.tile_cnt.push_back((uint32_t)(n ? (n + kHistTileBytes - 1) / kHistTileBytes : 1));
    }
    p.measure_fused = ((uint64_t)1 << I.h) <= kCalDirect && C <= kFusedMeasureChannels && p.tile_ch.size() <= kFusedMeasureTiles;
    I.n_segments = p.seg_ch.size();
    I.payload_cap_words = slot + 4;  // decode reads <= 3 words past the last chunk
    // per-wave-table kernels when the shared-table tasks would leave more than 1 wave in 16 idle
    // (channels of a few segments); their wave tasks run longest first
    p.use_wave_tasks = padded_waves * 15 > (uint64_t)I.n_segments * 16;
    if (tune.wave_tasks >= 0) p.use_wave_tasks = tune.wave_tasks != 0;
    const uint64_t lim = (uint64_t)1 << I.h;
    p.fused_calibration = p.use_wave_tasks && lim <= kCalDirect;
    if (p.use_wave_tasks) {
        std::vector<uint32_t> order(p.seg_ch.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return p.seg_n[a] > p.seg_n[b]; });
        std::vector<uint32_t> nseg_ch(C, 0);
        for (uint32_t c : p.seg_ch) ++nseg_ch[c];
        p.tickets_fit = true;
        for (uint32_t c = 0; c < C; ++c)
            if (nseg_ch[c] >= (1u << 24) || ch_len[c] >= ((uint64_t)1 << 36)) p.tickets_fit = false;
        p.wave_tasks.reserve(order.size() + C);
        auto record = [&](uint32_t c) {
            WaveTask t{};
            t.cal_off = ch_off[c];
            t.ch = c;
            t.cal_n = (uint32_t)(ch_len[c] < lim ? ch_len[c] : (lim <= kCalDirect ? lim : 0));
            t.nseg_ch = nseg_ch[c] ? nseg_ch[c] : 1;
            t.flags = p.skip[c] ? 2u : 0u;
            return t;
        };
        for (uint32_t s : order) {
            const uint32_t c = p.seg_ch[s];
            WaveTask t = record(c);
            const uint64_t first = p.w0[c] + p.seg_first[s];  // packed input: a multiple of 16 (whole-channel windows)
            t.src_off = ch_off[c] + src_bytes(first);
            t.dst_off = p.seg_off[s];
            t.n = (uint32_t)p.seg_n[s];
            t.seg = s;
            if (p.seg_first[s] == 0) t.flags |= 1u;
            p.wave_tasks.push_back(t);
        }
        for (uint32_t c = 0; c < C; ++c)  // calibrate-only records, after the real work
            if (!nseg_ch[c]) {
                WaveTask t = record(c);
                t.flags |= 1u;
                p.wave_tasks.push_back(t);
            }
    }
    // calibration: one wave per channel reads the window directly up to kCalDirect samples (the
    // reference's range is 2^2..2^10); longer windows go through the tiled histogram kernel
    if (lim > kCalDirect)
        for (uint32_t c = 0; c < C; ++c) {
            const uint64_t n = ch_len[c] < lim ? ch_len[c] : lim;
            for (uint64_t first = 0; first < n; first += kHistTileBytes) {
                p.cal_tile_ch.push_back(c);
                p.cal_tile_start.push_back(first);
                p.cal_tile_n.push_back((uint32_t)(n - first < kHistTileBytes ? n - first : kHistTileBytes));
            }
        }
    // decode table: K symbols per lookup, W index bits.  maxlen <= 5: W = K * maxlen (<= 10), every
    // entry holds K whole codewords; longer codes (and W capped below 2 * maxlen): hybrid pair table of 10 index bits and 31
    // staging registers, which keeps 4 workgroups per CU (tables + staging <= 40 KiB of LDS).
    p.dec_K = I.maxlen <= 2 ? 4 : 2;
    p.W = p.dec_K * I.maxlen;
    p.dec_NR = 32;
    if (p.dec_K == 2) {
        // (wave-task plans build the table once per WAVE: 256 entries instead of 1024 cost a few more flagged
        // entries but a quarter of the build and 3 KiB less LDS per wave -- 2400 x 72 000, S = 8: 66 -> 58 us)
        uint32_t cap = tune.dec_w_cap >= 8 && tune.dec_w_cap <= 12 ? (uint32_t)tune.dec_w_cap : p.use_wave_tasks ? 8u : 10u;
        if (cap < I.maxlen) cap = I.maxlen;  // a flagged entry still holds its first codeword
        if (p.W > cap) p.W = cap;
        if (p.W < 2 * I.maxlen) p.dec_NR = tune.dec_nr == 32 ? 32 : 31;
    }
    // codebooks by rank: bit-reversed code (first code bit at bit 0) | len << 16
    p.codes.assign((size_t)K * 16, 0);
    for (uint32_t k = 0; k < K; ++k) {
        uint16_t code[16];
        uint8_t ln[16];
        canonical_codes(sclv + (size_t)k * S, (int)S, code, ln);
        for (uint32_t r = 0; r < S; ++r) p.codes[k * 16 + r] = bitrev(code[r], ln[r]) | ((uint32_t)ln[r] << 16);
    }
    p.sclv.assign(sclv, sclv + (size_t)K * S);
}

}  // namespace mh
