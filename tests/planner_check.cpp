// planner_check.cpp -- runs the product's host planner (csrc/mh_planner.hpp, the code mh_plan_create
// executes before its uploads) as a plain host program, so that it can be built with
// -fsanitize=address,undefined.  Reads cases from stdin:
//     C S h mode window K seg_chunks  len[0..C)  sclv[0..K*S)
// checks the planner's internal invariants and prints, per case, the directory for the test to
// compare with the CPU oracle's:  "nseg cap seg_chunks wave_tasks" then four lines ch / first / n / off.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mh_planner.hpp"

#define CHECK(cond)                                                    \
    do {                                                               \
        if (!(cond)) {                                                 \
            fprintf(stderr, "invariant failed: %s (line %d)\n", #cond, __LINE__); \
            return 2;                                                  \
        }                                                              \
    } while (0)

int main()
{
    unsigned C, S, h, mode, window, K, sc;
    while (scanf("%u %u %u %u %u %u %u", &C, &S, &h, &mode, &window, &K, &sc) == 7) {
        std::vector<uint64_t> len(C), off(C);
        std::vector<uint8_t> sclv((size_t)K * S);
        uint64_t o = 0;
        for (unsigned c = 0; c < C; ++c) {
            unsigned long long v;
            if (scanf("%llu", &v) != 1) return 1;
            len[c] = v;
            off[c] = o;
            o += (v + 15) & ~15ull;
        }
        for (auto &b : sclv) {
            unsigned v;
            if (scanf("%u", &v) != 1) return 1;
            b = (uint8_t)v;
        }
        const char *msg = "";
        uint32_t arg = 0, maxlen = 0;
        const int rc = mh::plan_check_args(len.data(), C, S, h, mode, window, sclv.data(), K, &maxlen, &msg, &arg);
        if (rc != MH_OK) {
            printf("error %d\n", rc);
            continue;
        }
        mh::PlanHost p;
        p.info.C = C; p.info.S = S; p.info.h = h; p.info.mode = mode; p.info.window = window;
        p.info.K = K; p.info.seg_chunks = sc; p.info.maxlen = maxlen;
        mh::plan_host_build(p, off.data(), len.data(), sclv.data());
        const size_t n = p.seg_ch.size();
        CHECK(p.info.n_segments == n && p.seg_first.size() == n && p.seg_n.size() == n && p.seg_off.size() == n);
        // segments tile every window exactly, in order; slots do not overlap and fit the capacity
        uint64_t samples = 0;
        for (size_t s = 0; s < n; ++s) {
            const uint32_t c = p.seg_ch[s];
            CHECK(c < C && p.seg_n[s] > 0 && p.seg_n[s] <= (uint64_t)p.info.seg_chunks * MH_CHUNK);
            CHECK(p.seg_first[s] + p.seg_n[s] <= p.w1[c] - p.w0[c]);
            if (s + 1 < n) CHECK(p.seg_off[s] + mh::slot_words(p.seg_n[s], maxlen) == p.seg_off[s + 1]);
            CHECK(p.seg_off[s] % 32 == 0);
            samples += p.seg_n[s];
        }
        CHECK(samples == p.info.window_samples);
        if (n) CHECK(p.seg_off[n - 1] + mh::slot_words(p.seg_n[n - 1], maxlen) + 4 == p.info.payload_cap_words);
        // shared-table tasks: every segment once, <= 4 consecutive ones of one channel
        size_t covered = 0;
        for (size_t t = 0; t < p.task_seg0.size(); ++t) {
            CHECK(p.task_seg0[t] == covered && p.task_n[t] >= 1 && p.task_n[t] <= 4);
            for (unsigned k = 1; k < p.task_n[t]; ++k) CHECK(p.seg_ch[covered + k] == p.seg_ch[covered]);
            covered += p.task_n[t];
        }
        CHECK(covered == n);
        // the same tasks as self-contained records: what a wave derives by arithmetic equals the directory
        CHECK(p.wg_tasks.size() == p.task_seg0.size());
        const uint64_t seg_samples = (uint64_t)p.info.seg_chunks * MH_CHUNK;
        CHECK(p.seg_src_stride == seg_samples && p.slot_full == mh::slot_words(seg_samples, maxlen));
        for (size_t t = 0; t < p.wg_tasks.size(); ++t) {
            const mh::WgTask &w = p.wg_tasks[t];
            CHECK(w.seg0 == p.task_seg0[t] && w.nseg == p.task_n[t] && w.ch == p.seg_ch[w.seg0]);
            for (uint32_t k = 0; k < w.nseg; ++k) {
                const size_t sg = (size_t)w.seg0 + k;
                CHECK(w.src_off + k * p.seg_src_stride == off[w.ch] + p.w0[w.ch] + p.seg_first[sg]);
                CHECK(w.dst_off + k * p.slot_full == p.seg_off[sg]);
                CHECK((k + 1 < w.nseg ? seg_samples : (uint64_t)w.n_last) == p.seg_n[sg]);
            }
        }
        if (p.use_wave_tasks) {  // every segment once, longest first, then one record per channel without segments
            std::vector<uint8_t> seen(n, 0);
            std::vector<uint32_t> per_ch(C, 0), first_ch(C, 0);
            size_t nreal = 0;
            for (size_t i = 0; i < p.wave_tasks.size(); ++i) {
                const mh::WaveTask &t = p.wave_tasks[i];
                CHECK(t.ch < C && t.cal_off == off[t.ch]);
                CHECK(t.cal_n == (len[t.ch] < ((uint64_t)1 << h) ? len[t.ch] : (((uint64_t)1 << h) <= mh::kCalDirect ? ((uint64_t)1 << h) : 0)));
                CHECK(((t.flags >> 1) & 1u) == p.skip[t.ch]);
                ++per_ch[t.ch];
                first_ch[t.ch] += t.flags & 1u;
                if (t.n == 0) continue;
                CHECK(nreal == i);  // real work first
                ++nreal;
                CHECK(t.seg < n && !seen[t.seg]);
                seen[t.seg] = 1;
                CHECK(t.ch == p.seg_ch[t.seg] && t.n == p.seg_n[t.seg] && t.dst_off == p.seg_off[t.seg]);
                CHECK(t.src_off == off[t.ch] + p.w0[t.ch] + p.seg_first[t.seg]);  // byte input
                CHECK(t.src_off + t.n <= off[t.ch] + len[t.ch]);
                CHECK((t.flags & 1u) == (p.seg_first[t.seg] == 0 ? 1u : 0u));
                if (i) CHECK(p.wave_tasks[i - 1].n >= t.n);
            }
            CHECK(nreal == n);
            for (unsigned c = 0; c < C; ++c) CHECK(per_ch[c] >= 1 && first_ch[c] == 1);
            for (const mh::WaveTask &t : p.wave_tasks) CHECK(t.nseg_ch == per_ch[t.ch]);
            CHECK(p.fused_calibration == (((uint64_t)1 << h) <= mh::kCalDirect));
        }
        // histogram tiles cover the windows; calibration tiles cover min(2^h, T) when it is long
        uint64_t tiled = 0;
        std::vector<unsigned> tiles_of(C, 0);
        for (size_t t = 0; t < p.tile_ch.size(); ++t) {
            const unsigned c = p.tile_ch[t];
            CHECK(c < C && p.tile_n[t] <= mh::kHistTileBytes);
            CHECK(p.tile_n[t] > 0 || p.w0[c] == p.w1[c]);  // an empty tile only for an empty window
            CHECK(p.tile_start[t] + p.tile_n[t] <= p.w1[c]);
            tiled += p.tile_n[t];
            ++tiles_of[c];
        }
        CHECK(tiled == p.info.window_samples);
        CHECK(p.tile_cnt.size() == C);  // every channel has a tile: its last tile's workgroup finishes the channel
        for (unsigned c = 0; c < C; ++c) CHECK(tiles_of[c] >= 1 && tiles_of[c] == p.tile_cnt[c]);
        uint64_t cal = 0, cal_want = 0;
        for (size_t t = 0; t < p.cal_tile_ch.size(); ++t) cal += p.cal_tile_n[t];
        if (((uint64_t)1 << h) > mh::kCalDirect)
            for (unsigned c = 0; c < C; ++c) cal_want += len[c] < ((uint64_t)1 << h) ? len[c] : ((uint64_t)1 << h);
        CHECK(cal == cal_want);
        CHECK(p.W >= maxlen && p.W <= 12 && (p.dec_K == 2 || p.dec_K == 4));
        printf("%zu %llu %u %d %d\n", n, (unsigned long long)p.info.payload_cap_words, p.info.seg_chunks, (int)p.use_wave_tasks,
               (int)p.tickets_fit);
        for (size_t s = 0; s < n; ++s) printf("%u ", p.seg_ch[s]);
        printf("\n");
        for (size_t s = 0; s < n; ++s) printf("%llu ", (unsigned long long)p.seg_first[s]);
        printf("\n");
        for (size_t s = 0; s < n; ++s) printf("%llu ", (unsigned long long)p.seg_n[s]);
        printf("\n");
        for (size_t s = 0; s < n; ++s) printf("%llu ", (unsigned long long)p.seg_off[s]);
        printf("\n");
    }
    return 0;
}
