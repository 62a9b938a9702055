/*
 * mh_oracle.c -- CPU ORACLE (test infrastructure only; see mh_oracle.h for the contract and
 * for the reference file:line each function follows).  Plain C99, no dependencies.
 * Built by oracle/Makefile into oracle/libmh_oracle.so.
 */
#include "mh_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Reference restatements
 * ---------------------------------------------------------------------------------------- */

/* functions_1.py:27-68.  The reference keeps a dict keyed by str(value) seeded with {'0':0},
 * adds one count per sample, re-sums the dict after every sample and stops when the sum is
 * > cutoff-1 or the data is exhausted.  The sum of the dict is simply the number of samples
 * entered, so only that count is kept here; the in-place saturation (:45-46) is kept. */
uint64_t mho_online_cutoff_literal(uint8_t *data, uint64_t len, uint64_t cutoff, int max_rate)
{
    if (len == 0) return (uint64_t)-1; /* data_in[0] -> IndexError in the reference */
    uint64_t hist_count = 0, i = 0;
    int flag = 0;
    while (!flag) {
        if (data[i] >= max_rate) data[i] = (uint8_t)max_rate; /* :45-46 */
        hist_count += 1;                                      /* :50-53 + :56-58 */
        if (hist_count + 1 > cutoff) flag = 1;                /* :59  hist_count > cutoff-1 */
        if (i + 1 == len) flag = 1;                           /* :63 */
        i += 1;                                               /* :66 */
    }
    return i;
}

uint64_t mho_cutoff(uint64_t len, uint64_t cutoff) { return len < cutoff ? len : cutoff; }

/* get_BR_with_approx_sort.py:164  val_data[val_data > max_firing_rate] = max_firing_rate */
void mho_clip(uint8_t *x, uint64_t n, int S)
{
    for (uint64_t i = 0; i < n; ++i)
        if (x[i] > S - 1) x[i] = (uint8_t)(S - 1);
}

/* np.histogram(clipped, arange(-0.5, S+0.5)) == bincount of min(x, S-1)  (:139,171,189) */
void mho_hist(const uint8_t *x, uint64_t n, int S, uint64_t *hist)
{
    for (int s = 0; s < S; ++s) hist[s] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        int v = x[i];
        if (v > S - 1) v = S - 1;
        hist[v] += 1;
    }
}

/* np.argmax: first maximum (functions_1.py:77) */
int mho_argmax_first(const uint64_t *hist, int S)
{
    int p = 0;
    for (int s = 1; s < S; ++s)
        if (hist[s] > hist[p]) p = s;
    return p;
}

/* functions_1.py:75-90, statement by statement (np.arange / np.delete / np.flip / np.hstack /
 * np.argsort on small integer vectors). */
void mho_approx_sort_literal(const uint64_t *hist, int S, uint8_t *out_idx)
{
    int idx[32], left[32], right[32], cat[32];
    int n_idx = S, n_left = 0, n_right = 0;
    for (int i = 0; i < S; ++i) idx[i] = i;         /* :76 */
    int p = mho_argmax_first(hist, S);              /* :77 */
    if (2 * p > S) {                                /* :78  p_idx > len(hist)/2 */
        /* :79 right = arange(2, (S-1-p)*2+1, 2) */
        for (int v = 2; v < (S - 1 - p) * 2 + 1; v += 2) right[n_right++] = v;
        /* :80 idx = np.delete(idx, right) : remove POSITIONS listed in right */
        int keep[32], nk = 0;
        for (int i = 0; i < n_idx; ++i) {
            int del = 0;
            for (int j = 0; j < n_right; ++j) del |= (right[j] == i);
            if (!del) keep[nk++] = idx[i];
        }
        n_left = nk; /* :81 left = idx */
        for (int i = 0; i < nk; ++i) left[i] = keep[i];
    } else {
        /* :83 left = arange(1, (2p-1)+1, 2) */
        for (int v = 1; v < (2 * p - 1) + 1; v += 2) left[n_left++] = v;
        int keep[32], nk = 0;
        for (int i = 0; i < n_idx; ++i) { /* :84 */
            int del = 0;
            for (int j = 0; j < n_left; ++j) del |= (left[j] == i);
            if (!del) keep[nk++] = idx[i];
        }
        n_right = nk; /* :85 right = idx */
        for (int i = 0; i < nk; ++i) right[i] = keep[i];
    }
    int n = 0; /* :87 hstack((flip(left), right)) */
    for (int i = n_left - 1; i >= 0; --i) cat[n++] = left[i];
    for (int i = 0; i < n_right; ++i) cat[n++] = right[i];
    /* :88 argsort of a permutation of 0..S-1 : position of each value */
    for (int i = 0; i < n; ++i) out_idx[cat[i]] = (uint8_t)i;
}

/* closed form: p, p-1, p+1, p-2, p+2, ... skipping out-of-range */
void mho_approx_sort_rule(int S, int peak, uint8_t *idx)
{
    int n = 0;
    idx[n++] = (uint8_t)peak;
    for (int d = 1; n < S; ++d) {
        if (peak - d >= 0) idx[n++] = (uint8_t)(peak - d);
        if (peak + d < S && n < S) idx[n++] = (uint8_t)(peak + d);
    }
}

/* val_dot_prod = cal_hist^T . SCLVs^T ; np.argmin -> first minimum (:254,281).  All terms are
 * small integers, so integer arithmetic equals the reference's float64/object arithmetic. */
int mho_select_encoder(const uint32_t *cal_sorted, const uint8_t *sclv, int K, int S)
{
    int best = 0;
    uint64_t best_cost = 0;
    for (int k = 0; k < K; ++k) {
        uint64_t cost = 0;
        for (int r = 0; r < S; ++r) cost += (uint64_t)sclv[k * S + r] * cal_sorted[r];
        if (k == 0 || cost < best_cost) {
            best = k;
            best_cost = cost;
        }
    }
    return best;
}

int mho_calibrate(const uint8_t *x, uint64_t T, const mho_params *p, mho_chan *out)
{
    const int S = (int)p->S;
    if (T == 0) return -1; /* IndexError in the reference */
    memset(out, 0, sizeof(*out));
    const uint64_t c = mho_cutoff(T, (uint64_t)1 << p->h); /* functions_1.py:59-64 */
    uint64_t hist[16];
    mho_hist(x, c, S, hist); /* :171 */
    out->cutoff = c;
    if (p->mode == MHO_MODE_APPROX) {
        out->peak = (uint8_t)mho_argmax_first(hist, S);
        mho_approx_sort_literal(hist, S, out->idx); /* :175 */
    } else {
        out->peak = 0;
        for (int s = 0; s < S; ++s) out->idx[s] = (uint8_t)s; /* get_BR_no_sort.py:174 */
    }
    for (int k = 0; k < S; ++k) {
        out->cal_sorted[k] = (uint32_t)hist[out->idx[k]]; /* :176 */
        out->rank_of[out->idx[k]] = (uint8_t)k;
    }
    out->enc = (uint8_t)mho_select_encoder(out->cal_sorted, p->sclv, (int)p->K, S);
    const uint64_t e = c + T / 2; /* :180  int(len/2) */
    out->skipped = 0;
    switch (p->window & ~MHO_WIN_REV2_SEGMENTS) {
    case MHO_WIN_REF_HALF:
        if (e > T) { /* :183-185 */
            out->skipped = 1;
            out->w0 = out->w1 = c;
        } else {
            out->w0 = c;
            out->w1 = e;
        }
        break;
    case MHO_WIN_REF_HALF_TRUNC: /* test_chosen_system.py:99-103 : the slice truncates */
        out->w0 = c;
        out->w1 = e > T ? T : e;
        break;
    case MHO_WIN_AFTER_CAL:
        out->w0 = c;
        out->w1 = T;
        break;
    default:
        out->w0 = 0;
        out->w1 = T;
        break;
    }
    return 0;
}

void mho_measure_channel(const uint8_t *x, const mho_params *p, const mho_chan *ch,
                         uint64_t *post_mapped, uint64_t *bits)
{
    const int S = (int)p->S;
    uint64_t post[16];
    mho_hist(x + ch->w0, ch->w1 - ch->w0, S, post); /* :189 */
    uint64_t b = 0;
    for (int k = 0; k < S; ++k) {
        post_mapped[k] = post[ch->idx[k]];                         /* :193 */
        b += (uint64_t)p->sclv[ch->enc * S + k] * post_mapped[k]; /* :289 numerator */
    }
    *bits = b;
}

int mho_measure(const uint8_t *data, const uint64_t *ch_off, const uint64_t *ch_len,
                uint32_t C, const mho_params *p, uint64_t *cutoff, uint32_t *cal_sorted,
                uint8_t *peak, uint8_t *enc, uint64_t *post_mapped, uint64_t *bits,
                uint8_t *skipped, int nthreads)
{
    const int S = (int)p->S;
    int err = 0;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t c = 0; c < (int64_t)C; ++c) {
        mho_chan ch;
        if (mho_calibrate(data + ch_off[c], ch_len[c], p, &ch) != 0) {
            err = -1;
            continue;
        }
        cutoff[c] = ch.cutoff;
        peak[c] = ch.peak;
        enc[c] = ch.enc;
        skipped[c] = ch.skipped;
        for (int k = 0; k < S; ++k) cal_sorted[c * S + k] = ch.cal_sorted[k];
        mho_measure_channel(data + ch_off[c], p, &ch, post_mapped + (size_t)c * S, bits + c);
    }
    return err;
}

/* ------------------------------------------------------------------------------------------
 * Build-defined codec: canonical codebook, chunked container, decoder
 * ---------------------------------------------------------------------------------------- */

int mho_codebook(const uint8_t *len_in, int S, uint16_t *code, uint8_t *len)
{
    /* rows of Stored_SCLVs_S_<S>.pkl are non-decreasing with Kraft sum exactly 1 */
    uint32_t kraft = 0, maxlen = 0;
    for (int r = 0; r < S; ++r) {
        if (len_in[r] == 0 || len_in[r] > 15) return -1;
        if (r > 0 && len_in[r] < len_in[r - 1]) return -2;
        if (len_in[r] > maxlen) maxlen = len_in[r];
    }
    for (int r = 0; r < S; ++r) kraft += 1u << (maxlen - len_in[r]);
    if (kraft != (1u << maxlen)) return -3;
    uint32_t c = 0;
    for (int r = 0; r < S; ++r) {
        if (r > 0) c = (c + 1) << (len_in[r] - len_in[r - 1]);
        code[r] = (uint16_t)c; /* MSB-first codeword value */
        len[r] = len_in[r];
    }
    return 0;
}

static uint32_t bitrev(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

static uint32_t sclv_maxlen(const mho_params *p)
{
    uint32_t m = 0;
    for (uint32_t i = 0; i < p->K * p->S; ++i)
        if (p->sclv[i] > m) m = p->sclv[i];
    return m;
}

uint64_t mho_slot_words(uint64_t n, uint32_t maxlen)
{
    if (n == 0) return 0;
    const uint64_t full = n / MHO_CHUNK, rem = n % MHO_CHUNK;
    uint64_t w = (full + (rem ? 1 : 0)) * MHO_HDR_WORDS;
    w += full * (((uint64_t)MHO_CHUNK * maxlen + 31) / 32);
    if (rem) w += (rem * maxlen + 31) / 32;
    return (w + 31) & ~(uint64_t)31; /* slots start on 128-byte lines */
}

static void window_of(uint64_t T, const mho_params *p, uint64_t *w0, uint64_t *w1, int *skipped)
{
    const uint64_t c = mho_cutoff(T, (uint64_t)1 << p->h), e = c + T / 2;
    *skipped = 0;
    switch (p->window & ~MHO_WIN_REV2_SEGMENTS) {
    case MHO_WIN_REF_HALF:
        if (e > T) { *skipped = 1; *w0 = *w1 = c; } else { *w0 = c; *w1 = e; }
        break;
    case MHO_WIN_REF_HALF_TRUNC: *w0 = c; *w1 = e > T ? T : e; break;
    case MHO_WIN_AFTER_CAL: *w0 = c; *w1 = T; break;
    default: *w0 = 0; *w1 = T; break;
    }
}

uint64_t mho_plan_segments(const uint64_t *ch_len, uint32_t C, const mho_params *p,
                           uint32_t *seg_ch, uint64_t *seg_first, uint64_t *seg_n,
                           uint64_t *seg_off, uint64_t *cap_words)
{
    const uint32_t maxlen = sclv_maxlen(p);
    const uint64_t seg_samples = (uint64_t)p->seg_chunks * MHO_CHUNK;
    uint64_t nseg = 0, off = 0;
    for (uint32_t c = 0; c < C; ++c) {
        uint64_t w0, w1;
        int sk;
        window_of(ch_len[c], p, &w0, &w1, &sk);
        const uint64_t n = w1 - w0;
        /* container format revision 3: a long window that starts off a 128-sample boundary opens with a head
         * segment up to that boundary (include/muahuff.h, MH_WIN_REV2_SEGMENTS) */
        uint64_t head = 0;
        if (!(p->window & MHO_WIN_REV2_SEGMENTS) && n >= MHO_HEAD_MIN_WINDOW && w0 % MHO_HEAD_ALIGN)
            head = MHO_HEAD_ALIGN - w0 % MHO_HEAD_ALIGN;
        for (uint64_t first = 0; first < n; first += (first == 0 && head) ? head : seg_samples) {
            uint64_t m = n - first < seg_samples ? n - first : seg_samples;
            if (first == 0 && head) m = head;
            if (seg_ch) {
                seg_ch[nseg] = c;
                seg_first[nseg] = first;
                seg_n[nseg] = m;
                seg_off[nseg] = off;
            }
            off += mho_slot_words(m, maxlen);
            nseg++;
        }
    }
    if (cap_words) *cap_words = off;
    return nseg;
}

typedef struct {
    uint32_t code_rev[MHO_LUT]; /* bit-reversed codeword, first code bit at bit 0 */
    uint32_t len[MHO_LUT];
} enc_lut;

static void build_enc_lut(const mho_params *p, const uint8_t *rank_of, int enc, enc_lut *lut)
{
    const int S = (int)p->S;
    uint16_t code[16];
    uint8_t len[16];
    mho_codebook(p->sclv + enc * S, S, code, len);
    for (int v = 0; v < MHO_LUT; ++v) {
        const int sym = v > S - 1 ? S - 1 : v;
        const int r = rank_of[sym];
        lut->len[v] = len[r];
        lut->code_rev[v] = bitrev(code[r], len[r]);
    }
}

static inline void put_bits(uint32_t *w, uint64_t pos, uint32_t code_rev, uint32_t len)
{
    (void)len;
    const uint64_t v = (uint64_t)code_rev << (pos & 31);
    w[pos >> 5] |= (uint32_t)v;
    if (v >> 32) w[(pos >> 5) + 1] |= (uint32_t)(v >> 32);
}

/* Chunk header, format revision 2 (build-defined; the reference has no bitstream):
 *   bits [0,12)  min  = shortest sub-stream length of the chunk (<= 256 * 9 = 2304)
 *   bits [12,16) w    = bits needed for (longest - shortest), 0..12
 *   then 64 fields of w bits, field l = length of sub-stream l minus min, LSB-first from bit 16
 *   header words = ceil((16 + 64 * w) / 32) = 1..25; the payload starts at the next word. */
static uint32_t hdr_width(uint32_t range)
{
    uint32_t w = 0;
    while (range >> w) ++w;
    return w;
}

static uint32_t hdr_words(uint32_t w) { return (16u + 64u * w + 31u) >> 5; }

/* one chunk of m <= MHO_CHUNK samples -> words; returns words written */
static uint64_t encode_chunk(const uint8_t *x, uint32_t m, const enc_lut *lut, uint32_t *out,
                             uint64_t *bits)
{
    uint32_t lane_len[MHO_LANES];
    /* pass 1: sub-stream lengths.  Sample q belongs to piece q/16, lane (q/16)%64 */
    for (int l = 0; l < MHO_LANES; ++l) {
        uint32_t L = 0;
        for (int k = 0; k < MHO_ROWS; ++k) {
            const uint32_t base = ((uint32_t)k * MHO_LANES + l) * MHO_PIECE;
            for (int i = 0; i < MHO_PIECE; ++i) {
                const uint32_t q = base + i;
                if (q < m) {
                    const int v = x[q] > 15 ? 15 : x[q];
                    L += lut->len[v];
                }
            }
        }
        lane_len[l] = L;
    }
    uint64_t B = 0;
    for (int l = 0; l < MHO_LANES; ++l) B += lane_len[l];
    const uint64_t nw = (B + 31) / 32;
    uint32_t mn = lane_len[0], mx = lane_len[0];
    for (int l = 1; l < MHO_LANES; ++l) {
        if (lane_len[l] < mn) mn = lane_len[l];
        if (lane_len[l] > mx) mx = lane_len[l];
    }
    const uint32_t w = hdr_width(mx - mn), hw = hdr_words(w);
    memset(out, 0, (hw + nw) * sizeof(uint32_t));
    out[0] = mn | (w << 12);
    for (int l = 0; l < MHO_LANES; ++l)
        if (w) put_bits(out, 16 + (uint64_t)l * w, lane_len[l] - mn, w);
    /* pass 2: emit */
    uint32_t *pay = out + hw;
    uint64_t pos = 0;
    for (int l = 0; l < MHO_LANES; ++l) {
        for (int k = 0; k < MHO_ROWS; ++k) {
            const uint32_t base = ((uint32_t)k * MHO_LANES + l) * MHO_PIECE;
            for (int i = 0; i < MHO_PIECE; ++i) {
                const uint32_t q = base + i;
                if (q < m) {
                    const int v = x[q] > 15 ? 15 : x[q];
                    put_bits(pay, pos, lut->code_rev[v], lut->len[v]);
                    pos += lut->len[v];
                }
            }
        }
    }
    *bits += B;
    return hw + nw;
}

uint64_t mho_encode_segment(const uint8_t *x, uint64_t n, const mho_params *p,
                            const mho_chan *ch, uint32_t *out, uint64_t *bits)
{
    enc_lut lut;
    build_enc_lut(p, ch->rank_of, ch->enc, &lut);
    uint64_t w = 0;
    for (uint64_t q = 0; q < n; q += MHO_CHUNK) {
        const uint32_t m = (uint32_t)(n - q < MHO_CHUNK ? n - q : MHO_CHUNK);
        w += encode_chunk(x + q, m, &lut, out + w, bits);
    }
    return w;
}

typedef struct {
    uint8_t e[512]; /* sym | len<<4, indexed by the next maxlen stream bits */
    uint32_t maxlen;
} dec_tab;

static void build_dec_tab(const mho_params *p, uint8_t peak, uint8_t enc, dec_tab *t)
{
    const int S = (int)p->S;
    uint16_t code[16];
    uint8_t len[16], idx[16];
    mho_codebook(p->sclv + enc * S, S, code, len);
    if (p->mode == MHO_MODE_APPROX)
        mho_approx_sort_rule(S, peak, idx);
    else
        for (int s = 0; s < S; ++s) idx[s] = (uint8_t)s;
    uint32_t maxlen = 0;
    for (int r = 0; r < S; ++r)
        if (len[r] > maxlen) maxlen = len[r];
    t->maxlen = maxlen;
    for (int r = 0; r < S; ++r) {
        const uint32_t rev = bitrev(code[r], len[r]);
        for (uint32_t fill = 0; fill < (1u << (maxlen - len[r])); ++fill)
            t->e[rev | (fill << len[r])] = (uint8_t)(idx[r] | (len[r] << 4));
    }
}

static uint64_t decode_chunk(const uint32_t *in, uint32_t m, const dec_tab *t, uint8_t *out)
{
    uint64_t P[MHO_LANES + 1];
    P[0] = 0;
    const uint32_t mn = in[0] & 0xFFFu, w = (in[0] >> 12) & 15u, hw = hdr_words(w);
    for (int l = 0; l < MHO_LANES; ++l) {
        uint32_t f = 0;
        if (w) {
            const uint64_t fb = 16 + (uint64_t)l * w;
            uint64_t v = in[fb >> 5];
            if ((fb & 31) + w > 32) v |= (uint64_t)in[(fb >> 5) + 1] << 32;
            f = (uint32_t)(v >> (fb & 31)) & ((1u << w) - 1u);
        }
        P[l + 1] = P[l] + mn + f;
    }
    const uint64_t nw = (P[MHO_LANES] + 31) / 32;
    const uint32_t *pay = in + hw;
    const uint32_t mask = (1u << t->maxlen) - 1;
    for (int l = 0; l < MHO_LANES; ++l) {
        uint64_t pos = P[l];
        for (int k = 0; k < MHO_ROWS; ++k) {
            const uint32_t base = ((uint32_t)k * MHO_LANES + l) * MHO_PIECE;
            for (int i = 0; i < MHO_PIECE; ++i) {
                const uint32_t q = base + i;
                if (q >= m) continue;
                const uint64_t w = pos >> 5;
                uint64_t v = w < nw ? pay[w] : 0;
                if (w + 1 < nw) v |= (uint64_t)pay[w + 1] << 32;
                const uint8_t e = t->e[(uint32_t)(v >> (pos & 31)) & mask];
                out[q] = e & 15;
                pos += e >> 4;
            }
        }
    }
    return hw + nw;
}

uint64_t mho_decode_segment(const uint32_t *in, uint64_t n, const mho_params *p, uint8_t peak,
                            uint8_t enc, uint8_t *out)
{
    dec_tab t;
    build_dec_tab(p, peak, enc, &t);
    uint64_t w = 0;
    for (uint64_t q = 0; q < n; q += MHO_CHUNK) {
        const uint32_t m = (uint32_t)(n - q < MHO_CHUNK ? n - q : MHO_CHUNK);
        w += decode_chunk(in + w, m, &t, out + q);
    }
    return w;
}

int mho_encode(const uint8_t *data, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
               const mho_params *p, uint32_t *payload, uint64_t cap_words, uint64_t *seg_words,
               uint64_t *ch_bits, uint8_t *peak, uint8_t *enc, uint8_t *skipped, int nthreads)
{
    uint64_t cap = 0;
    const uint64_t nseg = mho_plan_segments(ch_len, C, p, NULL, NULL, NULL, NULL, &cap);
    if (cap > cap_words) return -2;
    uint32_t *seg_ch = (uint32_t *)malloc((nseg + 1) * sizeof(uint32_t));
    uint64_t *seg_first = (uint64_t *)malloc((nseg + 1) * 3 * sizeof(uint64_t));
    uint64_t *seg_n = seg_first + nseg + 1, *seg_off = seg_n + nseg + 1;
    mho_plan_segments(ch_len, C, p, seg_ch, seg_first, seg_n, seg_off, &cap);
    mho_chan *chs = (mho_chan *)malloc((C + 1) * sizeof(mho_chan));
    int err = 0;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t c = 0; c < (int64_t)C; ++c) {
        if (mho_calibrate(data + ch_off[c], ch_len[c], p, &chs[c]) != 0) err = -1;
        peak[c] = chs[c].peak;
        enc[c] = chs[c].enc;
        skipped[c] = chs[c].skipped;
        ch_bits[c] = 0;
    }
    if (!err) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
        for (int64_t s = 0; s < (int64_t)nseg; ++s) {
            const uint32_t c = seg_ch[s];
            uint64_t bits = 0;
            seg_words[s] = mho_encode_segment(data + ch_off[c] + chs[c].w0 + seg_first[s],
                                              seg_n[s], p, &chs[c], payload + seg_off[s], &bits);
#pragma omp atomic
            ch_bits[c] += bits;
        }
    }
    free(chs);
    free(seg_first);
    free(seg_ch);
    return err;
}

int mho_decode(const uint32_t *payload, const uint64_t *ch_off, const uint64_t *ch_len,
               uint32_t C, const mho_params *p, const uint8_t *peak, const uint8_t *enc,
               uint8_t *out, int nthreads)
{
    uint64_t cap = 0;
    const uint64_t nseg = mho_plan_segments(ch_len, C, p, NULL, NULL, NULL, NULL, &cap);
    uint32_t *seg_ch = (uint32_t *)malloc((nseg + 1) * sizeof(uint32_t));
    uint64_t *seg_first = (uint64_t *)malloc((nseg + 1) * 3 * sizeof(uint64_t));
    uint64_t *seg_n = seg_first + nseg + 1, *seg_off = seg_n + nseg + 1;
    mho_plan_segments(ch_len, C, p, seg_ch, seg_first, seg_n, seg_off, &cap);
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t s = 0; s < (int64_t)nseg; ++s) {
        const uint32_t c = seg_ch[s];
        uint64_t w0, w1;
        int sk;
        window_of(ch_len[c], p, &w0, &w1, &sk);
        mho_decode_segment(payload + seg_off[s], seg_n[s], p, peak[c], enc[c],
                           out + ch_off[c] + w0 + seg_first[s]);
    }
    free(seg_first);
    free(seg_ch);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic MUA: counter-based integer generator.  One splitmix64
 * finaliser per 4 consecutive bins, 16 uniform bits per bin, inverse-CDF by integer
 * thresholds thr[s] = floor(65536 * P(X <= s)), s < 15, computed once on the host.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t seed, uint64_t ch, uint64_t q)
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

void mho_synth(uint8_t *data, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
               const uint32_t *thr, uint64_t seed, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t c = 0; c < (int64_t)C; ++c) {
        uint8_t *x = data + ch_off[c];
        const uint32_t *th = thr + (size_t)c * 15;
        for (uint64_t t = 0; t < ch_len[c]; ++t) {
            const uint64_t z = mix64(seed, (uint64_t)c, t >> 2);
            const uint32_t u = (uint32_t)(z >> (16 * (t & 3))) & 0xFFFFu;
            uint8_t v = 0;
            for (int s = 0; s < 15; ++s) v += (u >= th[s]);
            x[t] = v;
        }
    }
}

/* functions_1.py:11-24 per channel: bin b sums samples [b*r, min(b*r+r, T)) */
void mho_rebin_u32(const uint8_t *x, uint64_t T, uint32_t r, uint32_t *out)
{
    const uint64_t nb = (T + r - 1) / r;
    for (uint64_t b = 0; b < nb; ++b) {
        uint32_t s = 0;
        for (uint64_t t = b * r; t < b * r + r && t < T; ++t) s += x[t];
        out[b] = s;
    }
}

/* MATLAB uint8(h.Values) saturates at 255 (Data/Load_and_bin_Sabes_store_as_mat_file.m:53) */
void mho_rebin_u8(const uint8_t *x, uint64_t T, uint32_t r, uint8_t *out)
{
    const uint64_t nb = (T + r - 1) / r;
    for (uint64_t b = 0; b < nb; ++b) {
        uint32_t s = 0;
        for (uint64_t t = b * r; t < b * r + r && t < T; ++t) s += x[t];
        out[b] = (uint8_t)(s > 255 ? 255 : s);
    }
}
