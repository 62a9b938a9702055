// mh_codec2.hpp -- second-generation encode / decode kernels (same container format).
//
// A workgroup (4 waves) owns up to 4 consecutive segments OF ONE CHANNEL, so the channel's
// tables sit once per workgroup at a fixed LDS address:
//   encode: 256-entry PAIR table {code, len} indexed by two clipped symbols (b0 | b1 << 4);
//           two symbols per LDS lookup, two lookups merged in 32-bit before they touch the
//           lane's 64-bit accumulator.  When no byte of a 16-byte piece exceeds 15 (checked
//           once per piece per wave) the two indices of a dword come from x | x >> 4 with no
//           per-byte clipping.
//   decode: multi-symbol table, one lookup yields up to 4 symbols already spread to bytes
//           {4 symbol bytes, n | bits << 8}; for S <= 3 (maxlen <= 2) every 8-bit window
//           holds >= 4 symbols, so one lookup == one output dword.  The chunk payload is
//           staged in LDS with coalesced loads; the symbol loop touches no global memory
//           except its 16-byte output stores.
// Partial (last) chunks of a channel go through the first-generation per-symbol routines.
#pragma once
#include "mh_kernels.hpp"
#include "mh_layout.hpp"
#include "mh_planner.hpp"

namespace mh {

struct TaskArgs {
    const WgTask *wg;           // shared-table kernels: one record per workgroup task (mh_planner.hpp)
    const WaveTask *wt;         // per-wave-table kernels: one record per wave task
    uint32_t ntask;
    // shared-table kernels: geometry of a full segment (every segment of a task but its last is one)
    uint32_t seg_samples;       // samples
    uint64_t seg_src_stride;    // bytes between the sources of consecutive segments
    uint64_t slot_full;         // words between their slots
};

// ------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------
struct Enc2Args {
    EncArgs e;
    TaskArgs t;
};

constexpr uint32_t kEncSharedDw = 512 + 32;  // pair table (256 x uint2) + single table (16 x uint2)

__host__ __device__ inline uint32_t enc2_wave_dwords(uint32_t stage_dw)
{
    return 96 + stage_dw * 64;  // carried tail (64) + header room (32) + staging / image area
}

// Slow path for a chunk (m <= 16384 samples) whose sub-streams do not fit the capped LDS staging
// (more than 3 bits/sample in some lane): two passes straight from global memory -- lengths,
// wave prefix sum, zero the chunk image in place, then OR every codeword into it with global
// atomics.  Correct for any data; only adversarial inputs ever get here.
// Returns {words written, code bits}.
// Input packing PK of the encoder (template parameter throughout): 0 = one byte per sample (the
// channel-major container), 4 / 2 = packed pieces written by k_deinterleave2<PK> for the time-major path.
template <int PK>
struct RawPiece { typedef u32x4 type; };
template <>
struct RawPiece<4> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <>
struct RawPiece<2> { typedef uint32_t type; };
template <int PK>
constexpr uint32_t piece_bytes() { return PK == 0 ? 16u : PK == 4 ? 8u : 4u; }

// sample `idx` (counted from `src`, which points at a piece boundary) of a byte / packed stream
template <int PK>
__device__ __forceinline__ uint32_t sample_at(const uint8_t *src, uint32_t idx)
{
    if (PK == 0) return src[idx];
    if (PK == 4) return (src[idx >> 1] >> (4 * (idx & 1u))) & 15u;  // little-endian bit packing (mh_layout.hpp)
    return (src[idx >> 2] >> (2 * (idx & 3u))) & 3u;
}

template <int PK>
__device__ __forceinline__ uint2 encode_chunk_slow_body(const uint8_t *__restrict__ src, uint32_t m,
                                                        const uint2 *lut1, uint32_t *__restrict__ gdst,
                                                        int lane)
{
    uint32_t tot = 0;
#pragma unroll 1
    for (int k = 0; k < kRows; ++k) {
        const uint32_t base = ((uint32_t)k * kLanes + lane) * MH_PIECE;
#pragma unroll 1
        for (int i = 0; i < MH_PIECE; ++i)
            if (base + i < m) {
                const uint32_t b = sample_at<PK>(src, base + i);
                tot += lut1[b > 15u ? 15u : b].y;
            }
    }
    const uint32_t incl = wave_scan_incl(tot, lane);
    const uint32_t P = incl - tot;
    const uint32_t B = __shfl(incl, 63, 64);
    const uint32_t nw = (B + 31) >> 5;
    const uint32_t mn = wave_min(tot), hwid = hdr_width(wave_max(tot) - mn), hw = hdr_words(hwid);
    uint32_t *gpay = gdst + hw;
    if ((uint32_t)lane < hw) gdst[lane] = lane == 0 ? (mn | (hwid << 12)) : 0u;
    for (uint32_t i = lane; i < nw; i += 64) gpay[i] = 0;
    __threadfence();  // the zeros are at L2 before any atomic below
    if (hwid) {
        const uint32_t fb = 16u + (uint32_t)lane * hwid;
        const uint64_t f = (uint64_t)(tot - mn) << (fb & 31);
        atomicOr(&gdst[fb >> 5], (uint32_t)f);
        if ((uint32_t)(f >> 32)) atomicOr(&gdst[(fb >> 5) + 1], (uint32_t)(f >> 32));
    }
    uint32_t pos = P;
#pragma unroll 1
    for (int k = 0; k < kRows; ++k) {
        const uint32_t base = ((uint32_t)k * kLanes + lane) * MH_PIECE;
#pragma unroll 1
        for (int i = 0; i < MH_PIECE; ++i)
            if (base + i < m) {
                const uint32_t b = sample_at<PK>(src, base + i);
                const uint2 e = lut1[b > 15u ? 15u : b];
                const uint64_t v = (uint64_t)e.x << (pos & 31);
                atomicOr(&gpay[pos >> 5], (uint32_t)v);
                if ((uint32_t)(v >> 32)) atomicOr(&gpay[(pos >> 5) + 1], (uint32_t)(v >> 32));
                pos += e.y;
            }
    }
    __threadfence();
    return make_uint2(hw + nw, B);
}

// Out of line for the full-chunk loop (keeps the hot kernel small).  The partial-chunk routine, itself
// out of line, inlines the body instead: a call inside a callee would need a stack frame, and with
// it scratch memory for every wave of the kernel.
template <int PK>
__device__ __noinline__ uint2 encode_chunk_slow(const uint8_t *__restrict__ src, uint32_t m,
                                                const uint2 *lut1, uint32_t *__restrict__ gdst,
                                                int lane)
{
    return encode_chunk_slow_body<PK>(src, m, lut1, gdst, lane);
}

// LC: accumulator checks.  0 maxlen<=2: one per piece; 1 maxlen<=4: one per 2 dwords (8 codewords
//     always fit 32 bits); 2 maxlen<=8: one per 2 dwords when 8 codewords fit 32 bits in every lane
//     of the wave, else one per dword; 3 maxlen==9: as 2, plus a per-pair route when four codewords
//     exceed a dword.  The escapes are wave-uniform branches on __any().
//     For LC <= 1 the staging (8 * maxlen dwords per lane, enc_stage_dw) holds the worst case of a chunk,
//     so the staging-full checks and the overflow route compile away.
// PB: bits per symbol in the pair index.  PB=3 (S<=8) keeps the hot entries (small symbols) on
//     distinct LDS banks; PB=4 (S=9,10) xor-swizzles the index for the same reason.
constexpr int kWin = 8;  // rows (1 KiB each) a wave keeps in flight

template <int PB>
__device__ __forceinline__ uint32_t pair_index_word(uint32_t x)
{
    // byte0 = b0 | b1 << PB, byte2 = b2 | b3 << PB  (requires every byte < 2^PB)
    uint32_t y = x | (x >> (8 - PB));
    if (PB == 4) y ^= (y >> 3) & 0x1F1F1F1Fu;  // bijective on each byte
    return y;
}

template <int PB>
__device__ __forceinline__ uint32_t clip_word(uint32_t d)
{
    const uint32_t lim = (1u << PB) - 1u;
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t b = (d >> (8 * i)) & 0xFFu;
        b = b > lim ? lim : b;
        r |= b << (8 * i);
    }
    return r;
}

// The two pair-table entries of a dword of four clipped symbols.  The index is formed pre-scaled by the
// entry size: for PB = 3, (x << 3 | x >> 2) has (b0 | b1 << 3) * 8 in bits 3..8 and (b2 | b3 << 3) * 8 in bits
// 19..24 -- two masks away from the two LDS byte offsets (4 VALU per dword instead of 6).
template <int PB>
__device__ __forceinline__ void pair_entries(uint32_t x, const uint2 *lut2, uint2 &e0, uint2 &e1)
{
    const char *base = reinterpret_cast<const char *>(lut2);
    if (PB == 3) {
        uint32_t y8;  // (x << 3) | (x >> 2) in two instructions (the compiler spends three and a fused and-or)
        asm("v_lshl_or_b32 %0, %1, 3, %2" : "=v"(y8) : "v"(x), "v"(x >> 2));
        e0 = *reinterpret_cast<const uint2 *>(base + (y8 & 0x1F8u));
        e1 = *reinterpret_cast<const uint2 *>(base + ((y8 >> 16) & 0x1F8u));
    } else {
        const uint32_t y8 = pair_index_word<PB>(x) << 3;
        e0 = *reinterpret_cast<const uint2 *>(base + (y8 & 0x7F8u));
        e1 = *reinterpret_cast<const uint2 *>(base + ((y8 >> 16) & 0x7F8u));
    }
}

// ---- the row body of the byte-input encoder -----------------------------------------------------------
// One 16-sample piece per lane (x = 4 dwords of counts) appended to the lane's 64-bit accumulator.  It exists
// as MACROS because two functions need the same statements -- encode_full_chunk's unrolled row loop and
// encode_row for partial chunks -- and routing the hot loop through a function cost the S = 10 kernel 6 %:
// one source text, two expansions.  The expansion site provides LC, PB, ABL, lut2, acc, nb, sp, st, cap.
__host__ __device__ constexpr int stage_ne(int LC) { return LC == 0 ? 4 : 8; }  // cap / 4 for the largest cap of the class

// Where a lane keeps its j-th spilled dword: staging ROW (j % NE) * 4 + j / NE of its column (NE = dwords one lane of
// the merge gathers per group, stage_ne(LC)).  The merge's lane (sub-stream s, quarter q) reads rows r * 4 + q,
// r = 0 .. NE-1 -- 64 consecutive LDS words per instruction, no bank conflict -- and with this permutation those are
// the CONSECUTIVE dwords q * NE + r of the sub-stream, so each output word is a funnel shift of two registers.
#define MH_STAGE_AT(j) ((((j) & (uint32_t)(MH_NE_ - 1)) << 2 | ((j) >> (MH_NE_ == 4 ? 2 : 3))) * 16)
#define MH_FLUSH()                                                                                  \
    if (nb >= 32) {                                                                                 \
        if ((ABL < 3 || ABL >= 5) && (LC <= 1 || sp < cap)) st[MH_STAGE_AT(sp)] = (uint32_t)acc;    \
        acc >>= 32;                                                                                 \
        nb -= 32;                                                                                   \
        ++sp;                                                                                       \
    }

// Short codes (LC 0: max length 2, LC 1: max length 4): the 16 codewords of a piece.  Everything that fits
// 32 bits is put together in 32-bit arithmetic first -- LC 0: the whole piece (<= 32 bits), LC 1: each half --
// so the lane's 64-bit accumulator is touched once resp. twice per piece.
#define MH_SHORT_CODES_ROW(x)                                                          \
    {                                                                                  \
        uint2 e0, e1, e2, e3, e4, e5, e6, e7;                                          \
        pair_entries<PB>((x).x, lut2, e0, e1);                                         \
        pair_entries<PB>((x).y, lut2, e2, e3);                                         \
        pair_entries<PB>((x).z, lut2, e4, e5);                                         \
        pair_entries<PB>((x).w, lut2, e6, e7);                                         \
        const uint32_t q0 = e0.x | (e1.x << e0.y), t0 = e0.y + e1.y;                   \
        const uint32_t q1 = e2.x | (e3.x << e2.y), t1 = e2.y + e3.y;                   \
        const uint32_t q2 = e4.x | (e5.x << e4.y), t2 = e4.y + e5.y;                   \
        const uint32_t q3 = e6.x | (e7.x << e6.y), t3 = e6.y + e7.y;                   \
        const uint32_t h0 = q0 | (q1 << t0), h1 = q2 | (q3 << t2);                     \
        if (LC == 0) {                                                                 \
            acc |= (uint64_t)(h0 | (h1 << (t0 + t1))) << nb;                           \
            nb += t0 + t1 + t2 + t3;                                                   \
            MH_FLUSH();                                                                \
        } else {                                                                       \
            acc |= (uint64_t)h0 << nb; nb += t0 + t1; MH_FLUSH();                      \
            acc |= (uint64_t)h1 << nb; nb += t2 + t3; MH_FLUSH();                      \
        }                                                                              \
    }

// Long codes: two dwords (8 codewords) share one accumulator check whenever they fit 32 bits in every lane of
// the wave; otherwise dword by dword, and for 9-bit codes pair by pair (wave-uniform escapes on __any()).
#define MH_LONG_CODES_ROW(x)                                                                           \
    _Pragma("unroll") for (int dp = 0; dp < 2; ++dp)                                                   \
    {                                                                                                  \
        const uint32_t y0 = pair_index_word<PB>((x)[2 * dp]), y1 = pair_index_word<PB>((x)[2 * dp + 1]); \
        const uint2 a0 = lut2[y0 & 0xFFu], a1 = lut2[(y0 >> 16) & 0xFFu];                              \
        const uint2 b0 = lut2[y1 & 0xFFu], b1 = lut2[(y1 >> 16) & 0xFFu];                              \
        const uint32_t t0 = a0.y + a1.y, t1 = b0.y + b1.y;                                             \
        /* MH_WT_ (wave-task encoders, partial chunks): ONE vote on the common path -- t0 > 32 implies t0 + t1 > 32 --  \
           2400 x 72 000 S = 10 encode 67.7 -> 64.5 us.  The long-channel kernels keep the two-vote form below, text   \
           and all: with one vote S = 10 encodes 1 % slower there, and folding both forms into one condition cost 10 % */ \
        if (MH_WT_) {                                                                                  \
            if (__builtin_expect(__any(t0 + t1 > 32u), 0)) {                                           \
                if (LC == 3 && __any(t0 > 32u || t1 > 32u)) {                                          \
                    acc |= (uint64_t)a0.x << nb; nb += a0.y; MH_FLUSH();                               \
                    acc |= (uint64_t)a1.x << nb; nb += a1.y; MH_FLUSH();                               \
                    acc |= (uint64_t)b0.x << nb; nb += b0.y; MH_FLUSH();                               \
                    acc |= (uint64_t)b1.x << nb; nb += b1.y; MH_FLUSH();                               \
                } else {                                                                               \
                    acc |= (uint64_t)(a0.x | (a1.x << a0.y)) << nb; nb += t0; MH_FLUSH();              \
                    acc |= (uint64_t)(b0.x | (b1.x << b0.y)) << nb; nb += t1; MH_FLUSH();              \
                }                                                                                      \
            } else {                                                                                   \
                const uint32_t q0 = a0.x | (a1.x << a0.y), q1 = b0.x | (b1.x << b0.y);                 \
                acc |= (uint64_t)(q0 | (q1 << t0)) << nb;                                              \
                nb += t0 + t1;                                                                         \
                MH_FLUSH();                                                                            \
            }                                                                                          \
        } else if (LC == 3 && __builtin_expect(__any(t0 > 32u || t1 > 32u), 0)) {                      \
            acc |= (uint64_t)a0.x << nb; nb += a0.y; MH_FLUSH();                                       \
            acc |= (uint64_t)a1.x << nb; nb += a1.y; MH_FLUSH();                                       \
            acc |= (uint64_t)b0.x << nb; nb += b0.y; MH_FLUSH();                                       \
            acc |= (uint64_t)b1.x << nb; nb += b1.y; MH_FLUSH();                                       \
        } else if (__any(t0 + t1 > 32u)) {                                                             \
            acc |= (uint64_t)(a0.x | (a1.x << a0.y)) << nb; nb += t0; MH_FLUSH();                      \
            acc |= (uint64_t)(b0.x | (b1.x << b0.y)) << nb; nb += t1; MH_FLUSH();                      \
        } else {                                                                                       \
            const uint32_t q0 = a0.x | (a1.x << a0.y), q1 = b0.x | (b1.x << b0.y);                     \
            acc |= (uint64_t)(q0 | (q1 << t0)) << nb;                                                  \
            nb += t0 + t1;                                                                             \
            MH_FLUSH();                                                                                \
        }                                                                                              \
    }

// rare: a count that does not fit PB bits somewhere in this KiB row -> clip the row; then the codewords
#define MH_ENCODE_ROW(x)                                                                   \
    {                                                                                      \
        constexpr uint32_t kHiMask_ = 0x01010101u * (0xFFu & ~((1u << PB) - 1u));          \
        if (__any((((x).x | (x).y | (x).z | (x).w) & kHiMask_) != 0)) {                    \
            (x).x = clip_word<PB>((x).x);                                                  \
            (x).y = clip_word<PB>((x).y);                                                  \
            (x).z = clip_word<PB>((x).z);                                                  \
            (x).w = clip_word<PB>((x).w);                                                  \
        }                                                                                  \
        if (LC >= 2) {                                                                     \
            MH_LONG_CODES_ROW(x)                                                           \
        } else {                                                                           \
            MH_SHORT_CODES_ROW(x)                                                          \
        }                                                                                  \
    }

template <int PK>
__device__ __forceinline__ typename RawPiece<PK>::type load_row(const uint8_t *p)
{
    if constexpr (PK == 0) {
#ifdef MH_ROW_PLAIN  // A/B builds: plain instead of non-temporal row loads
        return *reinterpret_cast<const u32x4_u *>(p);
#endif
        return __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
    } else if constexpr (PK == 4) {
        typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
        const u32x2_u v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_u *>(p));
        typename RawPiece<4>::type r = {v.x, v.y};
        return r;
    } else {
        return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
    }
}

// ---- packed input: the 256-entry table is indexed by the stream's own bytes ---------------------
// PK = 4: a byte is (b0 | b1 << 4), the PB = 4 pair index (xor-swizzled like the byte path's, for the banks).
// PK = 2: a byte holds FOUR symbols; the table entry is their four codewords back to back (<= 12 bits for the
//         S <= 4 this packing serves).  The hot entries -- all-zero byte, one non-zero symbol -- are 0 and the
//         powers of two; folding index bits 5..7 into bits 0..3 puts them on distinct banks.
template <int PK>
__device__ __forceinline__ uint32_t packed_index_word(uint32_t w)
{
    if (PK == 4) return w ^ ((w >> 3) & 0x1F1F1F1Fu);
    const uint32_t h = (w >> 5) & 0x07070707u;
    return w ^ h ^ (h << 1);  // every byte: low bits ^= h ^ 2h, h < 8: stays inside the byte
}

// entry of the PK = 2 table for the byte `v`; sym(b) -> {code, length} of symbol b (0..3)
template <class F>
__device__ __forceinline__ uint2 quad_entry(uint32_t v, F sym)
{
    uint32_t code = 0, len = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint2 e = sym((v >> (2 * i)) & 3u);
        code |= e.x << len;
        len += e.y;
    }
    return make_uint2(code, len);
}

// ---- per-wave LDS buffer of the encoder ------------------------------------------------------
// buf[0,64)   words carried over from the previous chunk (not yet a whole 256-byte block)
// buf[64,96)  room so that the 32-word chunk header always fits below R
// R = buf+96  staging AND image area, 64*cap dwords:
//   staging: lane l = 16*g + s keeps its j-th spilled dword at R[g*16*cap + j*16 + s]
//   image:   header at buf[pend .. pend+hw), hw <= 25 words, payload right behind it, growing upwards.
// The merge walks the four 16-lane groups in order.  A group's staged dwords (<= 16*cap) are
// first pulled into registers by all 64 lanes (cap/4 each), then ORed into the payload.  The
// payload written through group g ends at most at buf[pend+32 + 16*(g+1)*cap) <= R + 16*(g+1)*cap,
// the start of group g+1's staging, so nothing still needed is ever overwritten: the image is
// built IN PLACE and the wave needs half the LDS of a separate staging + image pair.
__device__ __forceinline__ uint32_t *stage_lane_base(uint32_t *buf, uint32_t cap, int lane)
{
    return buf + 96 + (uint32_t)(lane >> 4) * 16 * cap + (lane & 15);
}

#define MH_WAVE_SYNC()                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier()

// NE >= cap/4: staged dwords one lane gathers per group.
// FULL (a chunk of 16384 samples: every sub-stream has >= 256 bits, so a payload word is shared by at most TWO
// neighbouring sub-streams): a sub-stream is contiguous at bit offset P, so word P/32 + j of the image is the funnel
// shift (v[j] << s) | (v[j-1] >> (32 - s)), s = P % 32, of two CONSECUTIVE staged dwords -- which the gather holds in
// neighbouring registers (MH_STAGE_AT), the one at a quarter's start coming from the lane 16 below.  Interior words are
// written with plain ds_write; only a sub-stream's first and last word, which it may share with its neighbours, are
// OR-ed into words zeroed beforehand: 2 LDS atomics per sub-stream instead of up to 2 per staged dword (and no
// zero-fill of the image).  Partial chunks (sub-streams of any length, several to a word) keep the OR for every dword.
// DPP: the wave-wide scan / min / max of the sub-stream lengths as DPP ladders instead of ds_bpermute chains (mh_device.hpp,
// wave_scan_incl_dpp: for launches that wait on latency -- wave-task encoders, partial chunks).
template <int NE, int ABL, bool FULL, bool DPP = false>
__device__ __forceinline__ void merge_and_flush(uint32_t *buf, uint32_t cap, uint32_t tot, uint32_t sp,
                                                uint32_t *__restrict__ &dst, uint32_t &pend, int lane,
                                                uint32_t &words, uint32_t &bits)
{
    const uint32_t incl = DPP ? wave_scan_incl_dpp(tot) : wave_scan_incl(tot, lane);
    const uint32_t P = incl - tot;
    const uint32_t B = DPP ? wave_last(incl) : __shfl(incl, 63, 64);
    const uint32_t nw = (B + 31) >> 5;
    uint32_t *hdr = buf + pend;
    const uint32_t mn = DPP ? wave_min_dpp(tot) : wave_min(tot);
    const uint32_t hwid = hdr_width((DPP ? wave_max_dpp(tot) : wave_max(tot)) - mn), hw = hdr_words(hwid);
    uint32_t *pay = hdr + hw;  // hw <= 25 < kHdrWords: the image starts no higher than before
    if ((uint32_t)lane < hw) hdr[lane] = lane == 0 ? (mn | (hwid << 12)) : 0u;
    MH_WAVE_SYNC();
    if (hwid) {
        const uint32_t fb = 16u + (uint32_t)lane * hwid;
        const uint64_t f = (uint64_t)(tot - mn) << (fb & 31);
        atomicOr(&hdr[fb >> 5], (uint32_t)f);
        if ((uint32_t)(f >> 32)) atomicOr(&hdr[(fb >> 5) + 1], (uint32_t)(f >> 32));
    }
    const uint32_t *R = buf + 96;
    const int sl = lane & 15, jq = lane >> 4;
    uint32_t zeroed = 0;  // !FULL: payload words [0, zeroed) are initialised
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        const int src = g * 16 + sl;
        const uint32_t sp_s = __shfl(sp, src, 64), P_s = __shfl(P, src, 64);
        const uint32_t *sg = R + (uint32_t)g * 16 * cap;
        uint32_t vals[NE];  // staged dwords jq * NE + r of sub-stream `src` (0 past its end)
#pragma unroll
        for (int r = 0; r < NE; ++r) vals[r] = (uint32_t)(jq * NE + r) < sp_s ? sg[(uint32_t)(r * 4 + jq) * 16 + sl] : 0u;
        MH_WAVE_SYNC();
        if (FULL) {
            const uint32_t tot_s = __shfl(tot, src, 64);
            const uint32_t s = P_s & 31u, base = P_s >> 5;
            const uint32_t nword = (s + tot_s + 31) >> 5;          // image words the sub-stream touches (>= 8)
            const bool tail_shared = ((s + tot_s) & 31u) != 0;     // its last word also holds the next sub-stream's head
            // zero the words this group will OR into: every sub-stream's shared last word (its first word is the
            // previous sub-stream's last: zeroed by the same instruction, or by the previous group's)
            if (jq == 0 && tail_shared) pay[base + nword - 1] = 0;
            uint32_t prev = __shfl(vals[NE - 1], lane - 16, 64);   // dword jq * NE - 1 sits one quarter down
            if (jq == 0) prev = 0;
            MH_WAVE_SYNC();
#pragma unroll
            for (int r = 0; r <= NE; ++r) {
                // (r == NE: word 4 * NE, which exists only when the sub-stream fills the whole staging and does not
                // start on a word boundary -- the top quarter emits it from its last register)
                if (r == NE && jq != 3) break;
                const uint32_t j = (uint32_t)(jq * NE + r);
                const uint32_t cur = r < NE ? vals[r < NE ? r : 0] : 0u, lo = r ? vals[r - 1] : prev;
                const uint32_t w = s ? __builtin_amdgcn_alignbit(cur, lo, 32u - s) : cur;
                if (j < nword) {
                    if ((j == 0 && s != 0) || (j + 1 == nword && tail_shared))
                        atomicOr(&pay[base + j], w);
                    else
                        pay[base + j] = w;
                }
            }
        } else {
            const uint32_t w_end = (__shfl(incl, g * 16 + 15, 64) + 31) >> 5;
            for (uint32_t i = zeroed + lane; i < w_end; i += 64) pay[i] = 0;
            zeroed = w_end;
            MH_WAVE_SYNC();
#pragma unroll
            for (int r = 0; r < NE; ++r) {
                const uint32_t jj = (uint32_t)(jq * NE + r);
                if (jj < sp_s) {
                    const uint32_t pos = P_s + 32 * jj;
                    const uint64_t sh = (uint64_t)vals[r] << (pos & 31);
                    atomicOr(&pay[pos >> 5], (uint32_t)sh);
                    if ((uint32_t)(sh >> 32)) atomicOr(&pay[(pos >> 5) + 1], (uint32_t)(sh >> 32));
                }
            }
        }
        MH_WAVE_SYNC();
    }
    // only whole, 256-byte-aligned blocks go to HBM (16 B per lane, non-temporal); the rest waits
    const uint32_t total = pend + hw + nw;
    const uint32_t nflush = total & ~63u;
    // (ABL 8 only; an offset, not a rebuilt pointer -- that would turn the store into a FLAT access)
    const ptrdiff_t back_abl = (ptrdiff_t)((reinterpret_cast<uintptr_t>(dst) & (uintptr_t)32767) >> 2);
    if (ABL < 1 || ABL >= 5) {
        for (uint32_t i = lane * 4; i < nflush; i += 256) {
            u32x4 blk = {0u, 0u, 0u, 0u};
            if (ABL != 13) blk = *reinterpret_cast<const u32x4 *>(buf + i);
            u32x4_u *to = reinterpret_cast<u32x4_u *>(dst + i);
            // (tuning builds: ABL 5..7 time other flavours of this store -- plain, nt + sc1, sc0 sc1; 8 issues
            // the same stores but keeps them inside one L2-resident 4 KiB per wave: no DRAM writes)
            if (ABL == 8) to = reinterpret_cast<u32x4_u *>(dst - back_abl + (i & 1023u));
            if (ABL == 13) {  // image merged as usual, but the stores take register data: no ds_read -> store chain
                const u32x4 junk = {tot, sp, P, (uint32_t)lane};
                __builtin_nontemporal_store(junk, to);
                continue;
            }
            if (ABL == 5 || ABL == 8)
                *to = blk;
            else if (ABL == 6)
                asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(to), "v"(blk) : "memory");
            else if (ABL == 7)
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(to), "v"(blk) : "memory");
            else
                __builtin_nontemporal_store(blk, to);
        }
    }
    const uint32_t tail = total - nflush;
    uint32_t t = 0;
    if ((uint32_t)lane < tail) t = buf[nflush + lane];
    MH_WAVE_SYNC();
    if ((uint32_t)lane < tail) buf[lane] = t;
    MH_WAVE_SYNC();
    dst += nflush;
    pend = tail;
    words = hw + nw;
    bits = B;
}

// Ablation only (tuning builds): the flush of a typical chunk image (736 words = 1.44 bits/sample) with register data
// instead of the merged image -- same addresses, same block rule as merge_and_flush.
__device__ __forceinline__ void abl_flush_fixed(uint32_t seed, uint32_t *__restrict__ &dst, uint32_t &pend, int lane,
                                                uint32_t &words, uint32_t &bits)
{
    const uint32_t total = pend + 736u, nflush = total & ~63u;
    const u32x4 blk = {seed, seed ^ 1u, seed ^ 2u, (uint32_t)lane};
    for (uint32_t i = lane * 4; i < nflush; i += 256) __builtin_nontemporal_store(blk, reinterpret_cast<u32x4_u *>(dst + i));
    dst += nflush;
    pend = total - nflush;
    words = 736u;
    bits = 736u * 32u;
}

// chunk whose sub-streams outgrew the staging: flush the carried tail, then the global slow path
template <bool INLINE_SLOW, int PK>
__device__ __forceinline__ void overflow_chunk(const uint8_t *src, uint32_t m, const uint2 *lut1, uint32_t *buf,
                                               uint32_t *__restrict__ &dst, uint32_t &pend, int lane,
                                               uint32_t &words, uint32_t &bits)
{
    if ((uint32_t)lane < pend) dst[lane] = buf[lane];
    MH_WAVE_SYNC();
    dst += pend;
    pend = 0;
    const uint2 r = INLINE_SLOW ? encode_chunk_slow_body<PK>(src, m, lut1, dst, lane) : encode_chunk_slow<PK>(src, m, lut1, dst, lane);
    words = r.x;
    bits = r.y;
    dst += words;
}

// The row body as a function, for the partial-chunk encoder (ABL: see encode_full_chunk).
template <int LC, int PB, int ABL>
__device__ __forceinline__ void encode_row(u32x4 x, const uint2 *lut2, uint64_t &acc, uint32_t &nb, uint32_t &sp,
                                           uint32_t *st, uint32_t cap)
{
    constexpr int MH_NE_ = stage_ne(LC);
    constexpr bool MH_WT_ = true;  // (partial chunks: short launches)
    MH_ENCODE_ROW(x)
}

// The same for a piece of packed input (r = 2 dwords of 4-bit or 1 dword of 2-bit samples): every byte of
// the piece is a table index as it stands -- nothing is spread back to one byte per sample.  Used by the
// full-chunk and the partial-chunk encoder alike.
template <int LC, int PK, int ABL>
__device__ __forceinline__ void encode_row_packed(typename RawPiece<PK>::type r, const uint2 *lut2, uint64_t &acc,
                                                  uint32_t &nb, uint32_t &sp, uint32_t *st, uint32_t cap)
{
    constexpr int MH_NE_ = stage_ne(LC);
    if constexpr (PK == 2) {
        // 16 codewords in four lookups; LC 0 (max length 2): they always fit one dword, LC 1 (<= 4): 8 do
        const uint32_t y = packed_index_word<2>(r);
        const uint2 e0 = lut2[y & 0xFFu], e1 = lut2[(y >> 8) & 0xFFu], e2 = lut2[(y >> 16) & 0xFFu], e3 = lut2[y >> 24];
        const uint32_t t0 = e0.y + e1.y, t1 = e2.y + e3.y;
        const uint32_t q0 = e0.x | (e1.x << e0.y), q1 = e2.x | (e3.x << e2.y);
        if (LC == 0) {
            acc |= (uint64_t)(q0 | (q1 << t0)) << nb;
            nb += t0 + t1;
            MH_FLUSH();
        } else {
            acc |= (uint64_t)q0 << nb; nb += t0; MH_FLUSH();
            acc |= (uint64_t)q1 << nb; nb += t1; MH_FLUSH();
        }
    } else {
#pragma unroll
        for (int o = 0; o < 2; ++o) {  // a dword = 8 samples = four pair lookups
            const uint32_t y = packed_index_word<4>(o ? r.y : r.x);
            const uint2 a0 = lut2[y & 0xFFu], a1 = lut2[(y >> 8) & 0xFFu];
            const uint2 b0 = lut2[(y >> 16) & 0xFFu], b1 = lut2[y >> 24];
            const uint32_t t0 = a0.y + a1.y, t1 = b0.y + b1.y;
            if (LC == 3 && __builtin_expect(__any(t0 > 32u || t1 > 32u), 0)) {
                acc |= (uint64_t)a0.x << nb; nb += a0.y; MH_FLUSH();
                acc |= (uint64_t)a1.x << nb; nb += a1.y; MH_FLUSH();
                acc |= (uint64_t)b0.x << nb; nb += b0.y; MH_FLUSH();
                acc |= (uint64_t)b1.x << nb; nb += b1.y; MH_FLUSH();
            } else if (LC >= 2 && __any(t0 + t1 > 32u)) {
                acc |= (uint64_t)(a0.x | (a1.x << a0.y)) << nb; nb += t0; MH_FLUSH();
                acc |= (uint64_t)(b0.x | (b1.x << b0.y)) << nb; nb += t1; MH_FLUSH();
            } else {  // LC <= 1: 8 codewords of <= 4 bits always fit
                const uint32_t q0 = a0.x | (a1.x << a0.y), q1 = b0.x | (b1.x << b0.y);
                acc |= (uint64_t)(q0 | (q1 << t0)) << nb;
                nb += t0 + t1;
                if (LC != 0 || o == 1) { MH_FLUSH(); }
            }
        }
    }
}

// One full chunk.  v[] is a rolling window: row k of this chunk sits in v[k & 7]; after it is
// consumed the slot is refilled with the row 8 KiB further on (this chunk, then the next one).
// ABL (debug ablation, 0 in production): 1 no global stores, 2 also no merge, 3 also no staging
// writes, 4 loads only
// HAS_NEXT is a template constant on purpose: with a run-time flag the refill loads of the second half
// sit in a branch, the compiler's wait-count pass cannot count them, and every vmcnt in that half
// tightens by one per row down to vmcnt(0) -- the wave then drains its whole window at each chunk end.
// nxt = first byte of the chunk that follows `cur` in the stream (cur + one chunk in the plain layouts,
// elsewhere in the chunk-blocked intermediate of the time-major path); only read when HAS_NEXT
template <int LC, int PB, int ABL, bool HAS_NEXT, int PK, bool DPP = false>
__device__ __forceinline__ void encode_full_chunk(typename RawPiece<PK>::type (&v)[kWin], const uint8_t *__restrict__ cur,
                                                  const uint8_t *__restrict__ nxt, const uint2 *lut2, const uint2 *lut1,
                                                  uint32_t *buf, uint32_t cap, uint32_t *__restrict__ &dst,
                                                  uint32_t &pend, int lane, uint32_t &words, uint32_t &bits)
{
    constexpr int MH_NE_ = stage_ne(LC);
    constexpr bool MH_WT_ = DPP;  // (the wave-task kernels: see MH_LONG_CODES_ROW)
    uint64_t acc = 0;
    uint32_t nb = 0, sp = 0;
    uint32_t *st = stage_lane_base(buf, cap, lane);
    // (A/B builds: -DMH_ROW_GLOBAL = global loads for every encoder's rows, as before)
    // The rows of a chunk through a buffer resource on the chunk's first byte (wave-uniform: four scalar
    // registers) -- the row is an immediate / scalar offset and the lane's 16 bytes one vector register, where a
    // global load needs a 64-bit vector address per row (two VALU adds); and the cache policy is an operand of the
    // instruction (aux = 2: nt), not metadata an IR pass may drop (it did, for seven of a chunk's sixteen rows).
    // 1024 ch x 1e7 bins encode S = 5 / 8 -1.5 %; time-major 1024 x 1e7 block 4.04 -> 3.97 ms; 2400 x 72 000 S = 10 -3 %
    // (profiles/r03_dpp_reductions.txt).  Used by every encoder but the byte-input one of S <= 3: see kRowsByBuffer.
    // (A loop over the SAME few hundred MB reads slower this way -- 96 ch x 3.6e6 bins: +2..7 % -- because fewer of its
    // rows stay in the Infinity Cache between iterations: a property of re-reading one small input.)
    const auto rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(cur), 0, 0x7FFFFFFF, 0x00020000);
    const auto rs_nxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(nxt), 0, 0x7FFFFFFF, 0x00020000);
    const int voff = lane * (int)piece_bytes<PK>();
    constexpr int kRowBytes = kLanes * (int)piece_bytes<PK>();
    auto row_load = [&](const auto &rs, int row) {  // (packed pieces: 8 / 4 bytes per lane, the same form)
        if constexpr (PK == 0) {
            return (typename RawPiece<PK>::type)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, row * kRowBytes, 2);
        } else if constexpr (PK == 4) {
            return (typename RawPiece<PK>::type)__builtin_amdgcn_raw_buffer_load_b64(rs, voff, row * kRowBytes, 2);
        } else {
            return (typename RawPiece<PK>::type)__builtin_amdgcn_raw_buffer_load_b32(rs, voff, row * kRowBytes, 2);
        }
    };
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const typename RawPiece<PK>::type raw = v[k & (kWin - 1)];
        // (NOT the byte-input encoder of S <= 3: in the headline's alternation -- encode, decode, encode, ... -- the
        // decode that FOLLOWS that encoder runs 4 % slower when every row was read nt through the buffer form
        // (2.03 instead of 1.94-1.99 ms, tools/bench_with_lib.py; the encoder itself gains nothing at S = 3))
        constexpr bool kRowsByBuffer =
#ifndef MH_ROW_GLOBAL
            !(PK == 0 && LC == 0);
#else
            false;
#endif
        if constexpr (kRowsByBuffer) {
            if (k < kRows - kWin)
                v[k & (kWin - 1)] = row_load(rs_cur, k + kWin);
            else if (HAS_NEXT)
                v[k & (kWin - 1)] = row_load(rs_nxt, k + kWin - kRows);
        } else {
            if (k < kRows - kWin)
                v[k & (kWin - 1)] = load_row<PK>(cur + ((uint32_t)(k + kWin) * kLanes + lane) * piece_bytes<PK>());
            else if (HAS_NEXT)
                v[k & (kWin - 1)] = load_row<PK>(nxt + ((uint32_t)(k + kWin - kRows) * kLanes + lane) * piece_bytes<PK>());
        }
        if constexpr (PK != 0) {
            encode_row_packed<LC, PK, ABL>(raw, lut2, acc, nb, sp, st, cap);
            continue;
        }
        u32x4 x;
        if constexpr (PK == 0) x = raw;
        if (ABL == 4 || ABL == 11) {
            acc += x.x ^ x.y ^ x.z ^ x.w;
            continue;
        }
        MH_ENCODE_ROW(x)
    }
    if (ABL == 11 || ABL == 12) {  // 11: loads + a typical chunk's stores (the probe inside the real kernel);
        abl_flush_fixed((uint32_t)acc + nb + sp, dst, pend, lane, words, bits);  // 12: + the row arithmetic and staging
        return;
    }
    if (ABL >= 2 && ABL < 5) {  // keep the work alive, skip the rest
        words = 0;
        bits = (uint32_t)acc + nb + sp;
        return;
    }
    const uint32_t tot = sp * 32 + nb;
    if (nb > 0) {
        if (LC <= 1 || sp < cap) st[MH_STAGE_AT(sp)] = (uint32_t)acc;
        ++sp;
    }
    MH_WAVE_SYNC();
    if (LC >= 2 && __any(sp > cap)) {
        overflow_chunk<false, PK>(cur, kChunk, lut1, buf, dst, pend, lane, words, bits);
        return;
    }
    merge_and_flush<stage_ne(LC), ABL, true, DPP>(buf, cap, tot, sp, dst, pend, lane, words, bits);
}

// Last, partial chunk of a channel (m < 16384 samples).  Its full pieces (16 samples) take the same
// pair-table row routine as a full chunk -- all their loads issued up front, eight rows at a time --
// and only the one cut piece (m % 16 samples, in lane (m / 16) % 64, that lane's last piece) goes
// symbol by symbol.  Same staging and the same in-place merge.
// Returns {words, bits, new pend, words by which dst advanced}.
// (Out of line, so its pointer arguments carry their address spaces in the signature: as generic pointers every
// table lookup and staging access in here was a FLAT instruction -- slower than ds_read / global_load and counted
// on both wait counters.  Short channels spend a fifth of their samples in this function.)
#define MH_AS_GLOBAL __attribute__((address_space(1)))
#define MH_AS_LDS __attribute__((address_space(3)))
template <int LC, int PB, int PK>
__device__ __noinline__ uint4 encode_partial_chunk(const MH_AS_GLOBAL uint8_t *src_g, uint32_t m, const MH_AS_LDS uint2 *lut2_l,
                                                   const MH_AS_LDS uint2 *lut1_l, MH_AS_LDS uint32_t *buf_l, uint32_t cap,
                                                   MH_AS_GLOBAL uint32_t *dst0_g, uint32_t pend, int lane)
{
    constexpr int NE = stage_ne(LC), MH_NE_ = NE;
    const uint8_t *__restrict__ src = (const uint8_t *)src_g;
    const uint2 *lut2 = (const uint2 *)lut2_l, *lut1 = (const uint2 *)lut1_l;
    uint32_t *buf = (uint32_t *)buf_l;
    uint32_t *__restrict__ dst0 = (uint32_t *)dst0_g;
    uint32_t *__restrict__ dst = dst0;
    uint32_t words, bits;
    uint64_t acc = 0;
    uint32_t nb = 0, sp = 0;
    uint32_t *st = stage_lane_base(buf, cap, lane);
    const uint32_t nfp = m >> 4;               // full pieces
    const uint32_t nrows = (nfp + 63) >> 6;    // rows holding at least one of them (wave-uniform)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if ((uint32_t)(half * 8) >= nrows) break;
        typename RawPiece<PK>::type v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t piece = (uint32_t)(half * 8 + r) * kLanes + lane;
            v[r] = typename RawPiece<PK>::type{};
            if (piece < nfp) v[r] = load_row<PK>(src + piece * piece_bytes<PK>());
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t k = (uint32_t)(half * 8 + r);
            if (k < nrows && k * kLanes + lane < nfp) {  // the escapes inside ballot the active lanes only
                if constexpr (PK == 0)
                    encode_row<LC, PB, 0>(v[r], lut2, acc, nb, sp, st, cap);
                else
                    encode_row_packed<LC, PK, 0>(v[r], lut2, acc, nb, sp, st, cap);
            }
        }
    }
    const uint32_t cnt = m & 15u;
    if (cnt && (uint32_t)lane == (nfp & 63u)) {  // the cut piece
        for (uint32_t i = 0; i < cnt; ++i) {
            uint32_t b = sample_at<PK>(src, nfp * MH_PIECE + i);
            b = b > 15u ? 15u : b;
            const uint2 e = lut1[b];
            acc |= (uint64_t)e.x << nb;
            nb += e.y;
            if (nb >= 32) {
                if (LC <= 1 || sp < cap) st[MH_STAGE_AT(sp)] = (uint32_t)acc;
                acc >>= 32;
                nb -= 32;
                ++sp;
            }
        }
    }
    const uint32_t tot = sp * 32 + nb;
    if (nb > 0) {
        if (LC <= 1 || sp < cap) st[MH_STAGE_AT(sp)] = (uint32_t)acc;
        ++sp;
    }
    MH_WAVE_SYNC();
    if (LC >= 2 && __any(sp > cap))
        overflow_chunk<true, PK>(src, m, lut1, buf, dst, pend, lane, words, bits);
    else
        merge_and_flush<NE, 0, false, true>(buf, cap, tot, sp, dst, pend, lane, words, bits);
    return make_uint4(words, bits, pend, (uint32_t)(dst - dst0));
}

// One segment, one wave: the chunks of segment `seg` of channel `ch` through the wave's tables
// (lut2 pair table, lut1 single-symbol table) and its staging buffer.
// the first kWin rows of a segment that starts with a full chunk
template <int PK>
__device__ __forceinline__ void load_first_rows(typename RawPiece<PK>::type (&v)[kWin], const uint8_t *src, int lane)
{
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
        v[k] = load_row<PK>(src + ((uint32_t)k * kLanes + lane) * piece_bytes<PK>());
        asm volatile("" ::: "memory");  // keep the rows in issue order: vmcnt retires in order
    }
}

// src = first sample, n = samples, out = the segment's slot.  PRE: the caller has already issued
// load_first_rows into v (when n >= one chunk) -- the per-wave-table kernel does so before it builds
// its tables, so that the rows are in flight while the table entries are computed.
// PUBLISH: the wave stores the segment's word count and adds its bits to the channel's total itself (wave tasks);
// otherwise the caller does (the shared-table kernel publishes a whole task at once: one 32-byte store and one
// atomic per workgroup instead of four scattered 8-byte stores and four atomics -- 313 000 of each per launch on
// the 1024 x 1e7 set cost 0.06 ms, tools/ablate_encode.py level 15).
// PUBLISH (the wave-task kernels: a wave is a task of its own) also selects the DPP reductions in the chunks' merge.
template <int LC, int PB, int ABL, bool PRE, int PK, bool PUBLISH = true>
__device__ __forceinline__ void encode_segment(const EncArgs &e, uint32_t seg, uint32_t ch, const uint8_t *src, uint64_t n,
                                               uint32_t *__restrict__ out, typename RawPiece<PK>::type (&v)[kWin], const uint2 *lut2,
                                               const uint2 *lut1, uint32_t *buf, uint32_t cap, int lane,
                                               uint64_t &seg_bits, uint64_t *seg_words_out = nullptr)
{
    uint32_t pend = 0;  // words waiting in LDS behind `out` (the next unflushed word)
    const uint32_t nfull = (uint32_t)(n / kChunk);
    const uint32_t rem = (uint32_t)(n % kChunk);
    uint64_t words = 0, bits = 0;
    // bytes from one chunk of the channel to the next: contiguous, unless the plan says otherwise
    const size_t cstride = e.chunk_stride ? (size_t)e.chunk_stride : (size_t)(kChunk / MH_PIECE) * piece_bytes<PK>();
    if (nfull) {
        if (!PRE) load_first_rows<PK>(v, src, lane);
        uint32_t w, b;
        for (uint32_t c = 0; c + 1 < nfull; ++c) {
            encode_full_chunk<LC, PB, ABL, true, PK, PUBLISH>(v, src + (size_t)c * cstride, src + (size_t)(c + 1) * cstride, lut2, lut1, buf,
                                                     cap, out, pend, lane, w, b);
            words += w;
            bits += b;
        }
        encode_full_chunk<LC, PB, ABL, false, PK, PUBLISH>(v, src + (size_t)(nfull - 1) * cstride, src, lut2, lut1, buf, cap, out, pend,
                                                  lane, w, b);
        words += w;
        bits += b;
    }
    if (rem) {
        const uint4 r = encode_partial_chunk<LC, PB, PK>((const MH_AS_GLOBAL uint8_t *)(src + (size_t)nfull * cstride), rem,
                                                         (const MH_AS_LDS uint2 *)lut2, (const MH_AS_LDS uint2 *)lut1,
                                                         (MH_AS_LDS uint32_t *)buf, cap, (MH_AS_GLOBAL uint32_t *)out, pend, lane);
        words += r.x;
        bits += r.y;
        pend = r.z;
        out += r.w;
    }
    // (tuning builds: ABL 14 = everything but this partial-block store, 15 = also without seg_words / ch_bits)
    if ((ABL < 1 || ABL >= 5) && ABL != 14 && ABL != 15 && (uint32_t)lane < pend) out[lane] = buf[lane];  // segment tail (partial block)
    if (PUBLISH && lane == 0 && ABL != 15) {
        e.seg_words[seg] = words;
        if (!PRE || e.cal_mode == 0) atomicAdd(&e.ch_bits[ch], (unsigned long long)bits);  // zeroed by k_calibrate
    }
    if (!PUBLISH) *seg_words_out = words;
    seg_bits = bits;
}

// Long channels: a workgroup owns up to 4 consecutive segments OF ONE CHANNEL and shares its tables.
// Byte offset of sample t (a multiple of 16) in a channel stream of packing PK
template <int PK>
__device__ __forceinline__ uint64_t stream_bytes(uint64_t t) { return PK == 0 ? t : (t >> 4) * piece_bytes<PK>(); }

template <int LC, int PB, int ABL = 0, int PK = 0>
__global__ __launch_bounds__(256, 4) void k_encode2(Enc2Args a)
{
    // the workgroup's tables are STATIC shared memory: their addresses are compile-time constants that go into
    // the offset field of the ds_read, so a pre-scaled table index is the instruction's address operand as it is
    __shared__ __attribute__((aligned(16))) uint2 s_tab[kEncSharedDw / 2];
    __shared__ uint64_t s_pub[8];  // what the four waves report: words[4], bits[4]
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];  // the four waves' staging buffers
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t task = blockIdx.x;
    // ONE scalar load tells the workgroup everything (no task -> segment -> channel -> offsets chain); what the
    // tables need (the channel's 16 single-symbol entries) is requested next, then -- without waiting -- the
    // segment's first rows: two memory round trips from launch to the first codeword instead of five.
    const WgTask t = a.t.wg[task];
    const uint32_t nseg = t.nseg, ch = t.ch;
    uint2 *lut2 = s_tab;
    uint2 *lut1 = s_tab + 256;
    const uint2 *g = a.e.lut + (size_t)ch * kLut;
    const uint32_t b0 = threadIdx.x & ((1u << PB) - 1u), b1 = (threadIdx.x >> PB) & ((1u << PB) - 1u);
    uint2 q0, q1, q2, q3;  // the thread's table inputs (unconditional loads from valid addresses)
    if (PK == 2) {
        q0 = g[threadIdx.x & 3u], q1 = g[(threadIdx.x >> 2) & 3u], q2 = g[(threadIdx.x >> 4) & 3u], q3 = g[(threadIdx.x >> 6) & 3u];
    } else {
        q0 = g[b0], q1 = g[b1];
        q2 = q3 = make_uint2(0u, 0u);
    }
    const uint2 q1s = g[threadIdx.x & (kLut - 1)];
    // the wave's segment by arithmetic; a wave without one (wave >= nseg) keeps valid addresses and n = 0
    const uint32_t wv = (uint32_t)wave < nseg ? (uint32_t)wave : 0u;
    const uint32_t n = (uint32_t)wave + 1 < nseg ? a.t.seg_samples : ((uint32_t)wave + 1 == nseg ? t.n_last : 0u);
    const uint8_t *src = a.e.data + t.src_off + (uint64_t)wv * a.t.seg_src_stride;
    typename RawPiece<PK>::type v[kWin];
    {   // first rows: unconditional (a load inside a branch is waited for with vmcnt(0) at the join)
        const int frow = n >= (uint32_t)kChunk ? 1 : 0;
        // (a shorter segment will not use them and may end within 16 bytes of the caller's buffer: its lanes all
        // read the first 16 bytes of the task record instead -- memory that is certainly there)
        const uint8_t *first = frow ? src : reinterpret_cast<const uint8_t *>(a.t.wg + task);
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            v[k] = load_row<PK>(first + (size_t)frow * (((uint32_t)k * kLanes + lane) * piece_bytes<PK>()));
            asm volatile("" ::: "memory");  // keep the rows in issue order
        }
    }
    // pair table: entry for symbols (b0, b1) = code(b0) followed by code(b1)
    if (PK == 2) {  // four-symbol table indexed by the packed byte
        const uint32_t l01 = q0.y + q1.y, l012 = l01 + q2.y;
        lut2[packed_index_word<2>(threadIdx.x) & 0xFFu] =
            make_uint2(q0.x | (q1.x << q0.y) | (q2.x << l01) | (q3.x << l012), l012 + q3.y);
    } else if (threadIdx.x < (1u << (2 * PB))) {
        uint32_t idx = b0 | (b1 << PB);
        if (PB == 4) idx ^= (idx >> 3) & 0x1Fu;
        lut2[idx] = make_uint2(q0.x | (q1.x << q0.y), q0.y + q1.y);
    }
    if (threadIdx.x < kLut) lut1[threadIdx.x] = q1s;
    __syncthreads();
    const uint32_t cap = a.e.stage_dw;
    uint32_t *buf = smem + (size_t)wave * enc2_wave_dwords(cap);
    uint64_t bits = 0, words = 0;
    // (packed input exists for whole-channel windows only: w0 = 0 and segments start at chunk boundaries)
    if (n)
        encode_segment<LC, PB, ABL, true, PK, false>(a.e, t.seg0 + (uint32_t)wave, ch, src, n,
                                                     a.e.payload + t.dst_off + (uint64_t)wave * a.t.slot_full, v, lut2, lut1, buf,
                                                     cap, lane, bits, &words);
    // the task's results leave together: word counts of its <= 4 segments in one store, ONE atomic for the channel
    if (lane == 0) {
        s_pub[wave] = words;
        s_pub[4 + wave] = bits;
    }
    __syncthreads();
    if (ABL == 15) return;
    if (threadIdx.x < nseg) a.e.seg_words[t.seg0 + threadIdx.x] = s_pub[threadIdx.x];
    if (threadIdx.x == 0)
        atomicAdd(&a.e.ch_bits[ch], (unsigned long long)(s_pub[4] + s_pub[5] + s_pub[6] + s_pub[7]));  // zeroed by k_calibrate / k_lut_preset
}

// Short channels (the reference's real recordings at 50 ms bins are 2e4-7e4 samples per channel,
// Data/get_all_binned_data.py:16): one WAVE per segment, any channel, with the wave's own tables,
// so a workgroup packs segments of four different channels and no wave idles.  One 32-byte record
// per wave task (the planner orders them longest first) replaces the task -> segment -> channel
// chain of dependent loads, and the segment's first rows are requested before the tables are built.
// (a wave's tables: the pair table has 4^PB entries -- 64 for S <= 8 -- plus the 16 single-symbol entries; the
// four-symbol table of 2-bit input has 256.  Sized exactly: with 3-bit pairs four workgroups fit a CU at the
// largest staging as well.)
template <int PB, int PK>
__host__ __device__ constexpr uint32_t enc2w_table_dwords() { return (PK == 2 ? 512u : 2u << (2 * PB)) + 32u; }
template <int PB, int PK>
__host__ __device__ inline uint32_t enc2w_wave_dwords(uint32_t stage_dw) { return enc2w_table_dwords<PB, PK>() + enc2_wave_dwords(stage_dw); }

template <int LC, int PB, int PK = 0>
__global__ __launch_bounds__(256, 4) void k_encode2w(Enc2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 4 + (uint32_t)wave;
    if (slot >= a.t.ntask) return;
    const WaveTask t = a.t.wt[slot];
    const uint32_t cap = a.e.stage_dw;
    constexpr uint32_t kTabDw = enc2w_table_dwords<PB, PK>();
    uint32_t *wbase = smem + (size_t)wave * enc2w_wave_dwords<PB, PK>(cap);
    uint2 *lut2 = reinterpret_cast<uint2 *>(wbase);
    uint2 *lut1 = reinterpret_cast<uint2 *>(wbase + kTabDw - 32);
    const uint8_t *src = a.e.data + t.src_off;  // the planner's records hold byte offsets of the plan's input packing
    typename RawPiece<PK>::type v[kWin];
    uint2 el = make_uint2(0u, 0u);  // the 16 single-symbol entries {bit-reversed code, length}, one per lane
    // The segment's first rows are requested before the tables are built -- but AFTER the few small loads the
    // tables depend on: the memory counter retires in order, so those are waited for with the rows still in
    // flight.  The row loads are unconditional for the same reason (loads inside a branch cannot be counted,
    // and every wait after them would drain the queue).
    const bool full = t.n >= (uint32_t)kChunk;
    // (a record shorter than a chunk will not use the rows and may end within 16 bytes of the caller's buffer:
    // its lanes all read the first 16 bytes of the record itself -- memory that is certainly there)
    const uint8_t *first = full ? src : reinterpret_cast<const uint8_t *>(a.t.wt + slot);
    const int frow = full ? 1 : 0;
    auto first_rows = [&]() {
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            v[k] = load_row<PK>(first + (size_t)frow * (((uint32_t)k * kLanes + lane) * piece_bytes<PK>()));
            asm volatile("" ::: "memory");  // keep the rows in issue order
        }
    };
    if (a.e.cal_mode == 0) {
        if (lane < kLut) el = a.e.lut[(size_t)t.ch * kLut + lane];
        first_rows();
    } else {
        const int S = (int)a.e.S;
        const uint32_t K = a.e.K;
        // <= 4 encoders: the whole code table is one dword per lane, fetched before anything is known
        const bool codes_in_regs = K * 16u <= 64u;
        const uint32_t pre = a.e.codes[(uint32_t)lane < K * 16u ? (uint32_t)lane : 0u];  // (unconditional: see first_rows)
        // calibrating (mode 1) and preset (mode 2) records issue the SAME loads -- a branch between two sets of
        // loads makes the compiler wait at the join -- from addresses that are valid in either mode; the mode
        // decides afterwards which results are used
        const bool preset = a.e.cal_mode != 1;
        const uint8_t *cal = a.e.data + (preset ? (t.src_off & ~(uint64_t)15) : t.cal_off);
        const uint32_t cal_n = preset ? 1u : t.cal_n;
        const uint8_t *pk = preset ? a.e.peak_in + t.ch : cal, *en = preset ? a.e.enc_in + t.ch : cal;
        const CalLoads cl = wave_calibrate_issue(cal, cal_n, K, a.e.sclv16, lane);
        const int pi = *pk;
        const uint32_t ki = *en;
        first_rows();
        int p;
        uint32_t k;
        if (!preset) {
            wave_calibrate_finish(cl, cal, cal_n, S, a.e.mode, K, a.e.sclv, lane, p, k);
        } else {  // preset word; out-of-range values decode as 0 like k_lut_preset
            p = pi < S ? pi : 0;
            k = ki < K ? ki : 0;
        }
        {
            const int sym = lane > S - 1 ? S - 1 : lane;
            const uint32_t at = k * 16 + (uint32_t)rank_of_symbol((int)a.e.mode, S, p, sym & 15);
            uint32_t e = __shfl(pre, (int)(at & 63u), 64);
            if (!codes_in_regs && lane < kLut) e = a.e.codes[at];
            if (lane < kLut) el = make_uint2(e & 0xFFFFu, e >> 16);
        }
        if ((t.flags & 1u) && lane == 0) {  // the channel's first record publishes its word
            if (a.e.peak_out) a.e.peak_out[t.ch] = (uint8_t)p;
            if (a.e.enc_out) a.e.enc_out[t.ch] = (uint8_t)k;
            if (a.e.skip_out) a.e.skip_out[t.ch] = (uint8_t)((t.flags >> 1) & 1u);
        }
    }
    if (PK == 2) {
#pragma unroll
        for (uint32_t t0 = 0; t0 < 256; t0 += 64) {
            const uint32_t tt = t0 + (uint32_t)lane;
            lut2[packed_index_word<2>(tt) & 0xFFu] =
                quad_entry(tt, [&](uint32_t b) { return make_uint2(__shfl(el.x, (int)b, 64), __shfl(el.y, (int)b, 64)); });
        }
        if (lane < kLut) lut1[lane] = el;
    } else {
        constexpr uint32_t m = (1u << PB) - 1u;
#pragma unroll
        for (uint32_t t0 = 0; t0 < (1u << (2 * PB)); t0 += 64) {
            const uint32_t tt = t0 + (uint32_t)lane, b0 = tt & m, b1 = (tt >> PB) & m;
            const uint32_t ax = __shfl(el.x, (int)b0, 64), ay = __shfl(el.y, (int)b0, 64);
            const uint32_t bx = __shfl(el.x, (int)b1, 64), by = __shfl(el.y, (int)b1, 64);
            uint32_t idx = b0 | (b1 << PB);
            if (PB == 4) idx ^= (idx >> 3) & 0x1Fu;
            lut2[idx] = make_uint2(ax | (bx << ay), ay + by);
        }
        if (lane < kLut) lut1[lane] = el;
    }
    MH_WAVE_SYNC();
    uint64_t bits = 0;
    if (t.n)
        encode_segment<LC, PB, 0, true, PK>(a.e, t.seg, t.ch, src, t.n, a.e.payload + t.dst_off, v, lut2, lut1,
                                            wbase + kTabDw, cap, lane, bits);
    if (a.e.cal_mode != 0 && lane == 0) {
        // Bit total of the channel without a zeroing launch: every record adds {bits << 24 | 1} to the
        // channel's word in plan scratch with ONE returning device-scope atomic.  The record that sees
        // nseg_ch - 1 earlier tickets in the returned value is the last one: the returned total plus its
        // own bits is the channel's total; it stores it and leaves the scratch zero for the next launch.
        // (The planner only enables this when ticket count and bit total fit their 24 / 40 bits:
        // PlanHost::tickets_fit, checked at both edges by tests/planner_check.cpp.)
        // Ordering: there is nothing to order.  Count AND payload travel in the same 64-bit agent-scope RMW, and the
        // operations on acc[ch] form one total order: the value a record gets back is exactly the sum of the
        // records before it in that order, whichever XCDs they ran on.  The final ch_bits store is a plain store by
        // one wave, visible at the end of the kernel like every other output; the atomicExch re-arms the word.
        const unsigned long long before = atomicAdd(&a.e.acc[t.ch], ((unsigned long long)bits << 24) | 1ull);
        if ((uint32_t)(before & 0xFFFFFFull) + 1u == t.nseg_ch) {
            a.e.ch_bits[t.ch] = (before >> 24) + bits;
            atomicExch(&a.e.acc[t.ch], 0ull);
        }
    }
}

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------
#ifdef MH_TUNING
__device__ int d_dec_abl;  // see decode_staged_chunk
#endif

struct Dec2Args {
    DecArgs d;
    TaskArgs t;
    uint32_t W;  // table index bits
    // per-wave-table kernel: builds its tables itself from the (peak, encoder) word
    const uint8_t *peak, *enc;
    const uint32_t *codes;
    uint32_t S, mode, nK;
    uint32_t plan_slots;  // 1: segments sit in the plan's slots (offset in the task record)
};

// LDS dwords of the workgroup-shared tables: multi-symbol table (2 dwords per entry for K = 4,
// 1 for the pair table) + the 512-byte per-symbol table used by partial / oversize chunks
__host__ __device__ inline uint32_t dec2_shared_dwords(uint32_t W, uint32_t K)
{
    return (K == 1 ? 0u : (K == 4 ? 2u : 1u) << W) + kDtab / 4;  // K = 1: the per-symbol table is all there is
}

// staging dwords per wave: whole 16-byte-per-lane vectors (1 KiB each) covering NR * 64 words
__host__ __device__ inline uint32_t dec2_stage_dwords(uint32_t NR) { return ((NR + 3) / 4) * 256; }

// Per-chunk pipeline state: the scanned header of the chunk about to be decoded.
struct ChunkHdr {
    uint32_t P;   // this lane's sub-stream starts at bit P of the chunk payload
    uint32_t nw;  // payload words of the chunk
    uint32_t hw;  // header words of the chunk (1..25)
};

// hw32: word (lane & 31) of the chunk (a full chunk is longer than 32 words)
// DPP: the prefix sum of the 64 sub-stream lengths as a DPP ladder (mh_device.hpp): the wave-task decoders (2400 x 72 000
// S = 3 decode 43.3 -> 41.2 us) and the hybrid long-channel ones (S = 8 -1.5..3 % on 1024 ch x 1e7 bins).  The long-channel
// S <= 6 decoders keep the ds_bpermute form: with DPP the S = 3 decoder's best runs improve (1.92 ms) but it turns
// bimodal -- 1.93 or 2.0-2.05 ms from one process to the next on the same box, where the bpermute form holds
// 1.94-1.95 ms in every run (tools/bench_with_lib.py, profiles/r03_dpp_reductions.txt (11)).
template <bool DPP = false>
__device__ __forceinline__ ChunkHdr scan_header(uint32_t hw32, int lane)
{
    uint32_t wid;
    const uint32_t len = hdr_len_from_wave(hw32, lane, wid);
    const uint32_t incl = DPP ? wave_scan_incl_dpp(len) : wave_scan_incl(len, lane);
    ChunkHdr h;
    h.P = incl - len;
    // chunk sizes are wave-uniform: keep them in scalar registers (the stream pointers and the
    // bounds checks derived from them then cost no vector registers)
    h.nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((DPP ? wave_last(incl) : __shfl(incl, 63, 64)) + 31) >> 5));
    h.hw = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr_words(wid));
    return h;
}

// Decodes one full chunk whose whole payload (+3 words of read-ahead) sits in LDS `stage`.
// K symbols per table lookup (4 or 2), bit buffer topped up every M lookups.
// The loop issues no global load, so nothing in it waits on the vector-memory counter (which
// also counts the 16-byte output stores).
// HY (hybrid pair table, maxlen > W/2): an entry whose second codeword does not fit in the W
// index bits has bit 31 set and carries one symbol; the second comes from the per-symbol table
// `tab1` in a branch that whole waves skip (long codewords belong to rare symbols).
// PARTIAL: the chunk holds m < 16384 samples.  Rows past the last sample are skipped (wave-uniform),
// lanes whose piece is not complete skip the row, and the one cut piece (m % 16 samples) is decoded
// symbol by symbol from the same window.
// ST: how a decoded row leaves.  0 = global store (the long-channel kernels and the one-symbol decoder); 1 / 2 =
// through a buffer resource on the chunk's first output byte with the default / the nt cache policy (no 64-bit vector
// address per row).  The wave-task decoders of S <= 6 use 1: 2400 x 72 000 decode S = 3 41 -> 35.5 us, S = 5 47.7 -> 45 us;
// 10 000 x 20 000 S = 3 55 -> 40.6 us, S = 5 60 -> 53.5 us.  The long-channel kernels LOSE with either buffer form (S = 3:
// 1.95 -> 2.25 ms with the default policy, 2.0-2.1 with nt): a decoder that gets its rows out faster writes worse on
// this part (cf. the occupancy cap); the one-symbol decoder loses 4 % (profiles/r03_dpp_reductions.txt).
// (The source asks for non-temporal global stores; all but one of a kernel's 33 lose that metadata in an IR pass --
// seen in the ISA -- so "global store" means the default policy.)
template <int K, int M, int RL, bool HY, bool PARTIAL = false, int ST = 0>
__device__ __forceinline__ void decode_staged_chunk(ChunkHdr h, const uint32_t *tabw, uint32_t tbase, uint32_t maskW,
                                                    const uint8_t *tab1, uint32_t mask1,
                                                    const uint32_t *stage, uint8_t *__restrict__ out,
                                                    int lane, uint32_t m = kChunk)
{
    // Bit window: 64 bits starting at word `wi` of the staged payload, `bp` bits already used.
    // RL (reload): every M lookups the window is simply RE-READ from LDS at the lane's absolute bit
    //   position -- branch-free; a conditional refill runs for the whole wave almost every step
    //   because some lane always needs one (S=3 decode 2.29 -> 2.11 ms).
    // RL == 0: the window is shifted and topped up from a one-word read-ahead when 32 bits are
    //   used up; better where the extra LDS read in the dependent chain costs more than the
    //   divergent branch (chosen per variant by measurement, see dispatch_decode).
    // RL == 2: the same top-up written with selects instead of a branch (the read-ahead word is
    //   re-read every time, off the dependent chain).
    // Window bound (!RL, HY, M = 2, W <= 12, maxlen <= 9): bp < 32 after a top-up; a pair lookup
    //   needs bp + SH + W <= 64 and advances <= W, the flagged branch tops up first and advances
    //   <= 9, so bp stays below 56 and one top-up always restores bp < 32.
    static_assert(!HY || (K == 2 && RL != 1), "hybrid entries exist for the pair table only");
    static_assert(K != 1 || RL != 1, "the one-symbol decoder tops its window up (RL 0 or 2)");
    // Pre-scaled index: the window is kept SH bits "early" (bp = stream position - SH, counted
    //   from the word before the staged payload), so (window >> bp) & (mask << SH) is already the
    //   byte offset of the table entry -- no shift in the lookup's dependent chain.  The table
    //   sits at LDS byte address `tbase` (0 in the shared-table kernel, where the offset IS the
    //   address; the per-wave-table kernel pays one add).  Pair tables only: with K = 4 the
    //   last of the 4 lookups between reloads may start at window bit 55 and needs 8 more, which
    //   leaves no room for scale bits.
    constexpr bool kReload = RL == 1;
    constexpr uint32_t SH = K == 2 ? 2 : 0;  // log2(bytes per pair-table entry); K = 1 and K = 4 index unscaled
    typedef const __attribute__((address_space(3))) uint32_t lds_u1;
    const uint32_t maskS = maskW << SH;
    stage -= 1;
    uint32_t pos = h.P + 32 - SH, wi = pos >> 5, bp = pos & 31;
    uint64_t buf = (uint64_t)stage[wi] | ((uint64_t)stage[wi + 1] << 32);
    uint32_t nxt = kReload ? 0u : stage[wi + 2];
    const uint32_t nfp = m >> 4;                   // complete pieces (PARTIAL)
    const uint32_t nrows_any = (m + 1023u) >> 10;  // rows holding any sample (PARTIAL)
#ifdef MH_TUNING
    const int dec_abl = __builtin_amdgcn_readfirstlane(d_dec_abl);  // once per chunk: the row stores below may alias it
#endif
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);  // (ST != 0)
    auto row = [&](int k) {
        const uint32_t piece = (uint32_t)k * kLanes + lane;
        if (PARTIAL && piece >= nfp) {
            if (piece == nfp && (m & 15u)) {  // the cut piece: m % 16 symbols, one lane of the chunk
                uint8_t *q = out + piece * MH_PIECE;
                for (uint32_t i = 0; i < (m & 15u); ++i) {
                    const uint32_t e1 = tab1[(uint32_t)(buf >> (bp + SH)) & mask1];
                    q[i] = (uint8_t)(e1 & 15u);
                    bp += e1 >> 4;
                    if (kReload) {
                        pos += e1 >> 4;
                        const uint32_t w_ = pos >> 5;
                        buf = (uint64_t)stage[w_] | ((uint64_t)stage[w_ + 1] << 32);
                        bp = pos & 31;
                    } else if (bp >= 32) {
                        buf = (buf >> 32) | ((uint64_t)nxt << 32);
                        bp -= 32;
                        ++wi;
                        nxt = stage[wi + 2];
                    }
                }
            }
            return;
        }
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 4 / K; ++i) {
                const uint32_t off = (uint32_t)(buf >> bp) & maskS;
                uint32_t adv;
                if (K == 1) {  // one symbol per lookup from the 2^maxlen-entry byte table: no flagged entries, no branch
                    const uint32_t e1 = tab1[(uint32_t)(buf >> bp) & mask1];
                    w |= (e1 & 15u) << (8 * i);
                    adv = e1 >> 4;
                } else if (K == 4) {
                    const uint2 e = reinterpret_cast<const uint2 *>(tabw)[off];
                    w = e.x;
                    adv = e.y;
                } else {
                    const uint32_t e = *reinterpret_cast<lds_u1 *>(off + tbase);
                    w |= (e & 0xFFFFu) << (8 * K * i);
                    adv = HY ? (e >> 16) & 0x7FFFu : e >> 16;
#ifdef MH_HY_NEVER  // timing-only A/B build: what the decoders would cost if no entry were ever flagged (wrong output)
                    if (false) {
#else
                    if (HY && (int32_t)e < 0) {  // rare: second codeword reaches past the index bits
#endif
                        bp += adv;
                        if (bp >= 32) {
                            buf = (buf >> 32) | ((uint64_t)nxt << 32);
                            bp -= 32;
                            ++wi;
                            nxt = stage[wi + 2];
                        }
                        const uint32_t e1 = tab1[(uint32_t)(buf >> (bp + SH)) & mask1];
                        w |= (e1 & 15u) << (8 * K * i + 8);
                        adv = e1 >> 4;
                    }
                }
                bp += adv;
                if (kReload) pos += adv;
                if ((d * (4 / K) + i + 1) % M == 0) {
                    if (kReload) {
                        const uint32_t w_ = pos >> 5;
                        buf = (uint64_t)stage[w_] | ((uint64_t)stage[w_ + 1] << 32);
                        bp = pos & 31;
                    } else if (RL == 2) {
                        const bool t = bp >= 32;
                        const uint32_t lo = t ? (uint32_t)(buf >> 32) : (uint32_t)buf;
                        const uint32_t hi = t ? nxt : (uint32_t)(buf >> 32);
                        buf = (uint64_t)lo | ((uint64_t)hi << 32);
                        bp &= 31;
                        wi += t ? 1u : 0u;
                        nxt = stage[wi + 2];
                    } else if (bp >= 32) {
                        buf = (buf >> 32) | ((uint64_t)nxt << 32);
                        bp -= 32;
                        ++wi;
                        nxt = stage[wi + 2];
                    }
                }
            }
            o[d] = w;
        }
#ifdef MH_TUNING  // timing-only ablation (tools/ablate_decode.py): 1 = every row of a chunk lands on its first KiB
        const uint32_t krow = dec_abl == 1 ? 0u : (uint32_t)k;  // (1/16 of the DRAM writes), 2 = no row stores
        if (dec_abl == 2) {
            if ((o.x ^ o.y) == 0x12345678u && o.z == 77u) out[0] = 1;
            return;
        }
#else
        const uint32_t krow = (uint32_t)k;
#endif
#ifdef MH_TUNING
        if (dec_abl == 3) {  // plain instead of non-temporal row stores
            *reinterpret_cast<u32x4_u *>(out + (krow * kLanes + lane) * MH_PIECE) = o;
            return;
        }
#endif
        if (!PARTIAL && ST != 0) {
            // (the row offset is part of the VECTOR offset -- the compiler folds what fits into the immediate: with a scalar-
            // register offset hipcc 7.2 leaves no wait state between this store and a VALU write of its data
            // registers -- GCNHazardRecognizer assumes that form has no such hazard -- and gfx950 then stores the
            // new value in lanes 12..15 of every 16: seen as wrong symbols in row 14 of a chunk)
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, lane * MH_PIECE + (int)krow * kLanes * MH_PIECE, 0, ST == 2 ? 2 : 0);
            return;
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4_u *>(out + (krow * kLanes + lane) * MH_PIECE));
    };
    if (PARTIAL) {  // a rolled loop keeps the rarely run instance small (registers and code)
#pragma unroll 1
        for (int k = 0; k < (int)nrows_any; ++k) row(k);
    } else {
#pragma unroll
        for (int k = 0; k < kRows; ++k) row(k);
    }
}

// TWO full chunks of one segment decoded side by side by the ONE-SYMBOL decoder (K = 1).  With long codes the symbol
// loop is a dependent chain per lane -- index from the window, LDS lookup, advance -- whose round trip nothing in the
// lane can hide; the hybrid pair table halves the chain only on paper, because SOME lane of the wave meets a flagged
// entry in 50-85 % of the lookups and the whole wave then runs the second, dependent lookup (profiles/
// r03_hybrid_flag_upper_bound.txt).  The one-symbol table has no flagged entries and no branch in its loop, so two
// independent chains -- the same lane's sub-streams in two chunks -- interleave perfectly: their lookups are issued back
// to back and each LDS round trip serves both.  (The same pairing around the hybrid loop was slower: its branches cut
// the two chains into separate basic blocks, r03_pair_decoding_ab.txt.)
template <int M, int RL, int ST = 0>
__device__ __forceinline__ void decode_staged_pair1(ChunkHdr hA, ChunkHdr hB, const uint8_t *tab1, uint32_t mask1,
                                                    const uint32_t *stageA, const uint32_t *stageB,
                                                    uint8_t *__restrict__ outA, uint8_t *__restrict__ outB, int lane)
{
    static_assert(RL != 1, "top-up window (RL 0 or 2)");
    struct Chain {
        const uint32_t *stage;
        uint64_t buf;
        uint32_t wi, bp, nxt;
    } A, B;
    auto start = [&](Chain &c, const uint32_t *stage, uint32_t P) {
        c.stage = stage - 1;  // (bit positions count from the word before the staged payload, as in decode_staged_chunk)
        const uint32_t pos = P + 32;
        c.wi = pos >> 5;
        c.bp = pos & 31;
        c.buf = (uint64_t)c.stage[c.wi] | ((uint64_t)c.stage[c.wi + 1] << 32);
        c.nxt = c.stage[c.wi + 2];
    };
    start(A, stageA, hA.P);
    start(B, stageB, hB.P);
    const auto rs_A = __builtin_amdgcn_make_buffer_rsrc(outA, 0, 0x7FFFFFFF, 0x00020000);  // (ST != 0: see decode_staged_chunk)
    const auto rs_B = __builtin_amdgcn_make_buffer_rsrc(outB, 0, 0x7FFFFFFF, 0x00020000);
    auto top_up = [&](Chain &c) {
        if (RL == 2) {
            const bool t = c.bp >= 32;
            const uint32_t lo = t ? (uint32_t)(c.buf >> 32) : (uint32_t)c.buf;
            const uint32_t hi = t ? c.nxt : (uint32_t)(c.buf >> 32);
            c.buf = (uint64_t)lo | ((uint64_t)hi << 32);
            c.bp &= 31;
            c.wi += t ? 1u : 0u;
            c.nxt = c.stage[c.wi + 2];
        } else if (c.bp >= 32) {
            c.buf = (c.buf >> 32) | ((uint64_t)c.nxt << 32);
            c.bp -= 32;
            ++c.wi;
            c.nxt = c.stage[c.wi + 2];
        }
    };
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        u32x4 oA, oB;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t wA = 0, wB = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t eA = tab1[(uint32_t)(A.buf >> A.bp) & mask1];  // both lookups in flight together
                const uint32_t eB = tab1[(uint32_t)(B.buf >> B.bp) & mask1];
                wA |= (eA & 15u) << (8 * i);
                wB |= (eB & 15u) << (8 * i);
                A.bp += eA >> 4;
                B.bp += eB >> 4;
                if ((d * 4 + i + 1) % M == 0) {
                    top_up(A);
                    top_up(B);
                }
            }
            oA[d] = wA;
            oB[d] = wB;
        }
        if (ST != 0) {
            __builtin_amdgcn_raw_buffer_store_b128(oA, rs_A, lane * MH_PIECE + k * kLanes * MH_PIECE, 0, ST == 2 ? 2 : 0);
            __builtin_amdgcn_raw_buffer_store_b128(oB, rs_B, lane * MH_PIECE + k * kLanes * MH_PIECE, 0, ST == 2 ? 2 : 0);
        } else {
            __builtin_nontemporal_store(oA, reinterpret_cast<u32x4_u *>(outA + ((uint32_t)k * kLanes + lane) * MH_PIECE));
            __builtin_nontemporal_store(oB, reinterpret_cast<u32x4_u *>(outB + ((uint32_t)k * kLanes + lane) * MH_PIECE));
        }
    }
}

// Last, partial chunk of a segment through the same staged machinery: header (two dependent
// reads of this chunk's own words), the whole payload copied to LDS in one batch of loads, then
// decode_staged_chunk<PARTIAL>.  A chunk too large for the staging area takes the per-symbol
// routine on global memory.
template <int K, int M, int RL, bool HY>
__device__ __forceinline__ void decode_partial_chunk(const uint32_t *__restrict__ in, uint32_t m, const uint32_t *tabw,
                                                  uint32_t tbase, uint32_t maskW, const uint8_t *tab1, uint32_t mask1,
                                                  uint32_t *stage, uint32_t cap_words, uint8_t *__restrict__ out,
                                                  int lane, uint64_t avail, uint32_t *err, uint32_t epoch)
{
    // avail = words readable from `in` on; the chunk's header, payload and 3 words of read-ahead
    // must lie inside, else the chunk is abandoned (corrupt or truncated stream)
    if (avail < 1) { if (lane == 0) atomicMax(err, epoch); return; }
    const uint32_t w0 = in[0];
    const uint32_t mn = w0 & 0xFFFu, hwid = (w0 >> 12) & 15u;
    if (avail < hdr_words(hwid)) { if (lane == 0) atomicMax(err, epoch); return; }
    uint32_t len = mn;
    if (hwid) {
        const uint32_t fb = 16u + (uint32_t)lane * hwid;
        uint64_t v = in[fb >> 5];
        if ((fb & 31) + hwid > 32) v |= (uint64_t)in[(fb >> 5) + 1] << 32;
        len += (uint32_t)(v >> (fb & 31)) & ((1u << hwid) - 1u);
    }
    const uint32_t incl = wave_scan_incl(len, lane);
    ChunkHdr h;
    h.P = incl - len;
    h.nw = (__shfl(incl, 63, 64) + 31) >> 5;
    h.hw = hdr_words(hwid);
    const uint32_t ns = h.nw + 3;
    if (avail < (uint64_t)h.hw + ns) { if (lane == 0) atomicMax(err, epoch); return; }
    if (ns > cap_words) {
        decode_chunk<3, false>(in, m, tab1, mask1, out, lane);
        return;
    }
    const uint32_t *pay = in + h.hw;
    for (uint32_t j0 = 0; j0 < ns; j0 += 8 * 64) {
        uint32_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = j0 + (uint32_t)j * 64 + lane;
            if (j0 + (uint32_t)j * 64 < ns) r[j] = pay[i < ns ? i : ns - 1];  // never past nw + 2
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j0 + (uint32_t)j * 64 < ns) stage[j0 + (uint32_t)j * 64 + lane] = r[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    decode_staged_chunk<K, M, RL, HY, true>(h, tabw, tbase, maskW, tab1, mask1, stage, out, lane, m);
}

// One segment, one wave: the chunks of segment `seg` through the wave's tables (`tab` multi-symbol
// table at LDS byte address `tbase`, `tab1` per-symbol table) and its payload staging area.
//
// Pipeline of the full chunks.  While chunk c decodes out of LDS, the payload of chunk c+1 (NV
// 16-byte vectors per lane) and the first 32 words of chunk c+2 (its header) are on their way into
// registers; they are moved to LDS at the END of the iteration, behind chunk c's 16 output stores.
// Two rules keep the wave from draining those stores at every chunk (gfx9 retires loads and stores
// through ONE in-order counter, and the compiler's wait-count pass only counts operations that are
// certain to have been issued on every path to the wait):
//   * every global load of the loop is unconditional, its index clamped -- never branched around;
//   * the loads are consumed in the same straight-line stretch that issued the stores, so the
//     wait in front of the register -> LDS copy is vmcnt(16 + k), not vmcnt(0).  (Consuming them at
//     the top of the next iteration joins the loop entry -- no stores yet -- with the back edge, and
//     the join has to assume the worst: vmcnt(0).)
// The oversize-chunk slow path has loads of its own and therefore sits OUTSIDE the loop: the
// first chunk that does not fit the staging area ends it and the rest of the segment goes chunk
// by chunk through decode_chunk (adversarial data only).
template <int K, int M, int NR, int RL, bool HY, bool WT = false>
__device__ __forceinline__ void decode_segment(const DecArgs &d, uint64_t pos, uint8_t *__restrict__ out, uint64_t n,
                                               const uint32_t *tab, uint32_t tbase, uint32_t maskW, const uint8_t *tab1,
                                               uint32_t mask1, uint32_t *stage, int lane)
{
    constexpr bool kHdrDpp = WT || HY;  // see scan_header
    constexpr int kSt = WT && K != 1 ? 1 : 0;  // see decode_staged_chunk
    constexpr uint32_t kCap = NR * 64;         // words of payload (+3 read-ahead) a staged chunk may have
    constexpr int NV = (NR + 3) / 4;           // 16-byte vectors per lane that cover kCap words
    // Untrusted input: `pos` = word index of a chunk's first header word, `lim` = words that may be
    // read.  Before any read that a header value steers, the wave checks (wave-uniform, a few
    // scalar operations per chunk) that header + payload + 3 words of read-ahead (+ the next
    // chunk's first 32 words where there is one) lie below lim; otherwise the segment is abandoned
    // and *err raised.  Stores only ever go to the plan's own window positions.
    const uint64_t lim = d.payload_words;
    const uint32_t *in = d.payload + pos;
    const uint32_t nfull = (uint32_t)(n / kChunk);
    const uint32_t rem = (uint32_t)(n % kChunk);
#ifdef MH_DEC_NOCHECK  // A/B builds only: what the bounds checks cost
#define MH_DEC_BAIL() do { } while (0)
#else
#define MH_DEC_BAIL()                              \
    do {                                           \
        if (lane == 0) atomicMax(d.err, d.epoch);  \
        return;                                    \
    } while (0)
#endif
    // `need` words readable from word `at` on?  Subtractive, so that a wild 64-bit offset cannot wrap the sum.
    auto room = [&](uint64_t at, uint64_t need) { return at <= lim && lim - at >= need; };
    // A full chunk holds 16384 codewords of >= 1 bit: a header that announces fewer than 512 payload words is
    // corrupt.  (It is also what the prefetch relies on: fetch() and peek() clamp their indices to av - 4 resp.
    // av - 1, and re-read the chunk's first 4 words once nothing follows -- nw >= 512 keeps all of that inside.)
    constexpr uint32_t kMinFull = kChunk / 32;
    uint32_t c = 0;  // next chunk the per-symbol loop below would have to decode
    if (nfull) {
        if (!room(pos, 32)) MH_DEC_BAIL();
        ChunkHdr hc = scan_header<kHdrDpp>(in[lane & 31], lane);  // chunk whose payload is (about to be) in LDS
        if (hc.nw < kMinFull || !room(pos, (uint64_t)hc.hw + hc.nw + 3 + (nfull > 1 ? 32 : 0))) MH_DEC_BAIL();
        u32x4 R[NV];
        // NV x 1 KiB; lanes past the chunk's own av = nw + 3 words all re-read its last vector (one
        // address: no extra traffic), so the instruction count is fixed but the bytes are the chunk's
        auto fetch = [&](const uint32_t *p, uint32_t av) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                uint32_t i_ = (uint32_t)(j * 256 + lane * 4);
                i_ = i_ + 4 <= av ? i_ : av - 4;
                R[j] = *reinterpret_cast<const u32x4_u *>(p + i_);
            }
        };
        auto peek = [&](const uint32_t *p, uint32_t at, uint32_t av) {  // word (lane & 31) behind a chunk, clamped
            const uint32_t i_ = at + (uint32_t)(lane & 31);
            return p[i_ < av ? i_ : av - 1];
        };
        auto to_lds = [&]() {
#pragma unroll
            for (int j = 0; j < NV; ++j) *reinterpret_cast<u32x4 *>(stage + j * 256 + lane * 4) = R[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if (hc.nw + 3 <= kCap) {
            // ---- prologue: payload(0) -> LDS, header(1) scanned
            const uint32_t *pay_n = in + hc.hw;  // payload of the chunk `nx` describes (chunk 0 as a stand-in while nfull == 1)
            uint64_t pos_n = pos;                // that chunk's first header word
            // words the next fetch / peek may touch: the chunk's nw + 3 (+ 32 of the following header);
            // 4 once nothing follows (all lanes then re-read one vector)
            uint32_t avail_n = hc.nw + 3, peek_n = hc.nw + (nfull > 1 ? 32 : 0);
            ChunkHdr nx = hc;
            fetch(pay_n, avail_n);
            uint32_t hw_next = peek(pay_n, hc.nw, peek_n ? peek_n : 1);
            to_lds();
            // every load is consumed where it is certain to have landed, even when its value is not
            // needed: a load left pending makes the compiler guard the reuse of its register with a
            // full drain later on
            asm volatile("" ::"v"(hw_next));
            if (nfull > 1) {
                pos_n += hc.hw + hc.nw;
                nx = scan_header<kHdrDpp>(hw_next, lane);
                if (nx.nw < kMinFull || !room(pos_n, (uint64_t)nx.hw + nx.nw + 3 + (nfull > 2 ? 32 : 0))) MH_DEC_BAIL();
                pay_n += hc.nw + nx.hw;
                avail_n = nx.nw + 3;
                peek_n = nx.nw + (nfull > 2 ? 32 : 0);
            } else {
                avail_n = 4;
                peek_n = 1;
            }
            for (;;) {
                // payload(c+1) and the head of chunk c+2 (behind the last chunk: a harmless re-read), in
                // flight before this chunk's stores
                fetch(pay_n, avail_n);
                hw_next = peek(pay_n, nx.nw, peek_n);
                decode_staged_chunk<K, M, RL, HY, false, kSt>(hc, tab, tbase, maskW, tab1, mask1, stage, out + (size_t)c * kChunk, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                ++c;
                if (c == nfull) {  // `nx` / `pay_n` describe the last chunk (== hc)
                    in = pay_n + nx.nw;
                    pos = pos_n + nx.hw + nx.nw;
                    break;
                }
                if (nx.nw + 3 > kCap) {  // oversize chunk next: the per-symbol loop takes over
                    in = pay_n - nx.hw;
                    pos = pos_n;
                    break;
                }
                to_lds();  // payload(c) has landed behind the 16 stores: registers -> LDS
                asm volatile("" ::"v"(hw_next));
                hc = nx;
                if (c + 1 < nfull) {  // scalar bookkeeping: header(c+1) from hw_next
                    pos_n += nx.hw + nx.nw;
                    pay_n += nx.nw;
                    nx = scan_header<kHdrDpp>(hw_next, lane);
                    if (nx.nw < kMinFull || !room(pos_n, (uint64_t)nx.hw + nx.nw + 3 + (c + 2 < nfull ? 32 : 0))) MH_DEC_BAIL();
                    pay_n += nx.hw;
                    avail_n = nx.nw + 3;
                    peek_n = nx.nw + (c + 2 < nfull ? 32 : 0);
                } else {
                    avail_n = 4;  // nothing follows the chunk now in LDS: the next fetch is a one-vector re-read
                    peek_n = 1;
                }
            }
        }
    }
    for (; c < nfull; ++c) {  // (rest of) a segment that holds an oversize chunk: per-symbol routine
        if (!room(pos, 32)) MH_DEC_BAIL();
        const ChunkHdr h = scan_header<kHdrDpp>(in[lane & 31], lane);
        if (h.nw < kMinFull || !room(pos, (uint64_t)h.hw + h.nw + 3)) MH_DEC_BAIL();
        decode_chunk<3, true>(in, kChunk, tab1, mask1, out + (size_t)c * kChunk, lane);
        in += h.hw + h.nw;
        pos += h.hw + h.nw;
    }
    if (rem)
        decode_partial_chunk<K, M, RL, HY>(in, rem, tab, tbase, maskW, tab1, mask1, stage, kCap, out + (size_t)nfull * kChunk,
                                            lane, pos < lim ? lim - pos : 0, d.err, d.epoch);
#undef MH_DEC_BAIL
}

// One-symbol decoders: the full chunks of a segment two at a time (decode_staged_pair1).  Both headers are scanned
// first (the second one's position follows from the first), then ONE batch of loads brings payload A, header B and
// payload B -- contiguous in the stream -- into the wave's staging area, and the two chunks decode side by side.
// Whatever does not fit that scheme goes through decode_segment: a pair too large for the staging area, the odd full
// chunk, the partial chunk.  Same bounds rules: every header-steered read is checked against lim first.
template <int K, int M, int NR, int RL, bool HY, bool WT = false>
__device__ __forceinline__ void decode_segment_dual(const DecArgs &d, uint64_t pos, uint8_t *__restrict__ out, uint64_t n,
                                                    const uint32_t *tab, uint32_t tbase, uint32_t maskW, const uint8_t *tab1,
                                                    uint32_t mask1, uint32_t *stage, int lane)
{
    static_assert(K == 1 && !HY, "pairing is for the branch-free one-symbol loop");
    constexpr bool kHdrDpp = true;
    constexpr uint32_t kCap = NR * 64;
    constexpr int NV = (NR + 3) / 4;
    constexpr uint32_t kMinFull = kChunk / 32;
    const uint64_t lim = d.payload_words;
    auto room = [&](uint64_t at, uint64_t need) { return at <= lim && lim - at >= need; };
    const uint32_t nfull = (uint32_t)(n / kChunk);
    uint32_t c = 0;
    while (c + 2 <= nfull) {
        if (!room(pos, 32)) {
            if (lane == 0) atomicMax(d.err, d.epoch);
            return;
        }
        const uint32_t *in = d.payload + pos;
        const ChunkHdr hA = scan_header<kHdrDpp>(in[lane & 31], lane);
        // header A, payload A and the first 32 words of chunk B (a full chunk is longer than that) must be there
        if (hA.nw < kMinFull || !room(pos, (uint64_t)hA.hw + hA.nw + 32)) {
            if (lane == 0) atomicMax(d.err, d.epoch);
            return;
        }
        const uint32_t *payA = in + hA.hw;
        const ChunkHdr hB = scan_header<kHdrDpp>(payA[hA.nw + (uint32_t)(lane & 31)], lane);
        const uint64_t posB = pos + hA.hw + hA.nw;
        if (hB.nw < kMinFull || !room(posB, (uint64_t)hB.hw + hB.nw + 3)) {
            if (lane == 0) atomicMax(d.err, d.epoch);
            return;
        }
        const uint32_t total = hA.nw + hB.hw + hB.nw + 3;  // payload A, chunk B, 3 words of read-ahead
        if (total > kCap) break;  // (too many bits per sample in both chunks for the staging area: one at a time below)
        u32x4 R[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {  // fixed instruction count; lanes past the end re-read the last vector
            uint32_t i_ = (uint32_t)(j * 256 + lane * 4);
            i_ = i_ + 4 <= total ? i_ : total - 4;
            R[j] = *reinterpret_cast<const u32x4_u *>(payA + i_);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            uint32_t i_ = (uint32_t)(j * 256 + lane * 4);
            i_ = i_ + 4 <= total ? i_ : total - 4;
            *reinterpret_cast<u32x4_u *>(stage + i_) = R[j];  // (clamped lanes rewrite the last vector with the same values)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        decode_staged_pair1<M, RL, 0>(hA, hB, tab1, mask1, stage, stage + hA.nw + hB.hw, out + (size_t)c * kChunk,
                                   out + (size_t)(c + 1) * kChunk, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pos = posB + hB.hw + hB.nw;
        c += 2;
    }
    if ((uint64_t)c * kChunk < n)
        decode_segment<K, M, NR, RL, HY, WT>(d, pos, out + (size_t)c * kChunk, n - (uint64_t)c * kChunk, tab, tbase, maskW, tab1, mask1,
                                         stage, lane);
}

// NR payload registers per lane: the next chunk's payload (up to NR*64 words) is fetched into
// registers while the current chunk decodes, and lands in LDS at the top of the next iteration.
// All those loads are issued BEFORE the current chunk's 16 output stores, so waiting for them
// (in-order vmcnt) never waits for a store.  A chunk whose payload exceeds NR*64 words (more
// than 3 bits/sample when NR = 24; impossible when NR = 17 and maxlen <= 2) takes the
// per-symbol routine that reads the stream straight from global memory.
#ifndef MH_DEC_MIN_WAVES
#define MH_DEC_MIN_WAVES 1
#endif
// Long channels: up to 4 consecutive segments of ONE channel per workgroup, tables shared at LDS
// address 0 (the kernel has no static LDS; mh_plan_create verifies that from the code object).
// The decode tables of channel `ch` built in LDS from its (peak, encoder) word by NT cooperating threads (64: one
// wave -- the per-wave-table kernel; 256: the workgroup of the shared-table kernel): the 512-byte per-symbol table
// tab1 and the multi-symbol table tab.  t = index of the thread among the NT, lane = its lane (each wave derives
// the channel's rank list for itself by shuffles).  Returns the index mask of tab1.
template <int NT>
__device__ __forceinline__ void table_sync()
{
    if (NT == 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <int K, int NT>
__device__ __forceinline__ uint32_t build_decode_tables(const Dec2Args &a, uint32_t ch, uint32_t *tab, uint8_t *tab1, int t,
                                                        int lane)
{
    // (<= 4 encoders: the whole code table is one dword per lane, requested together with the channel's word
    // instead of after it -- one memory round trip before the tables instead of two)
    const uint32_t pre = a.codes[(uint32_t)lane < a.nK * 16u ? (uint32_t)lane : 0u];
    const int pi = a.peak[ch];
    const uint32_t ki = a.enc[ch];
    const int S = (int)a.S;
    const uint32_t W = a.W;
    const int p = pi < S ? pi : 0;
    const uint32_t k = ki < a.nK ? ki : 0;
    // rank r of the channel's code: bit-reversed code | len << 16 | symbol << 24, one rank per lane
    uint32_t code = __shfl(pre, (int)((k * 16 + (uint32_t)lane) & 63u), 64);
    if (a.nK * 16u > 64u && lane < S) code = a.codes[k * 16 + lane];
    const uint32_t mine = lane < S ? code | ((uint32_t)symbol_of_rank((int)a.mode, S, p, lane) << 24) : 0u;
    uint32_t rk[MH_LUT_SYMS];
#pragma unroll
    for (int r = 0; r < MH_LUT_SYMS; ++r) rk[r] = (uint32_t)__builtin_amdgcn_readlane((int)mine, r);  // (scalar: no LDS crossbar)
    uint32_t L = 0;  // rows are non-decreasing: the last rank has the longest code
#pragma unroll
    for (int r = 0; r < MH_LUT_SYMS; ++r)
        if (r == S - 1) L = (rk[r] >> 16) & 0xFFu;
    const uint32_t mask1 = (1u << L) - 1u;
    for (uint32_t j = (uint32_t)t; j < (1u << L); j += NT) {  // per-symbol table: symbol | len << 4
        uint32_t e = 0;
#pragma unroll
        for (int r = 0; r < MH_LUT_SYMS; ++r)
            if (r < S) {
                const uint32_t l = (rk[r] >> 16) & 0xFFu;
                if ((j & ((1u << l) - 1u)) == (rk[r] & 0xFFFFu)) e = (rk[r] >> 24) | (l << 4);
            }
        tab1[j] = (uint8_t)e;
    }
    if (K == 4) {  // the next W = 8 bits always hold 4 whole codewords -> {symbols spread to bytes, bits}
        // four lookups in the per-symbol table just built (as the pair table below) instead of matching every rank
        // against every one of the four positions: a wave that builds its own table (wave-task plans) spent 1.6 us
        // of a 7-us one-chunk record here
        table_sync<NT>();
        for (uint32_t idx = (uint32_t)t; idx < (1u << W); idx += NT) {
            uint32_t bpos = 0, bytes = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t e = tab1[(idx >> bpos) & mask1];
                bytes |= (e & 15u) << (8 * j);
                bpos += e >> 4;
            }
            reinterpret_cast<uint2 *>(tab)[idx] = make_uint2(bytes, bpos);
        }
    } else if (K == 2) {
        table_sync<NT>();
        for (uint32_t idx = (uint32_t)t; idx < (1u << W); idx += NT) {
            const uint32_t e1 = tab1[idx & mask1];
            const uint32_t l1 = e1 >> 4;
            const uint32_t e2 = tab1[(idx >> l1) & mask1];
            const uint32_t l2 = e2 >> 4;
            tab[idx] = l1 + l2 <= W ? (e1 & 15u) | ((e2 & 15u) << 8) | ((l1 + l2) << 16)
                                    : (e1 & 15u) | (l1 << 16) | 0x80000000u;
        }
    }
    table_sync<NT>();
    return mask1;
}

template <int K, int M, int NR, int RL, bool HY, bool DUAL = false>
__global__ __launch_bounds__(256, MH_DEC_MIN_WAVES) void k_decode2(Dec2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t task = blockIdx.x;
    // one record instead of the task -> segment -> channel chain; the segment's own directory entries are read
    // AFTER the table build on purpose: requesting them up front (and computing the slot by arithmetic) made this
    // kernel 3-7 % slower at S = 3..5 in three same-box alternations (profiles/r03_wgtask_ab.txt) -- the
    // encoder, which gains nothing either way, takes everything from the record
    const WgTask t = a.t.wg[task];
    const uint32_t seg0 = t.seg0, nseg = t.nseg, ch = t.ch;
    const uint32_t W = a.W;
    constexpr uint32_t kEntDw = K == 4 ? 2 : 1;  // dwords per table entry
    uint32_t *tab = smem;
    uint8_t *tab1 = reinterpret_cast<uint8_t *>(smem + (K == 1 ? 0u : kEntDw << W));
    // tables built by the workgroup itself from the channel's (peak, encoder) word: no table kernel in front of
    // the decoder, no per-channel tables in global memory
    const uint32_t mask1 = build_decode_tables<K, 256>(a, ch, tab, tab1, (int)threadIdx.x, lane);
    if ((uint32_t)wave >= nseg) return;
    uint32_t *stage = smem + dec2_shared_dwords(W, K) + (size_t)wave * dec2_stage_dwords(NR);
    const uint32_t seg = seg0 + (uint32_t)wave;
    if constexpr (DUAL)
        decode_segment_dual<K, M, NR, RL, HY>(a.d, a.d.seg_off[seg], a.d.out + a.d.ch_off[ch] + a.d.w0[ch] + a.d.seg_first[seg],
                                              a.d.seg_n[seg], tab, 0u, (1u << W) - 1u, tab1, mask1, stage, lane);
    else
        decode_segment<K, M, NR, RL, HY>(a.d, a.d.seg_off[seg], a.d.out + a.d.ch_off[ch] + a.d.w0[ch] + a.d.seg_first[seg],
                                         a.d.seg_n[seg], tab, 0u, (1u << W) - 1u, tab1, mask1, stage, lane);
}

// Short channels: one WAVE per segment of any channel, tables per wave (see k_encode2w).  The wave
// derives its tables from the channel's (peak, encoder) word and the plan's codebooks itself, so
// this decode is ONE launch: no table kernel in front of it.
template <int K, int M, int NR, int RL, bool HY, bool DUAL = false>
__global__ __launch_bounds__(256, MH_DEC_MIN_WAVES) void k_decode2w(Dec2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 4 + (uint32_t)wave;
    if (slot >= a.t.ntask) return;
    const WaveTask t = a.t.wt[slot];
    const uint32_t W = a.W;
    constexpr uint32_t kEntDw = K == 4 ? 2 : 1;
    const uint32_t wdw = dec2_shared_dwords(W, K) + dec2_stage_dwords(NR);  // dwords per wave
    uint32_t *tab = smem + (size_t)wave * wdw;
    uint8_t *tab1 = reinterpret_cast<uint8_t *>(tab + (K == 1 ? 0u : kEntDw << W));
    const uint64_t pos = a.plan_slots ? t.dst_off : a.d.seg_off[t.seg];
    // a (peak, encoder) word outside the plan's ranges (corrupt metadata) decodes as (0, 0)
    const uint32_t mask1 = build_decode_tables<K, 64>(a, t.ch, tab, tab1, lane, lane);
    const uint32_t tbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tab;
    if constexpr (DUAL)
        decode_segment_dual<K, M, NR, RL, HY, true>(a.d, pos, a.d.out + t.src_off, t.n, tab, tbase, (1u << W) - 1u, tab1, mask1,
                                              tab + dec2_shared_dwords(W, K), lane);
    else
        decode_segment<K, M, NR, RL, HY, true>(a.d, pos, a.d.out + t.src_off, t.n, tab, tbase, (1u << W) - 1u, tab1, mask1,
                                         tab + dec2_shared_dwords(W, K), lane);
}

// ------------------------------------------------------------------------------------------
// window histogram, second generation (S >= 4): one LDS lookup per TWO bytes.  The table maps
// a pair index (b0 | b1 << PB, see pair_index_word) to the sum of two one-hot 6-bit fields
// (field s = clipped symbol s) packed in 64 bits; lanes add entries into a 64-bit accumulator
// and drain it into 32-bit counters every 3 vectors (<= 48 per field < 64).  ~2.5 VALU ops
// per byte for any S, against 1.75 * (S-1) for the byte-compare kernel.
// ------------------------------------------------------------------------------------------
template <int PB>
__global__ __launch_bounds__(256) void k_hist2(HistArgs a, uint32_t S)
{
    __shared__ __attribute__((aligned(16))) unsigned long long tab[256];
    __shared__ uint32_t red[MH_LUT_SYMS][4];
    const int tid = threadIdx.x;
    // (the tile's directory entries are requested before the table is built: two dependent round trips that the
    // build and its barrier overlap)
    const uint32_t tile = blockIdx.x;
    const uint32_t ch = a.tile_ch[tile];
    const uint8_t *p = a.data + a.ch_off[ch] + a.tile_start[tile];
    const uint32_t n = a.tile_n[tile];
    {
        const uint32_t m = (1u << PB) - 1u;
        if ((uint32_t)tid < (1u << (2 * PB))) {
            uint32_t b0 = tid & m, b1 = (tid >> PB) & m;
            b0 = b0 > S - 1 ? S - 1 : b0;
            b1 = b1 > S - 1 ? S - 1 : b1;
            uint32_t idx = (tid & m) | (((tid >> PB) & m) << PB);
            if (PB == 4) idx ^= (idx >> 3) & 0x1Fu;
            tab[idx] = (1ull << (6 * b0)) + (1ull << (6 * b1));
        }
    }
    __syncthreads();
    uint32_t cnt[MH_LUT_SYMS];
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] = 0;
    unsigned long long acc = 0;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)p & 15)) & 15);
    head = head < n ? head : n;
    const uint32_t nvec = (n - head) >> 4;
    const uint32_t tail = (n - head) & 15;
    if ((uint32_t)tid < head) {
        uint32_t b = p[tid];
        b = b > S - 1 ? S - 1 : b;
        acc += 1ull << (6 * b);
    }
    if ((uint32_t)tid < tail) {
        uint32_t b = p[head + (nvec << 4) + tid];
        b = b > S - 1 ? S - 1 : b;
        acc += 1ull << (6 * b);
    }
    constexpr uint32_t kHiMask = 0x01010101u * (0xFFu & ~((1u << PB) - 1u));
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p + head);
    int pending = 0;
    // Four vectors are requested before the first is looked at: the wave-wide test for bytes that need clipping makes
    // every iteration wait for its load, and one vector per trip left a short tile (a 72 000-bin channel: 9 vectors
    // per thread) paying the memory latency nine times over -- measure, S = 4..10: 2400 x 72 000 bins 32.7-35.3 ->
    // 29.7-32.4 us, 1024 x 1e7 bins 1.69-1.77 -> 1.60-1.62 ms (profiles/r03_dpp_reductions.txt).
    constexpr int kAhead = 4;
    for (uint32_t i = tid; i < nvec; i += 256 * kAhead) {
        u32x4 xs[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const uint32_t j = i + 256u * (uint32_t)u;
            xs[u] = __builtin_nontemporal_load(q + (j < nvec ? j : i));  // (past the end: vector i again, not used)
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (i + 256u * (uint32_t)u < nvec) {
                u32x4 x = xs[u];
                const uint32_t hi = (x.x | x.y | x.z | x.w) & kHiMask;
                if (__any(hi != 0)) {
                    x.x = clip_word<PB>(x.x);
                    x.y = clip_word<PB>(x.y);
                    x.z = clip_word<PB>(x.z);
                    x.w = clip_word<PB>(x.w);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t y = pair_index_word<PB>(x[d]);
                    acc += tab[y & 0xFFu];
                    acc += tab[(y >> 16) & 0xFFu];
                }
                if (++pending == 3) {  // <= 1 (head/tail) + 3 * 16 = 49 per field
                    pending = 0;
#pragma unroll
                    for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] += (uint32_t)(acc >> (6 * s)) & 63u;
                    acc = 0;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) cnt[s] += (uint32_t)(acc >> (6 * s)) & 63u;
#pragma unroll
    for (int s = 0; s < MH_LUT_SYMS; ++s) {
        const uint32_t v = wave_sum_u32(cnt[s]);
        if ((tid & 63) == 0) red[s][tid >> 6] = v;
    }
    __syncthreads();
    if ((uint32_t)tid < S - 1) {  // the top bin is the window length minus the rest (k_finalize)
        const uint32_t v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        const uint32_t slot = a.tile_slot ? a.tile_slot[tile] : ch;
        if (v) hist_add(&a.hist[(size_t)slot * kHistStride + tid], v, a.tile_cnt != nullptr);
    }
    measure_tail(a, ch);
}

// [slot][16] u64 scratch (bins 0..8 counted) -> dense [slot][10], top bin = interval length - rest
__global__ __launch_bounds__(256) void k_sweep_finalize(const unsigned long long *scratch,
                                                        const uint64_t *slot_len, uint32_t nslots,
                                                        uint64_t *out)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    uint64_t rest = 0;
    for (int b = 0; b < MH_SWEEP_BINS - 1; ++b) {
        const uint64_t v = scratch[(size_t)s * kHistStride + b];
        out[(size_t)s * MH_SWEEP_BINS + b] = v;
        rest += v;
    }
    out[(size_t)s * MH_SWEEP_BINS + MH_SWEEP_BINS - 1] = slot_len[s] - rest;
}

}  // namespace mh
