/*
 * muahuff.h -- C ABI of libmuahuff.so, the MI355X (gfx950) data plane for static-Huffman
 * compression of binned multi-channel MUA spike counts.
 *
 * The reference (zhengzhang96/Hardware-efficient-MUA-compression) has NO FFI or plugin
 * interface: its hot path is a set of NumPy statements inside three scripts.  Each entry
 * point below therefore cites the reference statements it replaces (paths relative to the
 * reference checkout, directory names contain spaces).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns MH_OK (0) or a negative MH_ERR_* code and never throws;
 *     mh_last_error() returns a thread-local message for the last failure on this thread;
 *   - "host" pointers are plain host memory read synchronously during the call;
 *     "device" pointers are HIP device pointers (e.g. torch.Tensor.data_ptr()); the caller
 *     allocates and owns every buffer;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); device work is
 *     enqueued asynchronously on it and nothing in mh_measure/mh_encode/mh_decode/
 *     mh_compact/mh_synth_poisson/mh_rebin/mh_deinterleave/mh_interleave/mh_power_draws/
 *     mh_reduce_rows synchronises, allocates or frees, so they can be captured into a hipGraph;
 *   - a plan is bound to the device that was current when it was created and may be used
 *     from one stream at a time (it owns per-channel scratch tables).
 */
#ifndef MUAHUFF_H
#define MUAHUFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_VERSION 103 /* 0.1.3 ; container format revision 3 (reads revision 2) */

/* ---- error codes -------------------------------------------------------------------- */
#define MH_OK 0
#define MH_ERR_ARG (-1)           /* NULL pointer, S/h/K out of range, ... */
#define MH_ERR_EMPTY_CHANNEL (-2) /* a channel with 0 bins: the reference raises IndexError
                                     (Compressing data/functions_1.py:45, data_in[0]) */
#define MH_ERR_SCLV (-3)          /* SCLV row not non-decreasing / Kraft sum != 1 */
#define MH_ERR_CAPACITY (-4)      /* payload buffer smaller than mh_plan_info.payload_cap_words */
#define MH_ERR_HIP (-5)           /* a HIP runtime call failed (message has the HIP error) */
#define MH_ERR_NO_DEVICE (-6)     /* no gfx950 device visible: there is NO CPU fallback */
#define MH_ERR_STREAM (-7)        /* mh_validate_stream: the stored stream is inconsistent / corrupt */

/* ---- container geometry (format revision 2; the reference has no bitstream, so this is
 *      build-defined -- see DESIGN.md "Container") ----------------------------------- */
#define MH_PIECE 16                        /* samples per piece (one 16-byte vector)          */
#define MH_LANES 64                        /* sub-streams per chunk = lanes of a wavefront    */
#define MH_ROWS 16                         /* pieces per sub-stream                           */
#define MH_SUB (MH_PIECE * MH_ROWS)        /* 256 samples per sub-stream                      */
#define MH_CHUNK (MH_SUB * MH_LANES)       /* 16384 samples per chunk                         */
#define MH_HDR_WORDS (MH_LANES / 2)        /* upper bound of a chunk header (slot sizing); the header
                                              itself is 1..25 words: 12-bit minimum sub-stream length,
                                              4-bit field width w, 64 w-bit (length - minimum) fields */

/* mapper: which symbol -> rank permutation the calibration yields */
#define MH_MODE_NOSORT 0 /* identity          Compressing data/get_BR_no_sort.py:174,192        */
#define MH_MODE_APPROX 1 /* approx_sort       Compressing data/functions_1.py:75-90             */

/* measured / encoded window of a channel of T bins with calibration cutoff c = min(2^h,T) */
#define MH_WIN_REF_HALF 0       /* [c, c+T/2); channel skipped when c+T/2 > T
                                   (Compressing data/get_BR_with_approx_sort.py:180-189)       */
#define MH_WIN_REF_HALF_TRUNC 1 /* [c, min(c+T/2,T)), never skipped
                                   (Compressing data/test_chosen_system.py:99-103)             */
#define MH_WIN_AFTER_CAL 2      /* [c, T)  : compress everything after calibration             */
#define MH_WIN_FULL 3           /* [0, T)  : whole channel (training histograms,
                                   Compressing data/get_BR_with_approx_sort.py:140-147)        */
/* How a window is cut into segments (part of the container format; the reference has no bitstream).
 * Revision 3, the default: a window of at least MH_HEAD_MIN_WINDOW samples that does not start at a multiple of
 * MH_HEAD_ALIGN samples begins with a short HEAD segment ending at the next multiple; the regular segments
 * (seg_chunks chunks each) follow.  With the calibration cutoffs of the reference (c = 2^2 .. 2^6 samples) every
 * 1-KiB row a wavefront loads or stores would otherwise start 4 .. 64 bytes into a 128-byte line of a line-aligned
 * channel and straddle nine lines instead of eight (1024 ch x 1e7 bins: decode -2 .. -7 %, encode -1.5 .. -4 %).
 * MH_WIN_REV2_SEGMENTS, OR-ed into the `window` argument, cuts every window from its first sample as revision 2
 * did: that is how streams stored by revision 2 are read. */
#define MH_WIN_REV2_SEGMENTS 0x100u
#define MH_HEAD_ALIGN 128
#define MH_HEAD_MIN_WINDOW (16 * MH_CHUNK)

typedef struct mh_plan mh_plan; /* opaque */

typedef struct {
    uint32_t C, S, h, mode, window, K, seg_chunks, maxlen;
    uint64_t n_segments;        /* directory entries                                           */
    uint64_t payload_cap_words; /* u32 words the encode payload buffer must hold               */
    uint64_t window_samples;    /* sum over channels of the window length                      */
    uint64_t n_skipped;         /* channels skipped by the MH_WIN_REF_HALF rule                */
} mh_plan_info_t;

/* ---- library / device --------------------------------------------------------------- */
int mh_version(void);
const char *mh_last_error(void);
/* properties of HIP device `device`; any output pointer may be NULL */
int mh_device_info(int device, int *cu_count, uint64_t *hbm_bytes, char *name, int name_cap,
                   char *arch, int arch_cap);

/* ---- host-side helpers (no GPU needed) ----------------------------------------------- */
/* Canonical codewords (MSB-first values) of one sorted codeword-length vector; rank 0 gets
 * the shortest code.  S=3, [1,2,2] -> 0,10,11 == Compressing data/test_chosen_system.py:26.
 * Lengths come from Compressing data/Produce SCLVs/Stored_SCLVs_S_<S>.pkl. */
int mh_codebook(const uint8_t *sclv_row, int S, uint16_t *code, uint8_t *len);
/* idx[k] = symbol that gets rank k for a calibration histogram peaking at `peak`:
 * the result of Compressing data/functions_1.py:75-90 (approx_sort), closed form. */
int mh_approx_sort_perm(int S, int peak, uint8_t *idx);

/* ---- plan ---------------------------------------------------------------------------- */
/* Describes a set of C channels laid out channel-major in one uint8 device buffer
 * (channel i = bytes [ch_off[i], ch_off[i]+ch_len[i]) ; the in-memory form of
 * all_binned_data[BP][dataset][channel], Data/get_all_binned_data.py:62-64) and fixes the
 * design point: S (symbols 0..S-1), h (calibration window 2^h samples), mapper mode,
 * window rule (MH_WIN_*, optionally | MH_WIN_REV2_SEGMENTS), K candidate encoders given as SCLV rows (host, K*S bytes, row order =
 * encoder index, first-min tie-break as np.argmin).  Precomputes windows, the segment
 * directory and codebooks and uploads them.  ch_off, ch_len, sclv: host.
 * seg_chunks = chunks per segment (the unit one wavefront encodes / decodes); 0 lets the planner
 * choose (2, or 1 for small inputs -- mh_plan_info reports the choice, which a stored stream must
 * carry: segment boundaries are part of the format). */
int mh_plan_create(mh_plan **plan, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                   uint32_t S, uint32_t h, uint32_t mode, uint32_t window, const uint8_t *sclv,
                   uint32_t K, uint32_t seg_chunks);
/* The same for a PACKED input buffer, the intermediate of the time-major (implant-order) path:
 * input_bits = 4 or 2 means channel i is ceil(ch_len[i] / 16) pieces of 8 resp. 4 bytes at ch_off[i]
 * (bytes), as mh_deinterleave_packed writes them; input_bits = 8 is mh_plan_create.  Packed plans
 * cover whole channels (MH_WIN_FULL), 2-bit ones need S <= 4, and only mh_encode_preset reads
 * them (a calibrate-then-stream encoder has its (peak, encoder) word already); mh_measure, mh_encode
 * and mh_decode return MH_ERR_ARG on such a plan -- its stream decodes with an ordinary byte-layout plan
 * over the same channel lengths (segment boundaries depend on lengths and seg_chunks only).
 * chunk_stride = 0: a channel's pieces are contiguous.  chunk_stride = B (a multiple of 16, at least
 * one chunk = 1024 pieces): CHUNK-BLOCKED buffer -- the j-th 16384-sample chunk of channel i starts at
 * ch_off[i] + j * B.  With ch_off[i] = i * chunk bytes and B = C * chunk bytes, all channels' chunks
 * of one time range are neighbours: the de-interleaver then scatters each tile over one small region
 * instead of C far-apart streams, which is what makes it fast. */
int mh_plan_create_packed(mh_plan **plan, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                          uint32_t S, uint32_t h, uint32_t mode, uint32_t window, const uint8_t *sclv,
                          uint32_t K, uint32_t seg_chunks, uint32_t input_bits, uint64_t chunk_stride);
int mh_plan_destroy(mh_plan *plan);
int mh_plan_info(const mh_plan *plan, mh_plan_info_t *info);
/* host copies of the segment directory (each array n_segments long, any may be NULL):
 * channel, first sample (relative to the window start), sample count, slot offset (words) */
int mh_plan_segments(const mh_plan *plan, uint32_t *seg_ch, uint64_t *seg_first,
                     uint64_t *seg_n, uint64_t *seg_off);
/* The planner without a device (pure host arithmetic, no GPU needed): what mh_plan_create would
 * report for this layout, to size buffers ahead of time or to rebuild the directory of a stored
 * stream.  info is filled; the directory arrays (may be NULL) receive at most seg_cap entries. */
int mh_plan_query(const uint64_t *ch_len, uint32_t C, uint32_t S, uint32_t h, uint32_t mode,
                  uint32_t window, const uint8_t *sclv, uint32_t K, uint32_t seg_chunks,
                  mh_plan_info_t *info, uint32_t *seg_ch, uint64_t *seg_first, uint64_t *seg_n,
                  uint64_t *seg_off, uint64_t seg_cap);

/* ---- device operations ---------------------------------------------------------------- */
/* Everything the reference computes per validation channel at one (S, h):
 *   clip                       get_BR_with_approx_sort.py:164
 *   cutoff c                   functions_1.py:27-68  (== min(2^h, T))
 *   calibration histogram      get_BR_with_approx_sort.py:171
 *   peak / permutation         functions_1.py:75-90 ; identity for MH_MODE_NOSORT
 *   encoder = first argmin     get_BR_with_approx_sort.py:254,281
 *   window + post histogram    get_BR_with_approx_sort.py:180-193
 *   bits = SCLV[enc].post      get_BR_with_approx_sort.py:289 (numerator)
 * Outputs (device, any may be NULL): cutoff[C]; cal_hist[C*S] and post_hist[C*S] in RANK
 * order (what the reference stores in val_histograms / val_histograms_post); peak[C];
 * enc[C]; bits[C]; skipped[C] (post_hist row is all zero and bits 0 for a skipped channel,
 * which the reference's float formula turns into NaN). */
int mh_measure(mh_plan *plan, const uint8_t *data, uint64_t *cutoff, uint32_t *cal_hist,
               uint8_t *peak, uint8_t *enc, uint64_t *post_hist, uint64_t *bits,
               uint8_t *skipped, void *stream);

/* Calibrate every channel as mh_measure does, then emit the window of each channel as a
 * static-Huffman bitstream (NEW work: the reference only multiplies histograms by code
 * lengths, get_BR_with_approx_sort.py:133-137).  Segment s is written at word offset
 * seg_off[s] (mh_plan_segments) of `payload`; seg_words[s] receives the words used.
 * ch_bits[c] = exact code bits of channel c == the reference's SCLV[enc].post_hist. */
int mh_encode(mh_plan *plan, const uint8_t *data, uint32_t *payload, uint64_t payload_cap_words,
              uint64_t *seg_words, uint64_t *ch_bits, uint8_t *peak, uint8_t *enc,
              uint8_t *skipped, void *stream);

/* Compression phase of a calibrate-then-stream protocol: encode the plan's windows with a
 * PRESET per-channel (peak, encoder) word instead of calibrating on this buffer -- what the
 * reference's RTL does after its calibration phase, reading {max_rate, encoder_sel} from its RAM
 * (FPGA implementation/README.md:50-66, RAM.v:4).  peak/enc: device, C entries (values out of
 * range are treated as 0).  The stream decodes with mh_decode and the same peak/enc. */
int mh_encode_preset(mh_plan *plan, const uint8_t *data, const uint8_t *peak, const uint8_t *enc,
                     uint32_t *payload, uint64_t payload_cap_words, uint64_t *seg_words,
                     uint64_t *ch_bits, void *stream);

/* Inverse of mh_encode: writes clip(x) = min(x, S-1) for every window sample into `out`
 * (same channel layout as the plan's data buffer; bytes outside the windows are left
 * untouched).  seg_off (device, words) = where each segment starts in `payload`; NULL means
 * the plan's slot offsets.  payload_words = words readable at `payload` (the dense stream plus
 * 4 words of slack, or the plan's payload_cap_words).
 * Memory-safe on ANY stream: the kernel follows the chunk headers it finds, but every read it
 * derives from them is checked against payload_words first; a segment whose headers point outside
 * is abandoned (its remaining output is not written) and the plan's status word is set --
 * mh_decode_status.  Stores only go to the plan's own window positions.  (peak, enc) values out
 * of range decode as 0.  Streams from storage should still be checked with mh_validate_stream,
 * which also detects inconsistencies that stay inside the buffer. */
int mh_decode(mh_plan *plan, const uint32_t *payload, uint64_t payload_words, const uint64_t *seg_off,
              const uint8_t *peak, const uint8_t *enc, uint8_t *out, void *stream);
/* flags (host) <- 1 when any mh_decode on this plan since the previous mh_decode_status (or since
 * plan creation) had to abandon a segment, else 0; the flag is cleared by the call.  Sticky on purpose:
 * replays of a hipGraph that captured mh_decode report through it like direct calls.  Call it after the
 * decode whose stream you do not trust.  Synchronises `stream`. */
int mh_decode_status(mh_plan *plan, uint32_t *flags, void *stream);

/* Host-side structural check of a stored stream (no GPU needed; all pointers host): rebuilds the
 * directory from (ch_len, h, window, seg_chunks), walks every chunk header of every segment of the
 * DENSE payload (segment s starts at word sum(seg_words[:s])) and verifies that header sizes,
 * sub-stream lengths (possible for the channel's code) and chunk sizes add up exactly to
 * seg_words and payload_words, and that (peak, enc) are in range.  MH_ERR_STREAM + mh_last_error
 * name the first inconsistency. */
int mh_validate_stream(const uint64_t *ch_len, uint32_t C, uint32_t S, uint32_t h, uint32_t mode,
                       uint32_t window, const uint8_t *sclv, uint32_t K, uint32_t seg_chunks,
                       const uint32_t *payload, uint64_t payload_words, const uint64_t *seg_words,
                       uint64_t n_segments, const uint8_t *peak, const uint8_t *enc);

/* Pack the used words of all segments back to back (directory order) for storage or for
 * the RCCL gather: dense_off[s] = exclusive prefix of seg_words, total_words[0] = sum. */
int mh_compact(mh_plan *plan, const uint32_t *payload, const uint64_t *seg_words,
               uint32_t *dense, uint64_t dense_cap_words, uint64_t *dense_off,
               uint64_t *total_words, void *stream);

/* Synthetic Poisson-like MUA: x[ch][t] = #{s<15 : u16(seed,ch,t) >=
 * thr[ch*15+s]}, u16 from a splitmix64-finalised counter.  ch_off/ch_len/thr: device. */
int mh_synth_poisson(uint8_t *data, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                     uint64_t max_len, const uint32_t *thr, uint64_t seed, void *stream);

/* Per-channel re-binning: out[b] = sum of x[b*r .. min(b*r+r,T))
 * (Compressing data/functions_1.py:11-24, bin_MUA_data, applied along time).
 * saturate != 0: uint8 output saturating at 255 like MATLAB uint8()
 * (Data/Load_and_bin_Sabes_store_as_mat_file.m:53); else uint32 output.
 * in_off/in_len/out_off: device arrays of C entries (out_off in output elements). */
int mh_rebin(const uint8_t *data, const uint64_t *in_off, const uint64_t *in_len, uint32_t C,
             uint64_t max_len, uint32_t r, int saturate, void *out, const uint64_t *out_off,
             void *stream);

/* Time-major interleaved samples -> the channel-major layout of a plan.  `in` holds T rows of C
 * bytes, |CH1|CH2|...|CHN| per time step: the order in which an implant (and the reference's
 * RTL in its compression phase, FPGA implementation/README.md:31) emits binned counts.
 * Channel c is written to out + out_off[c] (T bytes).  in, out, out_off: device. */
int mh_deinterleave(const uint8_t *in, uint64_t T, uint32_t C, uint8_t *out, const uint64_t *out_off,
                    void *stream);
/* The same transposition with the output clipped and packed: min(x, 15) in 4 bits per sample
 * (bits = 4) or min(x, 3) in 2 bits (bits = 2; enough when S <= 4).  The stream encoder clips at
 * S-1 anyway, so this halves / quarters the intermediate's trip through HBM.  Channel c is written
 * as ceil(T / 16) pieces of 8 / 4 bytes at out + out_off[c] (a cut last piece is zero-padded;
 * out_off[c] must be a multiple of 16 and the region behind it writable up to the next multiple of 16
 * bytes past its last piece: the kernel stores 16 bytes at a time; chunk_stride as in
 * mh_plan_create_packed: 0 = contiguous pieces, else chunk j of channel c starts at out_off[c] + j * chunk_stride):
 * plain little-endian bit packing -- sample i of the channel in bits [i * bits, (i + 1) * bits) of its stream:
 *   4 bits: byte j = s[2j] | s[2j+1] << 4
 *   2 bits: byte j = s[4j] | s[4j+1] << 2 | s[4j+2] << 4 | s[4j+3] << 6 */
int mh_deinterleave_packed(const uint8_t *in, uint64_t T, uint32_t C, uint32_t bits, uint8_t *out,
                           const uint64_t *out_off, uint64_t chunk_stride, void *stream);
/* The inverse: channel c = T bytes at in + in_off[c]  ->  out[t*C + c] (what a decoder hands back
 * to a consumer of the implant-order stream).  in, in_off, out: device. */
int mh_interleave(const uint8_t *in, const uint64_t *in_off, uint64_t T, uint32_t C, uint8_t *out,
                  void *stream);

/* ---- fused sweep histograms (all design points from ONE pass over the data) --------------
 * The two BR scripts loop S = 2..10 and histogram sizes 2^h (get_BR_with_approx_sort.py:107,157)
 * and re-read every validation channel for each of the 81 combinations, for each CV split.
 * All of those histograms are differences of prefix histograms of min(x, 9) taken at
 *   0, c_h = min(2^h, T), e_h = min(c_h + T/2, T), T        (h in hist_bits[0..nh))
 * so one pass that histograms the 2*nh+1 intervals between the SORTED breakpoints of each
 * channel serves every (S, h, CV): calibration histogram = H(c_h), post histogram =
 * H(c_h + T/2) - H(c_h), training histogram = H(T); smaller S merge the top bins. */
typedef struct mh_sweep mh_sweep; /* opaque */
#define MH_SWEEP_BINS 10 /* histogram bins per interval: values clipped at 9 (S <= 10) */
int mh_sweep_create(mh_sweep **sweep, const uint64_t *ch_off, const uint64_t *ch_len, uint32_t C,
                    const uint32_t *hist_bits, uint32_t nh);
int mh_sweep_destroy(mh_sweep *sweep);
/* n_intervals = 2*nh+1; bounds (host, C*(n_intervals+1) entries, may be NULL) receives the
 * sorted breakpoints of every channel: interval j of channel c is [bounds[j], bounds[j+1]) */
int mh_sweep_info(const mh_sweep *sweep, uint32_t *n_intervals, uint64_t *bounds);
/* hist (device): C * n_intervals * MH_SWEEP_BINS u64 counts, [channel][interval][bin] */
int mh_sweep_run(mh_sweep *sweep, const uint8_t *data, uint64_t *hist, void *stream);

/* ---- result consumers (the reference's Analyse results/ scripts), float64, bit-exact with NumPy --
 * Sums follow NumPy's pairwise summation order, so results equal np.sum / np.mean bit for bit.
 *
 * mh_power_draws: Analyse results/max_nb_channels_p_value_power_budget.py:100-105.  For every
 * draw d < n_draws:  x[d*x_stride] += comm_energy * sum_j br[idx[j*n_draws + d]] + per_channels
 * + static_power  (per_channels = nb_channels * (ADC_power + chan_processing_power), computed by
 * the caller in float64).  idx: [Z][n_draws] int32 channel indices < n_br, drawn by the caller
 * (the reference draws them with np.random.choice; the stream of a seeded legacy RNG can only
 * be reproduced on the host).  br, idx, x: device. */
int mh_power_draws(const double *br, uint32_t n_br, const int32_t *idx, uint32_t Z, uint64_t n_draws,
                   double comm_energy, double per_channels, double static_power, double *x,
                   uint64_t x_stride, void *stream);
/* mh_reduce_rows: Analyse results/integrate_BR_and_BDP_results_into_excel.py:118-119.  Row r =
 * vals[row_off[r] .. row_off[r+1]): sum[r] = np.sum(row) (pairwise), mx[r] = np.max(row) (NaN if
 * any element is NaN, NaN for an empty row).  All pointers device; row_off has n_rows+1 entries. */
int mh_reduce_rows(const double *vals, const uint64_t *row_off, uint64_t n_rows, double *sum,
                   double *mx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MUAHUFF_H */
