/* beat_this_amd -- C ABI of the MI355X-native beat_this inference hot path.
 *
 * The reference (CPJKU/beat_this, pure Python) has no FFI; its seam is the Python
 * API of beat_this/inference.py.  These entry points are what a binding for that
 * seam calls -- each one cites the reference code it replaces.  All `d_*` pointers
 * are DEVICE pointers owned by the caller (torch tensors on the Python side);
 * `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream).  No call
 * allocates device memory, synchronises the device, or starts threads.  Return
 * value: BT_OK or a negative BT_ERR_* code; bt_last_error() gives the text.
 * The library is reentrant per engine handle, not concurrently on one handle; the caller's workspace belongs to ONE
 * stream at a time (two streams running one engine need two workspaces).
 */
#ifndef BEAT_THIS_AMD_H
#define BEAT_THIS_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_OK 0
#define BT_ERR_ARG (-1)      /* bad argument / unsupported shape  -> ValueError */
#define BT_ERR_HIP (-2)      /* HIP runtime error                 -> RuntimeError */
#define BT_ERR_WORKSPACE (-3)/* workspace too small               -> RuntimeError */

#define BT_PREC_F32 0  /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 activations: float16=False */
#define BT_PREC_HALF 1 /* half-precision MFMA operands (IEEE fp16; bfloat16 in a -DBT_HALF_BF16 build, see bt_half_is_bf16),
                        * fp32 accumulate + fp32 residual stream: float16=True */
/* (2 was BT_PREC_FP8, an experimental e4m3 feed-forward path of rounds 1-2: withdrawn in round 3 -- on every golden case
 * its logit error stayed far above the reference's own reduced-precision error and it was no faster) */

#define BT_PREC_F32X3 3 /* fp32-class results at half-MFMA speed -- the precision that carries the 1e-3 / identical-beats gate:
                        * every fp32 operand of every product (QKV, attention scores, P.V, out-projection, feed-forward,
                        * convolutions, frontend.linear) is a hi + lo pair of IEEE fp16 halves (a = hi + lo to 2^-22) and
                        * a.b ~ lo.hi + hi.lo + hi.hi on three half MFMAs, fp32 accumulate: 16/3 times the fp32 matrix rate.
                        * Residual stream, statistics, softmax sums, GELU (exact erf form), stem and head stay fp32.
                        * Activations travel between kernels as interleaved hi / lo half planes ("hl32": per 32 consecutive
                        * columns 32 hi halves then 32 lo halves), weights come packed the same way (bt_pair_weights.w_*_x3;
                        * where they are NULL or a shape does not fit, the GEMM falls back to the register-staged kernel on
                        * [hi | lo] weights, beat_this_amd/pack.py _mat).  IEEE fp16 builds only.
                        * RANGE: a hi part holds |a| <= 65504 (main layers, like the reference's own fp16 autocast) or
                        * |a| <= 1023 (register-chained frontend halves, whose operands are pre-scaled by 64).  Beyond that a
                        * split yields inf, and inf / NaN then spreads to every token of the chunk (attention).  The first
                        * int32 of the workspace is the forward's range flag: bt_forward zeroes it; the gemm3 / attention /
                        * QKV kernels OR 1 into it when a value beyond 65504 goes through a split, and whatever ends the call
                        * -- the head (logits), the final norm or the frontend's exit (bt_forward_stages with last < 2) --
                        * ORs 2 when its output is not finite, which covers the splits that do not test their operands (the
                        * register-chained frontend halves, the register-staged GEMM).  A caller that finds the word non-zero
                        * after the forward must repeat the batch in BT_PREC_F32 (beat_this_amd/pack.py: Engine does). */

#define BT_MAX_LAYERS 32

typedef struct bt_engine bt_engine;

/* One [RMSNorm -> gated RoPE attention -> +x ; RMSNorm -> FF -> +x] pair
 * (roformer.py:38-61,83-132,176-179).  Weight matrices come in two device copies,
 * index BT_PREC_F32 / BT_PREC_HALF, row-major [N padded to a multiple of 128][K]:
 *   w_qkvg : rows = q (heads*32, scaled by log2(e)/sqrt(32)) | k | v | gate rows (heads),
 *            every row multiplied by the attention RMSNorm gamma;
 *   w_out  : to_out.0.weight;  w_ff1 : net.1.weight * FF gamma;  w_ff2 : net.4.weight. */
typedef struct {
  int32_t dim, heads;
  const void* w_qkvg[2];
  const float* b_gates;
  const void* w_out[2];
  const void* w_ff1[2];
  const float* b_ff1;
  const void* w_ff2[2];
  const float* b_ff2;
  /* ("PERM32" below: columns reordered inside every block of 32, new[16 g + r] = old[(r & 3) + 8 (r >> 2) + 4 g],
   * g in {0,1}, r in [0,16) -- the order in which an MFMA accumulator holds them.)
   * FF weights for ff_fused_kernel, fragment-major: for each hidden block hb (32 hidden units):
   * dim/32 tiles of W1 (rows hb*32.., k-tile kt) then dim/32 tiles of PERM32'd W2 (rows mt*32..,
   * cols hb*32..); a tile is [half h][lane 0..63][8] (bf16) or [quarter][lane][4] (fp32) with
   * element (lane, i) = W[tile_row0 + (lane & 31)][tile_col0 + 16 (lane >> 5) + i], i in [0,16). */
  const void* w_ff_frag[2];
  /* bf16 QKV+gate weights for qkv_front_kernel (time-direction attention of the frontend, dim <= 128),
   * fragment-major tiles [half h][lane][8] as above, in the order: for every head hd the dim/32 k-tiles
   * of its 32 q rows, then of its k rows, then of its v rows; after the last head dim/32 tiles of the
   * gate rows (w_qkvg rows 3 dim .. 3 dim + 31).  NULL for dim > 128. */
  const void* w_qkv_frag;
  /* Weights of the fused out-projection + FF kernels (csrc/fused2.hip), fragment-major tiles as above, two
   * precisions, uniform steps of 2 dim/32 tiles: per 32-row block mt of to_out.0.weight its dim/32 k-tiles
   * (natural k order) + as many zero tiles, then
   * the w_ff_frag stream with W1's columns PERM32-ordered as well.  NULL for dim > 128. */
  const void* w_outff_frag[2];
  /* Weights of the fused frequency-direction kernel (attention + FF), fragment-major, steps of 2 dim/32 tiles:
   * [gate rows of w_qkvg | zero tiles], per head [q rows | k rows] [v rows | PERM32'd to_out tiles (row block
   * mt, the head's 32 columns)], then the FF steps as in w_outff_frag.  NULL for dim > 128. */
  const void* w_attnff_frag[2];
  /* Weights of the fused layer tail (csrc/tail.hip: x += to_out(ao); x += FF(x) in one launch; main layers with
   * dim = 256 / 512, half precision only; NULL elsewhere): fragment-major tiles [half h][lane][8] as above, steps of
   * 2 dim/32 tiles: for st = 0 .. dim/64 - 1 the dim/32 k-tiles of to_out.0.weight's row block 2 st, then of row block
   * 2 st + 1 (natural k order); then for i = -1 .. hidden/32 the step [A(i + 1) | B(i - 1)], A(j) = the dim/32 k-tiles of
   * (net.1.weight * FF gamma) rows 32 j .. (k columns PERM32-ordered), B(j) = the dim/32 row tiles of net.4.weight's
   * columns 32 j .. (PERM32-ordered inside the block); halves that do not exist (j < 0, j >= hidden/32) are zero tiles. */
  const void* w_tail_frag;
  /* BT_PREC_F32X3 forms of w_outff_frag / w_attnff_frag: the half fragment stream of 64 x the weight's hi part and of
   * its lo part, interleaved per 32 x 32 tile ([hi tile 2 KB | lo tile 2 KB]).  NULL: the fp32 kernels run instead. */
  const void* w_outff_frag_x3;
  const void* w_attnff_frag_x3;
  /* BT_PREC_F32X3 on the LDS-DMA kernels (csrc/gemm3.hip, csrc/qkv_front.hip): the four matrices as hl32 half arrays
   * [N padded to 256][2 K] (per 32 columns the 32 hi halves, then the 32 lo halves of w - hi), and the hi / lo fragment
   * stream of w_qkv_frag in the layout of w_outff_frag_x3 (64 x the weight).  NULL: the register-staged GEMM runs. */
  const void* w_qkvg_x3;
  const void* w_out_x3;
  const void* w_ff1_x3;
  const void* w_ff2_x3;
  const void* w_qkv_frag_x3;
  /* BT_OPT_X3_GEMM_FP8 (BASELINE config 5): the four matrices of w_*_x3 as "hl8" half arrays [N padded to 256][2 K] -- per 32
   * columns 32 hi halves, 32 hi bytes e4m3(w), 32 lo bytes e4m3(2^11 (w - hi)): 128 B like an hl32 group.  NULL: the hl32
   * form runs for that GEMM.  (An hl8 ACTIVATION, written by the kernels for each other, carries 2^-3 of that in its byte
   * sections -- e4m3(a / 8), e4m3(2^8 (a - hi)) -- so that it reaches |a| = 3584 where e4m3 alone ends at 448; beyond it the
   * producer raises the range flag.) */
  const void* w_qkvg_f8;
  const void* w_out_f8;
  const void* w_ff1_f8;
  const void* w_ff2_f8;
} bt_pair_weights;

/* Packed BeatThis weights (beat_tracker.py:38-106).  Host-side packing is done by
 * beat_this_amd/pack.py from a reference-layout state dict. */
typedef struct {
  int32_t transformer_dim, n_layers, sum_head, partial_transformers;
  /* stem (beat_tracker.py:108-126): BN1d as scale/shift per mel bin; conv weight
   * [32][df*3+dt] with BN2d scale folded; BN2d shift as bias. */
  const float* bn1_scale;
  const float* bn1_shift;
  const float* stem_w;
  const float* stem_b;
  bt_pair_weights front[3][2];  /* [block][0 = frequency direction, 1 = time direction] */
  /* frontend convs (beat_tracker.py:155-166): [2C padded][3 taps * 2 freq * C] with BN folded */
  const void* conv_w[3][2];
  const float* conv_b[3];
  /* frontend.linear (beat_tracker.py:76-77), columns permuted from (c f) to (f c) */
  const void* lin_w[2];
  const float* lin_b;
  bt_pair_weights layers[BT_MAX_LAYERS];
  const float* head_w; /* [2][D] task_heads weight * final RMSNorm gamma */
  float head_b[2];
  const float* rope;   /* [rope_len][16][2] cos/sin of pos * freqs (rotary-embedding-torch), fp32 products like the reference's */
  int32_t ff_mult;     /* hidden width of the main layers' FeedForward = ff_mult * transformer_dim (the frontend's partial
                        * transformers always use 4, beat_tracker.py:279,288) */
  /* for bt_forward_stages only: the final RMSNorm's gamma [D] and the task_heads weight [2][D] without it */
  const float* norm_out_g;
  const float* head_w_raw;
  /* BT_PREC_F32X3: hl32 forms (see bt_pair_weights.w_qkvg_x3) of conv_w / lin_w; NULL: the register-staged GEMM runs */
  const void* conv_w_x3[3];
  const void* lin_w_x3;
  int32_t rope_len;    /* rows of `rope` = the longest sequence (frames per item) bt_forward accepts; 0 means 1536 */
} bt_model_desc;

typedef struct {
  const float* window;   /* [1024] periodic Hann */
  const float* twiddle;  /* [1089][2]  (see csrc/logmel.hip) */
  const int32_t* mel_start; /* [128] first FFT bin of each mel filter */
  const int32_t* mel_len;   /* [128] */
  const float* mel_w;       /* [128][32], 16-byte aligned; a row is ZERO beyond the filter's mel_len (the kernel reads it four taps at a time) */
} bt_logmel_tables;

const char* bt_last_error(void);
/* ABI version of this header: bumped whenever an entry point's signature, a struct layout or a BT_PREC_* value changes; a
 * binding must see exactly the value it was written against (beat_this_amd/_lib.py does) */
#define BT_ABI_VERSION 600
int bt_version(void);
/* operand type of the half-precision path (BT_PREC_HALF slot of the weight arrays) this library was built
 * with: 0 = IEEE fp16 (default), 1 = bfloat16 (-DBT_HALF_BF16) */
int bt_half_is_bf16(void);
/* sizeof/offsetof of the structs above as this library was compiled (binding self-check):
 * out[9] = {pair_weights, model_desc, logmel_tables, gemm_args, attn_args, offsetof layers, offsetof rope,
 * attn_frag_args, gemm3_args} */
void bt_struct_sizes(int32_t* out);

/* BeatThis(**hparams) + load_state_dict (inference.py:56-87): keeps a copy of `desc`. */
int bt_engine_create(const bt_model_desc* desc, bt_engine** out);
void bt_engine_destroy(bt_engine* e);
/* Options of an engine (arithmetic variants of BT_PREC_F32X3 that stay inside north_star's 1e-3 / identical-beats gate but
 * are not bit-identical to each other; both are kept so that the choice can be measured: tools/flip_soak.py,
 * profiles/r05_flip_frontier.txt).  bt_engine_set_option returns BT_ERR_ARG for an unknown option / value.
 *   BT_OPT_X3_ATTN_P16  the attention probabilities enter P.V as their fp16 hi parts (two MFMAs per fragment pair instead of
 *                       three), row sums from the same rounded values:  1 (default since round 6) in the main layers only;
 *                       2 in the frontend's time-direction attention as well (round 5's default: +2 % throughput, 2.7 x the
 *                       exact path's beat flips on trained-like weights over the soak -- an opt-in, not the default: the default
 *                       is chosen by the flip-soak rule of DESIGN.md section 3);  0: three-term P.V with the probabilities
 *                       split hi + lo everywhere (rounds 3 - 4);  3: P16 in the frontend only (a variant for the flip soak: which half
 *                       of the attention launches the flips come from).
 *   BT_OPT_X3_GEMM_FP8  0 (default);  BASELINE config 5 -- GEMMs of the main layers run the two cross terms of every hi + lo product
 *                       (hi . lo + lo . hi: 2^-11 of the product) on ONE block-scaled fp8 MFMA per 32-k step instead of four fp16
 *                       ones, operands travelling as hl8 (bt_pair_weights.w_*_f8): 2 MFMA units per product instead of 3.
 *                       1: the feed-forward GEMMs (FF1, FF2);  2: out-projection and QKV as well.  An opt-in speed setting inside
 *                       the 1e-3 gate, reported beside the default (bench.py: configs.cfg5; flip rates: DESIGN.md section 3):
 *                       measured at level 2 -4 % forward time, 1.7e-4 .. 2.5e-4 on the logits against the oracle (default 7e-5),
 *                       1.1 x (outlier weights) to 14 x (freshly initialised) the default's beat flips over the 96-track soak.
 *                       Activations beyond 3584 (hl8's range) raise the range flag like those beyond 65504 on the default path. */
#define BT_OPT_X3_ATTN_P16 1
#define BT_OPT_X3_GEMM_FP8 2
int bt_engine_set_option(bt_engine* e, int option, int value);
int bt_engine_get_option(const bt_engine* e, int option, int* value);
/* bytes of scratch bt_forward needs for a [B,T,128] batch (its first int32 is the BT_PREC_F32X3 range flag) */
size_t bt_workspace_bytes(const bt_engine* e, int B, int T, int prec);

/* BeatThis.forward (beat_tracker.py:188-192): d_spect [B,T,128] fp32 ->
 * d_beat, d_downbeat [B,T] fp32 logits (SumHead applied).  Any T <= bt_model_desc.rope_len (the reference's chunks have
 * T = 1500; its module takes any length, and so does this one given a rotary table that long). */
int bt_forward(bt_engine* e, void* stream, int prec, const float* d_spect, int B, int T, void* d_ws,
               size_t ws_bytes, float* d_beat, float* d_downbeat);

/* The three stages of BeatThis.forward on their own (beat_tracker.py:188-192: x = frontend(x); x = transformer_blocks(x);
 * x = task_heads(x)), for callers that call or hook the sub-modules: stages first..last run (0 = frontend: [B,T,128] ->
 * [B,T,D]; 1 = transformer_blocks incl. its final RMSNorm: [B,T,D] -> [B,T,D]; 2 = task_heads: [B,T,D] -> logits).
 * d_in is the input of stage `first`; d_out [B,T,D] receives the output of stage `last` when last < 2, otherwise
 * d_beat / d_downbeat do.  Same kernels and workspace as bt_forward (== stages 0..2). */
int bt_forward_stages(bt_engine* e, void* stream, int prec, int first, int last, const float* d_in, int B, int T, void* d_ws,
                      size_t ws_bytes, float* d_out, float* d_beat, float* d_downbeat);

/* The sub-modules below the three stages (what a caller of the reference reaches as model.frontend.stem, .blocks[i],
 * .blocks[i].partial, .linear, model.transformer_blocks.layers[l][0] / [1], .norm -- beat_tracker.py:54-80,108-168,
 * roformer.py:138-181), one unit per call, fp32 tensors in THIS library's activation layout, on the generic kernels (not
 * the fused fast path of bt_forward): BT_PREC_F32 or BT_PREC_HALF (BT_PREC_F32X3 is taken as BT_PREC_F32).
 *   BT_UNIT_STEM     d_in spect [B,T,128]              -> d_out [B,T,32,32]   (b, t, f, c)
 *   BT_UNIT_PARTIAL  index = block: d_in [B,T,F,C]     -> d_out [B,T,F,C]     PartialFTTransformer (F = 32 >> index, C = 32 << index)
 *   BT_UNIT_CONV     index = block: d_in [B,T,F,C]     -> d_out [B,T,F/2,2C]  conv (2,3) / stride (2,1) + BatchNorm + GELU
 *   BT_UNIT_LINEAR   d_in [B,T,4,256] (= (f c) order)  -> d_out [B,T,D]
 *   BT_UNIT_ATTN     index = layer: d_in [B,T,D]       -> d_out = x + Attention(x)      (the residual form the layer computes)
 *   BT_UNIT_FF       index = layer: d_in [B,T,D]       -> d_out = x + FeedForward(x)
 *   BT_UNIT_NORM     d_in [B,T,D]                      -> d_out = RMSNorm(x)
 *   BT_UNIT_FRONT_ATTN / BT_UNIT_FRONT_FF   the four leaves of a PartialFTTransformer (beat_tracker.py:251-301: attnF, ffF, attnT,
 *                    ffT -- ordinary Attention / FeedForward modules of width C = 32 << block on "(b t) f c" / "(b f) t c" rows):
 *                    index = 2 block + direction (0 = F, 1 = T); here B = sequences, T = tokens per sequence:
 *                    d_in [B,T,C] -> d_out = x + Attention(x) / x + FeedForward(x)
 * d_out may equal d_in for the in-place units (PARTIAL, ATTN, FF, FRONT_*).  Workspace as for bt_forward. */
enum { BT_UNIT_STEM = 0, BT_UNIT_PARTIAL = 1, BT_UNIT_CONV = 2, BT_UNIT_LINEAR = 3, BT_UNIT_ATTN = 4, BT_UNIT_FF = 5, BT_UNIT_NORM = 6,
       BT_UNIT_FRONT_ATTN = 7, BT_UNIT_FRONT_FF = 8 };
int bt_forward_unit(bt_engine* e, void* stream, int prec, int unit, int index, const float* d_in, float* d_out, int B, int T,
                    void* d_ws, size_t ws_bytes);

/* split_piece + zeropad (inference.py:90-135): d_chunks[b,t,:] = d_spect[d_starts[b]+t,:] or 0 */
int bt_split_chunks(void* stream, const float* d_spect, int64_t n_frames, const int32_t* d_starts, int B, int T,
                    float* d_chunks);
/* aggregate_prediction, overlap_mode="keep_first" (inference.py:138-185) */
int bt_aggregate(void* stream, const float* d_chunk_beat, const float* d_chunk_downbeat, const int32_t* d_starts,
                 int B, int T, int border, int64_t n_frames, float* d_beat, float* d_downbeat);

/* LogMelSpect.forward (preprocessing.py:56-59): d_audio [n_samples] fp32 @22.05 kHz ->
 * d_spect [1 + n_samples/441, 128].  n_samples must exceed 512 (reflect padding). */
int bt_logmel(void* stream, const bt_logmel_tables* tables, const float* d_audio, int64_t n_samples,
              float* d_spect);

/* ---- several tracks per launch (the batch form of Audio2Beats.__call__, inference.py:269-303: one launch per stage for
 * a whole list of tracks instead of a Python loop of per-track calls; same arithmetic as the single-track entry points).
 * A DEVICE table of bt_span describes the tracks: input samples, their count, the offset of the track's output in the
 * concatenated output buffer (samples for the resampler, spectrogram rows for the log-mel), and its output count. */
typedef struct { const float* d_in; int64_t n_in; int64_t out_off; int64_t n_out; } bt_span;
/* resample every track (all at the same up / down), max_n_out = the largest n_out */
int bt_resample_batch(void* stream, const bt_span* d_tracks, int n_tracks, int64_t max_n_out, int up, int down,
                      const float* d_filter, int half_len, float* d_out);
/* log-mel of every track: n_out = 1 + n_in / 441 frames written at row out_off of d_spect; every n_in > 512 */
int bt_logmel_batch(void* stream, const bt_logmel_tables* tables, const bt_span* d_tracks, int n_tracks,
                    int64_t max_frames, float* d_spect);
/* split_piece over the concatenated spectrogram: d_chunk_table [B][4] int32 = {first source row of the chunk (may lie
 * before the piece), first row of its piece, end row of its piece, unused}, ABSOLUTE rows of d_spect; rows outside the
 * piece read as zeros */
int bt_split_chunks_batch(void* stream, const float* d_spect, const int32_t* d_chunk_table, int B, int T, float* d_chunks);
/* keep_first aggregation of all pieces: d_pieces [n_pieces][4] int32 = {first frame, end frame (absolute, of the
 * concatenated d_beat / d_downbeat), first chunk, end chunk}; chunk starts from d_chunk_table */
int bt_aggregate_batch(void* stream, const float* d_chunk_beat, const float* d_chunk_downbeat, const int32_t* d_chunk_table,
                       const int32_t* d_pieces, int n_pieces, int64_t max_frames, int T, int border, float* d_beat,
                       float* d_downbeat);

/* Audio2Frames.signal2spect's resampling step (inference.py:274-275, soxr.resample on the host in the reference):
 * rational polyphase FIR on the GPU, y[m] = sum_k x[k] h[m down + half_len - k up], d_filter = 2 half_len + 1 taps
 * (beat_this_amd/tables.py: resample_filter = up * firwin(.., 1/max(up,down), kaiser 5.0), scipy.signal.resample_poly's
 * design), n_out <= ceil(n_in up / down).  Not bit-compatible with libsoxr (parity unpinned). */
int bt_resample(void* stream, const float* d_in, int64_t n_in, int up, int down, const float* d_filter, int half_len,
                float* d_out, int64_t n_out);

/* Postprocessor.postp_minimal peak mask (postprocessor.py:93-99) + nonzero (:119-120):
 * d_logits [n_arrays][n] -> d_idx [n_arrays][n] ascending frame indices, d_count [n_arrays]. */
int bt_peaks(void* stream, const float* d_logits, int64_t n, int n_arrays, int32_t* d_idx, int32_t* d_count);
/* ragged form: array a = d_logits + d_spans[2 a], d_spans[2 a + 1] frames; its indices are written at d_idx + d_spans[2 a] */
int bt_peaks_batch(void* stream, const float* d_logits, const int32_t* d_spans, int n_arrays, int32_t* d_idx,
                   int32_t* d_count);
/* HOST: the same peak mask for logits in host memory (the reference's Postprocessor accepts CPU tensors,
 * postprocessor.py:58-83); idx must hold n entries */
int bt_peaks_host(const float* logits, int64_t n, int32_t* idx, int32_t* count);

/* HOST: deduplicate_peaks(peaks, width) on its own (postprocessor.py:176-197): groups of ascending frame indices not more than
 * `width` apart (measured from the running mean) are replaced by their mean; out must hold n doubles */
int bt_deduplicate_peaks_host(const int32_t* idx, int n, double width, double* out, int32_t* n_out);
/* HOST: deduplicate_peaks(width=1) (postprocessor.py:176-197), frame/fps, snap every
 * downbeat to the nearest beat, np.unique (postprocessor.py:121-136).  Output buffers
 * must hold n_beat_idx / n_down_idx doubles. */
int bt_postprocess_host(const int32_t* beat_idx, int n_beat_idx, const int32_t* down_idx, int n_down_idx,
                        double fps, double* beats, int32_t* n_beats, double* downbeats, int32_t* n_downbeats);

/* Audio2Beats.__call__ for ONE track in ONE call (inference.py:269-281,301-303 without the audio decoder and the host
 * post-processing): resample (when up != down) -> log-mel -> split_piece -> BeatThis.forward -> keep_first aggregation -> peak
 * mask -> ONE device-to-host copy of the result, all enqueued on `stream` by this function -- the host language is not in the
 * loop between the stages (single-file latency, BASELINE config 1; round 5: five places where the GPU waited for Python).
 * bt_audio2beats_plan sizes the call: a track of n_in samples at a rate with 22050 / rate = up / down in lowest terms gives
 * n22 samples at 22.05 kHz, n_frames = 1 + n22 / 441 spectrogram rows and B chunks of T frames (T = 1500, or n_frames + 12 for a
 * piece of <= 1488 frames); ws_bytes of device scratch; result_words int32 of PINNED host memory:
 *     h_result = [beat peak frames: n_frames slots | downbeat peak frames: n_frames slots | n_beat, n_down | range flag]
 * (ascending frame indices as bt_peaks writes them, then the two counts, then the BT_PREC_F32X3 range flag of the forward: non-zero
 * = repeat the call with BT_PREC_F32), valid once `stream` has drained.  d_audio: mono fp32 on the device (the caller's mono mix
 * and upload).  use_graph != 0: the forward's launches are replayed as a hipGraph the engine captures itself on first use and keeps
 * per (B, T, precision, d_ws) -- d_ws must then be the same allocation from call to call (T == 1500 and B <= 16 only; other shapes
 * launch plainly).  Same kernels in the same order as the separate entry points: bit-identical results.  Framewise logits of the
 * call stay readable in d_ws at plan.off_logits ([beat n_frames | downbeat n_frames] fp32) until the next call on d_ws. */
typedef struct {
  int64_t n22, n_frames, result_words;
  int32_t B, T;
  size_t ws_bytes, off_wave22, off_spect, off_chunks, off_chunk_logits, off_logits, off_result, off_forward, forward_bytes;
} bt_a2b_plan;
int bt_audio2beats_plan(const bt_engine* e, int64_t n_in, int up, int down, int prec, bt_a2b_plan* plan);
int bt_audio2beats_enqueue(bt_engine* e, void* stream, int prec, const bt_logmel_tables* tables, const float* d_audio, int64_t n_in,
                           int up, int down, const float* d_filter, int half_len, void* d_ws, size_t ws_bytes, int32_t* h_result,
                           int use_graph);

/* Per-launch timing of bt_forward with HIP events recorded on the caller's stream (bench.py's
 * roofline leg); per engine handle.  bt_profile_begin(e) arms it; every bt_forward(e, ...) until bt_profile_end(e, ...)
 * records one event pair per kernel launch; bt_profile_end() synchronises on the events and returns summed
 * milliseconds and launch counts per category (index = BT_CAT_*). */
#define BT_CAT_STEM 0
#define BT_CAT_QKV_GEMM 1
#define BT_CAT_ATTN_FLASH 2  /* time-direction + main attention */
#define BT_CAT_OUT_GEMM 3
#define BT_CAT_FF1_GEMM 4
#define BT_CAT_FF2_GEMM 5
#define BT_CAT_CONV_GEMM 6
#define BT_CAT_LINEAR_GEMM 7
#define BT_CAT_HEAD 8
#define BT_CAT_FF_FUSED 9         /* ff_fused_kernel (frontend FF blocks) */
#define BT_CAT_ATTN_FREQ_FUSED 10 /* attnff_fused_kernel: frequency-direction half (QKV + attention + out-proj + FF) */
#define BT_CAT_LAYER_TAIL 11      /* layer_tail_kernel (main layers: out-projection + FF1 + FF2 in one launch) */
#define BT_PROFILE_CATEGORIES 12
void bt_profile_begin(bt_engine* e);
int bt_profile_end(bt_engine* e, double* ms_by_category, int32_t* launches_by_category, int n_categories);

/* Single-operator entry points (used by the parity tests; same kernels bt_forward launches). */
typedef struct {
  const void* A; int64_t lda; const void* W; int32_t M, N, K; int32_t epi, flags;
  const float* bias; void* out; int64_t ldo; float* x; int64_t ldx;
  int32_t conv_C2, conv_T, conv_F;
  float* gates; int32_t inner, heads; const float* rope; int32_t pdiv, pmod, map_T, map_F;
} bt_gemm_args;
int bt_gemm(void* stream, int prec, const bt_gemm_args* a);

typedef struct {
  const void* qkv; int64_t ld; const float* gates; void* out;
  int32_t n_seq, L, heads, inner, o_div; int64_t o_outer, o_inner, o_tok;
} bt_attn_args;
int bt_attention(void* stream, int prec, const bt_attn_args* a);

/* half-precision attention on FRAGMENT-MAJOR operands (csrc/attn2.hip): per (sequence, head) `nbp` blocks of
 * 32 tokens, 2 KB each.  Q/K block: [quarter a][token][8 dims 8a..8a+7]; V block: [s][lane = 32 g + d]
 * [8 tokens 16 s + 8 (j >> 2) + 4 g + (j & 3)]; gates [n_seq * heads][nbp * 32] fp32.  q must be
 * pre-scaled by log2(e)/sqrt(32).  nbp >= bt_attn_frag_blocks(L).  Output as bt_attention (half).
 * x3 <= 0 (half operands): 0 / -1 = the 128-query kernel (what the forward runs); -2 = two query blocks per wave on a
 * hand-scheduled key loop (round 6: same bits for every query, 2.7 % slower at 0.9 % fewer joules -- the fp16 attention is
 * bound by its exponentials; kept for tests and probes).
 * x3 > 0 (BT_PREC_F32X3): blocks of 4 KB = [hi block | lo block] of the fp32 values, three MFMAs per product; output
 * fp32 [rows, inner] (out_f32 = 1), hl32 half [rows, 2 inner] (0) or hl8 rows of the same size (2: per 32 columns 32 hi halves |
 * 32 e4m3 bytes of value / 8 | 32 e4m3 bytes of 2^8 (value - hi), what bt_gemm3 reads as A with x3 flag 0x100); status (may be NULL) =
 * range flag of the hl32 / hl8 output;
 * x3 = 1: 128-key LDS tiles, x3 = 2: 64-key tiles, x3 = 5: two query blocks per wave on a hand-scheduled key loop -- the same
 * arithmetic in all three, bit-identical results; x3 = 4 (the forward's choice since round 4): 5 where its 256-query
 * workgroups cost fewer per-CU rounds than the 128-query ones of 2 (launch_attn_frag), 2 otherwise.  + 8 (BT_X3_P16): the P16
 * arithmetic (BT_OPT_X3_ATTN_P16) on the same kernel choice.
 * x3 > 0 needs `scratch`: n_seq * heads * nbp int32 words (the launch's overflow map: queries whose probabilities left fp16's
 * range in the fast pass are recomputed on their row maxima by a second, gathered launch; contents undefined afterwards). */
#define BT_X3_P16 8
typedef struct {
  const void* q; const void* k; const void* v; const float* gates; void* out;
  int32_t n_seq, L, heads, inner, nbp, o_div; int64_t o_outer, o_inner, o_tok;
  int32_t x3, out_f32; int32_t* status;
  int32_t* scratch;
} bt_attn_frag_args;
/* BASELINE.json config 5 ("final0 fp8 MFMA weights ... CDNA4 fp8 attention/FFN path") at OPERATOR level, report-only: the GEMM of a
 * main layer's to_qkv / to_out / FeedForward linears (roformer.py:38-61,99-132) with both operands in the OCP MX e4m3 format --
 * d_A [M][K] and d_W [N padded to 128][K] e4m3 bytes, d_SA [M][K / 32] and d_SW [N padded to 128][K / 32] E8M0 scale bytes (value
 * = 2^(byte - 127) x element), K = 512 | 1024 | 2048, N % 64 == 0 -- on v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulation,
 * d_out fp32 [M][ldo].  No forward calls it: it is measured beside the fp16 GEMM (tools/mx8_probe.py) and its arithmetic is
 * priced on the oracle (tools/flip_soak.py sim --schemes mxfp8) -- DESIGN.md, config 5. */
int bt_gemm_mx8(void* stream, const void* d_A, const void* d_SA, const void* d_W, const void* d_SW, float* d_out, int M, int N,
                int K, int64_t ldo);

/* half GEMM of the main layers (csrc/gemm3.hip), single-operator entry for the parity tests.
 * epi 0: out[M,ldo] (half) = gelu(rms(A) W^T + bias);  epi 1: x[M,ldx] (fp32) += A W^T + bias, half shadow xb,
 * partial row sums of squares ssq_out[N/64][M];  epi 2: q|k|v|gates = rms(A) W^T with RoPE / sigmoid, written
 * fragment-major (layout of bt_attention_frag) for n_seq sequences of L tokens (M = n_seq L).
 * rms(A) uses ssq_in[ssq_parts][M] (partial row sums of squares of the fp32 source of A), NULL = no RMSNorm.
 * x3 != 0 (BT_PREC_F32X3): A half [M, 2 lda], W half [N padded to 256, 2 K], out half [M, 2 ldo], xb half [M, 2 ldx]
 * are hl32 (interleaved hi / lo planes of the fp32 values; K, lda, ldo, ldx count fp32 elements), q / k / v blocks are
 * 4 KB, epi 0 uses the exact erf GELU; status (may be NULL) = range flag.  x3 + 0x100: A and W are hl8 (see
 * bt_pair_weights.w_ff1_f8); x3 + 0x200: the activation the launch writes for the next GEMM (epi 0: out,
 * epi 1: xb) is hl8. */
typedef struct {
  const void* A; int64_t lda; int32_t M, K; const void* W; int32_t N, epi; const float* bias;
  const float* ssq_in; int32_t ssq_parts; void* out; int64_t ldo; float* x; int64_t ldx; void* xb; float* ssq_out;
  int32_t n_seq, L, nbp, heads; const float* rope; void* qf; void* kf; void* vf; float* gates; const float* b_gates;
  int32_t no_resid; /* epi 1: x = A W^T + bias, x is only written (frontend.linear) */
  /* epi 1 as the frontend convolution (beat_tracker.py:155-166, BatchNorm folded): gelu != 0 -> x = gelu(.. + bias) (tanh
   * form; exact erf form with x3), x may be NULL (shadow output xb only); conv_C2 = 2 C > 0 -> A is the (b, t, f, c)
   * activation shadow seen as [M = B T F/2, conv_C2], lda = conv_C2, K = 3 conv_C2: the rows m - conv_F, m, m + conv_F
   * (time taps t-1, t, t+1 with conv_F = F/2 rows per time step), rows with t outside [0, conv_T) read as zeros.  Needs
   * no_resid. */
  int32_t gelu, conv_C2, conv_T, conv_F;
  int32_t x3; int32_t* status;
} bt_gemm3_args;
int bt_gemm3(void* stream, const bt_gemm3_args* a);
int bt_attn_frag_blocks(int L);
int bt_attention_frag(void* stream, const bt_attn_frag_args* a);
/* Time-direction QKV projection of a frontend block: d_x [B,T,F,C] fp32 -> fragment-major q, k, v, gates
 * (prec = BT_PREC_HALF: 2 KB blocks from w_qkv_frag; BT_PREC_F32X3: 4 KB [hi | lo] blocks from w_qkv_frag_x3) */
int bt_qkv_front(void* stream, int prec, const bt_pair_weights* w, const float* d_rope, const float* d_x, int B, int T, int F,
                 void* d_q, void* d_k, void* d_v, float* d_gates, int nbp);
/* x[M,C] += FF(x) with one bt_pair_weights, dim = C <= 128 */
int bt_ff_fused(void* stream, int prec, const bt_pair_weights* w, float* d_x, int64_t M);
/* fused halves (csrc/fused2.hip): x += to_out(ao) then x += FF(x);  x += AttnF(x) then x += FF(x).
 * d_xb (may be NULL): shadow of the new x for the following convolution -- half [M, dim] (BT_PREC_HALF) or hl32 half
 * [M, 2 dim] (BT_PREC_F32X3) */
int bt_outff_fused(void* stream, int prec, const bt_pair_weights* w, const void* d_ao, float* d_x, int64_t M, void* d_xb);
int bt_attnff_fused(void* stream, int prec, const bt_pair_weights* w, const float* d_rope, float* d_x, int64_t M);
/* fused tail of a main layer (csrc/tail.hip, BT_PREC_HALF, w->dim = 256 / 512, w->w_tail_frag set): d_x [M, dim] fp32
 * += to_out(d_ao [M, dim] half), then += FF(x); optional outputs: d_xb = half copy of the new x, d_ssq_out [dim/64][M] =
 * partial row sums of squares of the new x (what the next layer's QKV projection consumes) */
int bt_layer_tail(void* stream, const bt_pair_weights* w, int hidden, const void* d_ao, float* d_x, int64_t M, void* d_xb,
                  float* d_ssq_out);

#ifdef __cplusplus
}
#endif
#endif
