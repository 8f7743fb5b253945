// Internal launch interface between the engine (engine.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BT_PREC_F32
#define BT_PREC_F32 0
#define BT_PREC_HALF 1
#endif
#ifndef BT_PREC_F32X3
#define BT_PREC_F32X3 3   // (include/beat_this_amd.h)
#endif

// ---- GEMM: C[M,N] = epilogue( A[M,K] . W[N,K]^T ) ------------------------------------
enum {
  GEMM_EPI_STORE = 0,  // [*rowscale] [+bias] [gelu] -> out (compute dtype, or fp32 if OUT_F32)
  GEMM_EPI_RESID = 1,  // [+bias] + x -> x (fp32, in place)
  GEMM_EPI_QKV = 2     // *rowscale, RoPE on q|k, sigmoid(+bias) on gate columns
};
enum {
  GEMM_F_RMS = 1,      // accumulate row sum-of-squares of A while staging, scale rows in the epilogue
  GEMM_F_BIAS = 2,
  GEMM_F_GELU = 4,
  GEMM_F_OUT_F32 = 8,  // EPI_STORE writes fp32 even in half mode
  GEMM_F_A_F32 = 16,   // A is fp32 in memory even in half mode (residual stream)
  GEMM_F_CONV = 32,    // A rows are gathered: 3 time taps x C2 contiguous channels (implicit GEMM conv)
  GEMM_F_ROWMAP = 64   // QKV: store rows in (b,f,t) order instead of (b,t,f)
};

struct GemmP {
  const void* A;
  long lda;        // elements
  const void* W;   // [N padded to 128, K] compute dtype
  int M, N, K;     // N = number of real output columns
  int epi, flags;
  const float* bias;
  void* out;
  long ldo;
  float* x;        // EPI_RESID: residual stream, ld = ldx
  long ldx;
  void* xb;        // optional half shadow of the fp32 result (EPI_RESID: of x; OUT_F32: of out), same ld
  float* ssq_out;  // optional (gemm2 fp32 epilogues): [N / 64][M] partial row sums of squares of the fp32 result
  // conv gather (GEMM_F_CONV): output row m = (b, t, f'), A = x[b, t-1..t+1, f', 0..C2)
  int conv_C2, conv_T, conv_F;
  // QKV epilogue
  float* gates;    // [rows, heads] fp32
  int inner;       // heads * 32
  int heads;
  const float* rope;  // [pos][16][2] (cos, sin) fp32
  int pdiv, pmod;  // pos = (m / pdiv) % pmod
  int map_T, map_F;  // GEMM_F_ROWMAP
};
int launch_gemm(const GemmP& p, int prec, hipStream_t s);

// ---- half / hi + lo GEMM of the main layers (gemm3.hip): LDS-DMA ring, transposed product, register epilogues --------
enum { G3_FF1 = 0, G3_RESID = 1, G3_QKV = 2 };
struct Gemm3P {
  const void* A; long lda; int M, K;   // half [M, lda]
  const void* W; int N;                // half [N padded to 128, K]
  int epi;
  const float* bias;                   // FF1: [N]; RESID: [N] or null
  const float* ssq_in; int ssq_parts;  // RMSNorm of A: [parts][M] partial row sums of squares, or null
  void* out; long ldo;                 // FF1: half [M, ldo] = gelu(rms(A) W^T + bias)
  float* x; long ldx; void* xb;        // RESID: x += A W^T + bias (fp32, in place), xb = half shadow (or null)
  float* ssq_out;                      // RESID: [N / 64][M] partial sums of squares of the new x (or null)
  // QKV: rows are (sequence, token) with n_seq sequences of L tokens (M = n_seq * L); columns q | k | v | gates
  int n_seq, L, nblk, nbp, heads, inner;
  const float* rope; void* qf; void* kf; void* vf; float* gates; const float* b_gates;
  int no_resid;  // RESID: x = A W^T + bias (x is only written: frontend.linear, frontend convs)
  int gelu;      // RESID: x = gelu(... + bias) (tanh form; frontend convs).  x may be null then (half output xb only)
  // RESID, implicit-GEMM convolution (frontend convs, beat_tracker.py:155-166): conv_C2 > 0 -> A is the half shadow of the
  // (b, t, f, c) activation seen as [M = B T F/2, C2 = 2 C]; K = 3 C2 = the rows m - conv_F, m, m + conv_F (time taps
  // t-1, t, t+1; conv_F = F/2 rows per time step), rows outside 0 <= t < conv_T read as zeros
  int conv_C2, conv_T, conv_F;
  // BT_PREC_F32X3: x3 != 0 -> A, W and every activation output (FF1: out; RESID: the shadow xb; QKV: qf / kf / vf) are
  // fp32 values stored as interleaved hi / lo half planes ("hl32", gemm3.hip header): A = half [M, 2 lda], W = half
  // [N padded, 2 K], out = half [M, 2 ldo], xb = half [M, 2 ldx], attention blocks of 4 KB ([hi 2 KB | lo 2 KB]); K, lda,
  // ldo, ldx stay in fp32 elements.  FF1 uses the exact erf GELU.  status (may be null): set to 1 when a value beyond
  // the fp16 range went through a hi + lo split (the caller then repeats the forward in BT_PREC_F32).
  // x3 + G3_X3_F8: A and W are "hl8" (per 32 columns 32 hi halves | 32 hi bytes | 32 lo bytes: gemm3.hip, X3 = 2) -- FF1 and
  // RESID epilogues only; x3 + G3_X3_OUT_F8: the activation this launch writes for the next GEMM (FF1: out; RESID: xb) is hl8.
  int x3;
  int* status;
};
enum { G3_X3_F8 = 0x100, G3_X3_OUT_F8 = 0x200 };
bool gemm3_supported(const Gemm3P& p);
int launch_gemm3(const Gemm3P& p, hipStream_t s);

// ---- MX e4m3 GEMM (gemm_mx8.hip): BASELINE config 5 at operator level, report-only ------------------------------------------
struct GemmMx8P {
  const void* A; const void* SA;   // e4m3 bytes [M][K], E8M0 block scales [M][K / 32]
  const void* W; const void* SW;   // e4m3 bytes [N padded to 128][K], scales [N padded to 128][K / 32]
  float* out; long ldo;            // fp32 [M][ldo]
  int M, N, K;                     // K = 512 | 1024 | 2048
};
bool gemm_mx8_supported(const GemmMx8P& p);
int launch_gemm_mx8(const GemmMx8P& p, hipStream_t s);

// ---- attention ---------------------------------------------------------------------
struct AttnP {
  const void* qkv;    // [n_seq * L rows, ld] compute dtype; q | k | v column blocks of width inner
  long ld;
  const float* gates; // [n_seq * L, heads]
  void* out;          // compute dtype, [rows, inner]
  int n_seq, L, heads, inner;
  // output row of token t of sequence s:  (s / o_div) * o_outer + (s % o_div) * o_inner + t * o_tok
  int o_div;
  long o_outer, o_inner, o_tok;
};
int launch_attn_flash(const AttnP& p, int prec, hipStream_t s);

// half attention on fragment-major operands (attn2.hip; layout described there)
struct AttnFragP {
  const void* q; const void* k; const void* v;  // [n_seq * heads][nbp][1024] half
  const float* gates;                            // [n_seq * heads][nbp * 32] fp32
  void* out;                                     // half [rows, inner], row mapping as AttnP
  int n_seq, L, heads, inner, nbp;
  int o_div;
  long o_outer, o_inner, o_tok;
  // BT_PREC_F32X3: x3 != 0 -> q, k, v blocks are 4 KB ([hi block | lo block], attn2.hip); out is fp32 [rows, inner]
  // (out_f32 = 1), hl32 planes half [rows, 2 inner] (0) or hl8 rows (2; gemm3.hip X3 = 2); status: range flag of the hl32 output (may be null).
  // x3 = 1: 128-key LDS tiles, x3 = 2: 64-key tiles (same results)
  int x3, out_f32;
  int* status;
  // x3: the launch's overflow map, [n_seq * heads][nbp] words (one per query block, bit = query whose fast pass overflowed fp16:
  // written by the attention kernel, consumed by the fix-up launch that follows it inside launch_attn_frag; scratch, required)
  int* fix_mask;
};
int attn_frag_blocks(int L);  // 32-token blocks to allocate per (sequence, head): ceil(L/32) rounded up to a tile
int launch_attn_frag(const AttnFragP& p, hipStream_t s);

// time-direction QKV projection of the frontend (qkv_front.hip), half, fragment-major outputs
struct QkvFrontP {
  const float* x;        // residual stream [B, T, F, C] fp32
  int B, T, F, C;
  const void* wfrag;     // bt_pair_weights.w_qkv_frag
  const float* b_gates;  // [C / 32]
  const float* rope;     // [pos][16][2]
  void* q; void* k; void* v;  // [B * F * heads][nbp][1024] half
  float* gates;          // [B * F * heads][nbp * 32]
  int nbp;
  // BT_PREC_F32X3: x3 != 0 -> wfrag = bt_pair_weights.w_qkv_frag_x3, output blocks of 4 KB ([hi block | lo block]); status:
  // range flag (may be null)
  int x3;
  int* status;
};
int launch_qkv_front(const QkvFrontP& p, hipStream_t s);

// ---- register-chained fused frontend blocks (fused.hip), C in {32, 64, 128} ----------------------
struct FusedFFP {
  float* x; long M; int C;           // residual stream [M, C] fp32, updated in place
  const void* wfrag;                 // W1 (gamma folded) and PERM32'd W2 in fragment-major order (fused.hip)
  const float* b1; const float* b2;  // [4C], [C]
  void* xb;                          // optional half shadow of the updated x (same layout), may be null
};
int launch_ff_fused(const FusedFFP& p, int prec, hipStream_t s);

// ---- fused halves of a PartialFTTransformer (fused2.hip): x read once, written once -----------------------------
struct FusedOutFFP {  // x += Wout . ao ; x += FF(x)      (time direction, after the flash attention)
  float* x; long M; int C;
  const void* ao;                    // attention output [M, C], compute dtype
  const void* wfrag;                 // bt_pair_weights.w_outff_frag
  const float* b1; const float* b2;
  void* xb;                          // optional half shadow of the new x
  int abl;                           // development (BT_F2_ABL)
};
int launch_outff_fused(const FusedOutFFP& p, int prec, hipStream_t s);
struct FusedAttnFFP {  // x += AttnF(x) ; x += FF(x)      (frequency direction)
  float* x; long M; int C;
  const float* b_gates; const float* rope;
  const void* wfrag;                 // bt_pair_weights.w_attnff_frag
  const float* b1; const float* b2;
};
int launch_attnff_fused(const FusedAttnFFP& p, int prec, hipStream_t s);

// ---- tail of a main layer (tail.hip): x += to_out(ao); x += FF(x) in one launch, C = 256 / 512, half operands ------
struct LayerTailP {
  float* x; long M; int C; int hidden;   // residual stream [M, C] fp32 (in place); hidden = ff_mult * C
  const void* ao;                        // attention output [M, C], half
  const void* wfrag;                     // bt_pair_weights.w_tail_frag (stream layout: tail.hip header)
  const float* b1; const float* b2;      // [hidden], [C]
  void* xb;                              // half shadow of the new x [M, C] (or null)
  float* ssq_out;                        // [C / 64][M] partial row sums of squares of the new x (or null)
};
bool layer_tail_supported(int C, int hidden);
int launch_layer_tail(const LayerTailP& p, hipStream_t s);

// ---- small model kernels (frontend.hip) ---------------------------------------------
struct StemP {
  const float* spect;  // [B, T, 128]
  float* x;            // [B, T, 32, 32]  (b, t, f, c)
  const float* bn1_scale; const float* bn1_shift;  // [128]
  const float* w;      // [32][12] (co, df*3+dt) with bn2 scale folded
  const float* bias;   // [32] bn2 shift
  int B, T;
};
int launch_stem(const StemP& p, hipStream_t s);

struct HeadP {
  const float* x;      // [M, D]
  const float* w;      // [2, D] with final-norm gamma folded (prenorm: the plain task_heads weight)
  float b0, b1;
  float* beat; float* downbeat;  // [M]
  int M, D, sum_head;
  int prenorm;         // x is already normalised (BeatThis.task_heads called on its own)
  int* status;         // BT_PREC_F32X3 range flag (bit 1 is set when a logit is not finite), may be null
};
int launch_head(const HeadP& p, hipStream_t s);
int launch_clear_words(int* w, int n, hipStream_t s);   // n <= 64 words
// transformer_blocks' final RMSNorm as a pass of its own: y = x * sqrt(D) / |x| * gamma   (roformer.py:181, stage exit)
int launch_norm_out(const float* x, const float* gamma, float* y, long M, int D, hipStream_t s, int* status = nullptr);
int launch_finite_rows(const float* x, long M, int D, int* status, hipStream_t s);
// half shadow and per-64-column sums of squares [D/64][M] of a residual stream handed in from outside (stage entry)
// (hl32 != 0: the shadow in the BT_PREC_F32X3 form, half [M, 2 D] interleaved hi / lo planes)
int launch_shadow_ssq(const float* x, void* xb, float* ssq, long M, int D, hipStream_t s, int hl32 = 0);

// One track of a batched front-end launch: input samples, their count, and where the track's output starts in the
// concatenated output buffer (samples for the resampler, frames for the log-mel kernel) / how many outputs it has.
// Same layout as bt_span of include/beat_this_amd.h.
struct bt_span_t { const float* data; long n; long out_off; long n_out; };

// starts (one piece, frames relative to spect) or table ([B][4] absolute chunk table, see frontend.hip); neither: one piece
// whose chunk starts are computed on the device from (n_frames, B, T, border) -- launch_aggregate likewise
int launch_split(const float* spect, long n_frames, const int* starts, const int* table, int B, int T, float* chunks,
                 hipStream_t s, int border = 0);
int launch_aggregate(const float* cb, const float* cd, const int* starts, const int* table, const int* pieces, int n_pieces,
                     int B, int T, int border, long n_frames, float* beat, float* downbeat, hipStream_t s);
int launch_resample(const bt_span_t& one, const bt_span_t* tracks, int n_tracks, long max_n_out, int up, int down,
                    const float* h, int half, float* y, hipStream_t s);
// logits: n_arrays arrays of n frames (spans == nullptr) or arrays (offset, length) = spans[2 a], spans[2 a + 1];
// idx: ascending frame indices at the array's offset; count: [n_arrays]
int launch_peaks(const float* logits, long n, const int* spans, int n_arrays, int* idx, int* count, hipStream_t s);

// ---- log-mel -----------------------------------------------------------------------
struct LogmelP {
  bt_span_t one;               // single track: {audio, n_samples, 0, n_frames}
  const bt_span_t* tracks;     // batch: device table, blockIdx.y = track (one unused then)
  int n_tracks; long max_frames;
  const float* window;    // [1024]
  const float* twiddle;   // see logmel.hip
  const int* mel_start;   // [128]
  const int* mel_len;     // [128]
  const float* mel_w;     // [128][32]
  float* spect;           // [sum of n_frames, 128]: track k starts at row tracks[k].out_off
};
int launch_logmel(const LogmelP& p, hipStream_t s);
