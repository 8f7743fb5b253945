// Small bandwidth-bound kernels around the GEMM/attention core:
//   stem       BeatThis.make_stem   (beat_tracker.py:108-126)  BN1d -> conv(4x3,s(4,1)) -> BN2d -> GELU
//   head       final RMSNorm + SumHead (roformer.py:180, beat_tracker.py:315-330)
//   split      split_piece / zeropad  (inference.py:90-135)   chunk gather with zero padding
//   aggregate  aggregate_prediction, keep_first (inference.py:138-185)
//   peaks      postp_minimal peak mask + ordered compaction (postprocessor.py:93-99,119-120)
// Activation layout everywhere: (b, t, f, c) row-major, i.e. one (b,t) row = F*C = 1024 floats.
#include "common.h"
#include "kernels.h"

namespace {

// One workgroup per (b, 8 time steps): the BatchNorm1d'ed input rows t0-1 .. t0+8 are staged once in LDS (the three
// time taps of neighbouring steps share them); a thread owns 4 consecutive channels of one frequency position, so a
// wave-instruction stores 1 KB of contiguous output (8 positions x 32 channels).  (One workgroup per (b, t) with one
// output per lane and store -- the first version -- took 55 us for the 98 MB it writes.)
constexpr int STEM_TT = 8;
__global__ __launch_bounds__(256) void stem_kernel(const StemP p, int t_tiles) {
  __shared__ __attribute__((aligned(16))) float in[STEM_TT + 2][128];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / t_tiles, t0 = (blockIdx.x - b * t_tiles) * STEM_TT;
  for (int i = tid; i < (STEM_TT + 2) * 128; i += 256) {
    const int row = i >> 7, mel = i & 127;
    const int tt = t0 + row - 1;
    float v = 0.f;  // zero padding is applied AFTER BatchNorm1d (conv pads its own input)
    if (tt >= 0 && tt < p.T) v = fmaf(p.spect[((long)b * p.T + tt) * 128 + mel], p.bn1_scale[mel], p.bn1_shift[mel]);
    in[row][mel] = v;
  }
  const int c4 = tid & 7, f = tid >> 3;
  float w[4][12], bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 12; ++i) w[j][i] = p.w[(4 * c4 + j) * 12 + i];
    bias[j] = p.bias[4 * c4 + j];
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < STEM_TT; ++tt) {
    if (t0 + tt >= p.T) break;
    float a[4] = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&in[tt + dt][4 * f]);
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(w[j][df * 3 + dt], v[df], a[j]);
    }
    *reinterpret_cast<f32x4*>(p.x + ((long)b * p.T + t0 + tt) * 1024 + f * 32 + 4 * c4) =
        f32x4{gelu_erf(a[0]), gelu_erf(a[1]), gelu_erf(a[2]), gelu_erf(a[3])};
  }
}

// one wave per token row
__global__ __launch_bounds__(256) void head_kernel(const HeadP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const float* x = p.x + row * p.D;
  float ss = 0.f, d0 = 0.f, d1 = 0.f;
  if ((p.D & 3) == 0) {  // 16 bytes per lane: one wave-instruction covers 1 KB of the row
    for (int k = 4 * lane; k < p.D; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + k);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.w + k), w1 = *reinterpret_cast<const f32x4*>(p.w + p.D + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ss = fmaf(v[i], v[i], ss);
        d0 = fmaf(v[i], w0[i], d0);
        d1 = fmaf(v[i], w1[i], d1);
      }
    }
  } else {
    for (int k = lane; k < p.D; k += 64) {
      float v = x[k];
      ss = fmaf(v, v, ss);
      d0 = fmaf(v, p.w[k], d0);
      d1 = fmaf(v, p.w[p.D + k], d1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o);
    d0 += __shfl_xor(d0, o);
    d1 += __shfl_xor(d1, o);
  }
  if (lane == 0) {
    const float sc = p.prenorm ? 1.f : sqrtf((float)p.D) / fmaxf(sqrtf(ss), 1e-12f);
    const float y0 = d0 * sc + p.b0, y1 = d1 * sc + p.b1;
    p.beat[row] = p.sum_head ? y0 + y1 : y0;
    p.downbeat[row] = y1;
    // BT_PREC_F32X3 range guard, last line of defence: an operand that left the fp16 range upstream reaches the logits of
    // its chunk as inf / NaN (every token of a sequence meets every other in the attention)
    if (p.status && !(fabsf(y0) <= 3.0e38f && fabsf(y1) <= 3.0e38f)) atomicOr(p.status, 2);
  }
}

// one wave per token row: y = x * sqrt(D) / max(|x|, 1e-12) * gamma (F.normalize semantics, roformer.py:20-30)
// status (BT_PREC_F32X3 stage exit, may be null): bit 1 is set when a row's sum of squares is not finite (an operand that
// left the fp16 range of a hi half upstream reaches every token of its chunk as inf / NaN)
__global__ __launch_bounds__(256) void norm_out_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       float* __restrict__ y, long M, int D, int* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * D;
  float ss = 0.f;
  for (int k = lane; k < D; k += 64) ss = fmaf(xr[k], xr[k], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float sc = sqrtf((float)D) / fmaxf(sqrtf(ss), 1e-12f);
  for (int k = lane; k < D; k += 64) y[row * D + k] = xr[k] * sc * gamma[k];
  if (status && lane == 0 && !(ss <= 3.0e38f)) atomicOr(status, 2);
}

// BT_PREC_F32X3 stage exit after the frontend: the same test on the rows of the residual stream handed out as they are
__global__ __launch_bounds__(256) void finite_rows_kernel(const float* __restrict__ x, long M, int D, int* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float ss = 0.f;
  for (int k = lane; k < D; k += 64) ss = fmaf(x[row * D + k], x[row * D + k], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if (lane == 0 && !(ss <= 3.0e38f)) atomicOr(status, 2);
}

// one wave per token row, lane = column of a 64-column group: the half shadow of x and the group's sum of squares
// (hl32 != 0: the BT_PREC_F32X3 form of the shadow, per 32 columns [32 hi halves | 32 lo halves], gemm3.hip)
__global__ __launch_bounds__(256) void shadow_ssq_kernel(const float* __restrict__ x, hf* __restrict__ xb,
                                                         float* __restrict__ ssq, long M, int D, int hl32) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  for (int g = 0; g < D / 64; ++g) {
    const float v = x[row * D + g * 64 + lane];
    if (hl32) {
      const hf h = (hf)v;   // (v is a loaded value: nothing to contract into the conversions)
      hf* d = xb + (row * D + g * 64) * 2 + (lane >> 5) * 64 + (lane & 31);
      d[0] = h;
      d[32] = (hf)(v - (float)h);
    } else {
      xb[row * D + g * 64 + lane] = (hf)v;
    }
    float sq = v * v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (ssq && lane == 0) ssq[(long)g * M + row] = sq;
  }
}

// chunk table row (int32 x 4): {first source frame (may be negative / before the piece), first frame of the piece,
// end frame of the piece, unused}, all ABSOLUTE frame indices of the (concatenated) spectrogram buffer; frames of a
// chunk outside [lo, hi) read as zeros (split_piece / zeropad, inference.py:90-135).
// start frame of chunk b of a piece of n frames cut into B chunks of T frames (inference.py:120-125: arange(-border, n - border,
// T - 2 border), the last one moved back to n - (T - border) when the piece is longer than one chunk's fresh span)
__device__ __forceinline__ long piece_chunk_start(int b, int B, long n, int T, int border) {
  return (b == B - 1 && n > T - 2 * border) ? n - (T - border) : (long)b * (T - 2 * border) - border;
}

__global__ void split_kernel(const float* __restrict__ spect, const int* __restrict__ table, const int* __restrict__ starts,
                             long n_frames, int B, int T, float* __restrict__ chunks, int border) {
  const long total = (long)B * T * 32;  // float4 units
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i >> 5;
    int c4 = (int)(i & 31);
    int b = (int)(row / T), t = (int)(row - (long)b * T);
    long src, lo = 0, hi = n_frames;
    if (table) { src = (long)table[4 * b] + t; lo = table[4 * b + 1]; hi = table[4 * b + 2]; }
    else if (starts) src = (long)starts[b] + t;
    else src = piece_chunk_start(b, B, n_frames, T, border) + t;   // (one piece, starts computed here: bt_audio2beats_enqueue)
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (src >= lo && src < hi) v = reinterpret_cast<const f32x4*>(spect + src * 128)[c4];
    reinterpret_cast<f32x4*>(chunks + row * 128)[c4] = v;
  }
}

// aggregate_prediction, keep_first (inference.py:138-185).  Single piece: `starts` + n_frames.  Batch (blockIdx.y = piece):
// pieces[4 k ..] = {first frame, end frame (absolute, of the concatenated outputs), first chunk, end chunk}, chunk starts
// from the chunk table of split_kernel.
__global__ void aggregate_kernel(const float* __restrict__ cb, const float* __restrict__ cd,
                                 const int* __restrict__ starts, const int* __restrict__ table,
                                 const int* __restrict__ pieces, int B, int T, int border, long n_frames,
                                 float* __restrict__ beat, float* __restrict__ downbeat) {
  long f_lo = 0, f_hi = n_frames;
  int c_lo = 0, c_hi = B;
  if (pieces) {
    const int* pc = pieces + 4 * blockIdx.y;
    f_lo = pc[0]; f_hi = pc[1]; c_lo = pc[2]; c_hi = pc[3];
  }
  for (long i = f_lo + (long)blockIdx.x * blockDim.x + threadIdx.x; i < f_hi; i += (long)gridDim.x * blockDim.x) {
    float vb = -1000.0f, vd = -1000.0f;
    for (int c = c_lo; c < c_hi; ++c) {  // first (earliest) chunk whose kept span covers frame i wins
      long s = table ? table[4 * c] : starts ? starts[c] : piece_chunk_start(c, B, n_frames, T, border);
      if (i >= s + border && i < s + T - border) {
        vb = cb[(long)c * T + (i - s)];
        vd = cd[(long)c * T + (i - s)];
        break;
      }
    }
    beat[i] = vb;
    downbeat[i] = vd;
  }
}

// grid.x = number of logit arrays; one workgroup scans one array in order.  spans == nullptr: array a = logits + a n
// (n frames); else array a = logits + spans[2 a] with spans[2 a + 1] frames, indices written at idx + spans[2 a].
__global__ __launch_bounds__(1024) void peaks_kernel(const float* __restrict__ logits, long n_uniform,
                                                     const int* __restrict__ spans, int* __restrict__ idx,
                                                     int* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const long off = spans ? (long)spans[2 * blockIdx.x] : (long)blockIdx.x * n_uniform;
  const long n = spans ? (long)spans[2 * blockIdx.x + 1] : n_uniform;
  const float* x = logits + off;
  int* out = idx + off;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (long t0 = 0; t0 < n; t0 += 1024) {
    const long i = t0 + tid;
    bool flag = false;
    if (i < n) {
      const float v = x[i];
      float mx = v;
#pragma unroll
      for (int d = -3; d <= 3; ++d) {
        long j = i + d;
        if (j >= 0 && j < n) mx = fmaxf(mx, x[j]);
      }
      flag = (v == mx) && (v > 0.0f);
    }
    const unsigned long long mask = __ballot(flag);
    const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(mask);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      int c = wave_tot[w];
      if (w < wave) woff += c;
      tot += c;
    }
    const int base = base_s;
    if (flag) out[base + woff + prefix] = (int)i;
    __syncthreads();
    if (tid == 0) base_s = base + tot;
    __syncthreads();
  }
  if (tid == 0) count[blockIdx.x] = base_s;
}

// Rational resampler (replaces the host soxr.resample call of inference.py:274-275):
//   y[m] = sum_k x[k] * h[m * down + half - k * up],   h = up * (Kaiser-windowed sinc, beat_this_amd/tables.py),
// i.e. upsample by `up`, zero-phase FIR low-pass, keep every `down`-th sample (polyphase form, zero extension at the
// ends).  A workgroup produces 256 consecutive outputs of one track (blockIdx.y): the input window they need
// (256 down / up + 2 half / up samples) is staged in LDS once with coalesced loads, and -- for up == 1, the
// 44.1 -> 22.05 kHz case -- so is the filter; a thread then runs its taps out of LDS.  HBM traffic = the input once +
// the output once.
constexpr int RS_XMAX = 4096;  // staged input samples (floats)
constexpr int RS_HMAX = 1024;  // staged filter taps (up == 1 only)
__global__ __launch_bounds__(256) void resample_kernel(const bt_span_t one, const bt_span_t* __restrict__ tracks, int up,
                                                         int down, const float* __restrict__ h, int half,
                                                         float* __restrict__ y) {
  __shared__ float xs[RS_XMAX];
  __shared__ float hs[RS_HMAX];
  const bt_span_t tr = tracks ? tracks[blockIdx.y] : one;
  const float* __restrict__ x = tr.data;
  const long n_in = tr.n, n_out = tr.n_out;
  const long m0 = (long)blockIdx.x * 256;
  if (m0 >= n_out) return;
  const int tid = threadIdx.x;
  // inputs needed by outputs m0 .. m0 + 255: k in [ceil((m0 down - half) / up), floor(((m0 + 255) down + half) / up)]
  const long c_first = m0 * down - half, c_last = (m0 + 255) * down + half;
  long k0 = c_first <= 0 ? 0 : (c_first + up - 1) / up;
  long k1 = c_last / up;
  if (k1 > n_in - 1) k1 = n_in - 1;
  const int nx = (int)(k1 - k0 + 1);
  const bool staged = nx <= RS_XMAX;
  if (staged)
    for (int i = tid; i < nx; i += 256) xs[i] = x[k0 + i];
  const bool hstaged = up == 1 && 2 * half + 1 <= RS_HMAX;
  if (hstaged)
    for (int i = tid; i <= 2 * half; i += 256) hs[i] = h[i];
  __syncthreads();
  const long m = m0 + tid;
  if (m >= n_out) return;
  const long c = m * down + half;                 // h index of x[0]'s tap
  long k_lo = c - 2L * half <= 0 ? 0 : (c - 2L * half + up - 1) / up;  // smallest k with c - k up <= 2 half
  long k_hi = c / up;                             // largest k with c - k up >= 0
  if (k_hi > n_in - 1) k_hi = n_in - 1;
  float acc = 0.f;
  if (staged && hstaged) {
    const float* xp = xs + (k_lo - k0);
    const float* hp = hs + (c - k_lo);            // tap index falls by 1 per input sample
    const int nt = (int)(k_hi - k_lo + 1);
    for (int i = 0; i < nt; ++i) acc = fmaf(xp[i], hp[-i], acc);
  } else if (staged) {
    for (long k = k_lo; k <= k_hi; ++k) acc = fmaf(xs[k - k0], h[c - k * up], acc);
  } else {
    for (long k = k_lo; k <= k_hi; ++k) acc = fmaf(x[k], h[c - k * up], acc);
  }
  y[tr.out_off + m] = acc;
}


// Integer decimation (up == 1: 44.1 -> 22.05 kHz, the case of BASELINE's metric), register blocked.  Polyphase form:
//   y[m] = sum_r sum_q h_r[q] u_r[m - q],   h_r[q] = h[D q + r],   u_r[n] = x[n D + half - r]      (r = 0 .. D-1)
// A thread owns 8 consecutive outputs and walks the taps 8 at a time: the 15 inputs those 64 products need sit in two
// aligned 8-blocks of the phase-split LDS image, and the upper block of one tap group is the lower block of the next, so a
// group costs 2 ds_read_b128 of inputs + 2 (broadcast) of taps for 64 FMAs -- against 2 ds_read_b32 per FMA in
// resample_kernel, which spent 2.45 ms per 6 five-minute tracks with the 377-tap HQ filter.
constexpr int DEC_R = 8, DEC_OUT = 256 * DEC_R;  // outputs per thread / per workgroup
template <int D>
__global__ __launch_bounds__(256) void decimate_kernel(const bt_span_t one, const bt_span_t* __restrict__ tracks,
                                                         const float* __restrict__ h, int half, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float dec_lds[];
  const bt_span_t tr = tracks ? tracks[blockIdx.y] : one;
  const float* __restrict__ x = tr.data;
  const long n_in = tr.n, n_out = tr.n_out;
  const long mb = (long)blockIdx.x * DEC_OUT;
  if (mb >= n_out) return;
  const int tid = threadIdx.x;
  const int ntaps = 2 * half + 1;
  const int QP = ((ntaps + D - 1) / D + 7) & ~7;  // taps per phase, padded to whole groups of 8
  const int UW = DEC_OUT + QP;                     // staged samples per phase
  float* us = dec_lds;                             // [D][UW]:  us[r][k] = u_r[mb - QP + k]
  float* hs = dec_lds + D * UW;                    // [D][QP]:  h_r[q]
  for (int i = tid; i < D * QP; i += 256) {
    const int r = i / QP, q = i - r * QP;
    const int t = D * q + r;
    hs[i] = t < ntaps ? h[t] : 0.f;
  }
  // contiguous, coalesced sweep over the input window; sample g belongs to phase r = (half - g) mod D
  const long g0 = (mb - QP) * D + half - (D - 1);
  const int ng = UW * D;
  for (int i = tid; i < ng; i += 256) {
    const long g = g0 + i;
    const long e = g - half + (D - 1);            // = n D + (D - 1 - r),  n = k + mb - QP  ->  e - (mb - QP) D = i
    const int k = i / D, r = D - 1 - (i - k * D);
    (void)e;
    us[r * UW + k] = (g >= 0 && g < n_in) ? x[g] : 0.f;
  }
  __syncthreads();
  float acc[DEC_R];
#pragma unroll
  for (int j = 0; j < DEC_R; ++j) acc[j] = 0.f;
  const int groups = QP >> 3;
#pragma unroll 1
  for (int r = 0; r < D; ++r) {
    const float* ur = us + r * UW + 8 * tid + QP;   // hi block of tap group 0: u_r[mb + 8 tid .. + 7]
    const float* hr = hs + r * QP;
    f32x4 hi0 = *reinterpret_cast<const f32x4*>(ur), hi1 = *reinterpret_cast<const f32x4*>(ur + 4);
#pragma unroll 2   // (two groups per iteration: the hi <- lo hand-over becomes a renaming instead of four v_mov_b64 per 64 FMAs)
    for (int G = 0; G < groups; ++G) {
      const f32x4 lo0 = *reinterpret_cast<const f32x4*>(ur - 8 * (G + 1)), lo1 = *reinterpret_cast<const f32x4*>(ur - 8 * (G + 1) + 4);
      const f32x4 ha = *reinterpret_cast<const f32x4*>(hr + 8 * G), hb = *reinterpret_cast<const f32x4*>(hr + 8 * G + 4);
      const float w[16] = {lo0[0], lo0[1], lo0[2], lo0[3], lo1[0], lo1[1], lo1[2], lo1[3],
                           hi0[0], hi0[1], hi0[2], hi0[3], hi1[0], hi1[1], hi1[2], hi1[3]};
      const float hq[8] = {ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < DEC_R; ++j) acc[j] = fmaf(hq[e], w[8 + j - e], acc[j]);
      hi0 = lo0; hi1 = lo1;
    }
  }
  const long m0 = mb + 8L * tid;
  float* yo = y + tr.out_off + m0;
  if (m0 + 8 <= n_out && ((reinterpret_cast<uintptr_t>(yo) & 15) == 0)) {
    *reinterpret_cast<f32x4*>(yo) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(yo + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  } else {
#pragma unroll
    for (int j = 0; j < DEC_R; ++j)
      if (m0 + j < n_out) yo[j] = acc[j];
  }
}

}  // namespace

int launch_resample(const bt_span_t& one, const bt_span_t* tracks, int n_tracks, long max_n_out, int up, int down,
                    const float* h, int half, float* y, hipStream_t s) {
  const int ntaps = 2 * half + 1;
  if (up == 1 && (down == 2 || down == 3 || down == 4)) {  // integer decimation: the register-blocked kernel
    const int QP = ((ntaps + down - 1) / down + 7) & ~7;
    const size_t smem = (size_t)down * (DEC_OUT + QP) * 4 + (size_t)down * QP * 4;
    if (smem <= 64 * 1024) {
      dim3 grid((unsigned)((max_n_out + DEC_OUT - 1) / DEC_OUT), (unsigned)n_tracks);
      if (down == 2) hipLaunchKernelGGL(decimate_kernel<2>, grid, dim3(256), smem, s, one, tracks, h, half, y);
      else if (down == 3) hipLaunchKernelGGL(decimate_kernel<3>, grid, dim3(256), smem, s, one, tracks, h, half, y);
      else hipLaunchKernelGGL(decimate_kernel<4>, grid, dim3(256), smem, s, one, tracks, h, half, y);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((max_n_out + 255) / 256), (unsigned)n_tracks), dim3(256), 0, s, one,
                     tracks, up, down, h, half, y);
  return (int)hipGetLastError();
}

int launch_stem(const StemP& p, hipStream_t s) {
  const int t_tiles = (p.T + STEM_TT - 1) / STEM_TT;
  hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((long)p.B * t_tiles)), dim3(256), 0, s, p, t_tiles);
  return (int)hipGetLastError();
}
// (the range flag of a BT_PREC_F32X3 forward is cleared by a launch, not by hipMemsetAsync: see bt_forward_stages)
__global__ void clear_words_kernel(int* __restrict__ w, int n) {
  if ((int)threadIdx.x < n) w[threadIdx.x] = 0;
}
int launch_clear_words(int* w, int n, hipStream_t s) {
  hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(64), 0, s, w, n);
  return (int)hipGetLastError();
}
int launch_head(const HeadP& p, hipStream_t s) {
  hipLaunchKernelGGL(head_kernel, dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}
int launch_norm_out(const float* x, const float* gamma, float* y, long M, int D, hipStream_t s, int* status) {
  hipLaunchKernelGGL(norm_out_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, gamma, y, M, D, status);
  return (int)hipGetLastError();
}
int launch_finite_rows(const float* x, long M, int D, int* status, hipStream_t s) {
  hipLaunchKernelGGL(finite_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, M, D, status);
  return (int)hipGetLastError();
}
int launch_shadow_ssq(const float* x, void* xb, float* ssq, long M, int D, hipStream_t s, int hl32) {
  if (D % 64 != 0) return -2;
  hipLaunchKernelGGL(shadow_ssq_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, reinterpret_cast<hf*>(xb), ssq, M, D, hl32);
  return (int)hipGetLastError();
}
int launch_split(const float* spect, long n_frames, const int* starts, const int* table, int B, int T, float* chunks,
                 hipStream_t s, int border) {
  long total = (long)B * T * 32;
  unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, s, spect, table, starts, n_frames, B, T, chunks, border);
  return (int)hipGetLastError();
}
int launch_aggregate(const float* cb, const float* cd, const int* starts, const int* table, const int* pieces, int n_pieces,
                     int B, int T, int border, long n_frames, float* beat, float* downbeat, hipStream_t s) {
  unsigned grid = (unsigned)((n_frames + 255) / 256 < 2048 ? (n_frames + 255) / 256 : 2048);
  hipLaunchKernelGGL(aggregate_kernel, dim3(grid, (unsigned)(pieces ? n_pieces : 1)), dim3(256), 0, s, cb, cd, starts, table,
                     pieces, B, T, border, n_frames, beat, downbeat);
  return (int)hipGetLastError();
}
int launch_peaks(const float* logits, long n, const int* spans, int n_arrays, int* idx, int* count, hipStream_t s) {
  hipLaunchKernelGGL(peaks_kernel, dim3((unsigned)n_arrays), dim3(1024), 0, s, logits, n, spans, idx, count);
  return (int)hipGetLastError();
}
