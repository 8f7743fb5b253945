// Log-mel front end on the GPU: replaces LogMelSpect.forward (beat_this/preprocessing.py:56-59),
// i.e. torchaudio MelSpectrogram(sr 22050, n_fft 1024, hop 441, hann periodic, center/reflect,
// normalized="frame_length", power 1, 128 slaney mels 30..11000 Hz) followed by log1p(1000 x).
//
// One wavefront per frame.  The 1024-point real FFT is a 512-point complex FFT of the
// (even, odd) packed windowed samples plus the standard split step.  512 = 8 x 8 x 8: each of
// the 64 lanes holds 8 complex values and does three radix-8 butterflies in registers; data
// moves between the three stages through LDS (split re/im arrays, index padded by idx/8 so
// both the writes and the stride-9 reads are bank-conflict free).  |X|/32 goes to LDS, the
// mel projection is a banded gather (1004 non-zeros, <= 26 per filter), then log1p.
// HBM traffic = the waveform once (hop-overlapped re-reads hit L2) + the (frames,128) output.
#include "common.h"
#include "kernels.h"

namespace {

// Every LDS array of this kernel is private to ONE wave (index [wave]): the stages of a frame's FFT exchange data between
// the lanes of that wave only.  LDS instructions of a wave execute in issue order, so a write followed by another lane's
// read needs no s_barrier -- only that the compiler keeps the order (it must: the accesses may alias) and does not move
// them across the stage boundary.  Through round 2 each of the six exchanges was a workgroup-wide __syncthreads() that
// made four unrelated frames wait for each other; removing them changed nothing measurable (0.308 vs 0.311 ms for 6 x 300 s):
// the kernel is bound by its VALU instructions per frame (~1700 then, 598 of them the FFT's arithmetic; precise sqrtf /
// log1pf expansions, 64-bit reflect index math and a one-tap-per-iteration mel loop were the rest and are gone since),
// 24 waves per CU hide every wait.
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()

struct cplx { float re, im; };
DEVI cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
DEVI cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
DEVI cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
DEVI cplx mul_mi(cplx a) { return {a.im, -a.re}; }  // * (-i)

// in-place 8-point DFT, X[p] = sum_a x[a] exp(-2 pi i a p / 8)
DEVI void dft8(cplx (&x)[8]) {
  const float h = 0.70710678118654752440f;
  cplx t0 = cadd(x[0], x[4]), t1 = csub(x[0], x[4]);
  cplx t2 = cadd(x[2], x[6]), t3 = csub(x[2], x[6]);
  cplx t4 = cadd(x[1], x[5]), t5 = csub(x[1], x[5]);
  cplx t6 = cadd(x[3], x[7]), t7 = csub(x[3], x[7]);
  cplx e0 = cadd(t0, t2), e2 = csub(t0, t2);
  cplx e1 = cadd(t1, mul_mi(t3)), e3 = csub(t1, mul_mi(t3));
  cplx o0 = cadd(t4, t6), o2 = csub(t4, t6);
  cplx o1 = cadd(t5, mul_mi(t7)), o3 = csub(t5, mul_mi(t7));
  // twiddle the odd half: w^1 = (1 - i)/sqrt2, w^2 = -i, w^3 = (-1 - i)/sqrt2
  cplx w1 = {(o1.re + o1.im) * h, (o1.im - o1.re) * h};
  cplx w2 = mul_mi(o2);
  cplx w3 = {(o3.im - o3.re) * h, -(o3.re + o3.im) * h};
  x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
  x[1] = cadd(e1, w1); x[5] = csub(e1, w1);
  x[2] = cadd(e2, w2); x[6] = csub(e2, w2);
  x[3] = cadd(e3, w3); x[7] = csub(e3, w3);
}

constexpr int XPAD = 576;  // 512 + 512/8

// twiddle table layout (float2 each): [0,64) tw1[b*8+p] = e^{-2 pi i b p/64};
// [64,576) tw2[c*64+m] = e^{-2 pi i c m/512}; [576,1089) tw3[k] = e^{-2 pi i k/1024}, k=0..512
__global__ __launch_bounds__(256) void logmel_kernel(const LogmelP p) {
  __shared__ float sre[4][XPAD];
  __shared__ float sim[4][XPAD];
  __shared__ float smag[4][520];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bt_span_t tr = p.tracks ? p.tracks[blockIdx.y] : p.one;
  const float* __restrict__ audio = tr.data;
  if ((long)blockIdx.x * 4 >= tr.n_out) return;  // (workgroup-uniform: a shorter track of the batch)
  const long frame = (long)blockIdx.x * 4 + wave;
  const bool active = frame < tr.n_out;
  float* re = sre[wave];
  float* im = sim[wave];
  float* mag = smag[wave];
  const f32x2* tw = reinterpret_cast<const f32x2*>(p.twiddle);
  const long N = tr.n;
  const long s0 = (active ? frame : 0) * 441 - 512;

  // ---- load + window: lane holds z[64 a + lane], z[n] = x[2n] + i x[2n+1] -------------
  // Interior frames (all but the first two and the last two of a track; wave-uniform test): the 1024 samples are one
  // contiguous run, 32-bit indices, one 8-byte load per complex value (4-byte aligned: s0 is odd for odd frames).  Edge
  // frames take the reflecting path (center=True, pad_mode="reflect").
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  cplx x[8];
  if (s0 >= 0 && s0 + 1024 <= N) {
    const float* __restrict__ a0 = audio + s0 + 2 * lane;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const f32x2 w = *reinterpret_cast<const f32x2*>(p.window + 128 * a + 2 * lane);
      const f32x2u v = *reinterpret_cast<const f32x2u*>(a0 + 128 * a);
      x[a].re = v.x * w.x;
      x[a].im = v.y * w.y;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int n = 64 * a + lane;
      long j0 = s0 + 2 * n, j1 = j0 + 1;
      j0 = j0 < 0 ? -j0 : (j0 >= N ? 2 * (N - 1) - j0 : j0);
      j1 = j1 < 0 ? -j1 : (j1 >= N ? 2 * (N - 1) - j1 : j1);
      const f32x2 w = *reinterpret_cast<const f32x2*>(p.window + 2 * n);
      x[a].re = audio[j0] * w.x;
      x[a].im = audio[j1] * w.y;
    }
  }
  // ---- stage 1: DFT over a; lane = (b, c) = (lane >> 3, lane & 7) -----------------------
  dft8(x);
  {
    const int b = lane >> 3, c = lane & 7;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      f32x2 t = tw[b * 8 + q];
      x[q] = cmul(x[q], cplx{t.x, t.y});
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      re[q * 72 + b * 9 + c] = x[q].re;
      im[q * 72 + b * 9 + c] = x[q].im;
    }
  }
  WAVE_SYNC();
  // ---- stage 2: DFT over b; lane = (p, c) ------------------------------------------------
  {
    const int pp = lane >> 3, c = lane & 7;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      x[b].re = re[pp * 72 + b * 9 + c];
      x[b].im = im[pp * 72 + b * 9 + c];
    }
    dft8(x);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x2 t = tw[64 + c * 64 + pp + 8 * q];
      x[q] = cmul(x[q], cplx{t.x, t.y});
    }
    WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      re[q * 72 + pp * 9 + c] = x[q].re;
      im[q * 72 + pp * 9 + c] = x[q].im;
    }
  }
  WAVE_SYNC();
  // ---- stage 3: DFT over c; lane = (q, p) = (lane >> 3, lane & 7); Z[p + 8q + 64r] ---------
  {
    const int q = lane >> 3, pp = lane & 7;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      x[c].re = re[q * 72 + pp * 9 + c];
      x[c].im = im[q * 72 + pp * 9 + c];
    }
    dft8(x);
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // k = (p + 8 q) + 64 r = lane + 64 r
      re[lane + 64 * r] = x[r].re;
      im[lane + 64 * r] = x[r].im;
    }
  }
  WAVE_SYNC();
  // ---- split step: X[k], k = 0..512, magnitude / sqrt(1024) --------------------------------
  // (v_sqrt_f32 / v_log_f32 below are the 1-ulp hardware forms: the precise sqrtf / log1pf expansions were 200 of the
  // kernel's instructions per frame; log(1 + y) loses nothing that matters for y >= 0 -- its absolute error is one rounding
  // of 1 + y, 6e-8)
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int k = lane + 64 * j;
    if (k <= 512) {
      const int k0 = k & 511, m = (512 - k) & 511;
      const float a = re[k0], b = im[k0], c = re[m], d = im[m];
      const float er = 0.5f * (a + c), ei = 0.5f * (b - d);
      const float orr = 0.5f * (a - c), oi = 0.5f * (b + d);
      const f32x2 t = tw[576 + k];  // (cos, -sin)
      const float xr = er + (t.x * oi + t.y * orr);
      const float xi = ei - (t.x * orr - t.y * oi);
      mag[k] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi) * 0.03125f;
    } else if (k < 520) {
      mag[k] = 0.f;  // (the mel loop reads whole groups of four bins: up to three past a filter's last one, times weight 0)
    }
  }
  WAVE_SYNC();
  // ---- banded mel projection + log1p(1000 x) ------------------------------------------------
  // four taps per iteration: one 16-byte load of the filter's weight row (zero beyond its length), four LDS reads
  if (active) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = lane + 64 * h;
      const int st = p.mel_start[m], len = p.mel_len[m];
      const f32x4* wrow = reinterpret_cast<const f32x4*>(p.mel_w + m * 32);
      const float* mg = mag + st;
      float acc = 0.f;
      for (int i = 0; i < len; i += 4) {
        const f32x4 w = wrow[i >> 2];
        acc = fmaf(mg[i], w[0], acc);
        acc = fmaf(mg[i + 1], w[1], acc);
        acc = fmaf(mg[i + 2], w[2], acc);
        acc = fmaf(mg[i + 3], w[3], acc);
      }
      p.spect[(tr.out_off + frame) * 128 + m] = __builtin_amdgcn_logf(fmaf(1000.0f, acc, 1.0f)) * 0.69314718055994531f;
    }
  }
}

}  // namespace

int launch_logmel(const LogmelP& p, hipStream_t s) {
  if (p.max_frames <= 0 || p.n_tracks <= 0 || (!p.tracks && p.one.n <= 512)) return -2;
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)((p.max_frames + 3) / 4), (unsigned)p.n_tracks), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}
