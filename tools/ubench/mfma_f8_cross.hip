// The arithmetic a config-5 path would run, checked on one 32 x 32 x 32 tile:
//   a . w  ~  hi(a) . hi(w)            two v_mfma_f32_32x32x16_f16
//           + 2^-11 [ e4m3(a) . e4m3(2^11 lo(w)) + e4m3(2^11 lo(a)) . e4m3(w) ]
//                                       ONE v_mfma_scale_f32_32x32x64_f8f6f4: A' = [hi bytes | lo bytes], W' = [lo bytes | hi bytes]
//                                       (lane half 0 supplies the first 32 bytes, lane half 1 the second), the 2^-11 in the E8M0 scale
// into the SAME fp32 accumulator.  Questions: (1) does v_cvt_pk_fp8_f32 give OCP e4m3 on gfx950, (2) does the scale operand apply as
// 2^(byte - 127) per instruction, (3) how close is the result to the exact product next to the three-fp16-MFMA form, (4) the rate of the
// scaled fp8 MFMA against the fp16 one.   hipcc --offload-arch=gfx950 -O2 mfma_f8_cross.hip -o mfma_f8_cross
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 hf;
typedef __attribute__((ext_vector_type(8))) hf hfx8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned char to_e4m3(float v) {
  const int w = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
  return (unsigned char)(w & 0xff);
}

// a, w: [32 rows][32 k] fp32.  out: [32][32] = a . w^T in the three forms; bytes: the e4m3 encodings for the host check
__global__ void cross(const float* a, const float* w, float* d_f8, float* d_x3, unsigned char* bytes) {
  const int lane = threadIdx.x, g = lane >> 5, lr = lane & 31;
  // fp16 fragments: lane half g owns k [8 g, 8 g + 8) of each 16-k piece (the kernels' convention)
  hfx8 ah[2], al[2], wh[2], wl[2];
  for (int m = 0; m < 2; ++m)
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * m + 8 * g + i;
      const float av = a[lr * 32 + k], wv = w[lr * 32 + k];
      ah[m][i] = (hf)av; al[m][i] = (hf)(av - (float)(hf)av);
      wh[m][i] = (hf)wv; wl[m][i] = (hf)(wv - (float)(hf)wv);
    }
  // fp8 operands: A' lane half 0 = hi bytes of its row (32 k), half 1 = lo bytes; W' the other way round
  i32x8 a8, w8;
  for (int j = 0; j < 8; ++j) {
    unsigned pa = 0, pw = 0;
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * j + i;
      const float av = a[lr * 32 + k], wv = w[lr * 32 + k];
      const float alo = (av - (float)(hf)av) * 2048.f, wlo = (wv - (float)(hf)wv) * 2048.f;
      const unsigned char ba = g == 0 ? to_e4m3(av) : to_e4m3(alo);
      const unsigned char bw = g == 0 ? to_e4m3(wlo) : to_e4m3(wv);
      pa |= (unsigned)ba << (8 * i);
      pw |= (unsigned)bw << (8 * i);
      if (g == 0) { bytes[lr * 32 + k] = to_e4m3(av); bytes[2048 + lr * 32 + k] = to_e4m3(wlo); }
      else { bytes[1024 + lr * 32 + k] = to_e4m3(alo); bytes[3072 + lr * 32 + k] = to_e4m3(wv); }
    }
    a8[j] = (int)pa; w8[j] = (int)pw;
  }
  f32x16 c, x;
  for (int r = 0; r < 16; ++r) c[r] = x[r] = 0.f;
  // cross terms: scale_a = 2^-11 (E8M0 byte 116), scale_b = 1 (127)
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, c, 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
  for (int m = 0; m < 2; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], wh[m], c, 0, 0, 0);
  // the shipped form: lo . hi + hi . lo + hi . hi on fp16 MFMAs
  for (int m = 0; m < 2; ++m) x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], wh[m], x, 0, 0, 0);
  for (int m = 0; m < 2; ++m) x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], wl[m], x, 0, 0, 0);
  for (int m = 0; m < 2; ++m) x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], wh[m], x, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * g;   // A row; lr = W row
    d_f8[row * 32 + lr] = c[r];
    d_x3[row * 32 + lr] = x[r];
  }
}

template <int MODE>
__global__ void rate(float* out, int iters) {
  i32x8 a, b;
  hfx8 h0, h1;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101; b[i] = 0x3a3a3a3a + i; h0[i] = (hf)(0.5f + threadIdx.x); h1[i] = (hf)(1.5f + i); }
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // fp16: K = 16 per issue
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, c3, 0, 0, 0);
    } else {           // scaled fp8: K = 64 per issue
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const double f = e == 0 ? ldexp((double)m, -9) : ldexp(1.0 + m / 8.0, e - 7);
  return s ? -f : f;
}

int main() {
  std::vector<float> a(1024), w(1024);
  srand(3);
  auto rnd = [] { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return u - 6.0; };
  for (auto& v : a) v = (float)(1.3 * rnd());          // activations of order 1
  for (auto& v : w) v = (float)(0.05 * rnd());         // weights of order 1 / sqrt(K)
  float *da, *dw, *d8, *d3; unsigned char* db;
  hipMalloc(&da, 4096); hipMalloc(&dw, 4096); hipMalloc(&d8, 4096); hipMalloc(&d3, 4096); hipMalloc(&db, 4096);
  hipMemcpy(da, a.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), 4096, hipMemcpyHostToDevice);
  cross<<<1, 64>>>(da, dw, d8, d3, db);
  std::vector<float> r8(1024), r3(1024); std::vector<unsigned char> by(4096);
  hipMemcpy(r8.data(), d8, 4096, hipMemcpyDeviceToHost); hipMemcpy(r3.data(), d3, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(by.data(), db, 4096, hipMemcpyDeviceToHost);
  double e8 = 0, e3 = 0, esem = 0, mag = 0, enc = 0;
  for (int k = 0; k < 1024; ++k) enc = fmax(enc, fabs(e4m3(by[k]) - a[k]) / fmax(fabs(a[k]), 1e-3));   // hi bytes of a: <= 2^-4 relative
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double exact = 0, sem = 0;
      for (int k = 0; k < 32; ++k) {
        const double av = a[i * 32 + k], wv = w[j * 32 + k];
        exact += av * wv;
        const double ahi = (double)(float)(_Float16)a[i * 32 + k], whi = (double)(float)(_Float16)w[j * 32 + k];
        sem += ahi * whi + ldexp(e4m3(by[i * 32 + k]) * e4m3(by[2048 + j * 32 + k]) + e4m3(by[1024 + i * 32 + k]) * e4m3(by[3072 + j * 32 + k]), -11);
      }
      e8 = fmax(e8, fabs(r8[i * 32 + j] - exact)); e3 = fmax(e3, fabs(r3[i * 32 + j] - exact));
      esem = fmax(esem, fabs(r8[i * 32 + j] - sem)); mag = fmax(mag, fabs(exact));
    }
  printf("e4m3 encoding of a (v_cvt_pk_fp8_f32 vs OCP decode): max relative error %.3f (<= 0.0625 expected)\n", enc);
  printf("instruction semantics (device result vs host evaluation of the same bytes, scale 2^-11): max |diff| %.3e of max |a.w| %.3f\n", esem, mag);
  printf("against the exact product: fp8 cross terms %.3e, three fp16 MFMAs %.3e (relative to max |a.w|: %.2e, %.2e)\n", e8, e3, e8 / mag, e3 / mag);
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (mode == 0) rate<0><<<2048, 256>>>(out, 100); else rate<1><<<2048, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (mode == 0) rate<0><<<2048, 256>>>(out, iters); else rate<1><<<2048, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double kper = mode == 0 ? 16 : 64;
    const double flop = 2048.0 * 4 * iters * 4 * 2.0 * 32 * 32 * kper;
    printf("%s: %.1f TFLOP/s (%.2f ms); 32x32 tile-k per second %.3e\n", mode == 0 ? "v_mfma_f32_32x32x16_f16        " : "v_mfma_scale_f32_32x32x64_f8f6f4", flop / ms / 1e9, ms,
           2048.0 * 4 * iters * 4 * kper / (ms * 1e-3));
  }
  return 0;
}
