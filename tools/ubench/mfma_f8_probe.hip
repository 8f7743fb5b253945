// Operand layout + rate check of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit (E8M0 = 127) scales.
// Hypothesis: lane l supplies row (l & 31), k = 32 (l >> 5) + [0, 32) as 32 consecutive bytes (8 VGPRs); C/D as bf16 32x32.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const unsigned char* A, const unsigned char* B, float* D) {
  const int lane = threadIdx.x, g = lane >> 5, lr = lane & 31;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(A + lr * 64 + 32 * g + 4 * i);
    b[i] = *reinterpret_cast<const int*>(B + lr * 64 + 32 * g + 4 * i);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * g;  // A row
    D[row * 32 + lr] = c[r];                        // lr = B row
  }
}

__global__ void rate(float* out, int iters) {
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838; }
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}

int main() {
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  srand(1);
  for (auto& v : A) { do v = rand() & 255; while ((v & 0x7f) == 0x7f); }
  for (auto& v : B) { do v = rand() & 255; while ((v & 0x7f) == 0x7f); }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  std::vector<float> D(32 * 32);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, mag = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int k = 0; k < 64; ++k) ref += (double)e4m3(A[i * 64 + k]) * e4m3(B[j * 64 + k]);
      worst = fmax(worst, fabs(ref - D[i * 32 + j]));
      mag = fmax(mag, fabs(ref));
    }
  printf("layout check: max |D - ref| = %g (max |ref| = %g) -> %s\n", worst, mag, worst <= 1e-3 * mag ? "OK" : "MISMATCH");
  // rate: 256 CUs x 4 SIMDs x 2 waves
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  const int iters = 20000;
  rate<<<2048, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate<<<2048, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2048.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 64;
  printf("rate: %.1f TFLOP/s (e4m3, 32x32x64 scaled, unit scales)\n", flop / ms * 1e-9);
  return 0;
}
