// Probe: lane layout of v_mfma_f32_4x4x4_16b_bf16 (which B values are summed into which lane's D registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ short bf(float x) { unsigned u = __float_as_uint(x); return (short)(u >> 16); }
__global__ void k(float* out, int mode) {
  const int l = threadIdx.x;
  s16x4 a, b;
  for (int i = 0; i < 4; ++i) {
    a[i] = mode == 0 ? (short)0x3f80 : bf((float)(i == (mode - 1) ? 1 : 0));   // ones, or unit vector e_{mode-1} over k
    b[i] = bf((float)(l * 4 + i));       // exactly representable small ints (<= 255)
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  float h[256];
  for (int mode = 0; mode < 5; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d (A = %s): lane: D[0..3]\n", mode, mode == 0 ? "ones" : "e_k");
    for (int l = 0; l < 64; ++l) { if (l < 10 || l >= 60 || l == 16 || l == 32) printf("  lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  return 0;
}
