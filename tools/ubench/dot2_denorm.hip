// Does v_dot2_f32_f16 keep fp16 subnormal inputs (as the MFMAs do), or flush them?  Decides whether the P16 row sums may run
// on it: numerator (MFMA) and denominator must see the same values.   hipcc --offload-arch=gfx950 -O2 dot2_denorm.hip -o dot2_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 hf;
typedef __attribute__((ext_vector_type(2))) hf hfx2;
typedef __attribute__((ext_vector_type(4))) hf hfx4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out) {
  const hfx2 one2 = {(hf)1.0f, (hf)1.0f};
  const hf a = (hf)9.5367431640625e-07f;   // 2^-20: fp16 subnormal
  const hf b = (hf)5.9604644775390625e-08f; // 2^-24: smallest fp16 subnormal
  hfx2 v = {a, b};
  asm volatile("" : "+v"(v));
  float d;
  asm volatile("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(d) : "v"(v), "v"(one2));
  const hfx4 ones = {(hf)1.0f, (hf)1.0f, (hf)1.0f, (hf)1.0f};
  hfx4 w = {a, b, (hf)0.f, (hf)0.f};
  f32x4 l = {0.f, 0.f, 0.f, 0.f};
  l = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, w, l, 0, 0, 0);
  // normal-range check and a large accumulator
  hfx2 n = {(hf)0.5f, (hf)0.25f};
  float d2;
  asm volatile("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(d2) : "v"(n), "v"(one2), "v"(1000.0f));
  if (threadIdx.x == 0) { out[0] = d; out[1] = l[0]; out[2] = d2; }
}
int main() {
  float* o; hipMalloc(&o, 16);
  k<<<1, 64>>>(o);
  float h[3]; hipMemcpy(h, o, 12, hipMemcpyDeviceToHost);
  printf("dot2(2^-20, 2^-24) = %.10e  (kept: %.10e)\nmfma4x4x4 same       = %.10e\ndot2(0.5, 0.25) + 1000 = %.6f\n", h[0], 9.5367431640625e-07 + 5.9604644775390625e-08, h[1], h[2]);
  return 0;
}
