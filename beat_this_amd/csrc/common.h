// Shared device helpers for the beat_this MI355X (gfx950 / CDNA4) kernels.
//
// Everything is written for wave64 + the 32x32 MFMA shapes:
//   bf16 operands : v_mfma_f32_32x32x16_bf16  (2 issues per 32-deep k-tile)
//   fp32 operands : v_mfma_f32_32x32x2_f32    (16 issues per 32-deep k-tile, exact fp32)
// C/D layout of both (MI355X guide, "Fragment layout"):
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),  reg in [0,16)
// A/B: lane l supplies row/col (l & 31); lane-half g = l >> 5 supplies half of the
// k-values of each issue.  The dot product is invariant under any relabelling of k
// that A and B share, so every kernel here uses ONE convention for a 32-deep k-tile:
//   lane-half g owns the 16 CONTIGUOUS k values [16 g, 16 g + 16)
// which turns every fragment read into wide contiguous LDS reads for both dtypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define DEVI __device__ __forceinline__

// One lane's share of a [32 rows] x [32 k] operand tile: 16 contiguous k values.
template <typename T> struct Frag;
template <> struct Frag<float> { f32x4 v[4]; };
template <> struct Frag<bf16> { bf16x8 v[2]; };

// LDS row pitch (bytes) of a 32-deep k-tile: +16 B pad makes the 16 B column slots of any
// 16 rows distinct (pitch/16 is odd), i.e. ds_read_b128 fragment reads are conflict free.
template <typename T> struct Tile { static constexpr int PITCH = 32 * (int)sizeof(T) + 16; };

template <typename T> DEVI Frag<T> ld_frag(const char* row_ptr, int g);
template <> DEVI Frag<float> ld_frag<float>(const char* row_ptr, int g) {
  Frag<float> f;
  const f32x4* p = reinterpret_cast<const f32x4*>(row_ptr + g * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) f.v[i] = p[i];
  return f;
}
template <> DEVI Frag<bf16> ld_frag<bf16>(const char* row_ptr, int g) {
  Frag<bf16> f;
  const bf16x8* p = reinterpret_cast<const bf16x8*>(row_ptr + g * 32);
  f.v[0] = p[0];
  f.v[1] = p[1];
  return f;
}

DEVI void mma32(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[i][j], b.v[i][j], acc, 0, 0, 0);
}
DEVI void mma32(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
}

// row of C/D register r for lane-half g
DEVI int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// Exact-erf GELU  x * Phi(x)  with Phi from the Abramowitz-Stegun 7.1.26 erfc polynomial
// (|error of erf| <= 1.5e-7): max |error| of the whole expression 4.2e-7 over [-12, 12] in fp32,
// below torch's own fp32 GELU (1.2e-6 vs float64).  ~13 VALU ops instead of ~35 for erff().
DEVI float gelu_erf(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316418882f, fabsf(x), 1.0f));  // 0.3275911 / sqrt(2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 0.5f * p * t * __builtin_amdgcn_exp2f(-0.7213475204f * x * x);  // exp(-x^2 / 2)
  return x * (x < 0.f ? e : 1.0f - e);
}
// GELU for the bf16 path: x * sigmoid(1.5957691 x (1 + 0.044715 x^2))  (the tanh form), 7 VALU ops (2 transcendental)
// instead of 14.  |gelu_tanh - gelu_erf| <= 4.8e-4 absolute for all x (1.2e-4 relative to max(|x|, 1)), i.e. an order of
// magnitude below the bf16 rounding (2^-9 relative) the result gets anyway; the fp32 path keeps gelu_erf.  The FF1
// epilogue of the main layers spent as many VALU cycles on gelu_erf as the MFMA pipe spent on the K = 512 product.
DEVI float gelu_tanh(float x) {
  const float x2 = x * x;
  const float z = x * fmaf(x2, -0.1029432f, -2.3022082f);   // -log2(e) * 1.5957691 * (1 + 0.044715 x^2)
  const float e = __builtin_amdgcn_exp2f(z);                 // exp(-u), u = 1.5957691 x (1 + 0.044715 x^2)
  return x * __builtin_amdgcn_rcpf(1.0f + e);                // inf -> 0, 0 -> x: both limits are exact
}
template <typename T> DEVI float gelu_t(float x);
template <> DEVI float gelu_t<float>(float x) { return gelu_erf(x); }
template <> DEVI float gelu_t<bf16>(float x) { return gelu_tanh(x); }
DEVI float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T> DEVI T from_f32(float x);
template <> DEVI float from_f32<float>(float x) { return x; }
template <> DEVI bf16 from_f32<bf16>(float x) { return (bf16)x; }

DEVI void st16(float* dst, const float* v) {  // 16 floats, 64 B aligned enough for 16 B stores
  f32x4* d = reinterpret_cast<f32x4*>(dst);
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
}

// (b,t,f) row index  <->  (b,f,t) row index  (time-direction partial transformer)
DEVI long btf_to_bft(long m, int T, int F) {
  long tf = (long)T * F;
  long b = m / tf;
  int rem = (int)(m - b * tf);
  int t = rem / F, f = rem - t * F;
  return (b * F + f) * (long)T + t;
}
