// Helpers of the register-chained kernels (fused.hip, qkv_front.hip): operand fragments from global
// memory / fp32 activations / accumulator registers / fragment-major LDS tiles.
#pragma once
#include "common.h"

// 16 contiguous elements from global memory as one lane's operand fragment
template <typename T> DEVI Frag<T> ldg_frag(const T* p);
template <> DEVI Frag<float> ldg_frag<float>(const float* p) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.v[i] = reinterpret_cast<const f32x4*>(p)[i];
  return f;
}
template <> DEVI Frag<hf> ldg_frag<hf>(const hf* p) {
  Frag<hf> f;
  f.v[0] = reinterpret_cast<const hfx8*>(p)[0];
  f.v[1] = reinterpret_cast<const hfx8*>(p)[1];
  return f;
}

// eight fp32 activations -> half-fragment h of Frag<hl>: the (hi, lo) half pairs, pre-scaled (common.h: OpScale), on the
// three-instructions-per-pair split of common.h (split_hl4)
DEVI void split_hl8(Frag<hl>& f, int h, const float (&v)[8]) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; j += 2)
    split_hl4(v[2 * j] * OpScale<hl>::ACT, v[2 * j + 1] * OpScale<hl>::ACT, v[2 * j + 2] * OpScale<hl>::ACT,
              v[2 * j + 3] * OpScale<hl>::ACT, hi[j], lo[j], hi[j + 1], lo[j + 1]);
  f.v[h] = __builtin_bit_cast(hfx8, u32x4{hi[0], hi[1], hi[2], hi[3]});
  f.v[2 + h] = __builtin_bit_cast(hfx8, u32x4{lo[0], lo[1], lo[2], lo[3]});
}
template <> DEVI Frag<hl> ldg_frag<hl>(const hl* p) {   // (memory holds fp32: hl is the operand type, not a storage type)
  Frag<hl> f;
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 a = q[2 * h], b = q[2 * h + 1];
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    split_hl8(f, h, v);
  }
  return f;
}

// 16 fp32 activations (this lane's token, channels [16 g, 16 g + 16) of one k-tile) -> operand
// fragment, accumulating the sum of squares for RMSNorm
template <typename T> DEVI Frag<T> ldx_frag(const float* p, bool ok, float& ss);
template <> DEVI Frag<float> ldx_frag<float>(const float* p, bool ok, float& ss) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f.v[i] = ok ? reinterpret_cast<const f32x4*>(p)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) ss = fmaf(f.v[i][j], f.v[i][j], ss);
  }
  return f;
}
template <> DEVI Frag<hf> ldx_frag<hf>(const float* p, bool ok, float& ss) {
  Frag<hf> f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 v = ok ? reinterpret_cast<const f32x4*>(p)[2 * h + i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ss = fmaf(v[j], v[j], ss);
        f.v[h][4 * i + j] = (hf)v[j];
      }
    }
  }
  return f;
}

template <> DEVI Frag<hl> ldx_frag<hl>(const float* p, bool ok, float& ss) {
  Frag<hl> f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 a = ok ? reinterpret_cast<const f32x4*>(p)[2 * h] : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 b = ok ? reinterpret_cast<const f32x4*>(p)[2 * h + 1] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) ss = fmaf(v[j], v[j], ss);
    split_hl8(f, h, v);
  }
  return f;
}

// accumulator registers (already in C layout) -> operand fragment with k-slot r <-> register r
template <typename T> DEVI Frag<T> pack_frag(const float (&h)[16]);
template <> DEVI Frag<float> pack_frag<float>(const float (&h)[16]) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.v[i] = f32x4{h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]};
  return f;
}
template <> DEVI Frag<hf> pack_frag<hf>(const float (&h)[16]) {
  Frag<hf> f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[s][j] = (hf)h[8 * s + j];
  return f;
}

template <> DEVI Frag<hl> pack_frag<hl>(const float (&h)[16]) {
  Frag<hl> f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float v[8] = {h[8 * s], h[8 * s + 1], h[8 * s + 2], h[8 * s + 3], h[8 * s + 4], h[8 * s + 5], h[8 * s + 6], h[8 * s + 7]};
    split_hl8(f, s, v);
  }
  return f;
}

DEVI void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Operand tile staged FRAGMENT-MAJOR in LDS: [piece][lane 0..63][8 half | 4 fp32]; lane l reads its
// 16 k-values as 16-byte pieces at tile + piece * 1024 + l * 16 (conflict-free, no padding).
template <typename T> DEVI Frag<T> lds_frag(const char* tile, int lane);
template <> DEVI Frag<hf> lds_frag<hf>(const char* tile, int lane) {
  Frag<hf> f;
  f.v[0] = *reinterpret_cast<const hfx8*>(tile + lane * 16);
  f.v[1] = *reinterpret_cast<const hfx8*>(tile + 1024 + lane * 16);
  return f;
}
template <> DEVI Frag<hl> lds_frag<hl>(const char* tile, int lane) {  // 4 KB tile: [hi half tile 2 KB | lo half tile 2 KB]
  Frag<hl> f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.v[i] = *reinterpret_cast<const hfx8*>(tile + i * 1024 + lane * 16);
  return f;
}
template <> DEVI Frag<float> lds_frag<float>(const char* tile, int lane) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.v[i] = *reinterpret_cast<const f32x4*>(tile + i * 1024 + lane * 16);
  return f;
}

