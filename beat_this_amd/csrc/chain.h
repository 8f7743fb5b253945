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

// an fp32 activation -> the (hi, lo) half pair of Frag<hl> (pre-scaled, common.h: OpScale): element j of half-fragment h
DEVI void split_hl(Frag<hl>& f, int h, int j, float v) {
  v *= OpScale<hl>::ACT;
  const hf hi = (hf)v;
  f.v[h][j] = hi;
  f.v[2 + h][j] = (hf)(v - (float)hi);
}
template <> DEVI Frag<hl> ldg_frag<hl>(const hl* p) {   // (memory holds fp32: hl is the operand type, not a storage type)
  Frag<hl> f;
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = q[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_hl(f, i >> 1, 4 * (i & 1) + j, v[j]);
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
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = ok ? reinterpret_cast<const f32x4*>(p)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ss = fmaf(v[j], v[j], ss);
      split_hl(f, i >> 1, 4 * (i & 1) + j, v[j]);
    }
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
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) split_hl(f, s, j, h[8 * s + j]);
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

