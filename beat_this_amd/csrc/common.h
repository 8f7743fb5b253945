// Shared device helpers for the beat_this MI355X (gfx950 / CDNA4) kernels.
//
// Everything is written for wave64 + the 32x32 MFMA shapes:
//   half operands : v_mfma_f32_32x32x16_bf16  (2 issues per 32-deep k-tile)
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

// Half-precision operand type of the BT_PREC_HALF path.  Default: IEEE fp16 (11-bit significand) -- what the reference's
// float16=True means on a GPU (inference.py:245-246, cli.py:82) and 8x finer than bfloat16 at the same MFMA rate; the
// logit error against the fp32 path drops from ~5e-2 to ~6e-3 (tools/prec_study.py, tests/test_gpu_scale.py).
// -DBT_HALF_BF16 builds the bfloat16 variant (8-bit significand, fp32 exponent range) for A/B comparisons.
// Accumulation is fp32 in both; the residual stream, RMSNorm statistics and softmax sums stay fp32.
#ifdef BT_HALF_BF16
typedef __bf16 hf;
#define BT_HALF_IS_BF16 1
#define MFMA32_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#else
typedef _Float16 hf;
#define BT_HALF_IS_BF16 0
#define MFMA32_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) hf hfx8;
typedef __attribute__((ext_vector_type(4))) hf hfx4;
typedef __attribute__((ext_vector_type(2))) hf hfx2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define DEVI __device__ __forceinline__

// One lane's share of a [32 rows] x [32 k] operand tile: 16 contiguous k values.
template <typename T> struct Frag;
template <> struct Frag<float> { f32x4 v[4]; };
template <> struct Frag<hf> { hfx8 v[2]; };
// BT_PREC_F32X3 in the register-chained kernels: an fp32 operand as a (hi, lo) pair of halves, hi = half(s a),
// lo = half(s a - hi).  sizeof = 4 like float: pointers into fp32 activations and 4 KB weight tiles ([hi 2 KB | lo 2 KB])
// index the same way as the float instantiation.  The power-of-two pre-scale s keeps lo a NORMAL fp16 number for every
// operand above 2^-3 / s (unscaled, the lo half of anything below 0.125 is a subnormal with a few bits, and the
// frontend's weights and activations mostly are: 1.2e-4 instead of 2e-5 at the logits): activations (everything split in
// registers) are scaled by OpScale::ACT, packed weights by OpScale::WGT, and every product is scaled back where its
// accumulator is consumed (OpScale::PW for weight . activation, PA for activation . activation; 1 for float / half).
struct hl { hf hi, lo; };
template <> struct Frag<hl> { hfx8 v[4]; };  // v[0], v[1] the hi fragment (as Frag<hf>), v[2], v[3] the lo fragment
template <typename T> struct OpScale { static constexpr float ACT = 1.f, PW = 1.f, PA = 1.f; };
template <> struct OpScale<hl> { static constexpr float ACT = 32.f, PW = 1.f / (32.f * 64.f), PA = 1.f / (32.f * 32.f); };  // WGT = 64 (pack.py)

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
template <> DEVI Frag<hf> ld_frag<hf>(const char* row_ptr, int g) {
  Frag<hf> f;
  const hfx8* p = reinterpret_cast<const hfx8*>(row_ptr + g * 32);
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
DEVI void mma32(f32x16& acc, const Frag<hf>& a, const Frag<hf>& b) {
  acc = MFMA32_H(a.v[0], b.v[0], acc);
  acc = MFMA32_H(a.v[1], b.v[1], acc);
}
// (a_hi + a_lo) . (b_hi + b_lo) without the lo . lo term: three half products, small ones first
DEVI void mma32(f32x16& acc, const Frag<hl>& a, const Frag<hl>& b) {
  acc = MFMA32_H(a.v[2], b.v[0], acc);
  acc = MFMA32_H(a.v[3], b.v[1], acc);
  acc = MFMA32_H(a.v[0], b.v[2], acc);
  acc = MFMA32_H(a.v[1], b.v[3], acc);
  acc = MFMA32_H(a.v[0], b.v[0], acc);
  acc = MFMA32_H(a.v[1], b.v[1], acc);
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
// GELU for the half path: x * sigmoid(x (a + b x^2 + c x^4)) with a, b, c fitted (minimax over [-9, 9]) to the exact
// x Phi(x): |error| <= 2.6e-5 absolute for all x -- below fp16 operand rounding (2^-12 relative) -- in 9 VALU ops
// (2 transcendental) instead of gelu_erf's 14.  (The textbook tanh form, c = 0, is 4.8e-4 off: visible next to fp16.)
// x^2 is clamped at 49: beyond |x| = 7 the sigmoid is saturated to 1 - 2e-11 and the quartic would eventually change
// sign.  The FF1 epilogue of the main layers spends as many VALU cycles on the activation as the MFMA pipe spends on the
// K = 512 product, hence the care.  The fp32 path keeps gelu_erf.
DEVI float gelu_tanh(float x) {
  const float x2 = fminf(x * x, 49.0f);
  const float pol = fmaf(fmaf(x2, 0.001014263f, -0.106775716f), x2, -2.3011212f);  // -log2(e) (a + b x^2 + c x^4)
  const float e = __builtin_amdgcn_exp2f(x * pol);           // exp(-u), u = x (a + b x^2 + c x^4)
  return x * __builtin_amdgcn_rcpf(1.0f + e);                // inf -> 0, 0 -> x: both limits are exact
}
template <typename T> DEVI float gelu_t(float x);
template <> DEVI float gelu_t<float>(float x) { return gelu_erf(x); }
template <> DEVI float gelu_t<hf>(float x) { return gelu_tanh(x); }
template <> DEVI float gelu_t<hl>(float x) { return gelu_erf(x); }
DEVI float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T> DEVI T from_f32(float x);
template <> DEVI float from_f32<float>(float x) { return x; }
template <> DEVI hf from_f32<hf>(float x) { return (hf)x; }

// hi + lo split of fp32 values (BT_PREC_F32X3): per pair (a, b)  whi = packed (half(a), half(b)),  wlo = packed
// (half(a - hi_a), half(b - hi_b)), in THREE instructions -- one v_cvt_pk_f16_f32 and, per lo half, one
// v_fma_mix{lo,hi}_f16 that reads its hi part straight out of the packed word (op_sel), multiplies it by -1 (an SGPR: the
// operand is read as fp32), adds the fp32 value and rounds the exact difference to fp16 into its half of the destination.
// Inline assembly for two reasons: (1) hipcc's own sequence is eight instructions per pair (two single conversions, two
// conversions back, two subtractions, two packed conversions), which made the split the largest VALU item of the x3
// attention loop; (2) a value that is about to be split must be ONE materialised fp32 number -- with fp contraction hipcc
// folds the arithmetic that produced it into the conversions (hi = half(x * y) from the EXACT product, the stored hi from a
// separately rounded one: the two differ by one fp16 ulp where the double rounding bites and hi + lo is then 2^-11 off;
// found by the unit tests of round 3 as 2.5e-4 .. 5e-4 relative on 0.1 % of the elements) -- and an asm operand is that.
// The hazard recogniser does not look into inline assembly, so the blocks carry their own wait states (gfx940 rules, one
// wait state each): a transcendental's result read by a non-transcendental VALU (the inputs may be v_exp results: leading
// s_nop), and a half-register write (mixlo / mixhi) followed by a read of that register (two pairs interleaved, trailing
// s_nop).  Inputs are always VALU results here, never raw MFMA accumulators.
DEVI void split_hl4(float a0, float b0, float a1, float b1, unsigned& h0, unsigned& l0, unsigned& h1, unsigned& l1) {
  const float m1 = -1.0f;
  asm("s_nop 0\n\t"
      "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_cvt_pk_f16_f32 %2, %6, %7\n\t"
      "v_fma_mixlo_f16 %1, %0, %8, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %2, %8, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %0, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %2, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0"
      : "=&v"(h0), "=&v"(l0), "=&v"(h1), "=&v"(l1)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "s"(m1));
}
DEVI void split_hl(float a, float b, unsigned& whi, unsigned& wlo) {
  const float m1 = -1.0f;
  asm("s_nop 0\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %0, %4, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0\n\t"
      "v_fma_mixhi_f16 %1, %0, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0"
      : "=&v"(whi), "=&v"(wlo)
      : "v"(a), "v"(b), "s"(m1));
}

// "hl8" ACTIVATIONS (BT_OPT_X3_GEMM_FP8, gemm3.hip X3 = 2).  e4m3 ends at 448 and activations do not (residual outlier channels of
// 10^3 on trained-like weights), so the byte sections of an activation carry 2^-3 of what a weight's carry: hi byte = e4m3(v / 8),
// lo byte = e4m3(2^8 (v - hi)), good to |v| = HL8_ACT_MAX = 3584 (beyond it the producers raise the range flag and the forward is
// repeated in exact fp32 like past 65504 on the hl32 path); a weight's are e4m3(w) and e4m3(2^11 (w - hi)) (pack.py), and the
// cross-term MFMA applies 2^-8 as its block scale.  Values below 2^-3 land in e4m3's subnormals (absolute step 2^-6 after the
// shift).  The shift was chosen on the flip soak (profiles/r05_hl8_shift.txt): flips of 204 k / 209 k / 200 k decisions on the
// init / lively / outlier weight styles -- shift 0: 290 / 20 / (every batch past 448: exact re-run); 2^-2: 287 / 23 / 14;
// 2^-3: 321 / 23 / 18; 2^-4: 416 / 26 / 18.
#ifndef BT_HL8_ACT_EXP
#define BT_HL8_ACT_EXP 3
#endif
constexpr float HL8_ACT_HI = 1.f / (1 << BT_HL8_ACT_EXP), HL8_ACT_LO = (float)(2048 >> BT_HL8_ACT_EXP), HL8_ACT_MAX = 448.f * (1 << BT_HL8_ACT_EXP);
constexpr int HL8_SCALE_WORD = 0x01010101 * (116 + BT_HL8_ACT_EXP);   // E8M0 bytes of 2^-(11 - BT_HL8_ACT_EXP) for the cross-term MFMA
DEVI unsigned pk4_f8_raw(float a, float b, float c, float d) {   // four fp32 -> four e4m3 bytes (v_cvt_pk_fp8_f32: OCP e4m3 on gfx950)
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
DEVI unsigned pk4_f8(float a, float b, float c, float d) { return pk4_f8_raw(a * HL8_ACT_HI, b * HL8_ACT_HI, c * HL8_ACT_HI, d * HL8_ACT_HI); }
// ... and, given their packed hi halves, the lo bytes (difference and scaling are exact in fp32)
DEVI unsigned lo4_f8(float a, float b, float c, float d, unsigned h01, unsigned h23) {
  const hfx2 p = __builtin_bit_cast(hfx2, h01), q = __builtin_bit_cast(hfx2, h23);
  return pk4_f8_raw((a - (float)p[0]) * HL8_ACT_LO, (b - (float)p[1]) * HL8_ACT_LO, (c - (float)q[0]) * HL8_ACT_LO, (d - (float)q[1]) * HL8_ACT_LO);
}

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
