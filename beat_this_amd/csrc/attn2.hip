// half flash attention on FRAGMENT-MAJOR operands (head_dim = 32, non-causal, no mask), the
// BT_PREC_HALF replacement of attn_flash_kernel for the time-direction and main-transformer
// attention (roformer.py:67-80,125-131).
//
// Operand layout (written by the QKV producers, csrc/qkv_front.hip and the QKV epilogue of
// csrc/gemm3.hip, straight from their MFMA accumulators with fully contiguous stores):
// per (sequence, head) a run of 32-token blocks, one block = 2 KB = what the 64 lanes of a wave
// read as MFMA operand fragments, in lane order:
//   Q, K block : [quarter a = 0..3][token 0..31][8 dims 8a..8a+7]          (16 B per entry)
//                lane (token lr, half g) reads entries (2g, lr) and (2g+1, lr): dims [16g, 16g+16)
//   V block    : [s = 0..1][lane = 32 g + d][8 tokens crow(8s+j, g), j = 0..7]
//                i.e. V^T with the token order of MFMA C registers 8s..8s+7
// so a K/V tile is copied global -> LDS linearly (buffer_load ... lds, 16 B per lane, no VGPRs), every
// fragment is ONE conflict-free ds_read_b128, and nothing is transposed or shuffled anywhere.
//
// Softmax: S^T = K . Q^T puts one query per lane (16 of its 32 key scores per lane-half).  d = 32
// makes this kernel VALU-bound (4 MFMAs per 512 exp), so the per-score work is cut to
// exp + half a cvt_pk (the row sums ride on v_mfma_f32_4x4x4_16b_bf16 with an all-ones A operand):
//   * the running-max subtraction rides on the MFMA: the accumulator INPUT of the score MFMA is a
//     register block holding -m (q is pre-scaled by log2(e)/sqrt(32), RoPE applied by the producer);
//   * m is fixed per query from the first key block; later scores may exceed it, which is exact in
//     fp32/half (common factor 2^-m cancels in O / l) unless exp2 overflows.  Overflow or a sum
//     >= 1e30 is detected on l at the end and the whole workgroup then re-runs the classic
//     online-softmax loop (SAFE pass), so the result is always the exact softmax.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KB = 4;                    // 32-key blocks per LDS tile (128 keys)
constexpr int BLK_BYTES = 2048;          // one fragment-major block
constexpr int TILE_BYTES = KB * BLK_BYTES;
constexpr int SMEM_BYTES = 2 * 2 * TILE_BYTES + 16;  // [buffer][K | V] + fallback flag

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// The lane index again, from an operand hipcc cannot see through: values derived from it are recomputed where they are
// used instead of being kept live (or spilled -- the kernels here allow no scratch next to LDS-DMA) across a key loop that
// has no register to spare.
DEVI int lane_id_fresh() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}
DEVI unsigned pk2(float a, float b) {
  const hfx2 t = {(hf)a, (hf)b};
  return __builtin_bit_cast(unsigned, t);
}
DEVI void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// 8 probabilities -> 4 dwords of packed half (one v_cvt_pk_bf16_f32 each).  Operands are assembled from
// these dwords by bit casts only: half-vector shuffles make hipcc (ROCm 7.2) emit 3x the conversions.
#ifndef BT_ATTN_PKRTZ
#define BT_ATTN_PKRTZ 0
#endif
constexpr float L_OVERFLOW = (BT_ATTN_PKRTZ && !BT_HALF_IS_BF16) ? 65504.f : 1e30f;  // fast pass: row sums from here on re-run
DEVI u32x4 pack8(const f32x16& p, int s) {
  u32x4 w;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#if BT_ATTN_PKRTZ && !BT_HALF_IS_BF16
    // Round-toward-zero packing (v_cvt_pkrtz_f16_f32) issues faster than the round-to-nearest v_cvt_pk_f16_f32.  It is as
    // accurate for a softmax: numerator (P.V) and denominator (the row sums, taken from the same packed words) carry the
    // same mean relative bias of -2^-11, which cancels in O / l, and the spread around it is that of round-to-nearest.  But
    // it SATURATES at 65504 instead of producing inf, so the overflow test of the fast pass is "row sum >= 65504" here (one
    // saturated probability makes the sum at least that; a sum that large without one only costs the re-run).
    w[j] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p[8 * s + 2 * j], p[8 * s + 2 * j + 1]));
#else
    const hfx2 t = {(hf)p[8 * s + 2 * j], (hf)p[8 * s + 2 * j + 1]};
    w[j] = __builtin_bit_cast(unsigned int, t);
#endif
  }
  return w;
}
// l += sum of this lane's 8 half probabilities: D = ones(4x4) . B puts, in every output register of a
// lane, the sum of the 4 k-values that lane supplied as B (16 independent 4x4x4 blocks, column j of
// block b lives in lane 4b + j for B and D alike; checked by tools/ubench/mfma444_probe.hip), so the
// row sums cost no VALU issue slots.
// BT_ATTN_ROWSUM selects where the row sums run: 0 = matrix pipe (4x4x4 MFMAs, below), 1 = VALU adds of the fp32
// probabilities, 2 = v_dot2_f32_f16 on the packed words (A/B-measured on the MI355X).
#ifndef BT_ATTN_ROWSUM
#define BT_ATTN_ROWSUM 0
#endif
DEVI void rowsum8(f32x4& l, const u32x4& w) {
#if BT_ATTN_ROWSUM == 2
  const hfx2 one2 = {(hf)1.0f, (hf)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#if BT_HALF_IS_BF16
    l[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hfx2, w[i]), one2, l[0], false);
#else
    l[0] = __builtin_amdgcn_fdot2(__builtin_bit_cast(hfx2, w[i]), one2, l[0], false);
#endif
#elif BT_HALF_IS_BF16
  const s16x4 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
  l = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s16x4, u32x2{w[0], w[1]}), l, 0, 0, 0);
  l = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s16x4, u32x2{w[2], w[3]}), l, 0, 0, 0);
#else
  const hfx4 ones = {(hf)1.0f, (hf)1.0f, (hf)1.0f, (hf)1.0f};
  l = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(hfx4, u32x2{w[0], w[1]}), l, 0, 0, 0);
  l = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(hfx4, u32x2{w[2], w[3]}), l, 0, 0, 0);
#endif
}
// 16 fp32 probabilities of one lane added as a tree (BT_ATTN_ROWSUM == 1)
DEVI void rowsum16_valu(f32x4& l, const f32x16& p) {
  const float a = (p[0] + p[1]) + (p[2] + p[3]), b = (p[4] + p[5]) + (p[6] + p[7]);
  const float c = (p[8] + p[9]) + (p[10] + p[11]), d = (p[12] + p[13]) + (p[14] + p[15]);
  l[0] += (a + b) + (c + d);
}
// Fast pass: P = exp2(s - m_ref - P_SHIFT) with m_ref the query's maximum over the FIRST key block.  fp16 probabilities
// overflow at 2^16, so the reference point sits P_SHIFT octaves below 1: later keys may score up to 16 + P_SHIFT octaves
// (13.9 nats) above the first block's maximum before the SAFE pass has to re-run the workgroup; the price is that keys
// more than 14 - P_SHIFT octaves below the reference point are rounded as fp16 subnormals (absolute 2^-25: a relative
// error of 2^-10 in the row sum only if ALL 1500 keys sit there).  The common factor cancels in O / l.  bfloat16 has
// the fp32 exponent range and needs no shift.
constexpr float P_SHIFT = BT_HALF_IS_BF16 ? 0.f : 4.f;
// BT_ATTN_EXPT (development, variant builds only -- results are WRONG): what the key loop of the fast pass spends where.
//   1 = no exponentials, 2 = no packing either (VALU-free), 4 = no row-sum MFMAs, 8 = no fragment reads from LDS in the loop,
//   16 = no LDS-DMA / barriers after the prologue, 32 = one score MFMA per block instead of two, 64 = one P.V MFMA instead of two
#ifndef BT_ATTN_EXPT
#define BT_ATTN_EXPT 0
#endif

// Per-query-block softmax state of a wave (QB query blocks of 32 queries share every K / V fragment read).
struct QState {
  hfx8 q0, q1;   // Q^T operand: dims [16g, 16g+8) and [16g+8, 16g+16) of this lane's query
  f32x16 negm;     // -m splat: accumulator input of the score MFMA (fast pass)
  f32x16 acc;      // O^T accumulator
  f32x4 l;         // row sum (all four registers equal)
  float m;         // running max (SAFE pass) / reference max (fast pass)
};

struct KFrag { hfx8 k0, k1; };
struct VFrag { hfx8 v0, v1; };
DEVI KFrag ld_k(const char* kb, int g, int lr) {
  KFrag f;
  f.k0 = *reinterpret_cast<const hfx8*>(kb + ((2 * g) * 32 + lr) * 16);
  f.k1 = *reinterpret_cast<const hfx8*>(kb + ((2 * g + 1) * 32 + lr) * 16);
  return f;
}
DEVI VFrag ld_v(const char* vb, int lane) {
  VFrag f;
  f.v0 = *reinterpret_cast<const hfx8*>(vb + lane * 16);
  f.v1 = *reinterpret_cast<const hfx8*>(vb + 1024 + lane * 16);
  return f;
}

// One 32-key block against the QB query blocks of this wave: scores, probabilities, O^T += V^T . P^T (the plain,
// unpipelined form: SAFE pass and the ragged / masked last tile of the fast pass).
template <bool SAFE, bool MASK, int QB>
DEVI void do_block(const KFrag& kf, const VFrag& vf, int g, QState (&st)[QB], int key0, int L) {
  const hfx8 k0 = kf.k0, k1 = kf.k1, v0 = vf.v0, v1 = vf.v1;
  f32x16 sc[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    if constexpr (SAFE) {
      zero16(sc[j]);
      sc[j] = MFMA32_H(k0, st[j].q0, sc[j]);
    } else {
      sc[j] = MFMA32_H(k0, st[j].q0, st[j].negm);
    }
    sc[j] = MFMA32_H(k1, st[j].q1, sc[j]);
  }
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    if constexpr (SAFE) {
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + crow(r, g) >= L) sc[j][r] = -1e30f;
      }
      float bm = sc[j][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[j][r]);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float m_new = fmaxf(st[j].m, bm);
      const float alpha = __builtin_amdgcn_exp2f(st[j].m - m_new);
      st[j].m = m_new;
#pragma unroll
      for (int r = 0; r < 4; ++r) st[j].l[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j].acc[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[j][r] = __builtin_amdgcn_exp2f(sc[j][r] - m_new);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[j][r] = __builtin_amdgcn_exp2f(QB == 1 ? sc[j][r] : sc[j][r] + st[j].negm[0]);
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + crow(r, g) >= L) sc[j][r] = 0.f;
      }
    }
    const u32x4 w0 = pack8(sc[j], 0), w1 = pack8(sc[j], 1);
#if BT_ATTN_ROWSUM == 1
    rowsum16_valu(st[j].l, sc[j]);
#else
    rowsum8(st[j].l, w0);
    rowsum8(st[j].l, w1);
#endif
    st[j].acc = MFMA32_H(v0, __builtin_bit_cast(hfx8, w0), st[j].acc);
    st[j].acc = MFMA32_H(v1, __builtin_bit_cast(hfx8, w1), st[j].acc);
  }
}

// Copy K tile `tile` and V tile `tile` (KB blocks each, contiguous in global memory) into LDS buffer
// `buf`: buffer_load ... lds with ONE per-lane offset register (tid * 16) for the whole kernel, the
// tile / piece offset in an SGPR, so no address lives in (spillable) VGPRs.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
DEVI void stage_tile(rsrc_t rk, rsrc_t rv, int tile, char* smem, int buf, int tid, int wave) {
  char* kd = smem + buf * 2 * TILE_BYTES + wave * 1024;
  char* vd = kd + TILE_BYTES;
  const int so = tile * TILE_BYTES;
  // (the instruction's immediate offset would be added to the LDS address as well: keep it 0)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lptr_t)kd, 16, tid * 16, so, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lptr_t)(kd + 4096), 16, tid * 16, so + 4096, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lptr_t)vd, 16, tid * 16, so, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lptr_t)(vd + 4096), 16, tid * 16, so + 4096, 0, 0);
  static_assert(TILE_BYTES == 8192, "stage_tile copies two 4 KB pieces per operand");
}

// compute (wave-uniform): false = this wave only takes part in the staging and the barriers and keeps its state (the two-query-
// block kernel repeats HALF a workgroup -- the 128 queries that are one workgroup of attn_frag_kernel -- and nobody else)
template <bool SAFE, int QB>
DEVI void attn_pass(rsrc_t rk, rsrc_t rv, char* smem, int tid, int wave, int lane, int g, int lr, QState (&st)[QB],
                    int L, int nblk, bool compute = true) {
  const int ntiles = (nblk + KB - 1) / KB;
  const bool partial = (L & 31) != 0;
  stage_tile(rk, rv, 0, smem, 0, tid, wave);
  __syncthreads();
  if (compute) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    zero16(st[j].negm);
    zero16(st[j].acc);
    st[j].l = f32x4{0.f, 0.f, 0.f, 0.f};
    st[j].m = -1e30f;
  }
  }
  if constexpr (!SAFE) {  // reference max of each query: its scores against key block 0
    const KFrag k00 = ld_k(smem, g, lr);
    const hfx8 k0 = k00.k0, k1 = k00.k1;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      f32x16 sc;
      zero16(sc);
      sc = MFMA32_H(k0, st[j].q0, sc);
      sc = MFMA32_H(k1, st[j].q1, sc);
      float bm = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bm = fmaxf(bm, (crow(r, g) < L) ? sc[r] : -1e30f);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      st[j].m = bm;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j].negm[r] = -bm - P_SHIFT;
    }
  }
  // Fragment reads are software pipelined by hand: V of block c and K of block c+1 are issued before the
  // score MFMAs of block c (LDS latency under 16 waves of traffic is several hundred cycles; issued
  // just-in-time it was the largest stall of the loop).
  KFrag kf = ld_k(smem, g, lr);
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage_tile(rk, rv, t + 1, smem, (t + 1) & 1, tid, wave);
    const char* kb = smem + (t & 1) * 2 * TILE_BYTES;
    const char* vb = kb + TILE_BYTES;
    const int nb = min(KB, nblk - t * KB);
    if (!compute) {
      // (nothing to multiply here)
    } else if (nb == KB && !(partial && t == ntiles - 1)) {
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const VFrag vf = ld_v(vb + c * BLK_BYTES, lane);
        KFrag kn = kf;
        if (c + 1 < KB) kn = ld_k(kb + (c + 1) * BLK_BYTES, g, lr);
        __builtin_amdgcn_sched_barrier(0);
        do_block<SAFE, false, QB>(kf, vf, g, st, (t * KB + c) * 32, L);
        kf = kn;
      }
    } else {
      for (int c = 0; c < nb; ++c) {
        const int blk = t * KB + c;
        const VFrag vf = ld_v(vb + c * BLK_BYTES, lane);
        if (c > 0) kf = ld_k(kb + c * BLK_BYTES, g, lr);
        if (partial && blk == nblk - 1)
          do_block<SAFE, true, QB>(kf, vf, g, st, blk * 32, L);
        else
          do_block<SAFE, false, QB>(kf, vf, g, st, blk * 32, L);
      }
    }
    __syncthreads();  // tile t+1 has landed (every wave waited for its own copies), tile t is free
    if (t + 1 < ntiles) kf = ld_k(smem + ((t + 1) & 1) * 2 * TILE_BYTES, g, lr);
  }
}

// ---- software-pipelined fast pass ------------------------------------------------------------------------
// The scores of key block c+1 are issued BEFORE the exponentials of block c, so the matrix pipe works on
// the next block while this wave converts the current one (no MFMA -> VALU hazard bubbles), on top of the
// overlap between waves.  One barrier per tile, placed at the start of the tile's LAST block: every
// wave has then issued and received its last fragment reads of tile t, so the freed buffer takes tile t+2.
template <int QB>
DEVI void score_fast(const KFrag& kf, const QState (&st)[QB], f32x16 (&sc)[QB]) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    sc[j] = MFMA32_H(kf.k0, st[j].q0, st[j].negm);
#if !(BT_ATTN_EXPT & 32)
    sc[j] = MFMA32_H(kf.k1, st[j].q1, sc[j]);
#endif
  }
}
template <int QB>
DEVI void finish_fast(f32x16 (&sc)[QB], const VFrag& vf, QState (&st)[QB]) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
#if !(BT_ATTN_EXPT & 3)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[j][r] = __builtin_amdgcn_exp2f(sc[j][r]);
#endif
#if BT_ATTN_EXPT & 2
    const u32x4 w0 = {__builtin_bit_cast(unsigned, sc[j][0]), __builtin_bit_cast(unsigned, sc[j][1]), __builtin_bit_cast(unsigned, sc[j][2]), __builtin_bit_cast(unsigned, sc[j][3])};
    const u32x4 w1 = {__builtin_bit_cast(unsigned, sc[j][8]), __builtin_bit_cast(unsigned, sc[j][9]), __builtin_bit_cast(unsigned, sc[j][10]), __builtin_bit_cast(unsigned, sc[j][11])};
#else
    const u32x4 w0 = pack8(sc[j], 0), w1 = pack8(sc[j], 1);
#endif
#if BT_ATTN_EXPT & 4
    st[j].l[0] += __builtin_bit_cast(float, w0[0]);
#elif BT_ATTN_ROWSUM == 1
    rowsum16_valu(st[j].l, sc[j]);
#else
    rowsum8(st[j].l, w0);
    rowsum8(st[j].l, w1);
#endif
#if BT_ATTN_EXPT & 64
    st[j].acc = MFMA32_H(vf.v0, __builtin_bit_cast(hfx8, w0 ^ w1), st[j].acc);
#else
    st[j].acc = MFMA32_H(vf.v0, __builtin_bit_cast(hfx8, w0), st[j].acc);
    st[j].acc = MFMA32_H(vf.v1, __builtin_bit_cast(hfx8, w1), st[j].acc);
#endif
  }
}

template <int QB>
DEVI void attn_pass_pipe(rsrc_t rk, rsrc_t rv, char* smem, int tid, int wave, int lane, int g, int lr,
                         QState (&st)[QB], int L, int nblk, long long* t_loop = nullptr) {
  const int ntiles = (nblk + KB - 1) / KB;
  const bool partial = (L & 31) != 0;
  int nfull = nblk / KB;  // tiles of KB unmasked blocks
  if (partial && nfull * KB == nblk) --nfull;
  nfull = __builtin_amdgcn_readfirstlane(nfull);  // (hipcc kept the loop bound in a VGPR -- and spilled it across the loop)
  stage_tile(rk, rv, 0, smem, 0, tid, wave);
  __syncthreads();
  if (ntiles > 1) stage_tile(rk, rv, 1, smem, 1, tid, wave);
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    zero16(st[j].acc);
    st[j].l = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {  // reference max of each query: its scores against key block 0
    const KFrag k00 = ld_k(smem, g, lr);
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      f32x16 sc;
      zero16(sc);
      sc = MFMA32_H(k00.k0, st[j].q0, sc);
      sc = MFMA32_H(k00.k1, st[j].q1, sc);
      float bm = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bm = fmaxf(bm, (crow(r, g) < L) ? sc[r] : -1e30f);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      st[j].m = bm;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j].negm[r] = -bm - P_SHIFT;
    }
  }
  if (t_loop) *t_loop = wall_clock64();
  if (nfull > 0) {
    f32x16 sc[QB];
    KFrag kn = ld_k(smem, g, lr);
    score_fast<QB>(kn, st, sc);
    kn = ld_k(smem + BLK_BYTES, g, lr);
#if BT_ATTN_EXPT & 8
    const VFrag vf0 = ld_v(smem + TILE_BYTES, lane);
#define LD_V(addr, lane) vf0
#define LD_K(addr, g, lr) kn
#else
#define LD_V(addr, lane) ld_v(addr, lane)
#define LD_K(addr, g, lr) ld_k(addr, g, lr)
#endif
    // (Reading the V fragments one block ahead as well -- a second V register set, 127 VGPRs -- was measured and dropped:
    // main-layer shape 104.8 / 103.2 vs 108.9 / 105.0 us, frontend shape 237 / 235 vs 228 / 231 us, the x3 kernel 2 % slower:
    // the fragment reads cost LDS issue slots and energy, not exposed latency.)
    for (int t = 0; t < nfull; ++t) {
      const char* kb = smem + (t & 1) * 2 * TILE_BYTES;
      const char* vb = kb + TILE_BYTES;
      const char* kb_next = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const VFrag vf = LD_V(vb + c * BLK_BYTES, lane);
        if (c + 1 < KB) {
          f32x16 sn[QB];
          score_fast<QB>(kn, st, sn);
          if (c + 2 < KB) kn = LD_K(kb + (c + 2) * BLK_BYTES, g, lr);
          __builtin_amdgcn_sched_barrier(0);
          finish_fast<QB>(sc, vf, st);
#pragma unroll
          for (int j = 0; j < QB; ++j) sc[j] = sn[j];
        } else {
#if !(BT_ATTN_EXPT & 16)
          __syncthreads();  // tile t+1 has landed; nobody reads tile t any more (every fragment read of it has arrived)
          if (t + 2 < ntiles) stage_tile(rk, rv, t + 2, smem, t & 1, tid, wave);
#endif
          const bool more = t + 1 < nfull;  // (uniform)
          KFrag k0n = kn;
          if (more) {
            k0n = LD_K(kb_next, g, lr);
            kn = LD_K(kb_next + BLK_BYTES, g, lr);
          }
          __builtin_amdgcn_sched_barrier(0);
          finish_fast<QB>(sc, vf, st);
          if (more) score_fast<QB>(k0n, st, sc);
        }
      }
    }
  }
  if (nfull < ntiles) {  // last tile: fewer than KB blocks and / or a masked last block
    const char* kb = smem + (nfull & 1) * 2 * TILE_BYTES;
    const char* vb = kb + TILE_BYTES;
    const int nb = nblk - nfull * KB;
    const int lane2 = lane_id_fresh(), g2 = lane2 >> 5, lr2 = lane2 & 31;  // (not kept live across the key loop)
    for (int c = 0; c < nb; ++c) {
      const int blk = nfull * KB + c;
      const VFrag vf = ld_v(vb + c * BLK_BYTES, lane2);
      const KFrag kf = ld_k(kb + c * BLK_BYTES, g2, lr2);
      if (partial && blk == nblk - 1)
        do_block<false, true, QB>(kf, vf, g2, st, blk * 32, L);
      else
        do_block<false, false, QB>(kf, vf, g2, st, blk * 32, L);
    }
  }
  __syncthreads();
}

#undef LD_V
#undef LD_K

template <int ABL, int QB>
__global__ __launch_bounds__(256, (QB == 1 ? 4 : 2)) void attn_frag_kernel(const AttnFragP p, int nqt, int sh_total) {
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  // XCD-aware order: the q-tiles of one (sequence, head) run on one XCD (block b -> XCD b % 8), so its
  // K/V stream is fetched into one L2
  const long long t_entry = wall_clock64();
  const int bid = blockIdx.x;
  const int idx = bid >> 3;
  const int sh = (idx / nqt) * 8 + (bid & 7);
  const int qt = idx % nqt;
  if (sh >= sh_total) return;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g0 = lane0 >> 5, lr0 = lane0 & 31;
  const int L = p.L;
  const int nblk = (L + 31) >> 5;
  const long seq_off = (long)sh * p.nbp * BLK_BYTES;
  const char* kseq = reinterpret_cast<const char*>(p.k) + seq_off;
  const char* vseq = reinterpret_cast<const char*>(p.v) + seq_off;
  int* flag = reinterpret_cast<int*>(smem + 4 * TILE_BYTES);
  if (tid0 == 0) *flag = 0;

  QState st[QB];
  const int qb0 = (qt * 4 + wave) * QB;  // this wave's first query block
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qbc = min(qb0 + j, nblk - 1);
    const char* qblk = reinterpret_cast<const char*>(p.q) + seq_off + (long)qbc * BLK_BYTES;
    st[j].q0 = *reinterpret_cast<const hfx8*>(qblk + ((2 * g0) * 32 + lr0) * 16);
    st[j].q1 = *reinterpret_cast<const hfx8*>(qblk + ((2 * g0 + 1) * 32 + lr0) * 16);
  }
  const unsigned seq_bytes = (unsigned)p.nbp * BLK_BYTES;
  const rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kseq), 0, seq_bytes, 0x00020000);
  const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vseq), 0, seq_bytes, 0x00020000);
  const long long tc0 = clock64(), tw0 = wall_clock64();
  long long t_loop = tw0;
  attn_pass_pipe<QB>(rk, rv, smem, tid0, wave, lane0, g0, lr0, st, L, nblk, (ABL & 128) ? &t_loop : nullptr);
  // (lane-derived values again: nothing but the softmax state stays live across the key loop)
  const int lane = lane_id_fresh(), tid = wave * 64 + lane, g = lane >> 5, lr = lane & 31;
  if constexpr ((ABL & 128) != 0) {  // development: shader-clock ticks vs 100 MHz wall ticks of the pass
    if (lane == 0) {
      long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(p.gates));
      dbg[((long)bid * 4 + wave) * 4] = clock64() - tc0;
      dbg[((long)bid * 4 + wave) * 4 + 1] = wall_clock64() - tw0;
      dbg[((long)bid * 4 + wave) * 4 + 2] = tw0 - t_entry;   // kernel entry -> pass start (descriptor setup, Q loads issued)
      dbg[((long)bid * 4 + wave) * 4 + 3] = t_loop - tw0;    // pass start -> key loop start (tile 0 landed, reference maximum)
    }
  }
  float l_tot[QB];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    l_tot[j] = st[j].l[0] + __shfl_xor(st[j].l[0], 32);
    const bool valid = qb0 + j < nblk && (qb0 + j) * 32 + lr < L;
    bad = bad || (valid && !(l_tot[j] < L_OVERFLOW));  // overflow / NaN: this query needs the running-max pass
  }
  if (__any(bad) && lane == 0) *flag = 1;
  __syncthreads();
  if (*flag) {  // workgroup-uniform
    __syncthreads();
    attn_pass<true, QB>(rk, rv, smem, tid, wave, lane, g, lr, st, L, nblk);
#pragma unroll
    for (int j = 0; j < QB; ++j) l_tot[j] = st[j].l[0] + __shfl_xor(st[j].l[0], 32);
  }

  const int seq = sh / p.heads, head = sh - seq * p.heads;
  hf* st16[QB][2];
  bool okq[QB];
  float gatev[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qi = (qb0 + j) * 32 + lr;
    okq[j] = qb0 + j < nblk && qi < L;
    gatev[j] = 0.f;
    st16[j][0] = st16[j][1] = nullptr;
    if (okq[j]) {
      gatev[j] = p.gates[(long)sh * p.nbp * 32 + qi];
      const long orow = (long)(seq / p.o_div) * p.o_outer + (long)(seq % p.o_div) * p.o_inner + (long)qi * p.o_tok;
      hf* op = reinterpret_cast<hf*>(p.out) + orow * p.inner + head * 32 + 8 * g;
#pragma unroll
      for (int k = 0; k < 2; ++k) st16[j][k] = op + 16 * k;
    }
  }
  // A lane holds 4-feature runs 8 a + 4 g of its query; the two halves of a wave exchange runs (v_permlane32_swap) so
  // that every lane stores two 16-byte pieces (features 16 k + 8 g .. + 7) instead of four 8-byte ones: the 8-byte
  // row-strided stores showed 1.9x write amplification in WRITE_SIZE.  (The exchange is executed by all lanes.)
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const float scale = okq[j] ? gatev[j] / l_tot[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned x0 = pk2(st[j].acc[8 * k] * scale, st[j].acc[8 * k + 1] * scale);
      const unsigned x1 = pk2(st[j].acc[8 * k + 2] * scale, st[j].acc[8 * k + 3] * scale);
      const unsigned y0 = pk2(st[j].acc[8 * k + 4] * scale, st[j].acc[8 * k + 5] * scale);
      const unsigned y1 = pk2(st[j].acc[8 * k + 6] * scale, st[j].acc[8 * k + 7] * scale);
      auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
      if (okq[j]) *reinterpret_cast<u32x4*>(st16[j][k]) = u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
  }
}


// =====================================================================================================================
// Round 6: the fp16 attention on TWO query blocks per wave and a hand-scheduled key loop (tools/gen/attn_hq2_loop.py ->
// attn_hq2_loop.inc: one asm statement, 4 big + 4 small MFMAs beside 24 VALU instructions per step and query block, every K / V
// fragment read feeding both blocks), the design of attn_frag_x3q2_kernel without the lo terms.  256 queries per workgroup, a
// ring of four 8 KB [K tile | V tile] buffers of 64 keys, two workgroups per CU.
//
// SAME BITS AS attn_frag_kernel<0, 1> for every query (the forward picks the kernel by launch size; a chunk must not depend on
// its batch): same reference point (the query's maximum over key block 0, minus P_SHIFT, on the score MFMA's accumulator input),
// same order of the products, the packing, the row-sum MFMAs and the P.V MFMAs, the ragged last tile on the same plain code --
// and the same REPEAT UNITS: a fast pass that overflowed re-runs the 128 queries that are one workgroup of attn_frag_kernel
// (here: a wave pair) on the running-maximum pass, nobody else; the other wave pair takes part in that pass's staging and
// barriers only.
#if !BT_HALF_IS_BF16
#include "attn_hq2_loop.inc"

// the plain form of one key block for QB query blocks of the fast pass (do_block<false, MASK, 1>'s arithmetic for each of them)
template <bool MASK, int QB>
DEVI void block_fast_plain(const KFrag& kf, const VFrag& vf, int g, QState (&st)[QB], int key0, int L) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    f32x16 sc = MFMA32_H(kf.k0, st[j].q0, st[j].negm);
    sc = MFMA32_H(kf.k1, st[j].q1, sc);
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
    if constexpr (MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + crow(r, g) >= L) sc[r] = 0.f;
    }
    const u32x4 w0 = pack8(sc, 0), w1 = pack8(sc, 1);
    rowsum8(st[j].l, w0);
    rowsum8(st[j].l, w1);
    st[j].acc = MFMA32_H(vf.v0, __builtin_bit_cast(hfx8, w0), st[j].acc);
    st[j].acc = MFMA32_H(vf.v1, __builtin_bit_cast(hfx8, w1), st[j].acc);
  }
}

__global__ __launch_bounds__(256, 2) void attn_frag_hq2_kernel(const AttnFragP p, int nqt, int sh_total) {
  constexpr int QB = 2, KBX = ATTN_HQ2_KBX, NBUF = ATTN_HQ2_NBUF;
  constexpr int TILEH = KBX * BLK_BYTES, BUFH = 2 * TILEH;
  static_assert(KBX == 2 && NBUF == 4 && BUFH == 8192 && NBUF * BUFH == 4 * TILE_BYTES, "the generated loop is written for four 8 KB ring buffers");
  static_assert(BT_ATTN_ROWSUM == 0 && !BT_ATTN_PKRTZ, "the generated loop runs the row sums on the matrix pipe and packs round-to-nearest");
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];   // (the running-maximum pass re-uses it as two 16 KB stages)
  const int bid = blockIdx.x;
  const int idx = bid >> 3;
  const int sh = (idx / nqt) * 8 + (bid & 7);
  const int qt = idx % nqt;
  if (sh >= sh_total) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int nblk = (L + 31) >> 5;
  const long seq_off = (long)sh * p.nbp * BLK_BYTES;
  const char* kseq = reinterpret_cast<const char*>(p.k) + seq_off;
  const char* vseq = reinterpret_cast<const char*>(p.v) + seq_off;
  int* flag = reinterpret_cast<int*>(smem + 4 * TILE_BYTES);   // [2]: one per wave pair
  if (tid < 2) flag[tid] = 0;

  QState st[QB];
  const int qb0 = (qt * 4 + wave) * QB;  // this wave's first query block
  {
    const int ln = lane_id_fresh(), gg = ln >> 5, ll = ln & 31;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int qbc = min(qb0 + j, nblk - 1);
      const char* qblk = reinterpret_cast<const char*>(p.q) + seq_off + (long)qbc * BLK_BYTES;
      st[j].q0 = *reinterpret_cast<const hfx8*>(qblk + ((2 * gg) * 32 + ll) * 16);
      st[j].q1 = *reinterpret_cast<const hfx8*>(qblk + ((2 * gg + 1) * 32 + ll) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (VGPR-returning loads and LDS-DMA do not retire in one order)
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned seq_bytes = (unsigned)p.nbp * BLK_BYTES;   // (tiles beyond it read as zeros: the ring is always refilled)
  const rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kseq), 0, seq_bytes, 0x00020000);
  const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vseq), 0, seq_bytes, 0x00020000);
  const int ntiles = (nblk + KBX - 1) / KBX;
  const bool partial = (L & 31) != 0;
  int nfull = nblk / KBX;  // tiles of KBX unmasked blocks
  if (partial && nfull * KBX == nblk) --nfull;
  nfull = __builtin_amdgcn_readfirstlane(nfull);
  auto stage_ring = [&](int tile) {   // tile t lives in buffer t & 3 = [K tile | V tile]: one 1 KB piece of each per wave
    char* kd = smem + (tile & (NBUF - 1)) * BUFH + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lptr_t)kd, 16, tid * 16, tile * TILEH, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lptr_t)(kd + TILEH), 16, tid * 16, tile * TILEH, 0, 0);
  };
  stage_ring(0);
  stage_ring(1);
  stage_ring(2);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 (this wave's two pieces of it) has landed
  __syncthreads();
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    zero16(st[j].acc);
    st[j].l = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {  // reference point of each query: its maximum over key block 0 (attn_pass_pipe's prologue)
    const int ln = lane_id_fresh(), gg = ln >> 5, ll = ln & 31;
    const KFrag k00 = ld_k(smem, gg, ll);
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      f32x16 sc;
      zero16(sc);
      sc = MFMA32_H(k00.k0, st[j].q0, sc);
      sc = MFMA32_H(k00.k1, st[j].q1, sc);
      float bm = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bm = fmaxf(bm, (crow(r, gg) < L) ? sc[r] : -1e30f);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      st[j].m = bm;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j].negm[r] = -bm - P_SHIFT;
    }
  }
  if (nfull > 0) {
    // operand words of the two buffer descriptors as plain SGPR quads (an asm operand cannot be a __amdgpu_buffer_rsrc_t)
    const unsigned long long ka = (unsigned long long)kseq, va = (unsigned long long)vseq;
    const u32x4 dk = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((ka >> 32) & 0xffffu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)seq_bytes), 0x00020000u};
    const u32x4 dv = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)va), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((va >> 32) & 0xffffu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)seq_bytes), 0x00020000u};
    const int ln = lane_id_fresh();
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;   // LDS byte address of the ring
    const unsigned klane = lds0 + ((2 * (ln >> 5)) * 32 + (ln & 31)) * 16, vlane = lds0 + ln * 16, dmaoff = (wave * 64 + ln) * 16;
    const unsigned m0base = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * 1024));
    int t = 0, soff = 3 * TILEH;
    asm volatile(ATTN_HQ2_ASM
                 : "+v"(st[0].acc), "+v"(st[1].acc), "+v"(st[0].l), "+v"(st[1].l), "+s"(t), "+s"(soff)
                 : "v"(st[0].q0), "v"(st[0].q1), "v"(st[1].q0), "v"(st[1].q1), "v"(st[0].negm), "v"(st[1].negm),
                   "v"(klane), "v"(vlane), "v"(dmaoff), "s"(dk), "s"(dv), "s"(m0base), "s"(nfull)
                 : ATTN_HQ2_CLOBBERS);
  }
  // every piece of the ring this wave asked for has landed, and so has everybody else's: the last tile (fewer than KBX blocks
  // and / or a masked last block) is read from its ring buffer by the plain code
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nfull < ntiles) {
    const int lane2 = lane_id_fresh(), g2 = lane2 >> 5, lr2 = lane2 & 31;
    const char* kb = smem + (nfull & (NBUF - 1)) * BUFH;
    const char* vb = kb + TILEH;
    const int nb = nblk - nfull * KBX;
    for (int c = 0; c < nb; ++c) {
      const int blk = nfull * KBX + c;
      const KFrag kf = ld_k(kb + c * BLK_BYTES, g2, lr2);
      const VFrag vf = ld_v(vb + c * BLK_BYTES, lane2);
      if (partial && blk == nblk - 1) block_fast_plain<true, QB>(kf, vf, g2, st, blk * 32, L);
      else block_fast_plain<false, QB>(kf, vf, g2, st, blk * 32, L);
    }
    __syncthreads();
  }
  const int lane = lane_id_fresh(), g = lane >> 5, lr = lane & 31;
  float l_tot[QB];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    l_tot[j] = st[j].l[0] + __shfl_xor(st[j].l[0], 32);
    const bool valid = qb0 + j < nblk && (qb0 + j) * 32 + lr < L;
    bad = bad || (valid && !(l_tot[j] < L_OVERFLOW));  // overflow / NaN: this query needs the running-max pass
  }
  if (__any(bad) && lane == 0) flag[wave >> 1] = 1;
  __syncthreads();
  if (flag[0] | flag[1]) {  // workgroup-uniform: the wave pair(s) whose flag is up repeat, the other keeps what it has
    const bool mine = flag[wave >> 1] != 0;
    __syncthreads();
    attn_pass<true, QB>(rk, rv, smem, wave * 64 + lane, wave, lane, g, lr, st, L, nblk, mine);
    if (mine) {
#pragma unroll
      for (int j = 0; j < QB; ++j) l_tot[j] = st[j].l[0] + __shfl_xor(st[j].l[0], 32);
    }
  }

  const int seq = sh / p.heads, head = sh - seq * p.heads;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qi = (qb0 + j) * 32 + lr;
    const bool okq = qb0 + j < nblk && qi < L;
    const float gatev = okq ? p.gates[(long)sh * p.nbp * 32 + qi] : 0.f;
    const long orow = okq ? (long)(seq / p.o_div) * p.o_outer + (long)(seq % p.o_div) * p.o_inner + (long)qi * p.o_tok : 0;
    hf* op = reinterpret_cast<hf*>(p.out) + orow * p.inner + head * 32 + 8 * g;
    const float scale = okq ? gatev / l_tot[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {   // (the two halves of the wave exchange 4-feature runs: 16-byte stores, attn_frag_kernel)
      const unsigned x0 = pk2(st[j].acc[8 * k] * scale, st[j].acc[8 * k + 1] * scale);
      const unsigned x1 = pk2(st[j].acc[8 * k + 2] * scale, st[j].acc[8 * k + 3] * scale);
      const unsigned y0 = pk2(st[j].acc[8 * k + 4] * scale, st[j].acc[8 * k + 5] * scale);
      const unsigned y1 = pk2(st[j].acc[8 * k + 6] * scale, st[j].acc[8 * k + 7] * scale);
      auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
      if (okq) *reinterpret_cast<u32x4*>(op + 16 * k) = u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
  }
}
#endif   // !BT_HALF_IS_BF16


// =====================================================================================================================
// BT_PREC_F32X3: the same kernel design on hi + lo operands (fp32-class results, the path that carries the 1e-3 gate).
// A 32-token block of Q, K or V is [hi block 2 KB | lo block 2 KB] (written by the X3 QKV producers: gemm3.hip,
// qkv_front.hip): q = hi + lo to 2^-22.  Both products run on three MFMAs per fragment pair (lo . hi + hi . lo + hi . hi,
// fp32 accumulation); the probabilities are split in registers after the exponential (P = hi + lo), their row sums are
// plain fp32 adds of the unsplit values (the matrix pipe is the busy side here, the VALU has the slack), and the output
// leaves as fp32 (frontend: consumed by outff_fused_kernel<hl>) or as hl32 planes (main layers: A operand of the
// out-projection on gemm3.hip).  K / V tiles: 128 keys = 16 KB each, double buffered = 64 KB -> 2 workgroups per CU.
constexpr int BLKX_BYTES = 2 * BLK_BYTES;
constexpr float P_SHIFT_X = 3.f;
// (the tile size KBX -- 32-key blocks per LDS tile -- is a template parameter of this path: 4 = 128 keys, 64 KB of LDS,
// two workgroups per CU; 2 = 64 keys, 32 KB, three workgroups per CU at twice the barriers)

struct QStateX {
  hfx8 q0, q1, q0l, q1l;  // Q^T operand (hi, lo): dims [16g, 16g+8) and [16g+8, 16g+16) of this lane's query
  f32x16 acc;             // O^T accumulator
  float l;                // row sum of this lane's 16 keys per block (the two halves are added at the end)
  f32x4 l4;               // P16: the same sum taken from the ROUNDED probabilities on 4x4x4 MFMAs (four equal registers)
  float m;                // running max (SAFE pass)
  f32x16 negm;            // fast pass, one query block per wave: -(reference max) - P_SHIFT splat = accumulator input of
                          // the first score MFMA;  two query blocks per wave (no registers for a splat): only negm[0] is
                          // used, added to every score on the VALU
};
struct KFragX { hfx8 k0, k1, k0l, k1l; };
struct VFragX { hfx8 v0, v1, v0l, v1l; };
DEVI KFragX ld_kx(const char* kb, int g, int lr) {
  KFragX f;
  f.k0 = *reinterpret_cast<const hfx8*>(kb + ((2 * g) * 32 + lr) * 16);
  f.k1 = *reinterpret_cast<const hfx8*>(kb + ((2 * g + 1) * 32 + lr) * 16);
  f.k0l = *reinterpret_cast<const hfx8*>(kb + BLK_BYTES + ((2 * g) * 32 + lr) * 16);
  f.k1l = *reinterpret_cast<const hfx8*>(kb + BLK_BYTES + ((2 * g + 1) * 32 + lr) * 16);
  return f;
}
DEVI VFragX ld_vx(const char* vb, int lane) {
  VFragX f;
  f.v0 = *reinterpret_cast<const hfx8*>(vb + lane * 16);
  f.v1 = *reinterpret_cast<const hfx8*>(vb + 1024 + lane * 16);
  f.v0l = *reinterpret_cast<const hfx8*>(vb + BLK_BYTES + lane * 16);
  f.v1l = *reinterpret_cast<const hfx8*>(vb + BLK_BYTES + 1024 + lane * 16);
  return f;
}
// 8 fp32 probabilities -> packed hi halves and packed lo halves (p = hi + lo)
DEVI void split8(const f32x16& p, int s, u32x4& whi, u32x4& wlo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j += 2)   // (common.h)
    split_hl4(p[8 * s + 2 * j], p[8 * s + 2 * j + 1], p[8 * s + 2 * j + 2], p[8 * s + 2 * j + 3], h[j], l[j], h[j + 1], l[j + 1]);
  whi = u32x4{h[0], h[1], h[2], h[3]};
  wlo = u32x4{l[0], l[1], l[2], l[3]};
}
// Scores of one key block, S^T = K . Q^T (+ st.negm when INIT: the subtraction of the reference maximum rides on the
// accumulator input like in the half kernel -- starting from zero and subtracting on the VALU costs 16 adds and 16
// register clears per block and changed nothing measurable in the result), small terms first.
template <bool INIT, int QB>
DEVI void score_x(const KFragX& kf, const QStateX (&st)[QB], f32x16 (&sc)[QB]) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    if constexpr (INIT) sc[j] = st[j].negm; else zero16(sc[j]);
    sc[j] = MFMA32_H(kf.k0l, st[j].q0, sc[j]);
  }
#pragma unroll
  for (int j = 0; j < QB; ++j) sc[j] = MFMA32_H(kf.k1l, st[j].q1, sc[j]);
#pragma unroll
  for (int j = 0; j < QB; ++j) sc[j] = MFMA32_H(kf.k0, st[j].q0l, sc[j]);
#pragma unroll
  for (int j = 0; j < QB; ++j) sc[j] = MFMA32_H(kf.k1, st[j].q1l, sc[j]);
#pragma unroll
  for (int j = 0; j < QB; ++j) sc[j] = MFMA32_H(kf.k0, st[j].q0, sc[j]);
#pragma unroll
  for (int j = 0; j < QB; ++j) sc[j] = MFMA32_H(kf.k1, st[j].q1, sc[j]);
}
// scores -> probabilities -> row sums, split, O^T += V^T . P^T
// (Measured and rejected: accumulating each LDS tile's products in a fresh accumulator and adding it to the running output
// with VALU adds -- the attention output's 1.1e-5 relative error at L = 1500 on the outlier-key test did not move, nor did
// it when the scores stopped riding on the reference maximum: it is the 22-bit operand representation, 2^-22 |q| |k|
// per score, amplified by scores of magnitude 60, not the accumulation.)
// (PRESUB: the scores already carry the reference maximum -- it rode on the first score MFMA's accumulator input)
//
// P16 (round 5, the default since the flip-rate soak of profiles/r05_flip_frontier.txt): the probabilities enter the product
// as their fp16 hi parts only -- O^T += V_hi^T . P_hi^T + V_lo^T . P_hi^T, four MFMAs instead of six, no lo split -- and the
// row sums are taken from the SAME rounded values (v_mfma_f32_4x4x4_16b_f16 with an all-ones A operand, like the half
// kernel): numerator and denominator of the softmax see identical probabilities, fp16 subnormals included, so what the
// rounding leaves in O / l is sum_j p_j d_j (v_j - o) / sum_j p_j with |d_j| <= 2^-12 -- the spread of V around the output, not V
// itself.  An fp16 overflow of a probability is inf in the row sum (-> the query is left to attn_fix_x3_kernel).
DEVI unsigned cvt_pk_rn(float a, float b) {   // (the instruction the hand-scheduled loop uses; leading wait state: a and b may
  unsigned w;                                  // be v_exp results, which a non-transcendental VALU may not read right away)
  asm("s_nop 0\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(a), "v"(b));
  return w;
}
template <bool SAFE, bool MASK, int QB, bool PRESUB = (QB == 1), bool P16 = false>
DEVI void finish_x(f32x16 (&sc)[QB], const VFragX& vf, int g, QStateX (&st)[QB], int key0, int L) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    if constexpr (SAFE) {
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + crow(r, g) >= L) sc[j][r] = -1e30f;
      }
      float bm = sc[j][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[j][r]);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float m_new = fmaxf(st[j].m, bm);
      const float alpha = __builtin_amdgcn_exp2f(st[j].m - m_new);
      st[j].m = m_new;
      st[j].l *= alpha;
      if constexpr (P16) st[j].l4 *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j].acc[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[j][r] = __builtin_amdgcn_exp2f(sc[j][r] - m_new);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[j][r] = __builtin_amdgcn_exp2f(PRESUB ? sc[j][r] : sc[j][r] + st[j].negm[0]);
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + crow(r, g) >= L) sc[j][r] = 0.f;
      }
    }
    if constexpr (!P16) {
      const float a = (sc[j][0] + sc[j][1]) + (sc[j][2] + sc[j][3]), b = (sc[j][4] + sc[j][5]) + (sc[j][6] + sc[j][7]);
      const float c = (sc[j][8] + sc[j][9]) + (sc[j][10] + sc[j][11]), d = (sc[j][12] + sc[j][13]) + (sc[j][14] + sc[j][15]);
      st[j].l += (a + b) + (c + d);
    }
  }
  if constexpr (P16) {
    u32x4 w0[QB], w1[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w0[j][i] = cvt_pk_rn(sc[j][2 * i], sc[j][2 * i + 1]);
        w1[j][i] = cvt_pk_rn(sc[j][8 + 2 * i], sc[j][8 + 2 * i + 1]);
      }
      // (v_dot2_f32_f16 on the packed words instead -- it keeps fp16 subnormals like the MFMAs do, tools/ubench/dot2_denorm.hip --
      // was built and measured: 8 more VALU instructions per step against 32 fewer matrix-pipe cycles, attention 10.2 ms per
      // step against 9.9 ms, profiles/r05_ab_rowsum_dot2.txt: the VALU, not the pipe, is this loop's short side)
      rowsum8(st[j].l4, w0[j]);
      rowsum8(st[j].l4, w1[j]);
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v0l, __builtin_bit_cast(hfx8, w0[j]), st[j].acc);
#pragma unroll
    for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v1l, __builtin_bit_cast(hfx8, w1[j]), st[j].acc);
#pragma unroll
    for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v0, __builtin_bit_cast(hfx8, w0[j]), st[j].acc);
#pragma unroll
    for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v1, __builtin_bit_cast(hfx8, w1[j]), st[j].acc);
    return;
  }
  u32x4 h0[QB], l0[QB], h1[QB], l1[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    split8(sc[j], 0, h0[j], l0[j]);
    split8(sc[j], 1, h1[j], l1[j]);
  }
  // small terms first; consecutive MFMAs of different query blocks never share an accumulator
#pragma unroll
  for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v0l, __builtin_bit_cast(hfx8, h0[j]), st[j].acc);
#pragma unroll
  for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v1l, __builtin_bit_cast(hfx8, h1[j]), st[j].acc);
#pragma unroll
  for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v0, __builtin_bit_cast(hfx8, l0[j]), st[j].acc);
#pragma unroll
  for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v1, __builtin_bit_cast(hfx8, l1[j]), st[j].acc);
#pragma unroll
  for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v0, __builtin_bit_cast(hfx8, h0[j]), st[j].acc);
#pragma unroll
  for (int j = 0; j < QB; ++j) st[j].acc = MFMA32_H(vf.v1, __builtin_bit_cast(hfx8, h1[j]), st[j].acc);
}

template <int KBX>
DEVI void stage_tile_x(rsrc_t rk, rsrc_t rv, int tile, char* smem, int buf, int tid, int wave) {
  constexpr int TILEX_BYTES = KBX * BLKX_BYTES;
  char* kd = smem + buf * 2 * TILEX_BYTES + wave * 1024;
  char* vd = kd + TILEX_BYTES;
  const int so = tile * TILEX_BYTES;
  static_assert(TILEX_BYTES % 4096 == 0, "stage_tile_x copies 4 KB pieces");
#pragma unroll
  for (int i = 0; i < KBX; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lptr_t)(kd + i * 4096), 16, tid * 16, so + i * 4096, 0, 0);
#pragma unroll
  for (int i = 0; i < KBX; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lptr_t)(vd + i * 4096), 16, tid * 16, so + i * 4096, 0, 0);
}

// Plain pass, one key block at a time (SAFE: classic online softmax over all tiles; fast: only the tiles from `t0` on --
// the ragged / masked last tile): tile t + 1 is staged while tile t is consumed, one barrier per tile.
// On entry tile t0 must be readable in buffer t0 & 1 (and tile t0 + 1, if `next_staged`, on its way into the other one).
template <bool SAFE, int QB, int KBX, bool P16 = false>
DEVI void attn_tiles_x(rsrc_t rk, rsrc_t rv, char* smem, int tid, int wave, int lane, int g, int lr, QStateX (&st)[QB], int L,
                       int nblk, int t0, bool next_staged) {
  constexpr int TILEX_BYTES = KBX * BLKX_BYTES;
  const int ntiles = (nblk + KBX - 1) / KBX;
  const bool partial = (L & 31) != 0;
  for (int t = t0; t < ntiles; ++t) {
    if (t + 1 < ntiles && !(t == t0 && next_staged)) stage_tile_x<KBX>(rk, rv, t + 1, smem, (t + 1) & 1, tid, wave);
    const char* kb = smem + (t & 1) * 2 * TILEX_BYTES;
    const char* vb = kb + TILEX_BYTES;
    const int nb = min(KBX, nblk - t * KBX);
    for (int c = 0; c < nb; ++c) {
      const int blk = t * KBX + c;
      const KFragX kf = ld_kx(kb + c * BLKX_BYTES, g, lr);
      const VFragX vf = ld_vx(vb + c * BLKX_BYTES, lane);
      f32x16 sc[QB];
      score_x<!SAFE && QB == 1, QB>(kf, st, sc);
      if (partial && blk == nblk - 1) finish_x<SAFE, true, QB, (QB == 1), P16>(sc, vf, g, st, blk * 32, L);
      else finish_x<SAFE, false, QB, (QB == 1), P16>(sc, vf, g, st, blk * 32, L);
    }
    __syncthreads();  // tile t + 1 has landed (every wave waited for its own copies), tile t is free
  }
}

// Reference maximum of this wave's queries for the fast pass: their scores against the first TWO key blocks (tile 0 is in LDS
// buffer 0 whatever the tile size; one block for L <= 32) -> st[j].negm = -(max) - P_SHIFT on every register.  A later key
// may score up to 16 + P_SHIFT octaves above it before the fast pass overflows; 64 keys instead of 32 make that ~60 times
// rarer on Gaussian scores at no cost (the blocks are there).  Shared by every x3 kernel: the reference point is part of
// the arithmetic, and the kernels must agree bit for bit.
// (P_SHIFT_X: the x3 kernels' headroom above that maximum, in octaves ON TOP of the rounding-up below -- the effective shift is
// 3 .. 4 octaves where rounds 3 - 4 had exactly 4: every octave of shift pushes the lo halves of the probabilities one octave
// deeper into fp16's subnormal range, and with 4 + rounding the three-term kernels' error went from 6e-6 to 1e-5.)
// Round 5: the reference point is rounded UP to a whole octave.  Probabilities relative to two reference points that differ
// by whole octaves differ by a power of two, and both the hi + lo split and the fp16 rounding of P16 commute with that
// (outside fp16's subnormal range): the reference point is a function of the query's own scores, in whole octaves, whichever
// kernel form computes it -- and a query whose fast pass overflows is computed by attn_fix_x3_kernel on a reference point of
// its own (its row maximum), so no query's result depends on which other queries share its workgroup, 128 or 256 of them by
// the kernel a launch size selects.
// Round 5, second change: the query's OWN key block joins the two leading ones (`kdiag[j]`: the K fragments of key block
// qblk[j], fetched from global memory with the Q fragments).  With rotary positions the largest scores of a query sit near
// its own position far more often than among the first 64 frames of the piece: the workgroups of the benchmark's forward with an
// overflowing query fell from 1.1 % to the figure in DESIGN.md section 5 (21 % on the outlier stress weights before).
// (diag_max_x: that block's part, computed right behind the loads -- before any LDS-DMA is issued -- so that its fragments are
// dead when the passes start; dmax[j] = this lane's maximum over its 16 keys of the block.)
template <int QB>
DEVI void diag_max_x(const char* kseq, const int (&qblk)[QB], int g, int lr, const QStateX (&st)[QB], int L, float (&dmax)[QB]) {
  KFragX kd[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) kd[j] = ld_kx(kseq + (long)qblk[j] * BLKX_BYTES, g, lr);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (VGPR-returning loads: landed before the first LDS-DMA is issued)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    f32x16 sd;
    zero16(sd);
    sd = MFMA32_H(kd[j].k0l, st[j].q0, sd);
    sd = MFMA32_H(kd[j].k1l, st[j].q1, sd);
    sd = MFMA32_H(kd[j].k0, st[j].q0l, sd);
    sd = MFMA32_H(kd[j].k1, st[j].q1l, sd);
    sd = MFMA32_H(kd[j].k0, st[j].q0, sd);
    sd = MFMA32_H(kd[j].k1, st[j].q1, sd);
    float m = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, (32 * qblk[j] + crow(r, g) < L) ? sd[r] : -1e30f);
    dmax[j] = m;
  }
}
template <int QB>
DEVI void ref_max_x(const char* smem, int g, int lr, QStateX (&st)[QB], int L, int nblk, const float (&dmax)[QB]) {
  float bm[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) bm[j] = dmax[j];
  const int nref = min(2, nblk);
  for (int c = 0; c < nref; ++c) {
    const KFragX kf = ld_kx(smem + c * BLKX_BYTES, g, lr);
    f32x16 s0[QB];
    score_x<false, QB>(kf, st, s0);
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) bm[j] = fmaxf(bm[j], (32 * c + crow(r, g) < L) ? s0[j][r] : -1e30f);
  }
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const float m = ceilf(fmaxf(bm[j], __shfl_xor(bm[j], 32)));
#pragma unroll
    for (int r = 0; r < 16; ++r) st[j].negm[r] = -m - P_SHIFT_X;
  }
}

// Fast pass: reference maximum of every query from key block 0, then the key loop software-pipelined by hand over the
// tiles of KB unmasked blocks: the scores of block c + 1 are issued BEFORE the exponentials of block c (two score
// buffers alternate: the loop is unrolled over the tile, no register copies), one barrier per tile at its last block.
template <int QB, int KBX, bool P16 = false>
DEVI void attn_fast_x(rsrc_t rk, rsrc_t rv, char* smem, int tid, int wave, int lane, int g, int lr, QStateX (&st)[QB], int L,
                      int nblk, const float (&dmax)[QB]) {
  constexpr int TILEX_BYTES = KBX * BLKX_BYTES;
  static_assert(KBX % 2 == 0, "two score buffers alternate over the blocks of a tile");
  const int ntiles = (nblk + KBX - 1) / KBX;
  const bool partial = (L & 31) != 0;
  int nfull = nblk / KBX;  // tiles of KBX unmasked blocks
  if (partial && nfull * KBX == nblk) --nfull;
  stage_tile_x<KBX>(rk, rv, 0, smem, 0, tid, wave);
  __syncthreads();
  if (ntiles > 1) stage_tile_x<KBX>(rk, rv, 1, smem, 1, tid, wave);
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    zero16(st[j].acc);
    st[j].l = 0.f;
    st[j].l4 = f32x4{0.f, 0.f, 0.f, 0.f};
    st[j].m = -1e30f;
  }
  // PIPE (one query block per wave): the scores of block c + 1 are issued before the exponentials of block c, two score
  // buffers alternate.  Two query blocks per wave are two independent chains already (and a second score buffer does not
  // fit the register file): there the scores of the next block follow the current block's products.
  constexpr bool PIPE = QB == 1;
  f32x16 s2[PIPE ? 2 : 1][QB];  // scores of the current / the next block (compile-time indices: the tile loop is unrolled)
  ref_max_x<QB>(smem, g, lr, st, L, nblk, dmax);   // (workgroup-uniform)
  KFragX kf = ld_kx(smem, g, lr);
  // the scores of block 0 of tile 0 (one query block per wave: on the reference maximum, which rides on the accumulator input)
  if (QB == 1 && nfull > 0) score_x<true, QB>(kf, st, s2[0]);
  else if (QB != 1) score_x<false, QB>(kf, st, s2[0]);
  for (int t = 0; t < nfull; ++t) {
    const char* kb = smem + (t & 1) * 2 * TILEX_BYTES;
    const char* vb = kb + TILEX_BYTES;
    const char* kb_next = smem + ((t + 1) & 1) * 2 * TILEX_BYTES;
#pragma unroll
    for (int c = 0; c < KBX; ++c) {
      const VFragX vf = ld_vx(vb + c * BLKX_BYTES, lane);
      const int cur = PIPE ? (c & 1) : 0, nxt = PIPE ? ((c + 1) & 1) : 0;
      bool have_next = true;
      if (c + 1 < KBX) {
        kf = ld_kx(kb + (c + 1) * BLKX_BYTES, g, lr);
      } else {
        // every fragment read of tile t has arrived (the barrier's wait): barrier, refill, first block of tile t + 1
        __syncthreads();  // tile t + 1 has landed in every wave; nobody reads tile t any more
        if (t + 2 < ntiles) stage_tile_x<KBX>(rk, rv, t + 2, smem, t & 1, tid, wave);
        have_next = t + 1 < nfull;  // (uniform)
        if (have_next) kf = ld_kx(kb_next, g, lr);
      }
      if (PIPE && have_next) score_x<true, QB>(kf, st, s2[nxt]);
      finish_x<false, false, QB, (QB == 1), P16>(s2[cur], vf, g, st, 0, L);
      if (!PIPE && have_next) score_x<false, QB>(kf, st, s2[0]);
    }
  }
  if (nfull < ntiles)  // last tile: fewer than KBX blocks and / or a masked last block (staged by the loop / the prologue)
    attn_tiles_x<false, QB, KBX, P16>(rk, rv, smem, tid, wave, lane, g, lr, st, L, nblk, nfull, true);
  else
    __syncthreads();
}

// The rows of one query block (this lane: query qi, valid when okq) scaled by gate / row sum and stored in the launch's output form
// (OUT: 0 = hl32 / hl8 rows of the main layers, 1 = fp32 rows of the frontend); amax: range guard of the split.
template <int OUT>
DEVI void store_rows_x(const AttnFragP& p, const QStateX& st, float l_tot, int qi, bool okq, int sh, int g, float& amax) {
  const int seq = sh / p.heads, head = sh - seq * p.heads;
  const float gatev = okq ? p.gates[(long)sh * p.nbp * 32 + qi] : 0.f;
  const long orow = okq ? (long)(seq / p.o_div) * p.o_outer + (long)(seq % p.o_div) * p.o_inner + (long)qi * p.o_tok : 0;
  const float scale = okq ? gatev / l_tot : 0.f;
  if constexpr (OUT == 1) {
    float* op = reinterpret_cast<float*>(p.out) + orow * p.inner + head * 32 + 4 * g;
    if (okq) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        *reinterpret_cast<f32x4*>(op + 8 * a) = f32x4{st.acc[4 * a] * scale, st.acc[4 * a + 1] * scale,
                                                       st.acc[4 * a + 2] * scale, st.acc[4 * a + 3] * scale};
    }
  } else {
    // hl32 row: the head's 32 features are one [hi 32 | lo 32] group at half offset 64 head.  A lane holds 4-feature
    // runs 8 a + 4 g; the two halves of the wave exchange runs so that every lane stores 16-byte pieces (features
    // 16 k + 8 g .. + 7), for the hi and for the lo part.  (The exchange is executed by all lanes.)
    hf* op = reinterpret_cast<hf*>(p.out) + orow * 2 * p.inner + head * 64 + 8 * g;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned xh[2], xl[2], yh[2], yl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float a0 = st.acc[8 * k + 2 * i] * scale, a1 = st.acc[8 * k + 2 * i + 1] * scale;
        const float b0 = st.acc[8 * k + 4 + 2 * i] * scale, b1 = st.acc[8 * k + 4 + 2 * i + 1] * scale;
        split_hl4(a0, a1, b0, b1, xh[i], xl[i], yh[i], yl[i]);   // (common.h)
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(b0), fabsf(b1))));
        // (fmaxf drops NaNs: a NaN row of a VALID query -- acc(inf) x scale(0) -- must raise the range flag like an inf one,
        // ADVICE r5; lanes without a query (padding rows of the last block, queries left to the fix-up launch) hold whatever
        // their accumulators hold and are not stored)
        const float nan_probe = (a0 + a1) + (b0 + b1);
        amax = (okq && nan_probe != nan_probe) ? __builtin_inff() : amax;
      }
      auto h0 = __builtin_amdgcn_permlane32_swap(xh[0], yh[0], false, false);
      auto h1 = __builtin_amdgcn_permlane32_swap(xh[1], yh[1], false, false);
      if (p.out_f32 == 2) {
        // hl8 row (BT_OPT_X3_GEMM_FP8 = 2: the out-projection runs the fp8 cross terms): the group is [32 hi halves | 32 hi bytes |
        // 32 lo bytes]; this lane's 8 features 16 k + 8 g .. + 7 are 8 bytes in each byte section
        const float s0 = st.acc[8 * k] * scale, s1 = st.acc[8 * k + 1] * scale, s2 = st.acc[8 * k + 2] * scale, s3 = st.acc[8 * k + 3] * scale;
        const float t0 = st.acc[8 * k + 4] * scale, t1 = st.acc[8 * k + 5] * scale, t2 = st.acc[8 * k + 6] * scale, t3 = st.acc[8 * k + 7] * scale;
        auto b8 = __builtin_amdgcn_permlane32_swap(pk4_f8(s0, s1, s2, s3), pk4_f8(t0, t1, t2, t3), false, false);
        auto c8 = __builtin_amdgcn_permlane32_swap(lo4_f8(s0, s1, s2, s3, xh[0], xh[1]), lo4_f8(t0, t1, t2, t3, yh[0], yh[1]), false, false);
        if (okq) {
          *reinterpret_cast<u32x4*>(op + 16 * k) = u32x4{h0[0], h1[0], h0[1], h1[1]};
          char* gb = reinterpret_cast<char*>(op) - 16 * g;   // the group's first byte
          *reinterpret_cast<u32x2*>(gb + 64 + 16 * k + 8 * g) = u32x2{b8[0], b8[1]};
          *reinterpret_cast<u32x2*>(gb + 96 + 16 * k + 8 * g) = u32x2{c8[0], c8[1]};
        }
      } else {
      auto l0 = __builtin_amdgcn_permlane32_swap(xl[0], yl[0], false, false);
      auto l1 = __builtin_amdgcn_permlane32_swap(xl[1], yl[1], false, false);
      if (okq) {
        *reinterpret_cast<u32x4*>(op + 16 * k) = u32x4{h0[0], h1[0], h0[1], h1[1]};
        *reinterpret_cast<u32x4*>(op + 32 + 16 * k) = u32x4{l0[0], l1[0], l0[1], l1[1]};
      }
      }
    }
  }
}

// OUT: 0 = hl32 planes [rows, 2 inner] (main layers), 1 = fp32 [rows, inner] (frontend)
template <int QB, int OUT, int KBX, int MINW, bool P16>
__global__ __launch_bounds__(256, MINW) void attn_frag_x3_kernel(const AttnFragP p, int nqt, int sh_total) {
  constexpr int TILEX_BYTES = KBX * BLKX_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILEX_BYTES + 16];
  const int bid = blockIdx.x;
  const int idx = bid >> 3;
  const int sh = (idx / nqt) * 8 + (bid & 7);
  const int qt = idx % nqt;
  if (sh >= sh_total) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, lr = lane & 31;
  const int L = p.L;
  const int nblk = (L + 31) >> 5;
  const long seq_off = (long)sh * p.nbp * BLKX_BYTES;
  const char* kseq = reinterpret_cast<const char*>(p.k) + seq_off;
  const char* vseq = reinterpret_cast<const char*>(p.v) + seq_off;
  QStateX st[QB];
  int qbi[QB];
  const int qb0 = (qt * 4 + wave) * QB;  // this wave's first query block
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qbc = min(qb0 + j, nblk - 1);
    qbi[j] = qbc;
    const char* qblk = reinterpret_cast<const char*>(p.q) + seq_off + (long)qbc * BLKX_BYTES;
    st[j].q0 = *reinterpret_cast<const hfx8*>(qblk + ((2 * g) * 32 + lr) * 16);
    st[j].q1 = *reinterpret_cast<const hfx8*>(qblk + ((2 * g + 1) * 32 + lr) * 16);
    st[j].q0l = *reinterpret_cast<const hfx8*>(qblk + BLK_BYTES + ((2 * g) * 32 + lr) * 16);
    st[j].q1l = *reinterpret_cast<const hfx8*>(qblk + BLK_BYTES + ((2 * g + 1) * 32 + lr) * 16);
  }
  // (the Q loads return to VGPRs: they must have landed before the first LDS-DMA is in flight -- counted vmcnt waits are
  // wrong across the two kinds of loads)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const unsigned seq_bytes = (unsigned)p.nbp * BLKX_BYTES;
  const rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kseq), 0, seq_bytes, 0x00020000);
  const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vseq), 0, seq_bytes, 0x00020000);
  float dmax[QB];   // the queries' own key block in the reference maximum (diag_max_x: its loads land with the Q fragments')
  diag_max_x<QB>(kseq, qbi, g, lr, st, L, dmax);
  attn_fast_x<QB, KBX, P16>(rk, rv, smem, tid, wave, lane, g, lr, st, L, nblk, dmax);
  auto lane_sum = [&](int j) { return P16 ? st[j].l4[0] : st[j].l; };
  float l_tot[QB];
  bool bad[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    l_tot[j] = lane_sum(j) + __shfl_xor(lane_sum(j), 32);
    const bool valid = qb0 + j < nblk && (qb0 + j) * 32 + lr < L;
    // the row sum is taken from the UNSPLIT fp32 probabilities, so an fp16 overflow of a hi part (p > 65504) does not turn
    // it into inf: but such a p makes the sum exceed 65504 as well -> this query needs its row maximum as the reference point
    // (a sum that large without any single overflow only costs the repeat).  (P16: the sum is taken from the rounded values --
    // inf then.)  Such a query is not stored here: its bit goes into the launch's overflow map and attn_fix_x3_kernel, the
    // next launch on the stream, computes it (round 5; rounds 3 - 5 repeated the whole workgroup's pass in place).
    bad[j] = valid && !(l_tot[j] < 65504.f);
    const unsigned long long bw = __ballot(bad[j]);
    if (lane == 0 && qb0 + j < nblk) p.fix_mask[(long)sh * p.nbp + qb0 + j] = (int)(unsigned)bw;
  }
#ifdef BT_DEV   // development: workgroups of the launch (word 2 of the status block; attn_fix_x3_kernel counts words 1 and 3)
  if (p.status && tid == 0) atomicAdd(p.status + 2, 1);
#endif

  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qi = (qb0 + j) * 32 + lr;
    store_rows_x<OUT>(p, st[j], l_tot[j], qi, qb0 + j < nblk && qi < L && !bad[j], sh, g, amax);
  }
  if (OUT == 0 && p.status && __any(!(amax <= (p.out_f32 == 2 ? HL8_ACT_MAX : 65504.f))) && lane == 0) atomicOr(p.status, 1);
}


// =====================================================================================================================
// BT_PREC_F32X3, round 4: the same product on a HAND-SCHEDULED key loop (tools/gen/attn_x3_loop.py -> attn_x3_loop.inc).
// One wave = TWO 32-query blocks (every K / V fragment read from LDS feeds twice the matrix work), 256 queries per
// workgroup, two workgroups per CU (<= 256 registers per lane); K / V in a ring of four 64-key tiles (16 KB each: [K 8 KB |
// V 8 KB]) filled by LDS-DMA three tiles ahead.  The loop over the full tiles is ONE asm statement: the two query blocks run
// half a step apart, so that the 12 MFMAs of one block (its P.V products of the previous step and its scores of the next
// one) always sit beside the 56 VALU instructions of the other block's softmax step -- five hand-placed single-issue
// instructions per 32-cycle MFMA gap, which is what one wave can issue beside a saturated matrix pipe
// (MI355X_MICROARCH.md).  hipcc's own schedule of the same work (attn_frag_x3_kernel above) clusters the split blocks and
// leaves runs of four back-to-back MFMAs next to runs of twenty VALU instructions: 45 - 50 % matrix-pipe busy.
// Prologue (Q fragments, reference maximum), the ragged / masked last tile, the overflow map and the output
// stores are the C++ of the kernel above, instantiated for two query blocks per wave.
#include "attn_x3_loop.inc"

template <int OUT, bool P16>
__global__ __launch_bounds__(256, 2) void attn_frag_x3q2_kernel(const AttnFragP p, int nqt, int sh_total) {
  constexpr int QB = 2, KBX = ATTN_X3Q2_KBX, NBUF = ATTN_X3Q2_NBUF;
  constexpr int TILEX_BYTES = KBX * BLKX_BYTES, BUF_BYTES = 2 * TILEX_BYTES;
  static_assert(KBX == 2 && NBUF == 4 && BUF_BYTES == 16384, "the generated loop is written for four 16 KB ring buffers");
  __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF_BYTES + 16];
  const int bid = blockIdx.x;
  const int idx = bid >> 3;
  const int sh = (idx / nqt) * 8 + (bid & 7);
  const int qt = idx % nqt;
  if (sh >= sh_total) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, lr = lane & 31;
  const int L = p.L;
  const int nblk = (L + 31) >> 5;
  const long seq_off = (long)sh * p.nbp * BLKX_BYTES;
  const char* kseq = reinterpret_cast<const char*>(p.k) + seq_off;
  const char* vseq = reinterpret_cast<const char*>(p.v) + seq_off;
  QStateX st[QB];
  const int qb0 = (qt * 4 + wave) * QB;  // this wave's first query block
  // (the Q fragments are loaded twice: the asm statement below takes 140 of the 256 registers for itself, and what is
  // only needed again behind it -- Q for the last, ragged tile -- is cheaper fetched again than kept)
  auto load_q = [&]() {
    const int ln = lane_id_fresh(), gg = ln >> 5, ll = ln & 31;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int qbc = min(qb0 + j, nblk - 1);
      const char* qblk = reinterpret_cast<const char*>(p.q) + seq_off + (long)qbc * BLKX_BYTES;
      st[j].q0 = *reinterpret_cast<const hfx8*>(qblk + ((2 * gg) * 32 + ll) * 16);
      st[j].q1 = *reinterpret_cast<const hfx8*>(qblk + ((2 * gg + 1) * 32 + ll) * 16);
      st[j].q0l = *reinterpret_cast<const hfx8*>(qblk + BLK_BYTES + ((2 * gg) * 32 + ll) * 16);
      st[j].q1l = *reinterpret_cast<const hfx8*>(qblk + BLK_BYTES + ((2 * gg + 1) * 32 + ll) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (VGPR-returning loads and LDS-DMA do not retire in one order)
    __builtin_amdgcn_sched_barrier(0);
  };
  load_q();
  float dmax[QB];   // the queries' own key block in the reference maximum: before any LDS-DMA is in flight (diag_max_x)
  {
    const int qbi[QB] = {min(qb0, nblk - 1), min(qb0 + 1, nblk - 1)};
    diag_max_x<QB>(kseq, qbi, g, lr, st, L, dmax);
  }
  const unsigned seq_bytes = (unsigned)p.nbp * BLKX_BYTES;   // (tiles beyond it read as zeros: the ring is always refilled)
  const rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kseq), 0, seq_bytes, 0x00020000);
  const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vseq), 0, seq_bytes, 0x00020000);
  const int ntiles = (nblk + KBX - 1) / KBX;
  const bool partial = (L & 31) != 0;
  int nfull = nblk / KBX;  // tiles of KBX unmasked blocks
  if (partial && nfull * KBX == nblk) --nfull;
  nfull = __builtin_amdgcn_readfirstlane(nfull);
  // ring: tile t lives in buffer t & 3 = [K tile | V tile]; tiles 0, 1, 2 now, tile t + 3 from inside tile t's second block
  auto stage_ring = [&](int tile) {
    char* kd = smem + (tile & (NBUF - 1)) * BUF_BYTES + wave * 1024;
    const int so = tile * TILEX_BYTES;
#pragma unroll
    for (int i = 0; i < KBX; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lptr_t)(kd + i * 4096), 16, tid * 16, so + i * 4096, 0, 0);
#pragma unroll
    for (int i = 0; i < KBX; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lptr_t)(kd + TILEX_BYTES + i * 4096), 16, tid * 16, so + i * 4096, 0, 0);
  };
  // One fast pass: the ring holds (or is receiving) tiles 0, 1, 2, st[j].negm the reference points; the full tiles on the asm
  // loop, the last, ragged tile on the plain code.  On return every LDS-DMA of the workgroup has landed.
  auto fast_pass = [&]() {
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      zero16(st[j].acc);
      st[j].l = 0.f;
      st[j].l4 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float nm0 = st[0].negm[0], nm1 = st[1].negm[0];   // (what the plain code below needs of the splats)
    if (nfull > 0) {
    // operand words of the two buffer descriptors as plain SGPR quads (an asm operand cannot be a __amdgpu_buffer_rsrc_t)
    const unsigned long long ka = (unsigned long long)kseq, va = (unsigned long long)vseq;
    const u32x4 dk = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((ka >> 32) & 0xffffu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)seq_bytes), 0x00020000u};
    const u32x4 dv = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)va), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((va >> 32) & 0xffffu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)seq_bytes), 0x00020000u};
    const int ln = lane_id_fresh();
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;   // LDS byte address of the ring
    const unsigned klane = lds0 + ((2 * (ln >> 5)) * 32 + (ln & 31)) * 16, vlane = lds0 + ln * 16, dmaoff = (wave * 64 + ln) * 16;
    const unsigned m0base = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * 1024));
    const float m1 = -1.0f;
    int t = 0, soff = 3 * TILEX_BYTES;
    if constexpr (P16)
      asm volatile(ATTN_X3Q2P_ASM
                   : "+v"(st[0].acc), "+v"(st[1].acc), "+v"(st[0].l4), "+v"(st[1].l4), "+s"(t), "+s"(soff)
                   : "v"(st[0].q0), "v"(st[0].q1), "v"(st[0].q0l), "v"(st[0].q1l), "v"(st[1].q0), "v"(st[1].q1), "v"(st[1].q0l), "v"(st[1].q1l),
                     "v"(st[0].negm), "v"(st[1].negm), "v"(klane), "v"(vlane), "v"(dmaoff), "s"(dk), "s"(dv), "s"(m0base), "s"(m1), "s"(nfull)
                   : ATTN_X3Q2P_CLOBBERS);
    else
      asm volatile(ATTN_X3Q2_ASM
                   : "+v"(st[0].acc), "+v"(st[1].acc), "+v"(st[0].l), "+v"(st[1].l), "+s"(t), "+s"(soff)
                   : "v"(st[0].q0), "v"(st[0].q1), "v"(st[0].q0l), "v"(st[0].q1l), "v"(st[1].q0), "v"(st[1].q1), "v"(st[1].q0l), "v"(st[1].q1l),
                     "v"(st[0].negm), "v"(st[1].negm), "v"(klane), "v"(vlane), "v"(dmaoff), "s"(dk), "s"(dv), "s"(m0base), "s"(m1), "s"(nfull)
                   : ATTN_X3Q2_CLOBBERS);
    }
    // every piece of the ring this wave asked for has landed, and so has everybody else's: the last tile (fewer than KBX
    // blocks and / or a masked last block) is read from its ring buffer by the plain code
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nfull < ntiles) {
      load_q();
      // (the same arithmetic as the key loop and as attn_frag_x3_kernel, bit for bit: the reference maximum rides on the first
      // score MFMA's accumulator input -- results must not depend on which kernel a launch size selects)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[0].negm[r] = nm0;
        st[1].negm[r] = nm1;
      }
      const int lane2 = lane_id_fresh(), g2 = lane2 >> 5, lr2 = lane2 & 31;
      const char* kb = smem + (nfull & (NBUF - 1)) * BUF_BYTES;
      const char* vb = kb + TILEX_BYTES;
      const int nb = nblk - nfull * KBX;
      for (int c = 0; c < nb; ++c) {
        const int blk = nfull * KBX + c;
        const KFragX kf = ld_kx(kb + c * BLKX_BYTES, g2, lr2);
        const VFragX vf = ld_vx(vb + c * BLKX_BYTES, lane2);
        f32x16 sc[QB];
        score_x<true, QB>(kf, st, sc);
        if (partial && blk == nblk - 1) finish_x<false, true, QB, true, P16>(sc, vf, g2, st, blk * 32, L);
        else finish_x<false, false, QB, true, P16>(sc, vf, g2, st, blk * 32, L);
      }
      __syncthreads();
    }
  };
  stage_ring(0);
  stage_ring(1);
  stage_ring(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 (this wave's four pieces of it) has landed
  __syncthreads();
  ref_max_x<QB>(smem, g, lr, st, L, nblk, dmax);
  fast_pass();
  // (Lane-derived values are taken again from an operand hipcc cannot see through wherever they are needed: nothing but the
  // softmax state may stay live across an asm statement -- the three-term one leaves 18 registers -- and a value kept was a
  // spill, which the ISA lint does not allow next to LDS-DMA.)
  auto lane_sum = [&](int j) { return P16 ? st[j].l4[0] : st[j].l; };
  const int laneE = lane_id_fresh(), gE = laneE >> 5, lrE = laneE & 31;
  float l_tot[QB];
  bool bad[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    l_tot[j] = lane_sum(j) + __shfl_xor(lane_sum(j), 32);
    const bool valid = qb0 + j < nblk && (qb0 + j) * 32 + lrE < L;
    bad[j] = valid && !(l_tot[j] < 65504.f);   // (see attn_frag_x3_kernel: left to attn_fix_x3_kernel, through the overflow map)
    const unsigned long long bw = __ballot(bad[j]);
    if (laneE == 0 && qb0 + j < nblk) p.fix_mask[(long)sh * p.nbp + qb0 + j] = (int)(unsigned)bw;
  }
#ifdef BT_DEV
  if (p.status && tid == 0) atomicAdd(p.status + 2, 1);
#endif

  const int seq = sh / p.heads, head = sh - seq * p.heads;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qi = (qb0 + j) * 32 + lrE;
    const bool okq = qb0 + j < nblk && qi < L && !bad[j];
    const float gatev = okq ? p.gates[(long)sh * p.nbp * 32 + qi] : 0.f;
    const long orow = okq ? (long)(seq / p.o_div) * p.o_outer + (long)(seq % p.o_div) * p.o_inner + (long)qi * p.o_tok : 0;
    const float scale = okq ? gatev / l_tot[j] : 0.f;
    if constexpr (OUT == 1) {
      float* op = reinterpret_cast<float*>(p.out) + orow * p.inner + head * 32 + 4 * gE;
      if (okq) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          *reinterpret_cast<f32x4*>(op + 8 * a) = f32x4{st[j].acc[4 * a] * scale, st[j].acc[4 * a + 1] * scale,
                                                         st[j].acc[4 * a + 2] * scale, st[j].acc[4 * a + 3] * scale};
      }
    } else {
      hf* op = reinterpret_cast<hf*>(p.out) + orow * 2 * p.inner + head * 64 + 8 * gE;   // (hl32 row: see attn_frag_x3_kernel)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        unsigned xh[2], xl[2], yh[2], yl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float a0 = st[j].acc[8 * k + 2 * i] * scale, a1 = st[j].acc[8 * k + 2 * i + 1] * scale;
          const float b0 = st[j].acc[8 * k + 4 + 2 * i] * scale, b1 = st[j].acc[8 * k + 4 + 2 * i + 1] * scale;
          split_hl4(a0, a1, b0, b1, xh[i], xl[i], yh[i], yl[i]);
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(b0), fabsf(b1))));
        }
        auto h0 = __builtin_amdgcn_permlane32_swap(xh[0], yh[0], false, false);
        auto h1 = __builtin_amdgcn_permlane32_swap(xh[1], yh[1], false, false);
        if (p.out_f32 == 2) {
          // hl8 row (BT_OPT_X3_GEMM_FP8 = 2: the out-projection runs the fp8 cross terms): the group is [32 hi halves | 32 hi bytes |
          // 32 lo bytes]; this lane's 8 features 16 k + 8 g .. + 7 are 8 bytes in each byte section
          const float s0 = st[j].acc[8 * k] * scale, s1 = st[j].acc[8 * k + 1] * scale, s2 = st[j].acc[8 * k + 2] * scale, s3 = st[j].acc[8 * k + 3] * scale;
          const float t0 = st[j].acc[8 * k + 4] * scale, t1 = st[j].acc[8 * k + 5] * scale, t2 = st[j].acc[8 * k + 6] * scale, t3 = st[j].acc[8 * k + 7] * scale;
          auto b8 = __builtin_amdgcn_permlane32_swap(pk4_f8(s0, s1, s2, s3), pk4_f8(t0, t1, t2, t3), false, false);
          auto c8 = __builtin_amdgcn_permlane32_swap(lo4_f8(s0, s1, s2, s3, xh[0], xh[1]), lo4_f8(t0, t1, t2, t3, yh[0], yh[1]), false, false);
          if (okq) {
            *reinterpret_cast<u32x4*>(op + 16 * k) = u32x4{h0[0], h1[0], h0[1], h1[1]};
            char* gb = reinterpret_cast<char*>(op) - 16 * gE;   // the group's first byte
            *reinterpret_cast<u32x2*>(gb + 64 + 16 * k + 8 * gE) = u32x2{b8[0], b8[1]};
            *reinterpret_cast<u32x2*>(gb + 96 + 16 * k + 8 * gE) = u32x2{c8[0], c8[1]};
          }
        } else {
        auto l0 = __builtin_amdgcn_permlane32_swap(xl[0], yl[0], false, false);
        auto l1 = __builtin_amdgcn_permlane32_swap(xl[1], yl[1], false, false);
        if (okq) {
          *reinterpret_cast<u32x4*>(op + 16 * k) = u32x4{h0[0], h1[0], h0[1], h1[1]};
          *reinterpret_cast<u32x4*>(op + 32 + 16 * k) = u32x4{l0[0], l1[0], l0[1], l1[1]};
        }
        }
      }
    }
  }
  if (OUT == 0 && p.status && __any(!(amax <= (p.out_f32 == 2 ? HL8_ACT_MAX : 65504.f))) && laneE == 0) atomicOr(p.status, 1);
}


// =====================================================================================================================
// The queries whose fast pass overflowed fp16 (a key more than 16 + P_SHIFT octaves above their reference point), round 5.
// The kernels above leave them out and set their bits in the launch's overflow map (p.fix_mask: one word per (sequence, head)
// pair and query block); this launch follows on the stream, its workgroups walk the pairs: no bit -> next pair (188 bytes read); else the
// overflowed queries of the pair are GATHERED, 32 per round, and computed the way the in-place repeat of rounds 3 - 5 did --
// row maxima over all keys from the hi . hi scores, then the fast pass's own arithmetic (score_x / finish_x) on the reference
// point 14 - ceil(maximum) -- with the KEYS split over the four waves: wave w takes key blocks [w n, (w + 1) n), n =
// ceil(blocks / 4), fragments straight from global memory (the K / V blocks are fragment-major: no LDS staging, no barrier in
// the loops), the partial maxima, outputs and row sums meet in LDS and are added in wave order.  A query's result is a
// function of its own scores and of L alone -- not of its neighbours, the batch, or the kernel form in front.
// Why a launch of its own: the in-place repeat cost a workgroup 2.4 passes for a handful of queries (0.6 % of the queries of
// the outlier stress weights sit in 18 % of the frontend's workgroups: 11.8 % of all attention workgroups repeated); gathered,
// a pair's overflowed queries cost two quarter-passes of ONE query block per 32 of them, and the kernels in front lose the
// repeat's code, barrier and registers.
// (Four waves of <= 256 registers, two workgroups per CU: eight waves per workgroup -- half the key blocks per wave -- need a whole
// CU's registers at once and wait for the other stream's kernels to drain: +0.5 % on the benchmark step, same box, A/B.)
constexpr int FIX_GRID = 512, FIX_NW = 4;   // workgroups of the fix-up launch and waves per workgroup
template <int OUT, bool P16>
__global__ __launch_bounds__(64 * FIX_NW, 2) void attn_fix_x3_kernel(const AttnFragP p, int sh_total) {
  __shared__ float s_max[FIX_NW][32];
  __shared__ float s_part[FIX_NW][64][17];   // [wave][lane][16 outputs + row sum] (17: conflict-free columns)
  __shared__ int s_tot[FIX_NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, lr = lane & 31;
  const int L = p.L;
  const int nblk = (L + 31) >> 5;
  const int per = (nblk + FIX_NW - 1) / FIX_NW;
  const int b0 = min(wave * per, nblk), b1 = min(b0 + per, nblk);   // this wave's key blocks
  const bool partial = (L & 31) != 0;
  float amax = 0.f;
  for (int sweep = 0; sweep * FIX_GRID * FIX_NW < sh_total; ++sweep) {
    // every wave counts the bits of ONE pair's map (one load per sweep; L <= 2048 frames in one piece): pair blockIdx + FIX_GRID wave
    const int first = sweep * FIX_GRID * FIX_NW + blockIdx.x;
    {
      const int shw = first + FIX_GRID * wave;
      int total = 0;
      if (shw < sh_total) {
        const int* mww = p.fix_mask + (long)shw * p.nbp;
        for (int c0 = 0; c0 < nblk; c0 += 64) {
          int c = c0 + lane < nblk ? __popc((unsigned)mww[c0 + lane]) : 0;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
          total += c;
        }
      }
      if (lane == 0) s_tot[wave] = total;
    }
    __syncthreads();
    for (int w2 = 0; w2 < FIX_NW; ++w2) {
      const int total = s_tot[w2];   // (uniform)
      if (total == 0) continue;
      const int sh = first + FIX_GRID * w2;
      const int* mw = p.fix_mask + (long)sh * p.nbp;
#ifdef BT_DEV   // development: queries recomputed here / pairs with any (words 1, 3 of the status block)
      if (p.status && tid == 0) { atomicAdd(p.status + 1, total); atomicAdd(p.status + 3, 1); }
#endif
      const long seq_off = (long)sh * p.nbp * BLKX_BYTES;
      const char* kseq = reinterpret_cast<const char*>(p.k) + seq_off;
      const char* vseq = reinterpret_cast<const char*>(p.v) + seq_off;
      for (int base = 0; base < total; base += 32) {
        // this lane's query: the r-th set bit of the pair's map (lanes lr and lr + 32 hold the same query, as everywhere)
        const int r = base + lr;
        const bool okq = r < total;
        int blk = 0, tok = 0, pre = 0;
        unsigned wsel = 0;
        for (int c0 = 0; c0 < nblk; c0 += 64) {
          const int wl = c0 + lane < nblk ? mw[c0 + lane] : 0;
          const int nb = min(64, nblk - c0);
          for (int b = 0; b < nb; ++b) {
            const unsigned wb = (unsigned)__builtin_amdgcn_readlane(wl, b);
            const int c = __popc(wb);
            if (r >= pre && r < pre + c) { blk = c0 + b; wsel = wb; tok = r - pre; }
            pre += c;
          }
        }
        for (int i = tok; i > 0; --i) wsel &= wsel - 1;   // drop the lower set bits
        tok = okq ? __ffs((int)wsel) - 1 : 0;
        const int qi = blk * 32 + tok;
        QStateX st[1];
        const char* qblk = reinterpret_cast<const char*>(p.q) + seq_off + (long)blk * BLKX_BYTES;
        st[0].q0 = *reinterpret_cast<const hfx8*>(qblk + ((2 * g) * 32 + tok) * 16);
        st[0].q1 = *reinterpret_cast<const hfx8*>(qblk + ((2 * g + 1) * 32 + tok) * 16);
        st[0].q0l = *reinterpret_cast<const hfx8*>(qblk + BLK_BYTES + ((2 * g) * 32 + tok) * 16);
        st[0].q1l = *reinterpret_cast<const hfx8*>(qblk + BLK_BYTES + ((2 * g + 1) * 32 + tok) * 16);
        // row maxima over this wave's keys: hi . hi scores, two MFMAs per block (a reference point needs no more).  The loop is
        // latency, not work: the fragments of up to FIX_A blocks are requested at once and consumed as they arrive.
        float bm = -1e30f;
        constexpr int FIX_A = 6, FIX_B = 2;
        for (int c0 = b0; c0 < b1; c0 += FIX_A) {
          hfx8 ka[FIX_A][2];
#pragma unroll
          for (int c = 0; c < FIX_A; ++c)
            if (c0 + c < b1) {
              const char* kb = kseq + (long)(c0 + c) * BLKX_BYTES;
              ka[c][0] = *reinterpret_cast<const hfx8*>(kb + ((2 * g) * 32 + lr) * 16);
              ka[c][1] = *reinterpret_cast<const hfx8*>(kb + ((2 * g + 1) * 32 + lr) * 16);
            }
#pragma unroll
          for (int c = 0; c < FIX_A; ++c)
            if (c0 + c < b1) {
              f32x16 sc;
              zero16(sc);
              sc = MFMA32_H(ka[c][0], st[0].q0, sc);
              sc = MFMA32_H(ka[c][1], st[0].q1, sc);
#pragma unroll
              for (int i = 0; i < 16; ++i) bm = fmaxf(bm, ((c0 + c) * 32 + crow(i, g) < L) ? sc[i] : -1e30f);
            }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        if (g == 0) s_max[wave][lr] = bm;
        __syncthreads();
        {
          // The reference point puts the row maximum at 2^14 .. 2^15 (the largest probability an fp16 hi part holds is 65504;
          // the maximum comes from hi . hi scores and may sit 0.05 octaves low): the fast pass's own shift (P_SHIFT octaves
          // BELOW its reference point) buys headroom for keys that have not been seen yet; here all have been, and every octave
          // not spent on headroom keeps one more octave of small probabilities out of fp16's subnormal range.  Whole octaves:
          // see ref_max_x.
          float m = s_max[0][lr];
#pragma unroll
          for (int w = 1; w < FIX_NW; ++w) m = fmaxf(m, s_max[w][lr]);
          m = ceilf(m);
#pragma unroll
          for (int i = 0; i < 16; ++i) st[0].negm[i] = 14.0f - m;
        }
        zero16(st[0].acc);
        st[0].l = 0.f;
        st[0].l4 = f32x4{0.f, 0.f, 0.f, 0.f};
        st[0].m = -1e30f;
        for (int c0 = b0; c0 < b1; c0 += FIX_B) {   // (FIX_B blocks' K and V fragments in flight at once)
          KFragX kq[FIX_B];
          VFragX vq[FIX_B];
#pragma unroll
          for (int c = 0; c < FIX_B; ++c)
            if (c0 + c < b1) {
              kq[c] = ld_kx(kseq + (long)(c0 + c) * BLKX_BYTES, g, lr);
              vq[c] = ld_vx(vseq + (long)(c0 + c) * BLKX_BYTES, lane);
            }
#pragma unroll
          for (int c = 0; c < FIX_B; ++c)
            if (c0 + c < b1) {
              const int b = c0 + c;
              f32x16 sc[1];
              score_x<true, 1>(kq[c], st, sc);
              if (partial && b == nblk - 1) finish_x<false, true, 1, true, P16>(sc, vq[c], g, st, b * 32, L);
              else finish_x<false, false, 1, true, P16>(sc, vq[c], g, st, b * 32, L);
            }
        }
        {
          const float ls = P16 ? st[0].l4[0] : st[0].l;
#pragma unroll
          for (int i = 0; i < 16; ++i) s_part[wave][lane][i] = st[0].acc[i];
          s_part[wave][lane][16] = ls + __shfl_xor(ls, 32);
        }
        __syncthreads();
        if (wave == 0) {   // the partial results in wave order
          float l_tot = s_part[0][lane][16];
#pragma unroll
          for (int i = 0; i < 16; ++i) st[0].acc[i] = s_part[0][lane][i];
#pragma unroll
          for (int w = 1; w < FIX_NW; ++w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[0].acc[i] += s_part[w][lane][i];
            l_tot += s_part[w][lane][16];
          }
          // this kernel is the last resort of a query: a row sum that still is not a finite number (a probability beyond fp16
          // even relative to the hi . hi row maximum, P16: l4 = inf) would be stored as gate / inf = 0 times inf = NaN, which no
          // amax sees -- raise the range flag instead, the forward is then repeated on the exact fp32 path
          if (okq && !(l_tot < __builtin_inff())) amax = __builtin_inff();
          store_rows_x<OUT>(p, st[0], l_tot, qi, okq, sh, g, amax);
        }
        __syncthreads();   // (s_max / s_part are free for the next 32 queries)
      }
    }
    __syncthreads();   // (s_tot is free for the next sweep)
  }
  if (p.status && __any(!(amax <= (p.out_f32 == 2 ? HL8_ACT_MAX : 65504.f))) && lane == 0) atomicOr(p.status, 1);   // (OUT = 1: amax is 0 or inf)
}

}  // namespace

int attn_frag_blocks(int L) { return ((L + 31) / 32 + KB - 1) / KB * KB; }

template <int ABL, int QB>
static void launch_v(const AttnFragP& p, hipStream_t s) {
  const int nblk = (p.L + 31) / 32;
  const int nqt = (nblk + 4 * QB - 1) / (4 * QB);
  const long sh = (long)p.n_seq * p.heads;
  const long grid = (sh + 7) / 8 * 8 * nqt;
  hipLaunchKernelGGL((attn_frag_kernel<ABL, QB>), dim3((unsigned)grid), dim3(256), 0, s, p, nqt, (int)sh);
}

template <int QB, int OUT, int KBX, int MINW, bool P16>
static void launch_x3(const AttnFragP& p, hipStream_t s) {
  const int nblk = (p.L + 31) / 32;
  const int nqt = (nblk + 4 * QB - 1) / (4 * QB);
  const long sh = (long)p.n_seq * p.heads;
  const long grid = (sh + 7) / 8 * 8 * nqt;
  hipLaunchKernelGGL((attn_frag_x3_kernel<QB, OUT, KBX, MINW, P16>), dim3((unsigned)grid), dim3(256), 0, s, p, nqt, (int)sh);
}

template <int OUT, bool P16>
static void launch_x3q2(const AttnFragP& p, hipStream_t s) {
  const int nblk = (p.L + 31) / 32;
  const int nqt = (nblk + 7) / 8;   // 256 queries (8 blocks) per workgroup
  const long sh = (long)p.n_seq * p.heads;
  const long grid = (sh + 7) / 8 * 8 * nqt;
  hipLaunchKernelGGL((attn_frag_x3q2_kernel<OUT, P16>), dim3((unsigned)grid), dim3(256), 0, s, p, nqt, (int)sh);
}

int launch_attn_frag(const AttnFragP& p, hipStream_t s) {
  if (p.L <= 0 || p.n_seq <= 0 || p.heads <= 0 || p.inner != p.heads * 32 || p.nbp < attn_frag_blocks(p.L)) return -2;
  if ((long)p.n_seq * p.heads * ((p.L + 127) / 128) > 0x3fffffffL) return -3;
  if (p.x3 > 0) {
    if (BT_HALF_IS_BF16 || (long)p.nbp * 4096 >= 0x7fffffffL || !p.fix_mask) return -2;
    // x3 selects the LDS tile (tools/x3_probe.py, 16 chunks, +-3 % run to run):
    //   1 = 128-key tiles, 64 KB, two workgroups per CU:   main-layer shape 290 us, frontend shapes 590 us per launch;
    //   2 = 64-key tiles, 32 KB, three / four workgroups per CU (registers / LDS): main-layer shape 250 us, frontend shapes
    //       485 us -- the engine's choice.
    // (Two query blocks per wave -- QB = 2: half the fragment reads per MFMA, no score pipelining, 256 registers -- measured
    // 323 us on the main-layer shape and is not dispatched.)
    // (register cap of THREE workgroups per CU: capped for four, the kernel spilled two registers around the key loop and
    // their reloads raced the LDS-DMA in flight -- different results on every run of a 16-chunk forward; the ISA lint's
    // fourth rule now rejects any scratch access in a kernel with LDS-DMA)
    //   4 = round 4: two query blocks per wave on the hand-scheduled key loop (attn_frag_x3q2_kernel).
    // 4 takes it where it pays by the count below; otherwise the 64-key kernel's finer grain (128 queries per workgroup, three
    // per CU) fills the chip better.
    // + BT_X3_P16 (8): the P16 arithmetic (finish_x) on the same kernel selection; all kernels of one arithmetic agree bit for bit
    // Round 6: "at least two full rounds" became a count of per-CU rounds -- a CU works through its share of the grid at the
    // matrix pipe's pace, so a launch costs ceil(workgroups / 256) x (query blocks per wave), and the hand-scheduled loop does a
    // query block in 0.88 of the other kernel's time (tools/x3_probe.py B attn, B = 2 .. 14: the rule picks the faster kernel or one
    // within 4 % of it; the old rule lost 14 - 16 % at 4 - 5 chunks and 7 % on the 2-chunk main layers)
    const long pairs = (long)p.n_seq * p.heads, nb = (p.L + 31) / 32;
    const long rounds_q2 = (pairs * ((nb + 7) / 8) + 255) / 256, rounds_q1 = (pairs * ((nb + 3) / 4) + 255) / 256;
    const bool q2_pays = rounds_q2 * 2 * 88 <= rounds_q1 * 100;
    const int kern = p.x3 & 7;
    const int form = (kern == 4 && q2_pays) || kern == 5 ? 0 : (kern == 2 || kern == 4) ? 1 : 2;   // (5 = forced: tests, probes)
    switch (form * 4 + (p.out_f32 == 1 ? 2 : 0) + ((p.x3 & 8) ? 1 : 0)) {   // (out_f32 = 2: hl8 rows, a run-time branch of the hl32 epilogue)
      case 0: launch_x3q2<0, false>(p, s); break;
      case 1: launch_x3q2<0, true>(p, s); break;
      case 2: launch_x3q2<1, false>(p, s); break;
      case 3: launch_x3q2<1, true>(p, s); break;
      case 4: launch_x3<1, 0, 2, 3, false>(p, s); break;
      case 5: launch_x3<1, 0, 2, 3, true>(p, s); break;
      case 6: launch_x3<1, 1, 2, 3, false>(p, s); break;
      case 7: launch_x3<1, 1, 2, 3, true>(p, s); break;
      case 8: launch_x3<1, 0, 4, 2, false>(p, s); break;
      case 9: launch_x3<1, 0, 4, 2, true>(p, s); break;
      case 10: launch_x3<1, 1, 4, 2, false>(p, s); break;
      default: launch_x3<1, 1, 4, 2, true>(p, s); break;
    }
    {  // the queries the launch left to the fix-up (its overflow map): one workgroup per (sequence, head) pair, most return at once
      const unsigned sh = (unsigned)((long)p.n_seq * p.heads);
      switch ((p.out_f32 == 1 ? 2 : 0) + ((p.x3 & 8) ? 1 : 0)) {
        case 0: hipLaunchKernelGGL((attn_fix_x3_kernel<0, false>), dim3(sh < FIX_GRID ? sh : FIX_GRID), dim3(64 * FIX_NW), 0, s, p, (int)sh); break;
        case 1: hipLaunchKernelGGL((attn_fix_x3_kernel<0, true>), dim3(sh < FIX_GRID ? sh : FIX_GRID), dim3(64 * FIX_NW), 0, s, p, (int)sh); break;
        case 2: hipLaunchKernelGGL((attn_fix_x3_kernel<1, false>), dim3(sh < FIX_GRID ? sh : FIX_GRID), dim3(64 * FIX_NW), 0, s, p, (int)sh); break;
        default: hipLaunchKernelGGL((attn_fix_x3_kernel<1, true>), dim3(sh < FIX_GRID ? sh : FIX_GRID), dim3(64 * FIX_NW), 0, s, p, (int)sh); break;
      }
    }
    return (int)hipGetLastError();
  }
  // (Two query blocks per wave -- QB = 2, half the fragment reads per MFMA at half the occupancy -- measured equal.)
#ifdef BT_DEV
  // development builds only: BT_ATTN_ABL=128 dumps per-wave phase timings over the gates buffer (tools/attn_probe.py)
  static const int abl = getenv("BT_ATTN_ABL") ? atoi(getenv("BT_ATTN_ABL")) : 0;
  if (abl == 128) { launch_v<128, 1>(p, s); return (int)hipGetLastError(); }
  static const int qb = getenv("BT_ATTN_QB") ? atoi(getenv("BT_ATTN_QB")) : 1;   // two query blocks per wave (A/B only)
  if (qb == 2) { launch_v<0, 2>(p, s); return (int)hipGetLastError(); }
#endif
#if !BT_HALF_IS_BF16
  {
    // Round 6: two query blocks per wave on the hand-scheduled loop (attn_frag_hq2_kernel), bit-identical to the one-block kernel
    // (tested) -- and NOT dispatched: same box, three alternations of bench.py --prec half (profiles/r06_ab_hq2.txt), attention
    // 4.97 against 4.84 ms per 66-chunk step (+2.7 %; the first form of the loop, with the score pair back to back: +5 %) at -0.9 %
    // joules.  The fp16 attention is bound by its exponentials (16 v_exp_f32 + 8 conversions per MFMA quartet, d = 32: the kernel
    // header's "VALU-bound"), which four waves per SIMD of the compiler-scheduled kernel keep busier than two waves of a loop that
    // was scheduled for the matrix pipe.  Kept selectable (x3 = -2) for tests and probes; -DBT_HQ2_MIN_WG=1024 builds the
    // size-selected dispatch.  x3 = -1 forces the one-block kernel; 0 = the dispatch rule.
#ifndef BT_HQ2_MIN_WG
#define BT_HQ2_MIN_WG 2000000000
#endif
    const long wg2 = (long)p.n_seq * p.heads * (((p.L + 31) / 32 + 7) / 8);
    if (p.x3 == -2 || (p.x3 == 0 && wg2 >= BT_HQ2_MIN_WG)) {
      const long sh = (long)p.n_seq * p.heads;
      const int nqt = ((p.L + 31) / 32 + 7) / 8;
      const long grid = (sh + 7) / 8 * 8 * nqt;
      hipLaunchKernelGGL(attn_frag_hq2_kernel, dim3((unsigned)grid), dim3(256), 0, s, p, nqt, (int)sh);
      return (int)hipGetLastError();
    }
  }
#endif
  launch_v<0, 1>(p, s);
  return (int)hipGetLastError();
}
