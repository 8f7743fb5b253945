// Attention kernels (head_dim = 32, non-causal, no mask) replacing
// F.scaled_dot_product_attention + the per-head sigmoid gate of the reference
// (beat_this/model/roformer.py:67-80,125-131).
//
// q arrives pre-scaled: the QKV GEMM folds 1/sqrt(32) * log2(e) into the q rows of the
// weight, RoPE was applied in that GEMM's epilogue, so softmax here is exp2(s - max).
//
// attn_flash: one wave owns 32 queries, a 256-thread workgroup 128; keys stream through
// LDS in tiles of 64.  Scores are computed TRANSPOSED (S^T = K . Q^T) so that a lane holds
// 16 of the 32 key-scores of ONE query: row max / row sum are lane-local plus a single
// exchange with lane^32, and the probabilities are already in the B-operand layout of the
// second MFMA (O^T = V^T . P^T) -- no cross-lane traffic for P at all.  For half the V tile
// is transposed on its way into LDS so the A operand (V^T) is read with two ds_read_b64;
// for fp32 (k = 2 MFMA) V is read row-major.
// (The frequency-direction attention of the frontend lives in fused2.hip: attnff_fused_kernel.)
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT = 64;          // keys per tile
constexpr int VT_PITCH = 136;   // bytes per d-row of the transposed half V tile (64 keys * 2 + 8)

template <typename T> struct KV8;  // 8 contiguous elements staged in registers
template <> struct KV8<float> { f32x4 v[2]; };
template <> struct KV8<hf> { hfx8 v; };

template <typename T> DEVI KV8<T> ldg8(const T* p, bool ok);
template <> DEVI KV8<float> ldg8<float>(const float* p, bool ok) {
  KV8<float> r;
  if (ok) {
    r.v[0] = reinterpret_cast<const f32x4*>(p)[0];
    r.v[1] = reinterpret_cast<const f32x4*>(p)[1];
  } else {
    r.v[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.v[1] = r.v[0];
  }
  return r;
}
template <> DEVI KV8<hf> ldg8<hf>(const hf* p, bool ok) {
  KV8<hf> r;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  u32x4 z = {0, 0, 0, 0};
  r.v = ok ? *reinterpret_cast<const hfx8*>(p) : __builtin_bit_cast(hfx8, z);
  return r;
}
DEVI void sts8(char* dst, const KV8<float>& r) {
  reinterpret_cast<f32x4*>(dst)[0] = r.v[0];
  reinterpret_cast<f32x4*>(dst)[1] = r.v[1];
}
DEVI void sts8(char* dst, const KV8<hf>& r) { *reinterpret_cast<hfx8*>(dst) = r.v; }

template <typename T> struct VSize;
template <> struct VSize<float> { static constexpr int BYTES = KT * Tile<float>::PITCH; };
template <> struct VSize<hf> { static constexpr int BYTES = 32 * VT_PITCH; };

template <typename T>
__global__ __launch_bounds__(256) void attn_flash_kernel(const AttnP p) {
  constexpr int PITCH = Tile<T>::PITCH;
  __shared__ __attribute__((aligned(16))) char smem[KT * PITCH + VSize<T>::BYTES];
  char* Ks = smem;
  char* Vs = smem + KT * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const int seq = blockIdx.y / p.heads, head = blockIdx.y % p.heads;
  const int L = p.L;
  const long row0 = (long)seq * L;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const int qcol = head * 32, kcol = p.inner + head * 32, vcol = 2 * p.inner + head * 32;

  const int qi = blockIdx.x * 128 + wave * 32 + lr;
  const bool q_ok = qi < L;
  // Q^T as B operand: lane (q = lr, half g) holds d in [16 g, 16 g + 16)
  Frag<T> fq;
  {
    const T* qp = qkv + (row0 + (q_ok ? qi : L - 1)) * p.ld + qcol + 16 * g;
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fq.v[i] = reinterpret_cast<const f32x4*>(qp)[i];
    } else {
      fq.v[0] = reinterpret_cast<const hfx8*>(qp)[0];
      fq.v[1] = reinterpret_cast<const hfx8*>(qp)[1];
    }
  }

  // staging: thread -> (key = tid / 4, 8 d-values at 8 * (tid % 4))
  const int skey = tid >> 2, spart = tid & 3;
  auto loadKV = [&](int kt, KV8<T>& rk, KV8<T>& rv) {
    int key = kt * KT + skey;
    bool ok = key < L;
    const T* base = qkv + (row0 + (ok ? key : 0)) * p.ld + spart * 8;
    rk = ldg8<T>(base + kcol, ok);
    rv = ldg8<T>(base + vcol, ok);
  };

  f32x16 acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = (L + KT - 1) / KT;
  KV8<T> rk, rv;
  loadKV(0, rk, rv);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    sts8(Ks + skey * PITCH + spart * 8 * (int)sizeof(T), rk);
    if constexpr (std::is_same<T, float>::value) {
      sts8(Vs + skey * PITCH + spart * 8 * (int)sizeof(T), rv);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<hf*>(Vs + (spart * 8 + i) * VT_PITCH + skey * 2) = rv.v[i];
    }
    __syncthreads();
    if (kt + 1 < ntiles) loadKV(kt + 1, rk, rv);

    // ---- S^T = K . Q^T : two 32-key blocks ------------------------------------------
    f32x16 sc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[c][r] = 0.f;
      Frag<T> fk = ld_frag<T>(Ks + (c * 32 + lr) * PITCH, g);
      mma32(sc[c], fk, fq);
    }
    if (kt == ntiles - 1) {  // mask keys beyond L (last tile only)
      const int kbase = kt * KT;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + c * 32 + crow(r, g) >= L) sc[c][r] = -1e30f;
    }
    // ---- online softmax (per query = per lane pair (l, l^32)) ------------------------
    float mloc = sc[0][0];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sc[c][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __builtin_amdgcn_exp2f(sc[c][r] - m_new);
        sc[c][r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;

    // ---- O^T += V^T . P^T -----------------------------------------------------------
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float vv = *reinterpret_cast<const float*>(Vs + (c * 32 + crow(r, g)) * PITCH + lr * 4);
          acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sc[c][r], acc_o, 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const char* vp = Vs + lr * VT_PITCH + (c * 32 + s * 16 + 4 * g) * 2;
          hfx4 v0 = *reinterpret_cast<const hfx4*>(vp);
          hfx4 v1 = *reinterpret_cast<const hfx4*>(vp + 16);
          hfx8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          hfx8 pb;
#pragma unroll
          for (int j = 0; j < 8; ++j) pb[j] = (hf)sc[c][s * 8 + j];
          acc_o = MFMA32_H(va, pb, acc_o);
        }
    }
  }

  // ---- epilogue: normalise, gate, store ------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (q_ok) {
    const float gate = p.gates[(row0 + qi) * p.heads + head];
    const float scale = gate / l_tot;
    const long orow = (long)(seq / p.o_div) * p.o_outer + (long)(seq % p.o_div) * p.o_inner + (long)qi * p.o_tok;
    T* op = reinterpret_cast<T*>(p.out) + orow * p.inner + head * 32 + 4 * g;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if constexpr (std::is_same<T, float>::value) {
        *reinterpret_cast<f32x4*>(op + 8 * a) = f32x4{acc_o[4 * a] * scale, acc_o[4 * a + 1] * scale,
                                                       acc_o[4 * a + 2] * scale, acc_o[4 * a + 3] * scale};
      } else {
        hfx4 o = {(hf)(acc_o[4 * a] * scale), (hf)(acc_o[4 * a + 1] * scale),
                    (hf)(acc_o[4 * a + 2] * scale), (hf)(acc_o[4 * a + 3] * scale)};
        *reinterpret_cast<hfx4*>(op + 8 * a) = o;
      }
    }
  }
}

// ---- BT_PREC_F32X3: fp32 q|k|v in, fp32 out, both products on three half MFMAs (operands split into hi + lo halves) --
// Same structure as attn_flash_kernel (S^T = K . Q^T, online softmax in fp32, O^T = V^T . P^T); K and V^T sit in LDS as a
// hi and a lo half tile (split while they are staged), Q is split once, P is split in registers after the exponentials.
DEVI void split8(const KV8<float>& r, hfx8& hi, hfx8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = r.v[j >> 2][j & 3];
    const hf h = (hf)v;
    hi[j] = h;
    lo[j] = (hf)(v - (float)h);
  }
}

__global__ __launch_bounds__(256) void attn_flash_x3_kernel(const AttnP p) {
  constexpr int PITCH = Tile<hf>::PITCH;
  constexpr int KB = KT * PITCH, VB = 32 * VT_PITCH;
  __shared__ __attribute__((aligned(16))) char smem[2 * KB + 2 * VB];
  char* Ks = smem;             // hi, lo at + KB
  char* Vs = smem + 2 * KB;    // hi (transposed), lo at + VB

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, lr = lane & 31;
  const int seq = blockIdx.y / p.heads, head = blockIdx.y % p.heads;
  const int L = p.L;
  const long row0 = (long)seq * L;
  const float* qkv = reinterpret_cast<const float*>(p.qkv);
  const int qcol = head * 32, kcol = p.inner + head * 32, vcol = 2 * p.inner + head * 32;

  const int qi = blockIdx.x * 128 + wave * 32 + lr;
  const bool q_ok = qi < L;
  Frag<hf> fq, fq2;  // Q^T as B operand, hi and lo: lane (q = lr, half g) holds d in [16 g, 16 g + 16)
  {
    const float* qp = qkv + (row0 + (q_ok ? qi : L - 1)) * p.ld + qcol + 16 * g;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      KV8<float> r;
      r.v[0] = reinterpret_cast<const f32x4*>(qp)[2 * h];
      r.v[1] = reinterpret_cast<const f32x4*>(qp)[2 * h + 1];
      split8(r, fq.v[h], fq2.v[h]);
    }
  }
  const int skey = tid >> 2, spart = tid & 3;
  auto loadKV = [&](int kt, KV8<float>& rk, KV8<float>& rv) {
    int key = kt * KT + skey;
    bool ok = key < L;
    const float* base = qkv + (row0 + (ok ? key : 0)) * p.ld + spart * 8;
    rk = ldg8<float>(base + kcol, ok);
    rv = ldg8<float>(base + vcol, ok);
  };

  f32x16 acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const int ntiles = (L + KT - 1) / KT;
  KV8<float> rk, rv;
  loadKV(0, rk, rv);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    {
      hfx8 hi, lo;
      split8(rk, hi, lo);
      *reinterpret_cast<hfx8*>(Ks + skey * PITCH + spart * 16) = hi;
      *reinterpret_cast<hfx8*>(Ks + KB + skey * PITCH + spart * 16) = lo;
      split8(rv, hi, lo);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        *reinterpret_cast<hf*>(Vs + (spart * 8 + i) * VT_PITCH + skey * 2) = hi[i];
        *reinterpret_cast<hf*>(Vs + VB + (spart * 8 + i) * VT_PITCH + skey * 2) = lo[i];
      }
    }
    __syncthreads();
    if (kt + 1 < ntiles) loadKV(kt + 1, rk, rv);

    f32x16 sc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[c][r] = 0.f;
      const Frag<hf> fk = ld_frag<hf>(Ks + (c * 32 + lr) * PITCH, g), fk2 = ld_frag<hf>(Ks + KB + (c * 32 + lr) * PITCH, g);
      mma32(sc[c], fk2, fq);
      mma32(sc[c], fk, fq2);
      mma32(sc[c], fk, fq);
    }
    if (kt == ntiles - 1) {
      const int kbase = kt * KT;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + c * 32 + crow(r, g) >= L) sc[c][r] = -1e30f;
    }
    float mloc = sc[0][0];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sc[c][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __builtin_amdgcn_exp2f(sc[c][r] - m_new);
        sc[c][r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;

#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        hfx8 va[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const char* vp = Vs + w * VB + lr * VT_PITCH + (c * 32 + s * 16 + 4 * g) * 2;
          const hfx4 v0 = *reinterpret_cast<const hfx4*>(vp), v1 = *reinterpret_cast<const hfx4*>(vp + 16);
          va[w] = hfx8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        }
        hfx8 pb, pb2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = sc[c][s * 8 + j];
          const hf h = (hf)e;
          pb[j] = h;
          pb2[j] = (hf)(e - (float)h);
        }
        acc_o = MFMA32_H(va[1], pb, acc_o);
        acc_o = MFMA32_H(va[0], pb2, acc_o);
        acc_o = MFMA32_H(va[0], pb, acc_o);
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (q_ok) {
    const float gate = p.gates[(row0 + qi) * p.heads + head];
    const float scale = gate / l_tot;
    const long orow = (long)(seq / p.o_div) * p.o_outer + (long)(seq % p.o_div) * p.o_inner + (long)qi * p.o_tok;
    float* op = reinterpret_cast<float*>(p.out) + orow * p.inner + head * 32 + 4 * g;
#pragma unroll
    for (int a = 0; a < 4; ++a)
      *reinterpret_cast<f32x4*>(op + 8 * a) = f32x4{acc_o[4 * a] * scale, acc_o[4 * a + 1] * scale, acc_o[4 * a + 2] * scale,
                                                     acc_o[4 * a + 3] * scale};
  }
}

// ---------------------------------------------------------------------------------------
}  // namespace

int launch_attn_flash(const AttnP& p, int prec, hipStream_t s) {
  if (p.L <= 0 || p.n_seq <= 0 || p.inner != p.heads * 32) return -2;
  dim3 grid((p.L + 127) / 128, (unsigned)((long)p.n_seq * p.heads)), block(256);
  if (grid.y > 65535) return -3;
  if (prec == BT_PREC_F32X3 && !BT_HALF_IS_BF16)
    hipLaunchKernelGGL(attn_flash_x3_kernel, grid, block, 0, s, p);
  else if (prec == BT_PREC_F32)
    hipLaunchKernelGGL((attn_flash_kernel<float>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((attn_flash_kernel<hf>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}
