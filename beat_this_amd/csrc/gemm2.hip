// half GEMM for the wide layers (N >= 128, K % 64 == 0): main-transformer QKV / out / FF,
// frontend.linear and the frontend convolutions in BT_PREC_HALF.  Same contract and epilogues as
// gemm.hip (GemmP), different engine:
//   * 128 x 128 x 64 tiles, 4 waves as 2 x 2 (64 x 64 each = 2 x 2 MFMA 32x32 tiles, 16 MFMAs per
//     k-step between barriers);
//   * two LDS buffers, ONE barrier per k-step: global loads of step k+1 are issued before the
//     MFMAs of step k and written to the other buffer after them;
//   * XCD-aware tile order: the 1-D grid is remapped so that the (N/128) tiles sharing one
//     128-row A panel run on the same XCD (block b -> XCD b % 8), i.e. the panel is fetched into
//     one L2 instead of eight;
//   * half outputs leave through LDS: accumulators -> half tile in LDS -> 16-byte row-contiguous
//     stores (a 128-wide half row is 256 B = two full lines) instead of 2-byte scattered stores.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int PITCH2 = BK * 2 + 16;         // bytes per LDS tile row (half) -> conflict-free b128 reads
constexpr int OPITCH = BN * 2 + 16;         // bytes per staged output row
constexpr int BUF_BYTES = (BM + BN) * PITCH2;

// Staging is chunk-coalesced: consecutive lanes fetch consecutive 16-byte chunks of a row, so one
// wave-instruction covers whole 128-byte lines (fp32 A: 16 chunks = 256 B per row and k-step,
// 4 rows per instruction; half: 8 chunks = 128 B per row, 8 rows per instruction).  A
// row-per-lane mapping re-requests every line 8 times and is TA-bound at a quarter of the rate.
struct AF { f32x4 v[8]; };    // fp32 A: rows r0 + 16 p, p = 0..7, chunk c = tid % 16 (4 floats)
struct AH { hfx8 v[4]; };   // half A or W: rows r0 + 32 p, p = 0..3, chunk c = tid % 8 (8 half)

template <bool A_F32, int EPI>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const GemmP p, int n_tiles, int total_tiles, int per_xcd) {
  using EA = typename std::conditional<A_F32, float, hf>::type;
  constexpr int APASS = A_F32 ? 8 : 4;        // row passes of the A staging
  constexpr int AROWS = A_F32 ? 16 : 32;      // rows covered per pass
  constexpr int ACH = A_F32 ? 16 : 8;         // 16-byte chunks per row and k-step
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES + BM * 4];
  float* rs = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);

  // ---- XCD-aware tile assignment --------------------------------------------------------------
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * per_xcd + (bid >> 3);
  if (tile >= total_tiles) return;
  const int m_tile = tile / n_tiles, n_tile = tile - m_tile * n_tiles;
  const long m0 = (long)m_tile * BM;
  const int n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = p.K / BK;

  // ---- staging addresses ---------------------------------------------------------------------------
  const EA* A = reinterpret_cast<const EA*>(p.A);
  const bool conv = (p.flags & GEMM_F_CONV) != 0;
  const int ac = tid % ACH, ar0 = tid / ACH;
  constexpr int AEL = A_F32 ? 4 : 8;          // elements per chunk
  long a_off[APASS];                          // element offset of (row, k = 0, this chunk); conv: tap 1
  int a_t[APASS];                             // conv: time index of the row (validity of taps 0 / 2)
  bool a_ok[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const long gm = m0 + ar0 + AROWS * ps;
    a_ok[ps] = gm < p.M;
    a_t[ps] = 0;
    if (!conv) {
      a_off[ps] = (a_ok[ps] ? gm : 0) * p.lda + ac * AEL;
    } else {
      const long tf = (long)p.conv_T * p.conv_F;
      const long gmc = a_ok[ps] ? gm : 0;
      const long b = gmc / tf;
      const int rem = (int)(gmc - b * tf);
      const int t = rem / p.conv_F, f = rem - t * p.conv_F;
      a_t[ps] = t;
      a_off[ps] = ((b * p.conv_T + t) * p.conv_F + f) * (long)p.conv_C2 + ac * AEL;
    }
  }
  const hf* Wb = reinterpret_cast<const hf*>(p.W) + (long)(n0 + (tid >> 3)) * p.K + (tid & 7) * 8;
  using AReg = typename std::conditional<A_F32, AF, AH>::type;
  auto loadA = [&](int kt, AReg& ra) {
    long koff = (long)kt * BK;
    int dt = 0;
    if (conv) {
      const int k0 = kt * BK;
      const int tap = k0 / p.conv_C2;
      dt = tap - 1;
      koff = (long)dt * p.conv_F * p.conv_C2 + (k0 - tap * p.conv_C2);
    }
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const int tt = a_t[ps] + dt;
      const bool ok = a_ok[ps] && (!conv || (tt >= 0 && tt < p.conv_T));
      const EA* src = A + (ok ? a_off[ps] + koff : 0);
      if constexpr (A_F32) ra.v[ps] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      else {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const u32x4 z = {0, 0, 0, 0};
        ra.v[ps] = ok ? *reinterpret_cast<const hfx8*>(src) : __builtin_bit_cast(hfx8, z);
      }
    }
  };
  auto loadB = [&](int kt, AH& rb) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) rb.v[ps] = *reinterpret_cast<const hfx8*>(Wb + (long)ps * 32 * p.K + (long)kt * BK);
  };
  const bool rms_on = (p.flags & GEMM_F_RMS) != 0;
  float ssq[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) ssq[ps] = 0.f;
  auto storeA = [&](char* buf, const AReg& ra) {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      char* dst = buf + (ar0 + AROWS * ps) * PITCH2 + ac * (AEL * 2);
      if constexpr (A_F32) {
        const f32x4 v = ra.v[ps];
        ssq[ps] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], ssq[ps]))));
        *reinterpret_cast<hfx4*>(dst) = hfx4{(hf)v[0], (hf)v[1], (hf)v[2], (hf)v[3]};
      } else {
        if (rms_on) {  // sum of squares of the half operands themselves (v_dot2c_f32_bf16)
          const hfx8 v = ra.v[ps];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const hfx2 pr = {v[2 * q], v[2 * q + 1]};
            #if BT_HALF_IS_BF16
            ssq[ps] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ssq[ps], false);
#else
            ssq[ps] = __builtin_amdgcn_fdot2(pr, pr, ssq[ps], false);
#endif
          }
        }
        *reinterpret_cast<hfx8*>(dst) = ra.v[ps];
      }
    }
  };
  auto storeB = [&](char* buf, const AH& rb) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
      *reinterpret_cast<hfx8*>(buf + BM * PITCH2 + ((tid >> 3) + 32 * ps) * PITCH2 + (tid & 7) * 16) = rb.v[ps];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  {
    AReg ra;
    AH rb;
    loadA(0, ra);
    loadB(0, rb);
    storeA(smem, ra);
    storeB(smem, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * BUF_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * BUF_BYTES;
    AReg ra;
    AH rb;
    const bool more = kt + 1 < nk;
    if (more) {
      loadA(kt + 1, ra);
      loadB(kt + 1, rb);
    }
    const char* a_src = cur + (wm * 64 + lr) * PITCH2;
    const char* b_src = cur + BM * PITCH2 + (wn * 64 + lr) * PITCH2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag<hf> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = ld_frag<hf>(a_src + i * 32 * PITCH2 + ks * 64, g);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = ld_frag<hf>(b_src + j * 32 * PITCH2 + ks * 64, g);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[i], fb[j]);
    }
    if (more) {
      storeA(nxt, ra);
      storeB(nxt, rb);
    }
    __syncthreads();
  }

  const bool rms = rms_on;
  if (rms) {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {  // the ACH lanes sharing a row are consecutive
      float t = ssq[ps];
#pragma unroll
      for (int o = ACH / 2; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (ac == 0) rs[ar0 + AROWS * ps] = sqrtf((float)p.K) / fmaxf(sqrtf(t), 1e-12f);
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  // Phase 1: raw fp32 accumulators -> LDS tile (the k-loop buffers are free after the last barrier).
  // Phase 2: row-major pass, one thread = one 16-byte output chunk: every per-row / per-column
  // constant is hoisted, global loads are issued in batches, all global I/O is 16 B and coalesced.
  // (A per-element epilogue in the MFMA register layout serialises on ~64 dependent memory round
  // trips per lane: measured 70-110 us of fixed cost per GEMM.)
  constexpr int SP = BN * 4 + 16;  // staged row pitch (bytes)
  char* stage = smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float*>(stage + (wm * 64 + i * 32 + crow(r, g)) * SP + ((wn * 2 + j) * 32 + lr) * 4) =
            acc[i][j][r];
  __syncthreads();

  if (EPI == GEMM_EPI_QKV || (EPI == GEMM_EPI_STORE && !(p.flags & GEMM_F_OUT_F32))) {
    // ---- half output: 16 chunks of 8 columns per row ----------------------------------------------
    const int ch = tid & 15;
    const int col = n0 + ch * 8;
    float bias8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bias8[q] = 0.f;
    int cat = 0;
    if (EPI == GEMM_EPI_QKV) {
      cat = col / p.inner;  // 0 q, 1 k, 2 v, 3 gates / padding
    } else if ((p.flags & GEMM_F_BIAS) && col < p.N) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + col), b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { bias8[q] = b0[q]; bias8[4 + q] = b1[q]; }
    }
    const int ncols_store = (EPI == GEMM_EPI_QKV) ? 3 * p.inner : p.N;
    const int pair0 = (col & 31) >> 1;  // first RoPE pair index of this chunk
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
      const int row_l = (tid >> 4) + 16 * it;
      const long gm = m0 + row_l;
      if (gm >= p.M) continue;
      const float sc = rms ? rs[row_l] : 1.0f;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + row_l * SP + ch * 32);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + row_l * SP + ch * 32 + 16);
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] = s0[q] * sc; v[4 + q] = s1[q] * sc; }
      long orow = gm;
      if (EPI == GEMM_EPI_QKV) {
        if (p.flags & GEMM_F_ROWMAP) orow = btf_to_bft(gm, p.map_T, p.map_F);
        if (cat < 2) {
          const int pos = (int)((gm / p.pdiv) % p.pmod);
          const f32x4* cp = reinterpret_cast<const f32x4*>(p.rope + ((long)pos * 16 + pair0) * 2);
          const f32x4 c0 = cp[0], c1 = cp[1];  // (cos, sin) of pairs pair0 .. pair0 + 3
          const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float e = v[2 * q], o = v[2 * q + 1];
            v[2 * q] = e * cs[2 * q] - o * cs[2 * q + 1];
            v[2 * q + 1] = o * cs[2 * q] + e * cs[2 * q + 1];
          }
        } else if (cat >= 3) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int hc = col + q - 3 * p.inner;
            if (hc >= 0 && hc < p.heads) p.gates[orow * p.heads + hc] = sigmoidf(v[q] + p.bias[hc]);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v[q] += bias8[q];
          if (p.flags & GEMM_F_GELU) v[q] = gelu_erf(v[q]);
        }
      }
      if (col < ncols_store) {
        hfx8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (hf)v[q];
        *reinterpret_cast<hfx8*>(reinterpret_cast<hf*>(p.out) + orow * p.ldo + col) = o;
      }
    }
  } else {
    // ---- fp32 output / residual update: 32 chunks of 4 columns per row ---------------------------------
    const int ch = tid & 31;
    const int col = n0 + ch * 4;
    const bool col_ok = col < p.N;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if ((p.flags & GEMM_F_BIAS) && col_ok) b4 = *reinterpret_cast<const f32x4*>(p.bias + col);
    hf* xb = reinterpret_cast<hf*>(p.xb);
#pragma unroll
    for (int half8 = 0; half8 < 2; ++half8) {
      f32x4 xv[8];
      if (EPI == GEMM_EPI_RESID) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {  // batch the residual loads
          const long gm = m0 + (tid >> 5) + 8 * (half8 * 8 + it);
          xv[it] = (gm < p.M && col_ok) ? *reinterpret_cast<const f32x4*>(p.x + gm * p.ldx + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row_l = (tid >> 5) + 8 * (half8 * 8 + it);
        const long gm = m0 + row_l;
        const bool ok = gm < p.M && col_ok;  // wave-uniform in gm (a wave covers 2 rows x 32 chunks)
        const float sc = rms ? rs[row_l] : 1.0f;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + row_l * SP + ch * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q], sc, b4[q]);
        float* dst;
        long ld;
        if (EPI == GEMM_EPI_RESID) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += xv[it][q];
          dst = p.x; ld = p.ldx;
        } else {
          if (p.flags & GEMM_F_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
          }
          dst = reinterpret_cast<float*>(p.out); ld = p.ldo;
        }
        if (ok) {
          *reinterpret_cast<f32x4*>(dst + gm * ld + col) = v;
          if (xb) *reinterpret_cast<hfx4*>(xb + gm * ld + col) = hfx4{(hf)v[0], (hf)v[1], (hf)v[2], (hf)v[3]};
        }
        if (p.ssq_out) {  // partial sums of squares per 64 columns (16 consecutive lanes), for the consumer's RMSNorm
          float ss = ok ? fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3]))) : 0.f;
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
          if ((ch & 15) == 0 && gm < p.M && col < p.N) p.ssq_out[(long)(col >> 6) * p.M + gm] = ss;
        }
      }
    }
  }
}

template <bool A_F32>
int launch2(const GemmP& p, hipStream_t s) {
  const int n_tiles = (p.N + BN - 1) / BN;
  const long m_tiles = ((long)p.M + BM - 1) / BM;
  const long total = m_tiles * n_tiles;
  if (total > 0x3fffffffL) return -3;
  // each XCD takes a contiguous run of tiles, rounded to whole A panels when possible
  long per = (total + 7) / 8;
  per = (per + n_tiles - 1) / n_tiles * n_tiles;
  dim3 grid((unsigned)(per * 8)), block(256);
  switch (p.epi) {
    case GEMM_EPI_STORE:
      hipLaunchKernelGGL((gemm2_kernel<A_F32, GEMM_EPI_STORE>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      break;
    case GEMM_EPI_RESID:
      hipLaunchKernelGGL((gemm2_kernel<A_F32, GEMM_EPI_RESID>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      break;
    case GEMM_EPI_QKV:
      hipLaunchKernelGGL((gemm2_kernel<A_F32, GEMM_EPI_QKV>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace

bool gemm2_supported(const GemmP& p, int prec) {
  if (prec != BT_PREC_HALF || p.N < 128 || p.K % 64 != 0 || p.M <= 0) return false;
  if ((p.flags & GEMM_F_CONV) && p.conv_C2 % 64 != 0) return false;
  if (p.epi == GEMM_EPI_QKV && ((3 * p.inner) % 8 != 0 || p.ldo % 8 != 0)) return false;
  if (p.epi == GEMM_EPI_STORE && !(p.flags & GEMM_F_OUT_F32) && (p.N % 8 != 0 || p.ldo % 8 != 0)) return false;
  if ((p.epi == GEMM_EPI_RESID || (p.flags & GEMM_F_OUT_F32)) && p.N % 4 != 0) return false;
  if (p.epi == GEMM_EPI_RESID && p.ldx % 4 != 0) return false;
  return true;
}

int launch_gemm2(const GemmP& p, hipStream_t s) {
  return (p.flags & (GEMM_F_A_F32)) ? launch2<true>(p, s) : launch2<false>(p, s);
}
