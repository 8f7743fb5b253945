// Tiled MFMA GEMM with fused prologue/epilogues for every linear layer of BeatThis:
//   C[M,N] = epi( A[M,K] . W[N,K]^T )
// replaces nn.Linear / nn.Conv2d(+BatchNorm+GELU) calls of the reference
// (beat_this/model/roformer.py:51-58,114-131; beat_tracker.py:77,155-166).
//
// Tile: 128 rows x BN cols x 32 k per step, 256 threads = 4 waves, 32x32 MFMA tiles.
// Fusions (all in this one kernel, selected by GemmP):
//   * RMSNorm prologue: gamma is folded into W on the host; the per-row factor
//     sqrt(K)/max(||x||,1e-12) (roformer.py:22-32) is accumulated from the A tiles while
//     they are staged and applied in the epilogue.
//   * QKV epilogue: RoPE on the q and k column blocks (interleaved pairs, table lookup),
//     sigmoid(+bias) on the appended gate columns (roformer.py:117-129), optional
//     (b,t,f)->(b,f,t) row permutation of the store for the time-direction attention.
//   * bias / exact-erf GELU / residual add / fp32 or half output.
//   * implicit-GEMM A gather for the (2,3)/(2,1) frontend convolutions in (b,t,f,c) layout.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

template <typename E> struct Stg;
template <> struct Stg<float> { f32x4 v[4]; };
template <> struct Stg<hf> { hfx8 v[2]; };

template <typename E> DEVI Stg<E> ldg16(const E* p, bool ok);
template <> DEVI Stg<float> ldg16<float>(const float* p, bool ok) {
  Stg<float> s;
  if (ok) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = q[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  return s;
}
template <> DEVI Stg<hf> ldg16<hf>(const hf* p, bool ok) {
  Stg<hf> s;
  if (ok) {
    const hfx8* q = reinterpret_cast<const hfx8*>(p);
    s.v[0] = q[0];
    s.v[1] = q[1];
  } else {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 z = {0, 0, 0, 0};
    s.v[0] = __builtin_bit_cast(hfx8, z);
    s.v[1] = s.v[0];
  }
  return s;
}

DEVI float sumsq(const Stg<float>& s) {
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) a = fmaf(s.v[i][j], s.v[i][j], a);
  return a;
}
DEVI float sumsq(const Stg<hf>&) { return 0.f; }

// write 16 staged elements to an LDS tile row as compute dtype T
DEVI void sts16(char* dst, const Stg<float>& s, float) {
  f32x4* d = reinterpret_cast<f32x4*>(dst);
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = s.v[i];
}
DEVI void sts16(char* dst, const Stg<hf>& s, hf) {
  hfx8* d = reinterpret_cast<hfx8*>(dst);
  d[0] = s.v[0];
  d[1] = s.v[1];
}
DEVI void sts16(char* dst, const Stg<float>& s, hf) {
  hfx8* d = reinterpret_cast<hfx8*>(dst);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    hfx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (hf)s.v[2 * h + (j >> 2)][j & 3];
    d[h] = o;
  }
}

// fp32 -> hi + lo halves (BT_PREC_F32X3): hi = half(a), lo = half(a - hi); 16 staged elements to the two LDS tiles
DEVI void sts16_split(char* hi, char* lo, const Stg<float>& s) {
  hfx8* dh = reinterpret_cast<hfx8*>(hi);
  hfx8* dl = reinterpret_cast<hfx8*>(lo);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    hfx8 oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = s.v[2 * h + (j >> 2)][j & 3];
      const hf vh = (hf)v;
      oh[j] = vh;
      ol[j] = (hf)(v - (float)vh);
    }
    dh[h] = oh;
    dl[h] = ol;
  }
}

// SPLIT (BT_PREC_F32X3; T = half, A fp32): a second pair of LDS tiles holds the lo parts of A (split while it is staged) and
// of W (packed behind the hi part: p.W + rows_padded * K); three half MFMAs per product, small terms first.
template <typename T, bool A_F32, int BN, int EPI, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  using EA = typename std::conditional<A_F32, float, T>::type;
  using TO = typename std::conditional<SPLIT, float, T>::type;   // element type of q|k|v / plain outputs
  constexpr int WM = (BN == 128) ? 2 : 4, WN = 4 / WM;
  constexpr int TM = 128 / WM / 32, TN = BN / WN / 32;
  constexpr int PITCH = Tile<T>::PITCH;
  constexpr int TILES = (128 + BN) * PITCH;          // one A tile + one W tile
  constexpr int LO = SPLIT ? TILES : 0;              // offset of the lo tiles
  __shared__ __attribute__((aligned(16))) char smem[TILES + LO + 128 * 4 + (EPI == GEMM_EPI_QKV ? 128 * 12 : 0)];
  char* As = smem;
  char* Bs = smem + 128 * PITCH;
  float* rs = reinterpret_cast<float*>(smem + TILES + LO);
  // QKV epilogue: output row and rotary position of the tile's 128 rows, worked out ONCE per row (thread = row) -- they
  // take 64-bit divisions by run-time values, which the epilogue used to repeat for each of a lane's 32 (row, tile) pairs
  // (2 to 6 divisions each: more instructions than the whole k-loop)
  long* orow_t = reinterpret_cast<long*>(smem + TILES + LO + 128 * 4);
  int* pos_t = reinterpret_cast<int*>(smem + TILES + LO + 128 * 12);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const long m0 = (long)blockIdx.y * 128;
  const int n0 = blockIdx.x * BN;
  const int nk = p.K / 32;

  // ---- staging addresses -----------------------------------------------------------
  const int srow = tid >> 1, half = tid & 1;
  const long gm_s = m0 + srow;
  const bool a_valid = gm_s < p.M;
  const EA* A = reinterpret_cast<const EA*>(p.A);
  const bool conv = (p.flags & GEMM_F_CONV) != 0;
  long a_base = 0;  // element offset of k = 0 for this thread's row (plain mode)
  int cv_t = 0;
  long cv_bt = 0;   // (b*T) and f' pieces for the conv gather
  int cv_f = 0;
  if (!conv) {
    a_base = gm_s * p.lda + half * 16;
  } else if (a_valid) {
    long tf = (long)p.conv_T * p.conv_F;
    long b = gm_s / tf;
    int rem = (int)(gm_s - b * tf);
    cv_t = rem / p.conv_F;
    cv_f = rem - cv_t * p.conv_F;
    cv_bt = b * p.conv_T;
  }
  auto loadA = [&](int kt) -> Stg<EA> {
    if (!conv) return ldg16<EA>(A + a_base + (long)kt * 32, a_valid);
    int k0 = kt * 32;
    int tap = k0 / p.conv_C2;
    int j0 = k0 - tap * p.conv_C2 + half * 16;
    int tt = cv_t + tap - 1;
    bool ok = a_valid && tt >= 0 && tt < p.conv_T;
    long off = ((cv_bt + tt) * p.conv_F + cv_f) * (long)p.conv_C2 + j0;
    return ldg16<EA>(A + (ok ? off : 0), ok);
  };
  const bool b_thread = srow < BN;
  const T* Wp = reinterpret_cast<const T*>(p.W) + (long)(n0 + (b_thread ? srow : 0)) * p.K + half * 16;
  auto loadB = [&](int kt) -> Stg<T> { return ldg16<T>(Wp + (long)kt * 32, b_thread); };
  const long w_lo = (long)((p.N + 127) / 128 * 128) * p.K;  // (SPLIT) the lo part follows the hi part of the padded matrix
  auto loadB2 = [&](int kt) -> Stg<T> { return ldg16<T>(Wp + w_lo + (long)kt * 32, b_thread && SPLIT); };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ss = 0.f;
  Stg<EA> ra = loadA(0);
  Stg<T> rb = loadB(0), rb2;
  if (SPLIT) rb2 = loadB2(0);
  char* a_dst = As + srow * PITCH + half * 16 * (int)sizeof(T);
  char* b_dst = Bs + srow * PITCH + half * 16 * (int)sizeof(T);
  const char* a_src = As + (wm * TM * 32 + lr) * PITCH;
  const char* b_src = Bs + (wn * TN * 32 + lr) * PITCH;

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    ss += sumsq(ra);
    if constexpr (SPLIT) {
      sts16_split(a_dst, a_dst + LO, ra);
      if (b_thread) { sts16(b_dst, rb, T()); sts16(b_dst + LO, rb2, T()); }
    } else {
      sts16(a_dst, ra, T());
      if (b_thread) sts16(b_dst, rb, T());
    }
    __syncthreads();
    if (kt + 1 < nk) {
      ra = loadA(kt + 1);
      rb = loadB(kt + 1);
      if (SPLIT) rb2 = loadB2(kt + 1);
    }
    Frag<T> fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = ld_frag<T>(a_src + i * 32 * PITCH, g);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = ld_frag<T>(b_src + j * 32 * PITCH, g);
    if constexpr (SPLIT) {
      Frag<T> fa2[TM], fb2[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa2[i] = ld_frag<T>(a_src + LO + i * 32 * PITCH, g);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb2[j] = ld_frag<T>(b_src + LO + j * 32 * PITCH, g);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          mma32(acc[i][j], fa2[i], fb[j]);
          mma32(acc[i][j], fa[i], fb2[j]);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma32(acc[i][j], fa[i], fb[j]);
  }

  const bool rms = (p.flags & GEMM_F_RMS) != 0;
  if (rms) {
    float tot = ss + __shfl_xor(ss, 1);
    if (half == 0) rs[srow] = sqrtf((float)p.K) / fmaxf(sqrtf(tot), 1e-12f);
  }
  if (EPI == GEMM_EPI_QKV && tid < 128) {
    const long gm = m0 + tid;
    pos_t[tid] = (int)((gm / p.pdiv) % p.pmod);
    orow_t[tid] = (p.flags & GEMM_F_ROWMAP) ? btf_to_bft(gm < p.M ? gm : 0, p.map_T, p.map_F) : gm;
  }
  if (rms || EPI == GEMM_EPI_QKV) __syncthreads();

  // ---- epilogue --------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_l = wm * TM * 32 + i * 32 + crow(r, g);
      const long gm = m0 + row_l;
      const bool row_ok = gm < p.M;
      const float sc = rms ? rs[row_l] : 1.0f;
      if (EPI == GEMM_EPI_QKV) {
        const int pos = pos_t[row_l];
        const long orow = orow_t[row_l];
        const f32x2 cs = *reinterpret_cast<const f32x2*>(p.rope + ((long)pos * 16 + (lr >> 1)) * 2);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col0 = n0 + (wn * TN + j) * 32;
          const int col = col0 + lr;
          const int cat = col0 / p.inner;  // wave-uniform: 0 q, 1 k, 2 v, 3 gates / padding
          float v = acc[i][j][r] * sc;
          if (cat < 2) {
            float other = __shfl_xor(v, 1);
            v = (lr & 1) ? fmaf(other, cs.y, v * cs.x) : fmaf(-other, cs.y, v * cs.x);
          }
          if (cat < 3) {
            if (row_ok) reinterpret_cast<TO*>(p.out)[orow * p.ldo + col] = from_f32<TO>(v);
          } else {
            const int hc = col - 3 * p.inner;
            if (row_ok && hc < p.heads) p.gates[orow * p.heads + hc] = sigmoidf(v + p.bias[hc]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + (wn * TN + j) * 32 + lr;
          if (!row_ok || col >= p.N) continue;
          float v = acc[i][j][r] * sc;
          if (p.flags & GEMM_F_BIAS) v += p.bias[col];
          if (EPI == GEMM_EPI_RESID) {
            float* xp = p.x + gm * p.ldx + col;
            *xp = *xp + v;
          } else {
            if (p.flags & GEMM_F_GELU) v = gelu_erf(v);
            if (std::is_same<TO, float>::value || (p.flags & GEMM_F_OUT_F32))
              reinterpret_cast<float*>(p.out)[gm * p.ldo + col] = v;
            else
              reinterpret_cast<T*>(p.out)[gm * p.ldo + col] = from_f32<T>(v);
          }
        }
      }
    }
  }
}

template <typename T, bool A_F32, int EPI, bool SPLIT = false>
int launch_bn(const GemmP& p, hipStream_t s) {
  dim3 block(256);
  long mt = ((long)p.M + 127) / 128;
  if (p.N > 64) {
    dim3 grid((p.N + 127) / 128, (unsigned)mt);
    hipLaunchKernelGGL((gemm_kernel<T, A_F32, 128, EPI, SPLIT>), grid, block, 0, s, p);
  } else if (p.N > 32) {
    dim3 grid(1, (unsigned)mt);
    hipLaunchKernelGGL((gemm_kernel<T, A_F32, 64, EPI, SPLIT>), grid, block, 0, s, p);
  } else {
    dim3 grid(1, (unsigned)mt);
    hipLaunchKernelGGL((gemm_kernel<T, A_F32, 32, EPI, SPLIT>), grid, block, 0, s, p);
  }
  return (int)hipGetLastError();
}

template <typename T, bool A_F32, bool SPLIT = false>
int launch_epi(const GemmP& p, hipStream_t s) {
  switch (p.epi) {
    case GEMM_EPI_STORE: return launch_bn<T, A_F32, GEMM_EPI_STORE, SPLIT>(p, s);
    case GEMM_EPI_RESID: return launch_bn<T, A_F32, GEMM_EPI_RESID, SPLIT>(p, s);
    case GEMM_EPI_QKV: return launch_bn<T, A_F32, GEMM_EPI_QKV, SPLIT>(p, s);
  }
  return -1;
}

}  // namespace

bool gemm2_supported(const GemmP& p, int prec);
int launch_gemm2(const GemmP& p, hipStream_t s);

int launch_gemm(const GemmP& p, int prec, hipStream_t s) {
  if (p.K % 32 != 0 || p.M <= 0) return -2;
  if (gemm2_supported(p, prec)) return launch_gemm2(p, s);
  if ((p.flags & GEMM_F_CONV) && (p.conv_C2 % 32 != 0 || p.K != 3 * p.conv_C2)) return -2;
  if (p.epi == GEMM_EPI_QKV && (p.inner % 32 != 0)) return -2;
  if (prec == BT_PREC_F32) return launch_epi<float, true>(p, s);
  if (prec == BT_PREC_F32X3) {  // fp32 activations and outputs, hi + lo half operands (p.W = [hi | lo])
    if (BT_HALF_IS_BF16) return -2;
    return launch_epi<hf, true, true>(p, s);
  }
  if (p.flags & GEMM_F_A_F32) return launch_epi<hf, true>(p, s);
  return launch_epi<hf, false>(p, s);
}
