// Register-chained fused kernels for the frontend's small-channel blocks (C = 32 / 64 / 128).
//
// Idea: with the 32x32 MFMA, a product computed as  D^T = W . X^T  leaves lane (token = lane&31,
// half g) holding output features crow(r,g) of ITS token in registers r = 0..15.  That is exactly
// the B-operand form of the next MFMA (n = token, k-slots per lane-half) provided the next
// weight's k columns are stored in the order  new[16 g + r] = old[crow(r, g)]  inside every block
// of 32 (PERM32, done once by pack.py).  So chains of per-token linear maps run entirely in
// registers: no LDS, no barriers, no HBM round trip for intermediates; every per-token quantity
// (RMSNorm factor, RoPE angle, gate, softmax statistics) is lane-local.  One wave = 32 tokens.
//
//   ff_fused_kernel        x += W2 . gelu(W1 . rmsnorm(x) + b1) + b2          (roformer.py:38-61)
// Reached by bt_forward for main layers with transformer_dim <= 128 (small0) outside the gemm3 path (fp32).  The frontend
// halves run on the second generation of this idea, fused2.hip (whole halves per launch, LDS-DMA weight ring).
//
// Weights stream from L2/L1 straight into A-operand registers (<= 256 KB per block, shared by all
// waves of the launch); the residual stream x is read once (64 B per lane per k-tile) and
// written once (16 B per lane), so HBM traffic is the algorithmic minimum: 8 B per element of x.
#include <type_traits>

#include "common.h"
#include "chain.h"
#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Weights arrive FRAGMENT-MAJOR (pack.py: ff_fragment_major): for hidden block hb the 2*KT
// operand tiles [W1 rows hb*32.. x k-tile kt | W2p rows mt*32.. x cols hb*32..] are stored as
// [tile][half h][lane][8 elements], i.e. one contiguous block of 2*KT*32*32 elements per hb whose
// LDS image is exactly what the lanes read (lane l, half h: 16 B at tile + h*1024 B + l*16 B for
// half).  One workgroup (4 waves = 128 tokens) copies that block global -> LDS once per hb
// (coalesced 16-byte chunks, double buffered, one barrier per hb) and all four waves read their
// A fragments from LDS conflict-free.  L2 -> CU weight traffic drops 4x versus per-wave streaming.
template <typename T, int C>
__global__ __launch_bounds__(256) void ff_fused_kernel(const FusedFFP p) {
  constexpr int KT = C / 32;       // k-tiles of the first GEMM = m-tiles of the second
  constexpr int HB = 4 * C / 32;   // hidden 32-blocks
  constexpr int TILE_B = 32 * 32 * (int)sizeof(T);   // bytes of one operand tile
  constexpr int BLK_B = 2 * KT * TILE_B;             // weights of one hidden block
  constexpr int CHUNKS = BLK_B / 16 / 256;           // 16-byte chunks per thread per block
  __shared__ __attribute__((aligned(16))) char wl[2 * BLK_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const long tok = ((long)blockIdx.x * 4 + wave) * 32 + lr;
  const bool ok = tok < p.M;
  float* xrow = p.x + (ok ? tok : 0) * C;
  const char* Wf = reinterpret_cast<const char*>(p.wfrag);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  u32x4 stg[CHUNKS];
  auto wload = [&](int hb) {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) stg[c] = *reinterpret_cast<const u32x4*>(Wf + (long)hb * BLK_B + (c * 256 + tid) * 16);
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) *reinterpret_cast<u32x4*>(wl + buf * BLK_B + (c * 256 + tid) * 16) = stg[c];
  };
  wload(0);

  float ss = 0.f;
  Frag<T> xf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) xf[kt] = ldx_frag<T>(xrow + kt * 32 + 16 * g, ok, ss);
  ss += __shfl_xor(ss, 32);
  const float scale = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);

  f32x16 acc2[KT];
#pragma unroll
  for (int mt = 0; mt < KT; ++mt) zero16(acc2[mt]);

  wstore(0);
  __syncthreads();
#pragma unroll 1
  for (int hb = 0; hb < HB; ++hb) {
    const char* wb = wl + (hb & 1) * BLK_B;
    if (hb + 1 < HB) wload(hb + 1);
    f32x16 acc1;
    zero16(acc1);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) mma32(acc1, lds_frag<T>(wb + kt * TILE_B, lane), xf[kt]);
    float h[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + hb * 32 + 8 * a + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[4 * a + j] = gelu_erf(fmaf(acc1[4 * a + j], scale, b[j]));
    }
    const Frag<T> hf = pack_frag<T>(h);
#pragma unroll
    for (int mt = 0; mt < KT; ++mt) mma32(acc2[mt], lds_frag<T>(wb + (KT + mt) * TILE_B, lane), hf);
    if (hb + 1 < HB) wstore((hb + 1) & 1);
    __syncthreads();
  }
  if (ok) {
#pragma unroll
    for (int mt = 0; mt < KT; ++mt)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int f0 = mt * 32 + 8 * a + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + f0);
        f32x4* xp = reinterpret_cast<f32x4*>(xrow + f0);
        f32x4 v = *xp;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += acc2[mt][4 * a + j] + b[j];
        *xp = v;
        if (p.xb)
          *reinterpret_cast<hfx4*>(reinterpret_cast<hf*>(p.xb) + tok * C + f0) =
              hfx4{(hf)v[0], (hf)v[1], (hf)v[2], (hf)v[3]};
      }
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
int launch_ff_t(const FusedFFP& p, hipStream_t s) {
  dim3 grid((unsigned)((p.M + 127) / 128)), block(256);
  switch (p.C) {
    case 32: hipLaunchKernelGGL((ff_fused_kernel<T, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((ff_fused_kernel<T, 64>), grid, block, 0, s, p); break;
    case 128: hipLaunchKernelGGL((ff_fused_kernel<T, 128>), grid, block, 0, s, p); break;
    default: return -2;
  }
  return (int)hipGetLastError();
}
}  // namespace

int launch_ff_fused(const FusedFFP& p, int prec, hipStream_t s) {
  if (p.M <= 0) return -2;
  return prec == BT_PREC_F32 ? launch_ff_t<float>(p, s) : launch_ff_t<hf>(p, s);
}
