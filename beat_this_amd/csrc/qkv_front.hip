// Time-direction QKV projection of the frontend's partial transformers (C = 32 / 64 / 128) for the
// half path and (T = hl: hi + lo operands, three MFMAs per product, 4 KB [hi | lo] output blocks) for BT_PREC_F32X3: q|k|v|gates = RMSNorm(x) . W^T, RoPE over the time index, sigmoid gates
// (roformer.py:99-124 on the "(b f) t c" view of beat_tracker.py:297-299), written directly in the
// fragment-major block layout attn_frag_kernel consumes (csrc/attn2.hip).
//
// One wave = one 32-token block = 32 consecutive time steps of one (b, f) sequence, register-chained
// like csrc/fused.hip: lane = token for q/k/gates (D^T = W . X^T), lane = feature for v
// (D = X . W^T), so every store is a full contiguous run of the 2 KB destination block:
//   q/k: 4 x (32 lanes x 16 B = 512 B),  v: 2 x (64 lanes x 16 B = 1 KB),  gates: 128 B per head.
// x is read once (fp32, one or more full 128-byte lines per token), so HBM traffic is the
// algorithmic minimum: 4 B/element in, 3 x 2 B/element out.  Weights are staged fragment-major
// through double-buffered LDS, one head (q, k, v tiles) per step, shared by the 4 waves.
#include "chain.h"
#include "kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

DEVI void split2q(float a, float b, unsigned& whi, unsigned& wlo, float& amax) {
  split_hl(a, b, whi, wlo);   // (common.h)
  amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
}

template <typename T, int C>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? (C == 32 ? 4 : C == 64 ? 3 : 2) : (C == 32 ? 3 : 2)))
void qkv_front_kernel(const QkvFrontP p) {
  constexpr bool X3 = sizeof(T) == 4;    // T = hl
  constexpr int KT = C / 32;             // k-tiles
  constexpr int H = C / 32;              // heads
  constexpr int TILE_B = 1024 * (int)sizeof(T);  // one 32x32 operand tile, fragment-major (hl: [hi tile | lo tile])
  constexpr int BLK_E = X3 ? 2048 : 1024;        // half elements per 32-token output block (X3: [hi block | lo block])
  constexpr int STEP_B = 3 * KT * TILE_B;  // q, k, v tiles of one head (the gate step uses the first KT)
  constexpr int NCH = STEP_B / 16;       // 16-byte chunks per step
  // (hi, lo) operands at C = 128: a step is 48 KB -- double buffered that is ONE workgroup (one wave per SIMD) per CU for a
  // kernel that lives on loads in flight; single buffered (stage, barrier, multiply, barrier) two workgroups cover each other
  constexpr int NBUF = (X3 && C == 128) ? 1 : 2;
  __shared__ __attribute__((aligned(16))) char wl[NBUF * STEP_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const int nblk = (p.T + 31) >> 5;
  const long wb_id = (long)blockIdx.x * 4 + wave;          // global token-block index
  const long n_wb = (long)p.B * p.F * nblk;
  const bool wave_ok = wb_id < n_wb;
  const long wbc = wave_ok ? wb_id : n_wb - 1;
  const int sq = (int)(wbc / nblk), blk = (int)(wbc - (long)sq * nblk);  // sequence (b, f), block
  const int b = sq / p.F, f = sq - b * p.F;
  const int t = blk * 32 + lr;
  const bool ok = wave_ok && t < p.T;
  const float* xrow = p.x + (((long)b * p.T + (ok ? t : 0)) * p.F + f) * C;
  const char* Wf = reinterpret_cast<const char*>(p.wfrag);

  auto stage = [&](int step, int buf) {  // global -> LDS, image == global image (fragment-major tiles)
    const int nch = step < H ? NCH : KT * TILE_B / 16;
#pragma unroll
    for (int i = 0; i < (NCH + 255) / 256; ++i) {
      const int c0 = i * 256 + wave * 64;  // wave-uniform
      if (c0 < nch)
        __builtin_amdgcn_global_load_lds((gptr_t)(Wf + (long)step * STEP_B + (c0 + lane) * 16),
                                         (lptr_t)(wl + buf * STEP_B + c0 * 16), 16, 0, 0);
    }
  };
  if (NBUF == 2) stage(0, 0);

  float ss = 0.f, amax = 0.f;
  Frag<T> xf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) xf[kt] = ldx_frag<T>(xrow + kt * 32 + 16 * g, ok, ss);
  ss += __shfl_xor(ss, 32);
  // (T = hl: operands are pre-scaled, common.h OpScale -- every use of `scale` multiplies a weight . activation product)
  const float scale = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f) * OpScale<T>::PW;
  // RMSNorm factors of the 16 tokens whose V values this lane holds (register r <-> token crow(r,g))
  float sk[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sk[r] = __shfl(scale, crow(r, g));
  // RoPE factors of this lane's token (position = time index); features d = 8a + 4g + 2b (+1)
  f32x2 cs[8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
      cs[2 * a + bb] = *reinterpret_cast<const f32x2*>(p.rope + ((long)(ok ? t : 0) * 16 + 4 * a + 2 * g + bb) * 2);

  hf* qf = reinterpret_cast<hf*>(p.q);
  hf* kf = reinterpret_cast<hf*>(p.k);
  hf* vf = reinterpret_cast<hf*>(p.v);
  __syncthreads();
#pragma unroll 1
  for (int step = 0; step <= H; ++step) {
    const char* wb = wl + (NBUF == 2 ? (step & 1) * STEP_B : 0);
    if (NBUF == 2) {
      if (step < H) stage(step + 1, (step + 1) & 1);
    } else {
      stage(step, 0);
      __syncthreads();   // (this step's tiles have landed in every wave)
    }
    if (step < H) {
      const int hd = step;
      f32x16 aq, ak, av;
      zero16(aq); zero16(ak); zero16(av);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        mma32(aq, lds_frag<T>(wb + kt * TILE_B, lane), xf[kt]);
        mma32(ak, lds_frag<T>(wb + (KT + kt) * TILE_B, lane), xf[kt]);
        mma32(av, xf[kt], lds_frag<T>(wb + (2 * KT + kt) * TILE_B, lane));  // roles swapped: [token][feature]
      }
      if (wave_ok) {
        const long base = (((long)sq * H + hd) * p.nbp + blk) * BLK_E;  // elements
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float q[4], k[4];
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const f32x2 cth = cs[2 * a + bb];
            const int r = 4 * a + 2 * bb;
            const float q0 = aq[r] * scale, q1 = aq[r + 1] * scale, k0 = ak[r] * scale, k1 = ak[r + 1] * scale;
            q[2 * bb] = q0 * cth.x - q1 * cth.y; q[2 * bb + 1] = q1 * cth.x + q0 * cth.y;
            k[2 * bb] = k0 * cth.x - k1 * cth.y; k[2 * bb + 1] = k1 * cth.x + k0 * cth.y;
          }
          const long off = base + (a * 32 + lr) * 8 + 4 * g;
          if constexpr (X3) {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            unsigned qh[2], ql[2], kh[2], kl[2];
            split2q(q[0], q[1], qh[0], ql[0], amax); split2q(q[2], q[3], qh[1], ql[1], amax);
            split2q(k[0], k[1], kh[0], kl[0], amax); split2q(k[2], k[3], kh[1], kl[1], amax);
            *reinterpret_cast<u32x2*>(qf + off) = u32x2{qh[0], qh[1]}; *reinterpret_cast<u32x2*>(qf + off + 1024) = u32x2{ql[0], ql[1]};
            *reinterpret_cast<u32x2*>(kf + off) = u32x2{kh[0], kh[1]}; *reinterpret_cast<u32x2*>(kf + off + 1024) = u32x2{kl[0], kl[1]};
          } else {
            *reinterpret_cast<hfx4*>(qf + off) = hfx4{(hf)q[0], (hf)q[1], (hf)q[2], (hf)q[3]};
            *reinterpret_cast<hfx4*>(kf + off) = hfx4{(hf)k[0], (hf)k[1], (hf)k[2], (hf)k[3]};
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if constexpr (X3) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            unsigned oh[4], ol[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              split2q(av[8 * s + 2 * j] * sk[8 * s + 2 * j], av[8 * s + 2 * j + 1] * sk[8 * s + 2 * j + 1], oh[j], ol[j], amax);
            *reinterpret_cast<u32x4*>(vf + base + (s * 64 + lane) * 8) = u32x4{oh[0], oh[1], oh[2], oh[3]};
            *reinterpret_cast<u32x4*>(vf + base + 1024 + (s * 64 + lane) * 8) = u32x4{ol[0], ol[1], ol[2], ol[3]};
          } else {
            hfx8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (hf)(av[8 * s + j] * sk[8 * s + j]);
            *reinterpret_cast<hfx8*>(vf + base + (s * 64 + lane) * 8) = o;
          }
        }
      }
    } else {  // gate rows: one padded tile row block, gate hd = register hd of the g = 0 half
      f32x16 ag;
      zero16(ag);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) mma32(ag, lds_frag<T>(wb + kt * TILE_B, lane), xf[kt]);
      if (wave_ok && g == 0) {
#pragma unroll
        for (int hd = 0; hd < H; ++hd)
          p.gates[((long)sq * H + hd) * p.nbp * 32 + t] = sigmoidf(fmaf(ag[hd], scale, p.b_gates[hd]));
      }
    }
    __syncthreads();
  }
  // (X3 range guard: q, k, v beyond the fp16 range -- or activations beyond 65504 / OpScale::ACT, which come out as inf / NaN)
  if (X3 && p.status && __any(!(amax <= 65504.f)) && lane == 0) atomicOr(p.status, 1);
}

}  // namespace

int launch_qkv_front(const QkvFrontP& p, hipStream_t s) {
  if (p.B <= 0 || p.T <= 0 || p.F <= 0 || p.nbp < attn_frag_blocks(p.T)) return -2;
  const long n_wb = (long)p.B * p.F * ((p.T + 31) / 32);
  dim3 grid((unsigned)((n_wb + 3) / 4)), block(256);
  if (p.x3) {
    if (BT_HALF_IS_BF16) return -2;
    switch (p.C) {
      case 32: hipLaunchKernelGGL((qkv_front_kernel<hl, 32>), grid, block, 0, s, p); break;
      case 64: hipLaunchKernelGGL((qkv_front_kernel<hl, 64>), grid, block, 0, s, p); break;
      case 128: hipLaunchKernelGGL((qkv_front_kernel<hl, 128>), grid, block, 0, s, p); break;
      default: return -2;
    }
    return (int)hipGetLastError();
  }
  switch (p.C) {
    case 32: hipLaunchKernelGGL((qkv_front_kernel<hf, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((qkv_front_kernel<hf, 64>), grid, block, 0, s, p); break;
    case 128: hipLaunchKernelGGL((qkv_front_kernel<hf, 128>), grid, block, 0, s, p); break;
    default: return -2;
  }
  return (int)hipGetLastError();
}
