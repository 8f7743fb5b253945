// Second generation of the register-chained frontend kernels (C = 32 / 64 / 128): a whole half of a
// PartialFTTransformer per launch, the residual stream x read ONCE and written ONCE.
//
//   attnff_fused_kernel  (frequency direction, beat_tracker.py:293-296):
//       x += AttnF(x);  x += FF_F(x)          = attn_freq_fused_kernel + ff_fused_kernel of fused.hip
//   outff_fused_kernel   (time direction, after the flash attention, beat_tracker.py:297-300):
//       x += to_out(ao); x += FF_T(x)         = the out-projection GEMM + ff_fused_kernel
//
// Both end in the same FF tail.  After the first half the updated x sits in MFMA C-layout registers
// (lane = token, register r <-> feature crow(r, g) of every 32-block); that IS the B-operand form of
// the next MFMA when the weight's k columns are PERM32-ordered (see fused.hip), so W1 is packed with
// PERM32 columns as well and the FF consumes x straight from registers.  Weights stream through
// a 4-stage LDS ring (LDS-DMA), fragment-major, one step per barrier, shared by the 4 waves of a workgroup:
//   [out-proj: KT steps of KT tiles (rows mt*32.., k-tile kt) + KT zero tiles]  (outff only)
//   [FF: 4C/32 steps of KT tiles of W1p (rows hb*32..) + KT tiles of W2p (rows mt*32.., cols hb*32..)]
#include <cstdlib>
#include <type_traits>

#include "chain.h"
#include "kernels.h"

namespace {

// Development instrumentation (per-wave phase timing, load / store ablations) is compiled in with -DBT_DEV only
// (BT_DEV_BUILD=1 for beat_this_amd._lib.build): release builds read no environment variables and carry no debug hooks.

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
DEVI unsigned pk2_hf(float a, float b) {
  const hfx2 t = {(hf)a, (hf)b};
  return __builtin_bit_cast(unsigned, t);
}

// Weight stream: uniform steps of STEP_B bytes (2 KT fragment-major tiles), copied global -> LDS by LDS-DMA
// (buffer_load ... lds, no staging registers) into a ring of NST stages; step s + NST - 1 is issued while
// step s is consumed, so three steps of L2 latency are covered instead of none (the register-staged
// version waited for every step's load inside the step: ~2 k cycles x 25 steps per workgroup).
#ifdef BT_DEV
// development: per-wave phase timing of attnff_fused_kernel (bt_debug_fused2_buffer; null = off)
__device__ long long* g_f2_dbg = nullptr;
#define F2_DBG g_f2_dbg
#define F2_ABL(p) ((p).abl)
#else
#define F2_DBG ((long long*)nullptr)
#define F2_ABL(p) 0
#endif

template <typename T, int C>
struct WRing {
  static constexpr int KT = C / 32;
  static constexpr int TILE_B = 32 * 32 * (int)sizeof(T);
  static constexpr int STEP_B = 2 * KT * TILE_B;
#ifndef BT_F2_NST_HL128
#define BT_F2_NST_HL128 2   // (1.01 ms with two stages vs 1.13 with three, same measurement)
#endif
  // (C = 128: 3 stages = 51 KB, out-projection + FF halves 4 % faster inside the forward; C = 64: 3 stages only help back to
  //  back with itself; 3 workgroups per CU at C = 128: spills.  (hi, lo) operands at C = 128: a stage is 32 KB -- three of
  //  them leave ONE workgroup per CU, i.e. one wave per SIMD; two stages = two workgroups per CU)
#ifndef BT_F2_NST_HL64
#define BT_F2_NST_HL64 2   // (A/B on one box, x3 forward of 16 chunks: frequency + time halves 1.045 / 1.05 / 1.075 ms with 2 / 3 / 4 stages)
#endif
  static constexpr bool HL = sizeof(T) == 4 && !std::is_same<T, float>::value;
  static constexpr int NST = C == 128 ? (HL ? BT_F2_NST_HL128 : 3) : (HL && C == 64 ? BT_F2_NST_HL64 : 4);
  static constexpr int CH = STEP_B / 4096;  // buffer loads per thread per step (256 threads x 16 B = 4 KB each)
  rsrc_t rs;
  char* lds;
  int tid, wave, total;
  long long t_wait = 0, t_bar = 0;  // (timing dump only)
  bool timing = false;
  DEVI void issue(int s) {
    if (s >= total) return;
    char* dst = lds + (s % NST) * STEP_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + i * 4096), 16, tid * 16, s * STEP_B + i * 4096, 0, 0);
  }
  DEVI void prologue() {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s);
  }
  // make step s readable by every wave (all older LDS-DMA done in every wave), then refill the stage freed by step s-1
  DEVI const char* acquire(int s) {
    const int ahead = min(NST - 2, total - 1 - s);  // younger steps that may stay in flight
    long long q0 = 0, q1 = 0;
    if (timing) q0 = clock64();
    // lgkmcnt(0): every LDS read this wave issued has RETURNED before the barrier.  The stage refilled right after the
    // barrier is the one step s - 1 was read from, and hipcc sinks the MFMAs of a step's last tiles (with their still
    // outstanding ds_reads) below this wait: a fast wave's LDS-DMA then raced a slow wave's queued read (WAR; a few
    // garbage 32-token blocks per launch of 6000 workgroups, only under load -- found in round 2 by running the same batch
    // twice, tools/kernel_determinism.py).  A raw s_barrier orders nothing by itself.
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * CH) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (timing) q1 = clock64();
    __builtin_amdgcn_s_barrier();
    if (timing) { const long long q2 = clock64(); t_wait += q1 - q0; t_bar += q2 - q1; }
    issue(s + NST - 1);
    return lds + (s % NST) * STEP_B;
  }
};

// FF tail: xn[kt][r] = updated x of this lane's token in C layout; `step0` = stream step of the first hidden
// block.  b1s = first-layer bias in LDS (loaded once per workgroup; keeps ordinary global loads out of the ring's
// vmcnt accounting).
template <typename T, int C>
DEVI void ff_tail(WRing<T, C>& ws, int step0, float (&xn)[C / 32][16], const float* b1s, const float* b2,
                  float* xrow, hf* xbrow, bool ok, int lane, int g) {
  constexpr int KT = C / 32, HB = 4 * C / 32;
  constexpr int TILE_B = WRing<T, C>::TILE_B;
  float ss = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ss = fmaf(xn[kt][r], xn[kt][r], ss);
  ss += __shfl_xor(ss, 32);
  constexpr float PW = OpScale<T>::PW;   // (1 unless T = hl: products of pre-scaled operands are scaled back where consumed)
  const float scale = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f) * PW;
  Frag<T> xf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) xf[kt] = pack_frag<T>(xn[kt]);
  f32x16 acc2[KT];  // starts from x itself: the residual add costs nothing and xn's registers are free from here on
#pragma unroll
  for (int mt = 0; mt < KT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[mt][r] = xn[mt][r] * (1.f / PW);
#pragma unroll 1
  for (int hb = 0; hb < HB; ++hb) {
    const char* wb = ws.acquire(step0 + hb);
    f32x16 acc1;
    zero16(acc1);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) mma32(acc1, lds_frag<T>(wb + kt * TILE_B, lane), xf[kt]);
    float h[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + hb * 32 + 8 * a + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[4 * a + j] = gelu_t<T>(fmaf(acc1[4 * a + j], scale, b[j]));
    }
    const Frag<T> hf = pack_frag<T>(h);
#pragma unroll
    for (int mt = 0; mt < KT; ++mt) mma32(acc2[mt], lds_frag<T>(wb + (KT + mt) * TILE_B, lane), hf);
  }
#pragma unroll
  for (int mt = 0; mt < KT; ++mt) {
    f32x4 v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int f0 = mt * 32 + 8 * a + 4 * g;
      const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + f0);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[a][j] = fmaf(acc2[mt][4 * a + j], PW, b[j]);
      if (ok) *reinterpret_cast<f32x4*>(xrow + f0) = v[a];
    }
    if (xbrow) {  // half shadow: the two halves of the wave exchange 4-feature runs so that a lane stores 16 bytes
                  // (features 16 k + 8 g .. + 7); 8-byte row-strided stores cost 44 us per forward here.  Executed by all lanes.
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      if constexpr (std::is_same<T, hl>::value) {
        // BT_PREC_F32X3: the shadow is the hl32 form of the row (gemm3.hip: per 32 features [32 hi halves | 32 lo halves]),
        // UNSCALED (the GEMM that reads it applies no operand scale)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          unsigned xh[2], xl[2], yh[2], yl[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            split_hl4(v[2 * k][2 * i], v[2 * k][2 * i + 1], v[2 * k + 1][2 * i], v[2 * k + 1][2 * i + 1], xh[i], xl[i], yh[i], yl[i]);   // (common.h)
          }
          auto h0 = __builtin_amdgcn_permlane32_swap(xh[0], yh[0], false, false);
          auto h1 = __builtin_amdgcn_permlane32_swap(xh[1], yh[1], false, false);
          auto l0 = __builtin_amdgcn_permlane32_swap(xl[0], yl[0], false, false);
          auto l1 = __builtin_amdgcn_permlane32_swap(xl[1], yl[1], false, false);
          if (ok) {
            *reinterpret_cast<u32x4*>(xbrow + mt * 64 + 16 * k + 8 * g) = u32x4{h0[0], h1[0], h0[1], h1[1]};
            *reinterpret_cast<u32x4*>(xbrow + mt * 64 + 32 + 16 * k + 8 * g) = u32x4{l0[0], l1[0], l0[1], l1[1]};
          }
        }
      } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const unsigned x0 = pk2_hf(v[2 * k][0], v[2 * k][1]), x1 = pk2_hf(v[2 * k][2], v[2 * k][3]);
        const unsigned y0 = pk2_hf(v[2 * k + 1][0], v[2 * k + 1][1]), y1 = pk2_hf(v[2 * k + 1][2], v[2 * k + 1][3]);
        auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
        if (ok) *reinterpret_cast<u32x4*>(xbrow + mt * 32 + 16 * k + 8 * g) = u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void outff_fused_kernel(const FusedOutFFP p) {
  constexpr int KT = C / 32, HB = 4 * C / 32;
  constexpr int TILE_B = WRing<T, C>::TILE_B, STEP_B = WRing<T, C>::STEP_B, NST = WRing<T, C>::NST;
  __shared__ __attribute__((aligned(16))) char wl[NST * STEP_B + 4 * C * 4];
  float* b1s = reinterpret_cast<float*>(wl + NST * STEP_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const long tok = ((long)blockIdx.x * 4 + wave) * 32 + lr;
  const bool ok_st = tok < p.M && !(F2_ABL(p) & 2);
  const bool ok = tok < p.M && !(F2_ABL(p) & 1);
  float* xrow = p.x + (tok < p.M ? tok : 0) * C;
  for (int i = tid; i < 4 * C; i += 256) b1s[i] = p.b1[i];

  // attention output row of this token as B-operand fragments (natural k order)
  const T* arow = reinterpret_cast<const T*>(p.ao) + (ok ? tok : 0) * C;
  Frag<T> af[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) af[kt] = ldg_frag<T>(arow + kt * 32 + 16 * g);
  // residual stream in C layout
  float xn[KT][16];
#pragma unroll
  for (int mt = 0; mt < KT; ++mt)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(xrow + mt * 32 + 8 * a + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) xn[mt][4 * a + j] = v[j];
    }
  WRing<T, C> ws;
  ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, (unsigned)((KT + HB) * STEP_B), 0x00020000);
  ws.lds = wl; ws.tid = tid; ws.wave = wave; ws.total = KT + HB;
  __syncthreads();  // b1s is complete (the ring's raw barriers carry no LDS-write wait of their own)
  // No ordinary global load may still be in flight when the ring starts: hipcc waits for such loads with COUNTED
  // vmcnt values that assume in-order return, and LDS-DMA returns are not ordered against VGPR returns (observed:
  // RoPE factors consumed before they arrived, a few wrong rows per launch at C = 32).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ws.prologue();
  // ---- x += Wout . ao : one 32-feature row block of Wout per step (KT tiles + KT pad tiles) ----------------------
#pragma unroll
  for (int mt = 0; mt < KT; ++mt) {
    const char* wb = ws.acquire(mt);
    f32x16 acc;
    zero16(acc);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) mma32(acc, lds_frag<T>(wb + kt * TILE_B, lane), af[kt]);
#pragma unroll
    for (int r = 0; r < 16; ++r) xn[mt][r] = fmaf(acc[r], OpScale<T>::PW, xn[mt][r]);
  }
  hf* xbrow = p.xb ? reinterpret_cast<hf*>(p.xb) + tok * C * (std::is_same<T, hl>::value ? 2 : 1) : nullptr;
  ff_tail<T, C>(ws, KT, xn, b1s, p.b2, xrow, xbrow, ok_st, lane, g);
}

// ---------------------------------------------------------------------------------------------------------
// Frequency-direction half: attention (as attn_freq_fused_kernel of fused.hip, but with its weights staged
// through the same LDS stream instead of per-wave L2 reads) followed by the FF tail.  Stream steps, each
// 2 KT tiles:  [gate rows | pad]  then per head  [q rows | k rows]  [v rows | PERM32'd to_out tiles of the
// head's 32 columns]  then the FF steps (bt_pair_weights.w_attnff_frag).
template <typename T, int C>
__global__ __launch_bounds__(256) void attnff_fused_kernel(const FusedAttnFFP p) {
  constexpr int KT = C / 32;
  constexpr int H = C / 32;       // heads
  constexpr int F = 1024 / C;     // tokens per (b,t) row: 32, 16, 8
  constexpr int HB = 4 * C / 32;
  constexpr int TILE_B = WRing<T, C>::TILE_B, STEP_B = WRing<T, C>::STEP_B, NST = WRing<T, C>::NST;
  __shared__ __attribute__((aligned(16))) char wl[NST * STEP_B + 4 * C * 4];
  float* b1s = reinterpret_cast<float*>(wl + NST * STEP_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const long tok = ((long)blockIdx.x * 4 + wave) * 32 + lr;
  const bool ok = tok < p.M;
  float* xrow = p.x + (ok ? tok : 0) * C;
  long long* dbg = F2_DBG;
  const long long t_in = dbg ? clock64() : 0;
  for (int i = tid; i < 4 * C; i += 256) b1s[i] = p.b1[i];

  float ss = 0.f;
  Frag<T> xf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) xf[kt] = ldx_frag<T>(xrow + kt * 32 + 16 * g, ok, ss);
  ss += __shfl_xor(ss, 32);
  constexpr float PW = OpScale<T>::PW, PA = OpScale<T>::PA;  // (1 unless T = hl, see common.h)
  const float scale = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f) * PW;   // (every use multiplies a weight . activation product)
  // RMSNorm factors of the 16 tokens whose V rows this lane holds (register r <-> token crow(r,g))
  float sk[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sk[r] = __shfl(scale, crow(r, g));
  // RoPE factors for this lane's token: position = token index inside its (b,t) row
  const int pos = (int)(tok & (F - 1));
  f32x2 cs[8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)  // features d = 8a + 4g + 2b (+1): pair index d/2 = 4a + 2g + b
      cs[2 * a + b] = *reinterpret_cast<const f32x2*>(p.rope + ((long)pos * 16 + 4 * a + 2 * g + b) * 2);
  float bg[H];
#pragma unroll
  for (int hd = 0; hd < H; ++hd) bg[hd] = p.b_gates[hd];
  WRing<T, C> ws;
  ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, (unsigned)((1 + 2 * H + HB) * STEP_B), 0x00020000);
  ws.lds = wl; ws.tid = tid; ws.wave = wave; ws.total = 1 + 2 * H + HB;
  __syncthreads();  // b1s is complete (the ring's raw barriers carry no LDS-write wait of their own)
  // No ordinary global load may still be in flight when the ring starts: hipcc waits for such loads with COUNTED
  // vmcnt values that assume in-order return, and LDS-DMA returns are not ordered against VGPR returns (observed:
  // RoPE factors consumed before they arrived, a few wrong rows per launch at C = 32).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ws.timing = dbg != nullptr;
  const long long t_ring = dbg ? clock64() : 0;
  ws.prologue();

  // ---- step 0: gates of all heads; gate hd sits in register hd of the g = 0 half ----------------------------
  float gate[H];
  {
    const char* wb = ws.acquire(0);
    f32x16 ag;
    zero16(ag);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) mma32(ag, lds_frag<T>(wb + kt * TILE_B, lane), xf[kt]);
#pragma unroll
    for (int hd = 0; hd < H; ++hd) {
      float v = __shfl(ag[hd], lr);
      gate[hd] = sigmoidf(fmaf(v, scale, bg[hd]));
    }
  }

  f32x16 acco[KT];
#pragma unroll
  for (int mt = 0; mt < KT; ++mt) zero16(acco[mt]);
#pragma unroll 1
  for (int hd = 0; hd < H; ++hd) {
    // ---- step 1 + 2 hd: q^T, k^T (lane = token), scores, softmax -----------------------------------------------
    float pr[16], l = 0.f;
    {
      const char* wb = ws.acquire(1 + 2 * hd);
      f32x16 aq, ak;
      zero16(aq); zero16(ak);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        mma32(aq, lds_frag<T>(wb + kt * TILE_B, lane), xf[kt]);
        mma32(ak, lds_frag<T>(wb + (KT + kt) * TILE_B, lane), xf[kt]);
      }
      float q[16], k[16];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f32x2 t = cs[2 * a + b];
          const int r = 4 * a + 2 * b;
          const float q0 = aq[r] * scale, q1 = aq[r + 1] * scale, k0 = ak[r] * scale, k1 = ak[r + 1] * scale;
          q[r] = q0 * t.x - q1 * t.y; q[r + 1] = q1 * t.x + q0 * t.y;
          k[r] = k0 * t.x - k1 * t.y; k[r + 1] = k1 * t.x + k0 * t.y;
        }
      // S^T[key][query] = K . Q^T over d (k-slots = registers of both)
      f32x16 sc;
      zero16(sc);
      mma32(sc, pack_frag<T>(k), pack_frag<T>(q));
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[r] *= PA;
        if (F < 32 && ((crow(r, g) ^ lr) & ~(F - 1))) sc[r] = -1e30f;  // key and query in different rows
        mx = fmaxf(mx, sc[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(sc[r] - mx);
        l += pr[r];
      }
      l += __shfl_xor(l, 32);
    }
    // ---- step 2 + 2 hd: v (lane = feature), O^T = V^T . P^T, out-projection of this head ------------------------
    {
      const char* wb = ws.acquire(2 + 2 * hd);
      f32x16 av;
      zero16(av);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) mma32(av, xf[kt], lds_frag<T>(wb + kt * TILE_B, lane));  // roles swapped
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = av[r] * sk[r];
      f32x16 ao;
      zero16(ao);
      mma32(ao, pack_frag<T>(v), pack_frag<T>(pr));
      // (gate[hd] by selects: a run-time index into the register array put it into scratch memory -- three scratch accesses
      // in a kernel whose weight ring is LDS-DMA; tools/isa_lint.py, fourth rule)
      float gsel = gate[0];
#pragma unroll
      for (int k = 1; k < H; ++k) gsel = hd == k ? gate[k] : gsel;
      const float fin = gsel / l * PA;
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = ao[r] * fin;
      const Frag<T> of = pack_frag<T>(o);
#pragma unroll
      for (int mt = 0; mt < KT; ++mt) mma32(acco[mt], lds_frag<T>(wb + (KT + mt) * TILE_B, lane), of);
    }
  }
  // ---- x (C layout) += attention, then the FF tail (first FF step = stream step 1 + 2 H) -------------------------
  // These ordinary loads are issued while ring steps are in flight: wait for EVERYTHING before the first use
  // (a counted wait would be wrong, see the note at the ring prologue), and pin the wait before the uses.
  f32x4 xv[KT][4];
#pragma unroll
  for (int mt = 0; mt < KT; ++mt)
#pragma unroll
    for (int a = 0; a < 4; ++a)
      xv[mt][a] = ok ? *reinterpret_cast<const f32x4*>(xrow + mt * 32 + 8 * a + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  float xn[KT][16];
#pragma unroll
  for (int mt = 0; mt < KT; ++mt)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) xn[mt][4 * a + j] = fmaf(acco[mt][4 * a + j], PW, xv[mt][a][j]);
  const long long t_ff = dbg ? clock64() : 0;
  ff_tail<T, C>(ws, 1 + 2 * H, xn, b1s, p.b2, xrow, nullptr, ok, lane, g);
  if (dbg && lane == 0) {
    long long* d = dbg + ((long)blockIdx.x * 4 + wave) * 6;
    d[0] = t_ring - t_in; d[1] = t_ff - t_ring; d[2] = clock64() - t_ff; d[3] = ws.t_wait; d[4] = ws.t_bar; d[5] = 0;
  }
}

template <typename T>
int launch_outff_t(const FusedOutFFP& p, hipStream_t s) {
  dim3 grid((unsigned)((p.M + 127) / 128)), block(256);
  switch (p.C) {
    case 32: hipLaunchKernelGGL((outff_fused_kernel<T, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((outff_fused_kernel<T, 64>), grid, block, 0, s, p); break;
    case 128: hipLaunchKernelGGL((outff_fused_kernel<T, 128>), grid, block, 0, s, p); break;
    default: return -2;
  }
  return (int)hipGetLastError();
}
template <typename T>
int launch_attnff_t(const FusedAttnFFP& p, hipStream_t s) {
  dim3 grid((unsigned)((p.M + 127) / 128)), block(256);
  switch (p.C) {
    case 32: hipLaunchKernelGGL((attnff_fused_kernel<T, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((attnff_fused_kernel<T, 64>), grid, block, 0, s, p); break;
    case 128: hipLaunchKernelGGL((attnff_fused_kernel<T, 128>), grid, block, 0, s, p); break;
    default: return -2;
  }
  return (int)hipGetLastError();
}

}  // namespace

#ifdef BT_DEV
extern "C" int bt_debug_fused2_buffer(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_f2_dbg), &buf, sizeof buf);
}
#endif

int launch_outff_fused(const FusedOutFFP& p0, int prec, hipStream_t s) {
  if (p0.M <= 0) return -2;
  FusedOutFFP p = p0;
  p.abl = 0;
#ifdef BT_DEV
  static const int abl = getenv("BT_F2_ABL") ? atoi(getenv("BT_F2_ABL")) : 0;
  p.abl = abl;
#endif
  if (prec == BT_PREC_F32X3) return BT_HALF_IS_BF16 ? -2 : launch_outff_t<hl>(p, s);
  return prec == BT_PREC_F32 ? launch_outff_t<float>(p, s) : launch_outff_t<hf>(p, s);
}
int launch_attnff_fused(const FusedAttnFFP& p, int prec, hipStream_t s) {
  if (p.M <= 0) return -2;
  if (prec == BT_PREC_F32X3) return BT_HALF_IS_BF16 ? -2 : launch_attnff_t<hl>(p, s);
  return prec == BT_PREC_F32 ? launch_attnff_t<float>(p, s) : launch_attnff_t<hf>(p, s);
}
