// Tail of a main transformer layer in ONE launch (BT_PREC_HALF, transformer_dim C = 256 / 512):
//     x += to_out(ao)                         (roformer.py:130-132, the attention out-projection)
//     x += W2 . gelu(W1 . rmsnorm(x) + b1) + b2  (roformer.py:38-61, FeedForward)
// replacing three GEMM launches (out-projection, FF1, FF2 on gemm3.hip) that moved, per layer and 1500-frame chunk,
// the fp32 residual stream in and out twice (12 MB), its half shadow twice, the RMSNorm partials, and -- the big one --
// the 4C-wide hidden activation out and back in (12 MB): here x is read once, the hidden activation never leaves the
// registers, and x, its shadow and the statistics of the NEW x are written once.
//
// Register-chained like fused2.hip (lane = token): a wave owns 32 tokens, all C features of them: the residual row lives
// in the 16 (C = 512) accumulator tiles of the second FF GEMM for the whole kernel (C-layout: register r <-> feature
// crow(r, g) of every 32-block), which IS the B-operand form of the next MFMA when the weight's k columns are
// PERM32-ordered.  The weights stream through a 128 KB LDS ring (LDS-DMA, fragment-major tiles, one step = C / 32 tiles =
// one barrier), shared by the 4 waves (128 tokens) of the workgroup -- one workgroup per CU, one wave per SIMD, ~480 of
// the 512 registers.  Stream order (beat_this_amd/pack.py: tail_fragment_major):
//     [out-proj row blocks (2 st, 2 st + 1), st = 0 .. KT/2-1: KT k-tiles each]   then S(-1) .. S(HB), S(i) = [A(i+1) | B(i-1)]
// with A(hb) = the KT k-tiles of W1 rows hb*32.. (columns PERM32), B(hb) = the KT row tiles of PERM32'd W2, columns hb*32..
// (absent halves are zero tiles).  Step i multiplies the first FF product of hidden block i + 1 and the second product of
// block i - 1 -- two interleaved MFMA streams -- while the VALU evaluates the activation of block i in their shadow (one
// wave per SIMD: nobody else would fill the gaps).
#include "chain.h"
#include "kernels.h"

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

DEVI unsigned pk2(float a, float b) {
  const hfx2 t = {(hf)a, (hf)b};
  return __builtin_bit_cast(unsigned, t);
}

constexpr int TILE_B = 2048;  // one 32 x 32 half tile, fragment-major: [piece 0..1][lane][8]

// The 256 accumulator registers of the second FF product are PINNED to AGPRs with inline assembly ("+a"): left to
// itself hipcc (ROCm 7.2) put temporaries into the AGPRs and spilled the operand fragments to scratch (1.8 KB per lane,
// a scratch reload in front of every MFMA).  Inline assembly is invisible to the compiler's MFMA hazard recogniser:
//   * consecutive MFMAs accumulating into the SAME registers need no wait states (hardware back-to-back accumulate);
//   * the operands come from LDS reads / VALU conversions issued long before (waitcnt insertion does see the operands);
//   * consumers of the results outside this assembly: the statistics / operand packing after the out-projection and the
//     epilogue, both fenced by mfma_results_ready(); the activation reads the first product's chain one whole ring
//     step (>= 2 later MFMAs on the in-order matrix pipe, a barrier, LDS reads) after its last MFMA was issued.
#if BT_HALF_IS_BF16
#define TAIL_MFMA "v_mfma_f32_32x32x16_bf16"
#else
#define TAIL_MFMA "v_mfma_f32_32x32x16_f16"
#endif
// the chain of the first FF product: VGPRs, by assembly as well (a builtin MFMA's destination is the register
// allocator's choice, and it chose AGPRs -- evicting accumulator tiles to scratch inside the loop)
DEVI void mma32_vgpr_first(f32x16& acc, const Frag<hf>& a, const Frag<hf>& b) {
  asm volatile(TAIL_MFMA " %0, %1, %2, 0" : "=&v"(acc) : "v"(a.v[0]), "v"(b.v[0]));
  asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+v"(acc) : "v"(a.v[1]), "v"(b.v[1]));
}
DEVI void mma32_vgpr(f32x16& acc, const Frag<hf>& a, const Frag<hf>& b) {
  asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+v"(acc) : "v"(a.v[0]), "v"(b.v[0]));
  asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+v"(acc) : "v"(a.v[1]), "v"(b.v[1]));
}
DEVI void mma32_agpr(f32x16& acc, const Frag<hf>& a, const Frag<hf>& b) {
  asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+a"(acc) : "v"(a.v[0]), "v"(b.v[0]));
  asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+a"(acc) : "v"(a.v[1]), "v"(b.v[1]));
}
// XDL write -> VALU / memory read of the result: (passes + 3) wait states for an 8-pass MFMA on gfx94x/95x, rounded up generously
DEVI void mfma_results_ready() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
DEVI void agpr_fence(f32x16& acc) { asm volatile("" : "+a"(acc)); }

// DS reads and LDS-DMA issues are pinned where they are written (ALU / MFMA may move across): the scheduler otherwise
// hoists all 32 fragment reads of a step in front of its MFMAs -- 128 VGPRs of landing space that do not exist here --
// and bunches the LDS-DMA issues.
#define TAIL_PIN_DS() __builtin_amdgcn_sched_barrier(0x1 | 0x2 | 0x4 | 0x8 | 0x400)

// (What the kernel spends where was measured in round 2 with ablated builds of it -- M = 24000 (188 workgroups) / 32768 (256):
// whole kernel 161 / 186 us; without the LDS-DMA 119 / 120; without the fragment reads 157 / 161; without the activation 147 /
// 160; MFMAs + prologue + epilogue only 108 / 109 -- DESIGN.md section 5; the switches are gone from the source.)
DEVI Frag<hf> tail_frag(const char* p, int lane) {
  return lds_frag<hf>(p, lane);
}

template <int C>
struct TRing {
  static constexpr int KT = C / 32;
  static constexpr int STEP_B = 2 * KT * TILE_B;             // a step = 2 KT tiles (C = 512: 64 KB)
  // TWO stages for both widths (C = 512: 2 x 64 KB, C = 256: 2 x 32 KB): every acquire() is then the same branch-free
  // `vmcnt(0) lgkmcnt(0); barrier; vmcnt(0); burst` sequence.  The 4-stage ring C = 256 had through round 2 chose its
  // counted vmcnt wait in a branch on the step index and returned NaNs -- the same symptom as the half-step refill
  // experiments of DESIGN.md section 5 as soon as their wait sat in such a branch (hipcc's code around the pinned inline
  // assembly, not the hardware: the LDS-DMA order itself is verified by tools/ubench/ldsdma_order.hip).
  static constexpr int NST = 2;
  static constexpr int CH = STEP_B / 4096;  // buffer loads per thread per step (256 threads x 16 B = 4 KB each)
  static_assert(NST == 2, "ring shape");
  rsrc_t rs;
  char* lds;
  int tid, wave, total;
  DEVI void issue(int s) {
    if (s >= total) return;
    char* dst = lds + (s % NST) * STEP_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + i * 4096), 16, tid * 16, s * STEP_B + i * 4096, 0, 0);
  }
  // (Issuing a step's CH pieces one per tile iteration instead of as one burst behind the barrier was measured in round 2:
  // 1.028 - 1.036 ms per forward against 1.048 for the burst when pinned between full scheduling barriers, GARBAGE when left
  // free to move around the masked one -- hipcc's placement, not the hardware.  The placement hardly matters: an LDS-DMA
  // instruction blocks the issuing wave until the load path takes it, and with one wave per SIMD nobody else issues MFMAs
  // meanwhile.  The burst ships; the variants are gone from the source.)
  DEVI void prologue() {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s);
  }
  // make step s readable by every wave (all older LDS-DMA done in every wave); the stage of step s - 1 is free from here on
  DEVI const char* acquire(int s) {
    // (lgkmcnt(0): this wave's fragment reads of step s - 1 have returned before the barrier: its stage is refilled next;
    // vmcnt(0): step s -- the only one in flight in a two-stage ring -- has landed)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // hipcc reloads spilled registers (scratch_load -> VGPR) wherever it likes and waits for them with COUNTED vmcnt
    // values that assume LDS-DMA and VGPR returns retire in one order -- they do not (fused2.hip): a reload placed between
    // this barrier and the burst and used behind it (the build of round 2 had one, an LDS address) would be "waited for"
    // with vmcnt(16) while any one of the 16 younger LDS-DMA instructions may retire first.  Nothing of the ring is in
    // flight here (2-stage ring: vmcnt(0) above), so draining whatever the compiler has put here costs a reload's latency
    // at most, and nothing can move across: the burst sits between full scheduling barriers.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue(s + NST - 1);
    __builtin_amdgcn_sched_barrier(0);
    return lds + (s % NST) * STEP_B;   // (the caller issues step s + NST - 1 piece by piece while it multiplies step s)
  }
};

// One FF step = tiles [A(i + 1): KT k-tiles of W1 rows (i+1)*32..] [B(i - 1): KT row tiles of W2, columns (i-1)*32..]:
//   ne    += A . xf            first product of hidden block i + 1                      (32 MFMAs, one chain)
//   acc2  += B . hprev         second product of hidden block i - 1                     (32 MFMAs, KT chains)
//   hcur   = gelu(ce * scale + b1)   activation of hidden block i, one element per tile pair, in the MFMAs' shadow
// The two MFMA streams alternate, so the dependent chain on `ne` never issues back to back; fragments are read one tile
// pair ahead.
template <int KT, bool HAS_A, bool HAS_G, bool HAS_B, typename RING>
DEVI void ff_step(RING& ws, int step, int lane, const Frag<hf> (&xf)[KT], f32x16 (&acc2)[KT], const f32x16& ce, f32x16& ne,
                  const Frag<hf>& hprev, Frag<hf>& hcur, const float* b1lane, float scale) {
  static_assert(RING::CH == KT, "one LDS-DMA piece per tile iteration");
  const char* wb = ws.acquire(step);
  Frag<hf> fa, fb, na, nb;
  if (HAS_A) fa = tail_frag(wb, lane);
  if (HAS_B) fb = tail_frag(wb + KT * TILE_B, lane);
  unsigned hw[8];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (t + 1 < KT) {
      if (HAS_A) na = tail_frag(wb + (t + 1) * TILE_B, lane);
      if (HAS_B) nb = tail_frag(wb + (KT + t + 1) * TILE_B, lane);
    }
    TAIL_PIN_DS();
    // the two MFMAs (k-halves) of a product back to back
    if (HAS_A) {
      if (t == 0) mma32_vgpr_first(ne, fa, xf[0]);
      else mma32_vgpr(ne, fa, xf[t]);
    }
    if (HAS_G && (t * 8) % KT == 0) {
#pragma unroll
      for (int q = 0; q < (KT >= 8 ? 1 : 8 / KT); ++q) {  // elements r, r + 1 -> one packed dword of the next B operand
        const int r = 2 * (t * 8 / KT + q);
        const f32x2 b = *reinterpret_cast<const f32x2*>(b1lane + 8 * (r >> 2) + (r & 3));
        hw[r >> 1] = pk2(gelu_tanh(fmaf(ce[r], scale, b[0])), gelu_tanh(fmaf(ce[r + 1], scale, b[1])));
      }
    }
    if (HAS_B) mma32_agpr(acc2[t], fb, hprev);
    if (t + 1 < KT) { fa = na; fb = nb; }
  }
  if (HAS_G) {
    hcur.v[0] = __builtin_bit_cast(hfx8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    hcur.v[1] = __builtin_bit_cast(hfx8, u32x4{hw[4], hw[5], hw[6], hw[7]});
  }
}

template <int C>
__global__ __launch_bounds__(256, 1) void layer_tail_kernel(const LayerTailP p) {
  constexpr int KT = C / 32;
  using Ring = TRing<C>;
  constexpr int STEP_B = Ring::STEP_B, NST = Ring::NST;
  extern __shared__ __attribute__((aligned(16))) char wl[];
  float* b1s = reinterpret_cast<float*>(wl + NST * STEP_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, lr = lane & 31;
  const int HB = p.hidden >> 5;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
  const long tok = row0 + lr;
  const bool ok = tok < p.M;
  const float* xrow = p.x + (ok ? tok : 0) * C;
  for (int i = tid; i < p.hidden; i += 256) b1s[i] = p.b1[i];

  // Residual row in C layout, in two batches of KT / 2 tiles: all 64 loads at once would need 256 VGPRs of landing space
  // next to the 128 of the attention row (the accumulator tiles themselves live in AGPRs); the prologue is an HBM burst
  // of every CU at once anyway (bandwidth, not latency, bound).
  f32x16 acc2[KT];  // the residual row, then the accumulators of the second FF product
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int mt = half * (KT / 2); mt < (half + 1) * (KT / 2); ++mt)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(xrow + mt * 32 + 8 * a + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[mt][4 * a + j] = v[j];
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int mt = half * (KT / 2); mt < (half + 1) * (KT / 2); ++mt) agpr_fence(acc2[mt]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // attention output row of this token as B-operand fragments (natural k order)
  const hf* arow = reinterpret_cast<const hf*>(p.ao) + (ok ? tok : 0) * C;
  Frag<hf> af[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) af[kt] = ldg_frag<hf>(arow + kt * 32 + 16 * g);
  Ring ws;
  const int total = KT / 2 + HB + 2;
  ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, (unsigned)((long)total * STEP_B), 0x00020000);
  ws.lds = wl; ws.tid = tid; ws.wave = wave; ws.total = total;
  __syncthreads();  // b1s is complete (the ring's raw barriers carry no LDS-write wait of their own)
  // No ordinary global load may still be in flight when the ring starts (LDS-DMA and VGPR returns are not ordered
  // against each other, and hipcc's counted vmcnt waits assume they are): see fused2.hip.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  ws.prologue();

  // ---- x += Wout . ao : TWO 32-feature row blocks of Wout per step (two independent accumulation chains) -----------
#pragma unroll
  for (int st = 0; st < KT / 2; ++st) {
    const char* wb = ws.acquire(st);
    Frag<hf> f0 = tail_frag(wb, lane), f1 = tail_frag(wb + KT * TILE_B, lane), n0, n1;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt + 1 < KT) {
        n0 = tail_frag(wb + (kt + 1) * TILE_B, lane);
        n1 = tail_frag(wb + (KT + kt + 1) * TILE_B, lane);
      }
      TAIL_PIN_DS();
      // straight into the residual row's accumulator tiles (AGPRs): x += Wout . ao costs no VALU and no extra registers
      asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+a"(acc2[2 * st]) : "v"(f0.v[0]), "v"(af[kt].v[0]));
      asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+a"(acc2[2 * st + 1]) : "v"(f1.v[0]), "v"(af[kt].v[0]));
      asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+a"(acc2[2 * st]) : "v"(f0.v[1]), "v"(af[kt].v[1]));
      asm volatile(TAIL_MFMA " %0, %1, %2, %0" : "+a"(acc2[2 * st + 1]) : "v"(f1.v[1]), "v"(af[kt].v[1]));
      if (kt + 1 < KT) { f0 = n0; f1 = n1; }
    }
  }
  mfma_results_ready();
#pragma unroll
  for (int mt = 0; mt < KT; ++mt) agpr_fence(acc2[mt]);

  // ---- RMSNorm statistics of the new x, operand fragments of the first FF product -------------------------------------
  float ss = 0.f;
  Frag<hf> xf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { t[r] = acc2[kt][r]; ss = fmaf(t[r], t[r], ss); }
    xf[kt] = pack_frag<hf>(t);
    agpr_fence(acc2[kt]);
  }
  ss += __shfl_xor(ss, 32);
  const float scale = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);

  // ---- FF: steps S(-1) .. S(HB): S(i) = [A(i + 1) | B(i - 1)] (absent halves are zero tiles), activation of block i ----
  const float* b1lane = b1s + 4 * g;
  f32x16 c0, c1;
  Frag<hf> h0, h1;
  const int s0 = KT / 2;  // first FF step of the stream
  ff_step<KT, true, false, false>(ws, s0, lane, xf, acc2, c0, c0, h0, h0, b1lane, scale);             // S(-1): c0 = A(0)
  ff_step<KT, true, true, false>(ws, s0 + 1, lane, xf, acc2, c0, c1, h0, h0, b1lane, scale);          // S(0): c1 = A(1), h0 = act(0)
#pragma unroll 1
  for (int i = 1; i + 2 < HB; i += 2) {  // S(i): ce = c1, ne = c0, hprev = h0, hcur = h1;  S(i + 1): roles swapped
    ff_step<KT, true, true, true>(ws, s0 + 1 + i, lane, xf, acc2, c1, c0, h0, h1, b1lane + 32 * i, scale);
    ff_step<KT, true, true, true>(ws, s0 + 2 + i, lane, xf, acc2, c0, c1, h1, h0, b1lane + 32 * (i + 1), scale);
  }
  // HB is even: S(HB - 1) has no A half, S(HB) neither A nor activation
  ff_step<KT, false, true, true>(ws, s0 + HB, lane, xf, acc2, c1, c1, h0, h1, b1lane + 32 * (HB - 1), scale);
  ff_step<KT, false, false, true>(ws, s0 + HB + 1, lane, xf, acc2, c0, c0, h1, h1, b1lane, scale);
  mfma_results_ready();
#pragma unroll
  for (int mt = 0; mt < KT; ++mt) agpr_fence(acc2[mt]);

  // ---- epilogue: + b2, x / shadow / statistics leave through LDS as whole rows (see gemm3.hip, RESID epilogue) ---------
  __syncthreads();  // every wave is past its last fragment read: the ring area is free
  char* wst = wl + wave * 8192;
  hf* xb = reinterpret_cast<hf*>(p.xb);
  const int r4 = lane >> 4, cp = lane & 15;
#pragma unroll
  for (int gq = 0; gq < KT / 2; ++gq) {  // 64 features at a time: 32 token rows x 256 B in this wave's private 8 KB
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + gq * 64 + a * 32 + 8 * q + 4 * g);
        const f32x16& t = acc2[2 * gq + a];
        *reinterpret_cast<f32x4*>(wst + lr * 256 + (((8 * a + 2 * q + g) ^ (lr & 15)) << 4)) =
            f32x4{t[4 * q] + bb[0], t[4 * q + 1] + bb[1], t[4 * q + 2] + bb[2], t[4 * q + 3] + bb[3]};
      }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {  // 4 rows x 256 B per wave-instruction
      const int r = ps * 4 + r4;
      const long row = row0 + r;
      const bool okr = row < p.M;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wst + r * 256 + (cp << 4));
      const long off = row * C + gq * 64 + ((cp ^ (r & 15)) << 2);
      if (okr) {
        *reinterpret_cast<f32x4*>(p.x + off) = v;
        if (xb) *reinterpret_cast<u32x2*>(xb + off) = u32x2{pk2(v[0], v[1]), pk2(v[2], v[3])};
      }
      float sq = okr ? fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3]))) : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (p.ssq_out && okr && cp == 0) p.ssq_out[(long)gq * p.M + row] = sq;
    }
  }
}

template <int C>
int launch_t(const LayerTailP& p, hipStream_t s) {
  using Ring = TRing<C>;
  const size_t smem = (size_t)Ring::NST * Ring::STEP_B + (size_t)p.hidden * 4;
  if (smem > 160 * 1024) return -2;
  static bool attr_set = false;  // (per template instantiation; idempotent)
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_tail_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((layer_tail_kernel<C>), dim3((unsigned)((p.M + 127) / 128)), dim3(256), smem, s, p);
  return (int)hipGetLastError();
}

}  // namespace

bool layer_tail_supported(int C, int hidden) {
  return (C == 512 || C == 256) && hidden % 64 == 0 && hidden >= 128 && hidden <= 4096;
}

int launch_layer_tail(const LayerTailP& p, hipStream_t s) {
  if (p.M <= 0 || !layer_tail_supported(p.C, p.hidden) || !p.x || !p.ao || !p.wfrag) return -2;
  return p.C == 512 ? launch_t<512>(p, s) : launch_t<256>(p, s);
}
