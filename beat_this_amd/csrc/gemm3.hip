// half GEMM of the BT_PREC_HALF forward and hi + lo GEMM of the BT_PREC_F32X3 forward:  C[M,N] = epilogue(A[M,K] . W[N,K]^T),
// A = half activations (shadow of the residual stream / attention output / FF hidden), W = half weights.  Main-layer QKV,
// out-projection, FF1, FF2; frontend.linear; the second and third frontend convolution (implicit-GEMM gather).
//
// X3 (BT_PREC_F32X3, the path that carries the 1e-3 / identical-beats gate): both operands are fp32 values stored as
// INTERLEAVED hi / lo half planes ("hl32": per 32 consecutive k the 32 hi halves, then the 32 lo halves; a = hi + lo to
// 2^-22), so a k-step of 32 is ONE 128-byte row per operand row -- the LDS-DMA ring, the swizzle and the fragment reads
// of the 64-deep half configuration carry it unchanged -- and the product is three MFMAs per fragment pair (lo . hi +
// hi . lo + hi . hi, fp32 accumulation): 3x the matrix work for 2x the operand bytes of the half GEMM, on the same
// engine.  The epilogues keep fp32 results exact (erf GELU, fp32 residual stream) and write activations for the next
// GEMM / the attention as hl32 planes again.
//
// Engine (differences from gemm2.hip):
//   * 128 x 128 x 32 tiles, 4 waves as 2 x 2 (64 x 64 each = 2 x 2 MFMA 32x32 tiles);
//   * operands go global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, no staging VGPRs) into a
//     3-stage ring (48 KB -> 3 workgroups per CU): two k-steps are in flight while one is multiplied,
//     ONE raw s_barrier per k-step, counted s_waitcnt vmcnt(4) so the prefetch survives the barrier;
//   * LDS rows are 64 B; the 16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3) (the XOR goes
//     on the per-lane SOURCE address, the LDS image stays lane-linear), which makes every ds_read_b128
//     fragment read conflict free without padding;
//   * the product is formed TRANSPOSED (C^T = W . A^T) so that a lane owns ONE output row (token) and its registers
//     hold 4-feature runs of that row: RMSNorm factor, RoPE angle, bias runs are all lane-local.  QKV results are
//     stored straight from that layout (fragment-major attention operands: contiguous 512 B / 1 KB per instruction);
//     FF1 / residual results leave through LDS (free after the k-loop) so that every global access covers whole
//     lines.  Only the V columns of the QKV projection use the normal orientation (lane = feature), which is
//     exactly the V^T fragment layout of attn2.hip.
//   * RMSNorm: the producer of the residual stream (EPI_RESID here, gemm2's fp32 epilogues for
//     frontend.linear) writes per-row partial sums of squares per 64 columns; the consumers (QKV, FF1)
//     add the partials -- no pass over A for the statistics (LDS-DMA data never visits registers).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

// Tile configurations.  S: 128 x 128 x 32, 4 waves (2 x 2), 3-stage ring, 48 KB -> 3 workgroups per CU: every
// epilogue, small / ragged shapes.  B: 256 x 256 x 64, 8 waves (2 token x 4 feature, 128 x 64 each), 2 stages,
// 128 KB -> 1 workgroup per CU: half the L2 -> LDS operand traffic per flop (the 128^2 FF GEMMs were bound by
// LDS-DMA fill rate, not MFMA: 84 us with the MFMAs removed vs 95 us with them), and 128-byte LDS rows, i.e. every
// fetched cache line is used whole.  FF1 / RESID only (the V columns of QKV need a square wave tile).
struct G3CfgS { static constexpr int BM = 128, BN = 128, BK = 32, WGM = 2, WGN = 2, NST = 3, OCC = 3; };
struct G3CfgB { static constexpr int BM = 256, BN = 256, BK = 64, WGM = 2, WGN = 4, NST = 2, OCC = 1; };
// T = B with 192 token rows (waves of 96 x 64): picked when it covers M in fewer CU-rounds x rows -- final0's FF2 at
// M = 24000 is 125 x 2 = 250 tiles = one round on 98 % of the CUs instead of 188 tiles on 73 % (72 -> 65 us).
// (A 128^2 2-stage variant at 4 workgroups per CU, 1024 slots so that QKV's 2444 tiles take 3 rounds, was no faster.)
struct G3CfgT { static constexpr int BM = 192, BN = 256, BK = 64, WGM = 2, WGN = 4, NST = 2, OCC = 1; };
// X3: BK counts HALF elements of an hl32 row, i.e. a k-step of 32 (hi | lo = 64 halves = 128 B per row).  SX: 32 KB per
// stage, 2 stages -> 2 workgroups per CU (a third stage would leave one workgroup per CU alone with its epilogue);
// BX / TX: the 256- / 192-row tiles of the long-K residual GEMM, 64 KB per stage.
struct G3CfgSX { static constexpr int BM = 128, BN = 128, BK = 64, WGM = 2, WGN = 2, NST = 2, OCC = 2; };
struct G3CfgBX { static constexpr int BM = 256, BN = 256, BK = 64, WGM = 2, WGN = 4, NST = 2, OCC = 1; };
// (Round 6: the 256 x 256 tiles on k16 steps with four stages -- twice the prefetch distance in 128 KB -- ran FF2 6.8 % SLOWER inside the
// bench step, 3.93 -> 4.20 ms, three alternations on one box: the long-K launches at batch size are not the chain of L2 round trips the
// single-file ones were.)
struct G3CfgTX { static constexpr int BM = 192, BN = 256, BK = 64, WGM = 2, WGN = 4, NST = 2, OCC = 1; };
// MX (round 4): 256 x 128 tiles on k-steps of 16 (half an hl32 group per LDS row: 64 B = [16 hi | 16 lo]), 4 waves of 128 x 64,
// 24 KB per stage, 3 stages -> still TWO workgroups per CU (147 KB), i.e. the second workgroup keeps hiding the epilogue,
// with 3/4 of the L2 -> LDS operand bytes per flop of the 128 x 128 tiles (the stream every 128 x 128 GEMM here is bound by,
// DESIGN.md section 5) and 3/4 of the fragment reads per MFMA (a 128 x 64 wave tile: 12 reads per 24 MFMAs).
struct G3CfgMX { static constexpr int BM = 256, BN = 128, BK = 32, WGM = 2, WGN = 2, NST = 3, OCC = 2; };
// HX (round 5): 64 x 128 tiles (waves of 32 x 64), 24 KB per stage: the residual GEMMs of a
// single-file forward (M = 3000 rows: out-projection and FF2 are 24 x 4 = 96 tiles of 128 x 128 on 256 CUs) get twice the
// workgroups at half the work each.  Same MFMAs on the same operand pieces in the same k order per output element:
// bit-identical to the other configurations (tested).
// Round 6: THREE stages (72 KB, two workgroups per CU -- the grids this configuration is chosen for have at most 512 tiles).  With
// two stages the k-loop of a launch with one workgroup per CU is a chain of L2 round trips: FF2 of a 2-chunk forward (K = 2048) ran
// 64 steps of 0.85 us with 0.16 us of MFMA issue each.  30 s file 1.71 -> 1.58 ms (profiles/r06_ab_small_m_tiles.txt; deeper rings,
// k16 steps and the other tile shapes measured there are no better).
struct G3CfgHX { static constexpr int BM = 64, BN = 128, BK = 64, WGM = 2, WGN = 2, NST = 3, OCC = 2; };

// BT_PREC_F32X3: XCD groups a large weight matrix is split over (1 = off; launch_cfg).  tools/x3_probe.py, M = 24000:
// FF1 (W = 4 MB of hl32) 206 / 189 / 202 us with 1 / 2 / 4 groups, FF2 175 / 165 / 164, frontend.linear and the
// out-projection (W <= 2 MB: below the threshold) unchanged.
#ifndef X3_NSPLIT
#define X3_NSPLIT 2
#endif

namespace {

typedef G3CfgS CfgS;
typedef G3CfgB CfgB;

constexpr unsigned OOB = 0x80000000u;         // voffset of a lane that must read zeros (beyond num_records)

// The big streaming outputs (FF1's hidden activation, the q / k / v fragment blocks of the hi + lo QKV projection) leave with
// the non-temporal hint (global_store ... nt): they are read by the NEXT launch, and written plainly they displace the weight
// matrix -- 2 - 4 MB of hl32, the size of an XCD's L2 -- that every tile of this launch re-reads.  Round 6, same-box A/B, three
// alternations (profiles/r06_ab_ntstore.txt): FF1 4.37 -> 4.20 ms per 66-chunk step (-3.8 %), QKV -0.5 %, FF2 -0.6 %, step
// 29.61 -> 29.47 ms (-0.5 %), joules -0.45 %.  -DBT_NT_STORE=0 builds the plain stores.
#ifndef BT_NT_STORE
#define BT_NT_STORE 1
#endif
#if BT_NT_STORE
#define ST_STREAM(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define ST_STREAM(ptr, val) (*(ptr) = (val))
#endif

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

DEVI unsigned pk2(float a, float b) {
  const hfx2 t = {(hf)a, (hf)b};
  return __builtin_bit_cast(unsigned, t);
}

// 16 values of one lane (features crow(r, g) of its row) -> two 16-byte pieces of 8 consecutive features each:
// after the half exchange lane g = 0 holds features 16k .. 16k+7, lane g = 1 features 16k+8 .. 16k+15 (k = 0, 1).
DEVI void pack_row_hf(const float (&v)[16], u32x4 (&piece)[2]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    unsigned x0 = pk2(v[8 * k], v[8 * k + 1]), x1 = pk2(v[8 * k + 2], v[8 * k + 3]);
    unsigned y0 = pk2(v[8 * k + 4], v[8 * k + 5]), y1 = pk2(v[8 * k + 6], v[8 * k + 7]);
    auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    piece[k] = u32x4{r0[0], r1[0], r0[1], r1[1]};
  }
}

// (a, b) -> packed hi halves and packed lo halves of the hi + lo split: hi = half(v), lo = half(v - hi); amax tracks the
// largest magnitude that went through a split (range guard of BT_PREC_F32X3: a hi part beyond the fp16 range is inf)
DEVI void split2(float a, float b, unsigned& whi, unsigned& wlo, float& amax) {
  split_hl(a, b, whi, wlo);   // (common.h)
  amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
}
// pack_row_hf for both parts of the split
DEVI void pack_row_hl(const float (&v)[16], u32x4 (&hi)[2], u32x4 (&lo)[2], float& amax) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    unsigned xh[2], xl[2], yh[2], yl[2];
    split2(v[8 * k], v[8 * k + 1], xh[0], xl[0], amax);
    split2(v[8 * k + 2], v[8 * k + 3], xh[1], xl[1], amax);
    split2(v[8 * k + 4], v[8 * k + 5], yh[0], yl[0], amax);
    split2(v[8 * k + 6], v[8 * k + 7], yh[1], yl[1], amax);
    auto h0 = __builtin_amdgcn_permlane32_swap(xh[0], yh[0], false, false);
    auto h1 = __builtin_amdgcn_permlane32_swap(xh[1], yh[1], false, false);
    auto l0 = __builtin_amdgcn_permlane32_swap(xl[0], yl[0], false, false);
    auto l1 = __builtin_amdgcn_permlane32_swap(xl[1], yl[1], false, false);
    hi[k] = u32x4{h0[0], h1[0], h0[1], h1[1]};
    lo[k] = u32x4{l0[0], l1[0], l0[1], l1[1]};
  }
}
// pack_row_hl for the hl8 form: hi halves as there; hi bytes / lo bytes of the lane's 8 features 16 k + 8 g .. + 7 after the exchange
DEVI void pack_row_f8(const float (&v)[16], u32x4 (&hi)[2], u32x2 (&h8)[2], u32x2 (&l8)[2], float& amax) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    unsigned xh[2], xl[2], yh[2], yl[2];
    split2(v[8 * k], v[8 * k + 1], xh[0], xl[0], amax);
    split2(v[8 * k + 2], v[8 * k + 3], xh[1], xl[1], amax);
    split2(v[8 * k + 4], v[8 * k + 5], yh[0], yl[0], amax);
    split2(v[8 * k + 6], v[8 * k + 7], yh[1], yl[1], amax);
    const unsigned x8 = pk4_f8(v[8 * k], v[8 * k + 1], v[8 * k + 2], v[8 * k + 3]);
    const unsigned y8 = pk4_f8(v[8 * k + 4], v[8 * k + 5], v[8 * k + 6], v[8 * k + 7]);
    const unsigned xl8 = lo4_f8(v[8 * k], v[8 * k + 1], v[8 * k + 2], v[8 * k + 3], xh[0], xh[1]);
    const unsigned yl8 = lo4_f8(v[8 * k + 4], v[8 * k + 5], v[8 * k + 6], v[8 * k + 7], yh[0], yh[1]);
    auto h0 = __builtin_amdgcn_permlane32_swap(xh[0], yh[0], false, false);
    auto h1 = __builtin_amdgcn_permlane32_swap(xh[1], yh[1], false, false);
    auto b8 = __builtin_amdgcn_permlane32_swap(x8, y8, false, false);
    auto c8 = __builtin_amdgcn_permlane32_swap(xl8, yl8, false, false);
    hi[k] = u32x4{h0[0], h1[0], h0[1], h1[1]};
    h8[k] = u32x2{b8[0], b8[1]};
    l8[k] = u32x2{c8[0], c8[1]};
  }
}
// range guard: one flag word per forward (Gemm3P.status), set when a value beyond the fp16 range went through a split
DEVI void flag_range(int* status, float amax, float limit = 65504.f) {
  if (status && __any(!(amax <= limit)) && (threadIdx.x & 63) == 0) atomicOr(status, 1);
}

// ABL (development, BT_G3_ABL = 8): per-wave timing dump (k-loop, waits, epilogue) read by tools/gemm3_probe.py;
// bits 0 - 2 (no LDS-DMA after the prologue / no GELU / no MFMAs) are ablations that can be instantiated by hand
// X3: 0 = half operands, 1 = hl32 operands (BT_PREC_F32X3), 2 = "hl8" operands (round 5, BASELINE config 5: the cross terms
// of the hi + lo product on ONE block-scaled fp8 MFMA per 32-k step).  An hl8 group of 32 columns is 128 B like an hl32 one:
// [32 hi halves | 32 hi bytes = e4m3(v) | 32 lo bytes = e4m3(2^11 (v - hi))] for a weight, 2^-3 of that in the byte sections of an
// activation (common.h: range); per 32-k step and tile pair
//   acc += 2^-8 [hi bytes(P) | lo bytes(P)] . [lo bytes(Q) | hi bytes(Q)]      v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 = both
//                                                                               cross terms, the 2^-8 in its E8M0 scale operand
//   acc += hi(P) . hi(Q)                                                        two v_mfma_f32_32x32x16_f16
// i.e. 64 + 64 matrix-pipe cycles where the three-term form spends 192 (the scaled fp8 MFMA runs at 2.3 x the fp16 rate:
// tools/ubench/mfma_f8_cross.hip), on the same ring, swizzle, fragment-read count and operand registers.
template <int EPI, typename CFG, int X3 = 0, int ABL = 0>
__global__ __launch_bounds__(64 * CFG::WGM * CFG::WGN, (CFG::OCC * CFG::WGM * CFG::WGN + 3) / 4)
void gemm3_kernel(const Gemm3P p, int n_tiles, int total_tiles, int per_xcd, int nsplit) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, NST = CFG::NST;
  constexpr int NW = CFG::WGM * CFG::WGN, NT = 64 * NW;
  constexpr int TB = BM / CFG::WGM / 32;       // 32-token blocks per wave
  constexpr int FB = BN / CFG::WGN / 32;       // 32-feature blocks per wave
  constexpr int ROWB = BK * 2;                 // bytes per LDS row (BK half elements)
  constexpr int KR = X3 ? BK / 2 : BK;         // k values per k-step (X3: a row is BK / 2 hi + BK / 2 lo halves)
  constexpr bool HL16 = X3 && BK == 32;        // X3 on 64-byte rows: a k-step is HALF an hl32 group (16 hi + 16 lo halves)
  constexpr int EB = X3 ? 4 : 2;               // operand bytes per k value
  constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (4 or 8)
  constexpr int RPI = 64 / CPR;                // rows covered by one wave-instruction (1 KB)
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, ST_BYTES = A_BYTES + W_BYTES;
  constexpr int APC = A_BYTES / (NT * 16), WPC = W_BYTES / (NT * 16);  // LDS-DMA pieces per thread and k-step
  static_assert(EPI != G3_QKV || TB == FB, "the V columns swap the operand roles: square wave tile needed");
  static_assert(BN / CFG::WGN == 64, "ssq partials are per 64 columns = one wave");
  static_assert(!X3 || ROWB == 128 || ROWB == 64, "hl32: a k-step is a whole group (128 B per row) or half of one (64 B)");
  static_assert(X3 != 2 || ROWB == 128, "hl8: a k-step is a whole 32-column group");
  static_assert(NST * ST_BYTES >= NW * 8192, "the epilogues stage 8 KB per wave in the ring's LDS");
  __shared__ __attribute__((aligned(16))) char smem[NST * ST_BYTES];
  // XCD-aware tile order (block b -> XCD b % 8).  nsplit = 1: the n-tiles sharing one A panel run on the same XCD, an XCD
  // walks whole panels -- every XCD streams ALL of W once per panel, which is free while W stays in its 4 MB L2 (half
  // precision) and is not when W is an hl32 matrix of 3 - 4 MB (x3 FF1: 7x the algorithmic fetch bytes, the kernel bound
  // by that stream at ~10 TB/s).  nsplit = 2 / 4: the XCDs form nsplit groups, group g owns the n-tiles
  // [g, g + 1) * n_tiles / nsplit (its slice of W stays in L2), the 8 / nsplit XCDs of a group share out the A panels;
  // an A panel is then fetched by nsplit XCDs instead of one.
  const int bid = blockIdx.x;
  int m_tile, n_tile;
  if (nsplit <= 1) {
    const int tile = (bid & 7) * per_xcd + (bid >> 3);
    if (tile >= total_tiles) return;
    m_tile = tile / n_tiles;
    n_tile = tile - m_tile * n_tiles;
  } else {
    const int xcd = bid & 7, slot = bid >> 3;
    const int g = xcd % nsplit, j = xcd / nsplit, per_m = 8 / nsplit, nt = n_tiles / nsplit;
    const int m_local = slot / nt;
    m_tile = m_local * per_m + j;
    n_tile = g * nt + (slot - m_local * nt);
    if ((long)m_tile * n_tiles >= total_tiles) return;   // (total_tiles = m_tiles * n_tiles)
  }
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave index in an SGPR: uniform index math stays scalar)
  const int g = lane >> 5, lr = lane & 31;
  const int wm = wave / CFG::WGN, wn = wave % CFG::WGN;
  const int nk = p.K / KR;
  auto swz = [](int r) { return ROWB == 64 ? (r >> 2) & 3 : (r >> 1) & 7; };  // chunk XOR of LDS row r
  const int Lv = p.nblk * 32;  // QKV: rows are addressed per sequence, padded to whole 32-token blocks

  // QKV column tile kind: 0 = q, 1 = k (lane = token, RoPE), 2 = v (lane = feature), 3 = gates
  int kind = 0;
  if constexpr (EPI == G3_QKV) kind = n0 < 3 * p.inner ? n0 / p.inner : 3;
  const bool normal = EPI == G3_QKV && kind == 2;

  // ---- staging: per-lane source offsets (bytes) of the two 4 KB pieces of each operand ---------------------
  const unsigned a_bytes = (unsigned)((long)p.M * p.lda * EB), w_bytes = (unsigned)((long)n_tiles * BN * p.K * EB);
  static_assert(APC >= 1 && WPC >= 1, "tile too small for the workgroup");
  // (fixed-size arrays: an array whose size depends on a template parameter, used as an argument of the LDS-DMA
  // builtin, is what makes the HOST pass drop the kernel stub)
  static_assert(APC <= 4 && WPC <= 4, "voffA / voffW hold at most 4 pieces");
  unsigned voffA[4], voffW[4];
  int tapmask[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < APC; ++i) {
    const int r = (i * NW + wave) * RPI + lane / CPR;
    const int c = (lane % CPR) ^ swz(r);
    long row = (long)m0 + r;
    bool ok = row < p.M;
    if constexpr (EPI == G3_QKV) {
      const int seq = (int)(row / Lv), t = (int)(row - (long)seq * Lv);
      ok = seq < p.n_seq && t < p.L;
      row = (long)seq * p.L + t;
    }
#ifdef BT_ABL_HID_WRAP   // development ablation (tools/build_variant.py): see the FF1 store below
    if (EPI == G3_RESID && X3 && p.K >= 1024 && p.conv_C2 == 0) row &= BT_ABL_HID_WRAP - 1;
#endif
    // (HL16: LDS chunks 0, 1 = the hi halves of the step's 16 k values, 2, 3 = their lo halves, 64 B further in the group)
    voffA[i] = ok ? (unsigned)(row * p.lda * EB + (HL16 ? (c >> 1) * 64 + (c & 1) * 16 : c * 16)) : OOB;
    if constexpr (EPI == G3_RESID) {  // conv: which of the three time taps exist for this row
      const int t = p.conv_C2 > 0 ? (int)((row / p.conv_F) % p.conv_T) : 1;
      tapmask[i] = ok ? ((t >= 1 ? 1 : 0) | 2 | (t + 1 < p.conv_T ? 4 : 0)) : 0;
    }
  }
#pragma unroll
  for (int i = 0; i < WPC; ++i) {
    const int r = (i * NW + wave) * RPI + lane / CPR;
    const int c = (lane % CPR) ^ swz(r);
    voffW[i] = (unsigned)((long)(n0 + r) * p.K * EB + (HL16 ? (c >> 1) * 64 + (c & 1) * 16 : c * 16));
  }
  // LDS-DMA of one k-step: APC pieces of the A tile, WPC of the W tile (1 KB per wave-instruction; the pieces of a
  // thread are NW KB apart).  The instruction's immediate offset would also move the LDS address, so it stays 0 and
  // the k offset goes into the scalar offset.
  // conv: the descriptor starts one time step (conv_F rows) BEFORE A so that the tap offset dt * conv_F rows is >= 0
  const bool conv = EPI == G3_RESID && p.conv_C2 > 0;
  const unsigned tap_bytes = conv ? (unsigned)(p.conv_F * p.conv_C2 * EB) : 0u;
  const int c2_shift = conv ? __builtin_ctz((unsigned)p.conv_C2) : 0;
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) - tap_bytes, 0,
                                                      a_bytes + 2 * tap_bytes, 0x00020000);
  const rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);
#define G3_ISSUE(kt, stage)                                                                                          \
  do {                                                                                                                \
    char* st_ = smem + (stage) * ST_BYTES + wave * 1024;                                                              \
    const int so_ = HL16 ? ((kt) >> 1) * 128 + ((kt) & 1) * 32 : (kt) * ROWB;                                         \
    if (conv) {                                                                                                       \
      const int tap_ = ((kt) * KR) >> c2_shift;                                                                       \
      const int kin_ = ((kt) * KR) & (p.conv_C2 - 1);                                                                 \
      const int soa_ = tap_ * (int)tap_bytes + (HL16 ? (kin_ >> 5) * 128 + ((kin_ >> 4) & 1) * 32 : kin_ * EB);      \
      _Pragma("unroll") for (int i_ = 0; i_ < APC; ++i_)                                                              \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(st_ + i_ * NW * 1024), 16,                            \
                                                   ((tapmask[i_] >> tap_) & 1) ? voffA[i_] : OOB, soa_, 0, 0);        \
    } else {                                                                                                          \
      _Pragma("unroll") for (int i_ = 0; i_ < APC; ++i_)                                                              \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(st_ + i_ * NW * 1024), 16, voffA[i_], so_, 0, 0);     \
    }                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < WPC; ++i_)                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(st_ + A_BYTES + i_ * NW * 1024), 16, voffW[i_], so_, 0, 0); \
  } while (0)

  // ---- fragment addresses: P = rows that become accumulator ROWS (registers), Q = rows that become LANES ----
  // transposed (default): P = W tile, Q = A tile  -> lane = token;   normal (V columns): P = A, Q = W.
  constexpr int NP = FB, NQ = TB;  // (equal when `normal` is possible)
  const int pofs = (normal ? wm * (TB * 32) * ROWB : A_BYTES + wn * (FB * 32) * ROWB) + lr * ROWB;
  const int qofs = (normal ? A_BYTES + wn * (FB * 32) * ROWB : wm * (TB * 32) * ROWB) + lr * ROWB;
  const int sw = swz(lr);
  constexpr int MS = ROWB / 32;  // 32-byte (k16) pieces of a row: half = MS MFMA steps; X3 = MS / 2 steps of hi (m) and lo (m + MS / 2)
  int kc[MS];
#pragma unroll
  for (int m = 0; m < MS; ++m) kc[m] = ((2 * m + g) ^ sw) * 16;  // piece m: chunk 2 m + g

  f32x16 acc[NP][NQ];
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NQ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // token rows of this wave: block j (32 rows), this lane's token = row0 + 32 j + lr  (lane = token view)
  const int row0 = m0 + wm * (TB * 32);
  long trow[TB];    // real row index (A / x / ssq addressing), -1 if the row does not exist
  int tseq[TB], tblk[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    const long v = (long)row0 + 32 * j + lr;
    if constexpr (EPI == G3_QKV) {
      const int vb = row0 + 32 * j;  // wave-uniform
      tseq[j] = vb / Lv;
      tblk[j] = (vb - tseq[j] * Lv) >> 5;
      const int t = tblk[j] * 32 + lr;
      trow[j] = (tseq[j] < p.n_seq && t < p.L) ? (long)tseq[j] * p.L + t : -1;
    } else {
      tseq[j] = tblk[j] = 0;
      trow[j] = v < p.M ? v : -1;
    }
  }
  // RMSNorm factors: the partial sums of squares are requested BEFORE the LDS-DMA prologue and consumed right after
  // it behind an explicit vmcnt(0) (ordinary loads and LDS-DMA do not return in order, so no counted wait may separate
  // them): their latency overlaps the first tiles' instead of sitting in the epilogue (~2 k cycles of a 31 k wave life).
  float rs[TB], part[TB][8];
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    rs[j] = 1.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      part[j][q] = (EPI != G3_RESID && p.ssq_in && trow[j] >= 0 && q < p.ssq_parts) ? p.ssq_in[(long)q * p.M + trow[j]] : 0.f;
  }
  // FF1: this lane's bias values (MFMA layout: 4-feature runs 8 q + 4 g of each 32-feature block), requested here for the
  // same reason -- in the epilogue each block's four loads were a memory round trip in front of its GELU
  f32x4 bqv[EPI == G3_FF1 ? FB : 1][4];
  if constexpr (EPI == G3_FF1) {
#pragma unroll
    for (int a = 0; a < FB; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) bqv[a][q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + a * 32 + 8 * q + 4 * g);
  }
  // QKV, q / k tiles: (cos, sin) of this lane's token for its pairs 4 q + 2 g, + 1 (the same for every head), likewise
  f32x4 csv[EPI == G3_QKV ? TB : 1][4];
  if constexpr (EPI == G3_QKV) {
    if (kind < 2) {
#pragma unroll
      for (int b = 0; b < TB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          csv[b][q] = tseq[b] < p.n_seq ? *reinterpret_cast<const f32x4*>(p.rope + ((long)(tblk[b] * 32 + lr) * 16 + 4 * q + 2 * g) * 2)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  constexpr int LPS = APC + WPC;  // LDS-DMA instructions per thread and k-step
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) G3_ISSUE(s0, s0);
  if (EPI == G3_FF1 || EPI == G3_QKV || (EPI != G3_RESID && p.ssq_in)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  if (EPI != G3_RESID && p.ssq_in) {
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += part[j][q];
      for (int q = 8; q < p.ssq_parts; ++q) sum += trow[j] >= 0 ? p.ssq_in[(long)q * p.M + trow[j]] : 0.f;  // D > 512
      rs[j] = trow[j] >= 0 ? sqrtf((float)p.K) / fmaxf(sqrtf(sum), 1e-12f) : 0.f;
    }
  }
  int stage = 0, stage2 = NST - 1;
  long long t_wait = 0, t_bar = 0, t_loop0 = 0;
  if constexpr ((ABL & 8) != 0) t_loop0 = clock64();
  for (int kt = 0; kt < nk; ++kt) {
    long long tq0 = 0, tq1 = 0;
    if constexpr ((ABL & 8) != 0) tq0 = clock64();
    // tile kt has landed in every wave; NST - 2 younger tiles may stay in flight across the barrier
    // lgkmcnt(0): the fragment reads of tile kt - 1 have returned in THIS wave before the barrier -- the stage they came
    // from is refilled right after it (WAR: see WRing::acquire in fused2.hip)
    if (NST > 2 && kt + 1 < nk && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr ((ABL & 8) != 0) tq1 = clock64();
    __builtin_amdgcn_s_barrier();
    if constexpr ((ABL & 8) != 0) { const long long tq2 = clock64(); t_wait += tq1 - tq0; t_bar += tq2 - tq1; }
    if (kt + NST - 1 < nk && !(ABL & 1)) G3_ISSUE(kt + NST - 1, stage2);
    const char* st = smem + stage * ST_BYTES;
    if constexpr (X3 == 2) {
      // chunks of a 128-byte row: 0 .. 3 the hi halves (k16 piece m = chunks 2 m, 2 m + 1), 4, 5 the hi bytes, 6, 7 the lo
      // bytes.  P side: lane half 0 supplies its row's hi bytes, half 1 its lo bytes; Q side the other way round.
      typedef __attribute__((ext_vector_type(8))) int i32x8;
      const int cP = 4 + 2 * g, cQ = 6 - 2 * g;
      const int kP0 = (cP ^ sw) * 16, kP1 = ((cP + 1) ^ sw) * 16, kQ0 = (cQ ^ sw) * 16, kQ1 = ((cQ + 1) ^ sw) * 16;
      {
        i32x8 p8[NP], q8[NQ];
#pragma unroll
        for (int a = 0; a < NP; ++a) {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(st + pofs + a * 32 * ROWB + kP0);
          const u32x4 hi = *reinterpret_cast<const u32x4*>(st + pofs + a * 32 * ROWB + kP1);
          p8[a] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          const u32x4 lo = *reinterpret_cast<const u32x4*>(st + qofs + b * 32 * ROWB + kQ0);
          const u32x4 hi = *reinterpret_cast<const u32x4*>(st + qofs + b * 32 * ROWB + kQ1);
          q8[b] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b)   // scale of the first operand 2^-8 (E8M0 byte 119: activation bytes carry 2^-3, weight lo bytes 2^11; common.h), of the second 1 (127)
            acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(p8[a], q8[b], acc[a][b], 0, 0, 0, HL8_SCALE_WORD, 0, 0x7f7f7f7f);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        hfx8 ph[NP], qh[NQ];
#pragma unroll
        for (int a = 0; a < NP; ++a) ph[a] = *reinterpret_cast<const hfx8*>(st + pofs + a * 32 * ROWB + kc[m]);
#pragma unroll
        for (int b = 0; b < NQ; ++b) qh[b] = *reinterpret_cast<const hfx8*>(st + qofs + b * 32 * ROWB + kc[m]);
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b) acc[a][b] = MFMA32_H(ph[a], qh[b], acc[a][b]);
      }
    } else if constexpr (!X3) {
#pragma unroll
      for (int m = 0; m < ((ABL & 4) ? 0 : MS); ++m) {
        hfx8 fp[NP], fq[NQ];
#pragma unroll
        for (int a = 0; a < NP; ++a) fp[a] = *reinterpret_cast<const hfx8*>(st + pofs + a * 32 * ROWB + kc[m]);
#pragma unroll
        for (int b = 0; b < NQ; ++b) fq[b] = *reinterpret_cast<const hfx8*>(st + qofs + b * 32 * ROWB + kc[m]);
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b) acc[a][b] = MFMA32_H(fp[a], fq[b], acc[a][b]);
      }
    } else {
#pragma unroll
      for (int m = 0; m < ((ABL & 4) ? 0 : MS / 2); ++m) {  // k16 piece m: hi at chunk pair m, lo at chunk pair m + MS / 2
        hfx8 ph[NP], pl[NP], qh[NQ], ql[NQ];
#pragma unroll
        for (int a = 0; a < NP; ++a) {
          ph[a] = *reinterpret_cast<const hfx8*>(st + pofs + a * 32 * ROWB + kc[m]);
          pl[a] = *reinterpret_cast<const hfx8*>(st + pofs + a * 32 * ROWB + kc[m + MS / 2]);
        }
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          qh[b] = *reinterpret_cast<const hfx8*>(st + qofs + b * 32 * ROWB + kc[m]);
          ql[b] = *reinterpret_cast<const hfx8*>(st + qofs + b * 32 * ROWB + kc[m + MS / 2]);
        }
        // three passes over the wave's tiles (small terms first): consecutive MFMAs never share an accumulator
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b) acc[a][b] = MFMA32_H(pl[a], qh[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b) acc[a][b] = MFMA32_H(ph[a], ql[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b) acc[a][b] = MFMA32_H(ph[a], qh[b], acc[a][b]);
      }
    }
    stage = stage == NST - 1 ? 0 : stage + 1;
    stage2 = stage2 == NST - 1 ? 0 : stage2 + 1;
  }

#undef G3_ISSUE
  long long t_loop1 = 0;
  if constexpr ((ABL & 8) != 0) t_loop1 = clock64();
  // ---- epilogue ---------------------------------------------------------------------------------------------
  // FF1 / RESID results leave through LDS (free after the k-loop), one private 8 KB area per wave, one 32-token
  // block at a time: lanes write their token's pieces with the 16-byte chunk index XORed by the row (conflict free),
  // then read the block back ROW-major so that every global access of a wave-instruction covers whole 128-byte lines
  // (8 lanes = one 128 B row piece).  Storing straight from the MFMA layout (lane = token: 32 rows x 32 B at a 1-4 KB
  // stride per instruction, every line written in 4 pieces) ran at ~1.2 TB/s: FF1 took 81 us with its MFMAs removed.
  __syncthreads();
  char* wst = smem + wave * 8192;
  float amax = 0.f;  // (X3) largest magnitude that went through a hi + lo split
  if constexpr (EPI == G3_FF1 && X3) {
    // hl32 hidden activation: a token row of this wave is 64 features = [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63] =
    // 256 B; 16-byte chunk c = 8 a + 4 (lo) + 2 k + g is staged at chunk c ^ (row & 15) like the RESID rows below
    char* out8 = reinterpret_cast<char*>(p.out);
    const int nb0 = n0 + wn * 64;  // first feature of this wave
    const int r4 = lane >> 4, cp = lane & 15;
#pragma unroll
    for (int b = 0; b < TB; ++b) {
#pragma unroll
      for (int a = 0; a < FB; ++a) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = gelu_erf(fmaf(acc[a][b][r], rs[b], bqv[a][r >> 2][r & 3]));
        if (p.x3 & G3_X3_OUT_F8) {   // hl8 hidden activation (the next GEMM runs the fp8 cross terms): [hi halves | hi bytes | lo bytes]
          u32x4 hi[2];
          u32x2 h8[2], l8[2];
          pack_row_f8(v, hi, h8, l8, amax);
#pragma unroll
          for (int k = 0; k < 2; ++k) {   // chunks 8 a + 0 .. 3 the halves, 8 a + 4, 5 the hi bytes (features 16 k ..), 8 a + 6, 7 the lo bytes
            *reinterpret_cast<u32x4*>(wst + lr * 256 + (((8 * a + 2 * k + g) ^ (lr & 15)) << 4)) = hi[k];
            *reinterpret_cast<u32x2*>(wst + lr * 256 + (((8 * a + 4 + k) ^ (lr & 15)) << 4) + 8 * g) = h8[k];
            *reinterpret_cast<u32x2*>(wst + lr * 256 + (((8 * a + 6 + k) ^ (lr & 15)) << 4) + 8 * g) = l8[k];
          }
        } else {
        u32x4 hi[2], lo[2];
        pack_row_hl(v, hi, lo, amax);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          *reinterpret_cast<u32x4*>(wst + lr * 256 + (((8 * a + 2 * k + g) ^ (lr & 15)) << 4)) = hi[k];
          *reinterpret_cast<u32x4*>(wst + lr * 256 + (((8 * a + 4 + 2 * k + g) ^ (lr & 15)) << 4)) = lo[k];
        }
        }
      }
      static_assert(FB == 2, "row = 64 features = 256 B of hl32");
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {  // 4 rows x 256 B per wave-instruction
        const int r = ps * 4 + r4;
        const u32x4 w = *reinterpret_cast<const u32x4*>(wst + r * 256 + (cp << 4));
        const long row = (long)row0 + 32 * b + r;
#ifdef BT_ABL_HID_WRAP
        // Development ablation, results are garbage by construction: the hidden activation of row r lives at row r mod
        // BT_ABL_HID_WRAP (a power of two; 2048 rows = 16.8 MB of hl32: resident in L2 / MALL), FF2 reads it from there -- the
        // forward then runs every MFMA, LDS-DMA and epilogue instruction of the real one but the hidden activation's HBM round
        // trip: an UPPER BOUND on what a fused layer tail that never writes it could save.
        if (row < p.M) *reinterpret_cast<u32x4*>(out8 + ((row & (BT_ABL_HID_WRAP - 1)) * p.ldo + nb0) * 4 + ((cp ^ (r & 15)) << 4)) = w;
#else
        if (row < p.M) ST_STREAM(reinterpret_cast<u32x4*>(out8 + (row * p.ldo + nb0) * 4 + ((cp ^ (r & 15)) << 4)), w);
#endif
      }
    }
  } else if constexpr (EPI == G3_FF1) {
    hf* out = reinterpret_cast<hf*>(p.out);
    const int nb0 = n0 + wn * 64;  // first feature of this wave
#pragma unroll
    for (int b = 0; b < TB; ++b) {
#pragma unroll
      for (int a = 0; a < FB; ++a) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float u = fmaf(acc[a][b][r], rs[b], bqv[a][r >> 2][r & 3]);
          v[r] = (ABL & 2) ? u : gelu_tanh(u);
        }
        u32x4 piece[2];
        pack_row_hf(v, piece);
#pragma unroll
        for (int k = 0; k < 2; ++k)  // 128 B per token row: chunk c = 4 a + 2 k + g, stored at chunk c ^ (row & 7)
          *reinterpret_cast<u32x4*>(wst + lr * 128 + (((4 * a + 2 * k + g) ^ (lr & 7)) << 4)) = piece[k];
      }
      static_assert(FB == 2, "row = 64 features = 128 B");
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {  // 8 rows x 128 B per wave-instruction
        const int r = ps * 8 + (lane >> 3), cp = lane & 7;
        const u32x4 w = *reinterpret_cast<const u32x4*>(wst + r * 128 + (cp << 4));
        const long row = (long)row0 + 32 * b + r;
        if (row < p.M) *reinterpret_cast<u32x4*>(out + row * p.ldo + nb0 + ((cp ^ (r & 7)) << 3)) = w;
      }
    }
  } else if constexpr (EPI == G3_RESID) {
    hf* xb = reinterpret_cast<hf*>(p.xb);
    const int nb0 = n0 + wn * 64;
    if (nb0 < p.N) {  // (N = 64: the first frontend conv -- the upper column half of the tile is weight padding)
    // The x loads of a token block are all requested before the first is used: one at a time (load x, add, store,
    // next pass) they cost a memory round trip EACH -- 16 of them were 31 k of a 122 k-cycle wave life in FF2.
    if (p.bias) {  // bias in the MFMA layout (lane = token, 4-feature runs): 8 small loads, no extra live registers
#pragma unroll
      for (int a = 0; a < FB; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + nb0 + a * 32 + 8 * q + 4 * g);
#pragma unroll
          for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][b][4 * q + i] += bb[i];
        }
    }
    // Row-major phase addressing: lane (r4 = lane >> 4, cp = lane & 15) handles row 32 b + 4 ps + r4 and the 16-byte
    // chunk cp ^ (row & 15) of its 256 B.  ONE per-lane multiply (32-bit element offsets: M * ldx < 2^31 is checked by the
    // launcher); per access a wave-uniform term (32 b + 4 ps) * ldx and one of four column terms are added -- the
    // straightforward `row * ldx` per access cost ~150 quarter-rate integer multiplies / 64-bit ops per wave.
    const int r4 = lane >> 4, cp = lane & 15;
    const unsigned ldx = (unsigned)p.ldx;
    const unsigned off_lane = (unsigned)(row0 + r4) * ldx + (unsigned)nb0;
    unsigned colq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) colq[q] = (unsigned)((cp ^ (4 * q + r4)) << 2);  // (row & 15) = 4 (ps & 3) + r4
    const int rows_left = p.M - row0 - r4;  // row (32 b + 4 ps + r4) exists iff 32 b + 4 ps < rows_left
    if (p.gelu) {  // frontend convs: BatchNorm is folded into W / bias; GELU in the tanh form of the half path, exact for X3
#pragma unroll
      for (int a = 0; a < FB; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = X3 ? gelu_erf(acc[a][b][r]) : gelu_tanh(acc[a][b][r]);
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      f32x4 xv[8];  // (per token block: more at once -- all blocks, or a prefetch of the next one -- spills next to the
                    // accumulators and measured slower)
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const unsigned off = off_lane + (unsigned)(32 * b + 4 * ps) * ldx + colq[ps & 3];
        xv[ps] = (32 * b + 4 * ps < rows_left && !p.no_resid && p.x) ? *reinterpret_cast<const f32x4*>(p.x + off)
                                                               : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int a = 0; a < FB; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)  // 256 B per token row: 16-byte chunk c = 8 a + 2 q + g at chunk c ^ (row & 15)
          *reinterpret_cast<f32x4*>(wst + lr * 256 + (((8 * a + 2 * q + g) ^ (lr & 15)) << 4)) =
              f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {  // 4 rows x 256 B per wave-instruction
        const int r = ps * 4 + r4;
        const unsigned off = off_lane + (unsigned)(32 * b + 4 * ps) * ldx + colq[ps & 3];
        const bool ok = 32 * b + 4 * ps < rows_left;
        f32x4 v = *reinterpret_cast<const f32x4*>(wst + r * 256 + (cp << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += xv[ps][i];
        if (ok) {
          if (p.x) *reinterpret_cast<f32x4*>(p.x + off) = v;
          if constexpr (X3) {
            // hl32 shadow [M, 2 ldx]: features f .. f + 3 (f = nb0 + 4 (cp ^ (row & 15)), inside one 32-block): the hi run
            // at half offset 2 (off - f) + 64 (f / 32) + f % 32 = 2 off - (f & 31), the lo run 32 halves further
            if (xb) {
              const unsigned f = (unsigned)nb0 + colq[ps & 3];
              unsigned wh[2], wl[2];
              split2(v[0], v[1], wh[0], wl[0], amax);
              split2(v[2], v[3], wh[1], wl[1], amax);
              hf* d = xb + (2u * off - (f & 31u));
              *reinterpret_cast<u32x2*>(d) = u32x2{wh[0], wh[1]};
              if (p.x3 & G3_X3_OUT_F8) {   // hl8 shadow: the group's hi bytes start 64 B, its lo bytes 96 B behind its first hi half
                char* gb = reinterpret_cast<char*>(d) - (f & 31u);   // = group base + (f & 31)
                *reinterpret_cast<unsigned*>(gb + 64) = pk4_f8(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<unsigned*>(gb + 96) = lo4_f8(v[0], v[1], v[2], v[3], wh[0], wh[1]);
              } else {
                *reinterpret_cast<u32x2*>(d + 32) = u32x2{wl[0], wl[1]};
              }
            }
          } else {
            if (xb) *reinterpret_cast<u32x2*>(xb + off) = u32x2{pk2(v[0], v[1]), pk2(v[2], v[3])};
          }
        }
        float ssq = ok ? fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3]))) : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o);
        if (p.ssq_out && ok && cp == 0) p.ssq_out[(unsigned)(n0 / 64 + wn) * (unsigned)p.M + (unsigned)(row0 + 32 * b + r)] = ssq;
      }
    }
    }
  } else {  // G3_QKV
    // fragment-major attention operands; X3: a 32-token block is [hi block 2 KB | lo block 2 KB] (attn2.hip)
    constexpr int BLK_E = X3 ? 2048 : 1024;  // half elements per block
    if (kind < 2) {  // q / k: RoPE, fragment-major [quarter][token][8 dims]
      hf* dst = reinterpret_cast<hf*>(kind == 0 ? p.qf : p.kf);
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        if (tseq[b] >= p.n_seq) continue;  // wave-uniform
        const f32x4 (&cs)[4] = csv[b];  // (cos, sin) of this token's pairs 4 q + 2 g, +1: loaded before the k-loop
#pragma unroll
        for (int a = 0; a < FB; ++a) {
          const int head = (n0 - kind * p.inner + wn * 64 + a * 32) >> 5;
          hf* blk = dst + (((long)tseq[b] * p.heads + head) * p.nbp + tblk[b]) * BLK_E;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float e0 = acc[a][b][4 * q] * rs[b], o0 = acc[a][b][4 * q + 1] * rs[b];
            const float e1 = acc[a][b][4 * q + 2] * rs[b], o1 = acc[a][b][4 * q + 3] * rs[b];
            const float r0 = e0 * cs[q][0] - o0 * cs[q][1], r1 = o0 * cs[q][0] + e0 * cs[q][1];
            const float r2 = e1 * cs[q][2] - o1 * cs[q][3], r3 = o1 * cs[q][2] + e1 * cs[q][3];
            if constexpr (X3) {
              unsigned wh[2], wl[2];
              split2(r0, r1, wh[0], wl[0], amax);
              split2(r2, r3, wh[1], wl[1], amax);
              ST_STREAM(reinterpret_cast<u32x2*>(blk + (q * 32 + lr) * 8 + 4 * g), (u32x2{wh[0], wh[1]}));
              ST_STREAM(reinterpret_cast<u32x2*>(blk + 1024 + (q * 32 + lr) * 8 + 4 * g), (u32x2{wl[0], wl[1]}));
            } else {
              *reinterpret_cast<u32x2*>(blk + (q * 32 + lr) * 8 + 4 * g) = u32x2{pk2(r0, r1), pk2(r2, r3)};
            }
          }
        }
      }
    } else if (kind == 2) {  // v: lane = feature (dim), registers = tokens -> V^T fragments
      hf* dst = reinterpret_cast<hf*>(p.vf);
#pragma unroll
      for (int a = 0; a < TB; ++a) {  // token block a of this wave (accumulator rows)
        if (tseq[a] >= p.n_seq) continue;
        float sk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sk[r] = __shfl(rs[a], crow(r, g));
#pragma unroll
        for (int b = 0; b < FB; ++b) {  // feature block b (lanes)
          const int head = (n0 - 2 * p.inner + wn * 64 + b * 32) >> 5;
          hf* blk = dst + (((long)tseq[a] * p.heads + head) * p.nbp + tblk[a]) * BLK_E;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            unsigned w[4], wl[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v0 = acc[a][b][8 * s + 2 * i] * sk[8 * s + 2 * i], v1 = acc[a][b][8 * s + 2 * i + 1] * sk[8 * s + 2 * i + 1];
              if constexpr (X3) split2(v0, v1, w[i], wl[i], amax);
              else w[i] = pk2(v0, v1);
            }
            ST_STREAM(reinterpret_cast<u32x4*>(blk + (s * 64 + lane) * 8), (u32x4{w[0], w[1], w[2], w[3]}));
            if constexpr (X3) ST_STREAM(reinterpret_cast<u32x4*>(blk + 1024 + (s * 64 + lane) * 8), (u32x4{wl[0], wl[1], wl[2], wl[3]}));
          }
        }
      }
    } else {  // gates: features 3 inner + h
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        if (tseq[b] >= p.n_seq) continue;
        const int t = tblk[b] * 32 + lr;
#pragma unroll
        for (int a = 0; a < FB; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int h = n0 - 3 * p.inner + wn * 64 + a * 32 + crow(r, g);
            if (h < p.heads)
              p.gates[((long)tseq[b] * p.heads + h) * p.nbp * 32 + t] = sigmoidf(fmaf(acc[a][b][r], rs[b], p.b_gates[h]));
          }
      }
    }
  }
  if constexpr (X3) flag_range(p.status, amax, (p.x3 & G3_X3_OUT_F8) ? HL8_ACT_MAX : 65504.f);   // (hl8 activations end earlier: common.h)
  if constexpr ((ABL & 8) != 0) {  // development timing dump over the head of the (already written) output
    const long long t_end = clock64();
    __syncthreads();
    if (lane == 0) {
      long long* dbg = reinterpret_cast<long long*>(EPI == G3_RESID ? p.out : (void*)const_cast<float*>(p.ssq_in));
      long long* d = dbg + ((long)blockIdx.x * NW + wave) * 4;
      d[0] = t_loop1 - t_loop0; d[1] = t_wait; d[2] = t_bar; d[3] = t_end - t_loop1;
    }
  }
}

template <int EPI, typename CFG, int X3 = 0, int ABL = 0>
void launch_cfg(const Gemm3P& p, hipStream_t s) {
  const int n_tiles = (p.N + CFG::BN - 1) / CFG::BN;
  const long rows = p.epi == G3_QKV ? (long)p.n_seq * p.nblk * 32 : (long)p.M;
  const long m_tiles = (rows + CFG::BM - 1) / CFG::BM;
  const long total = m_tiles * n_tiles;
  // W split over XCD groups (see the kernel): forced through x3 >> 4 by tools/x3_probe.py, otherwise for an x3 weight matrix
  // beyond ~2.5 MB whose n-tiles divide evenly
  int nsplit = (p.x3 >> 4) & 15;
  if (nsplit == 0) nsplit = (X3 && X3_NSPLIT > 1 && (long)n_tiles * CFG::BN * p.K * 4 > (5L << 19) && n_tiles % X3_NSPLIT == 0) ? X3_NSPLIT : 1;
  if (nsplit != 2 && nsplit != 4 && nsplit != 8) nsplit = 1;
  if (n_tiles % nsplit) nsplit = 1;
  long per;
  if (nsplit == 1) {
    per = (total + 7) / 8;
    per = (per + n_tiles - 1) / n_tiles * n_tiles;
  } else {
    const long per_m = 8 / nsplit;
    per = (m_tiles + per_m - 1) / per_m * (n_tiles / nsplit);
  }
  dim3 grid((unsigned)(per * 8)), block(64 * CFG::WGM * CFG::WGN);
  hipLaunchKernelGGL((gemm3_kernel<EPI, CFG, X3, ABL>), grid, block, 0, s, p, n_tiles, (int)total, (int)per, nsplit);
}

}  // namespace

bool gemm3_supported(const Gemm3P& p) {
  const int eb = p.x3 ? 4 : 2;  // operand bytes per k value (x3: hl32 = a hi and a lo half)
  if (p.M <= 0 || p.K % 64 != 0 || p.K < 128 || p.lda % 8 != 0) return false;
  if ((long)p.M * p.lda * eb >= 0x7fffffffL || (long)(p.N + 255) / 256 * 256 * p.K * eb >= 0x7fffffffL) return false;
  if (p.x3 && BT_HALF_IS_BF16) return false;
  if (p.conv_C2 > 0 && (p.epi != G3_RESID || (p.conv_C2 & (p.conv_C2 - 1)) || p.conv_C2 % 32 != 0 || p.lda != p.conv_C2 ||
                        p.K != 3 * p.conv_C2 || p.conv_F <= 0 || p.conv_T <= 0 || p.M % (p.conv_T * p.conv_F) != 0 || !p.no_resid))
    return false;
  if (p.epi == G3_RESID && !p.x && !p.xb) return false;
  if (p.epi == G3_QKV) return p.inner % 128 == 0 && p.inner == p.heads * 32 && p.L > 0 && (long)p.n_seq * ((p.L + 31) / 32 * 32) < 0x7fffffffL;   // (p.rope must have L rows)
  if (p.epi == G3_FF1) return p.N % 128 == 0 && p.ldo % 8 == 0;
  if (p.epi == G3_RESID) return p.N % 64 == 0 && p.ldx % 8 == 0 && (long)(p.M + 256) * p.ldx * (p.x3 ? 2 : 1) < 0x7fffffffL &&
                                (long)(p.N / 64) * p.M < 0x7fffffffL;  // (32-bit element offsets in the epilogue)
  return false;
}

// BT_PREC_F32X3, FF1 of the main layers on the 256 x 256 tiles (half the L2 -> LDS bytes per flop of the 128 x 128 ones)
#ifndef X3_BIG_FF1
#define X3_BIG_FF1 0
#endif
// BT_PREC_F32X3: the 256 x 128 k16 configuration (G3CfgMX) for FF1 / out-projection / convolutions / frontend.linear
#ifndef X3_MX
#define X3_MX 1
#endif

int launch_gemm3(const Gemm3P& p, hipStream_t s) {
  if (!gemm3_supported(p)) return -2;
#ifdef BT_DEV
  // development builds only: BT_G3_ABL (timing dump), BT_G3_BIG = 0 / 1 forces the tile configuration
  static const int abl = getenv("BT_G3_ABL") ? atoi(getenv("BT_G3_ABL")) : 0;
  static const int force_big = getenv("BT_G3_BIG") ? atoi(getenv("BT_G3_BIG")) : -1;
#else
  constexpr int abl = 0, force_big = -1;
#endif
  // Measured on the final0 shapes (M = 24000): for K = 512 the 256^2 configuration is no faster in isolation (FF1 85 vs
  // 86 us) and slower inside the forward (one workgroup per CU cannot hide its epilogue behind another workgroup's
  // k-loop); for the long-K residual GEMM (FF2, K = 4 D: half the operand traffic per flop, epilogue amortised over 32
  // k-steps) it wins inside the forward as well, 0.452 vs 0.487 ms per step.
  const bool big_ok = p.epi != G3_QKV && p.N % 256 == 0;
  // (x3 = 2 / 3 through the single-operator entry bt_gemm3 forces the 256-row / the 128-row configuration: tools/x3_probe.py)
  const int force = (p.x3 & 15) == 2 ? 1 : (p.x3 & 15) == 3 ? 0 : force_big;
  // (x3 = 4 / 5 force the 256 x 128 k16 / the 64 x 128 tiles further down)
  const bool forced_small = (p.x3 & 15) == 4 || (p.x3 & 15) == 5;
  // BT_PREC_F32X3, round 6: the 256-column tiles only when they fill the chip (>= 160 tiles of 192 rows: M >= 15 k rows at N = 512).  A
  // 4 - 8 chunk single-file forward (M = 6000 - 12000) ran FF2 on 64 - 126 workgroups: 103 - 115 us against 58 - 91 us on the 128 x 128
  // tiles (tools/x3_probe.py B cfgs; profiles/r06_ab_small_m_tiles.txt)
  const long tiles192 = ((long)p.M + 191) / 192 * (p.N / 256);
  const bool fills = !p.x3 || tiles192 >= 160;
  const bool big = big_ok && !forced_small && (force == 1 || (force != 0 && p.epi == G3_RESID && p.K >= 1024 && p.M >= 4096 && fills) ||
                              (force != 0 && p.x3 && X3_BIG_FF1 && p.epi == G3_FF1 && p.M >= 4096));
  // 256 or 192 token rows per tile: fewer (rounds over the 256 CUs) x (rows per tile) wins
  auto cost = [&](int bm) { const long t = ((long)p.M + bm - 1) / bm * (p.N / 256); return (t + 255) / 256 * bm; };
  const bool rows192 = big && p.epi == G3_RESID && force != 1 && cost(192) < cost(256);
  // BT_PREC_F32X3: everything but QKV (square wave tiles) and the long-K residual GEMM (256-column tiles) on the 256 x 128
  // k16 configuration; x3 & 15 = 4 forces it, 3 forces the 128 x 128 tiles (tools/x3_probe.py, tests)
  // (taken when its 256-row tiles still give every CU its two workgroups: a 2-chunk single-file forward stays on the
  // 128 x 128 tiles, whose grid is twice as large)
  const long mx_tiles = ((long)p.M + 255) / 256 * ((p.N + 127) / 128);
  const bool f8 = (p.x3 & G3_X3_F8) != 0;   // A and W are hl8 (the kernel's X3 = 2 form: 128-byte rows only)
  // (round 6: ... and from 20 k rows on.  Below that -- single files of up to ~ 13 chunks -- its grid is a round and a fraction of the
  // 512 slots: FF1 at M = 9000 is 576 tiles, 86 us against 74 us on the 128 x 128 tiles; equal at 12 k and 16.5 k rows)
  const bool mx = p.x3 && !f8 && p.epi != G3_QKV && !big && !rows192 && force != 0 &&
                  ((p.x3 & 15) == 4 || (X3_MX && mx_tiles >= 512 && p.M >= 20000));
  // the 64-row tiles for residual GEMMs whose 128 x 128 grid leaves CUs idle (a single-file forward); x3 & 15 = 5 forces them
  const long sx_tiles = ((long)p.M + 127) / 128 * ((p.N + 127) / 128);
  // (round 6, on their three-stage ring: also the short-K residual GEMMs -- the out-projection -- of every forward below 20 k rows:
  // 19.5 / 20.5 / 30.1 / 34.0 / 47.1 us against 22.7 / 23.4 / 32.6 / 35.5 / 51.1 us on the 128 x 128 tiles at M = 4.5 / 6 / 9 / 12 / 16.5 k)
  const bool hx = p.x3 && p.epi == G3_RESID && !big && !rows192 && !mx &&
                  ((p.x3 & 15) == 5 || ((p.x3 & 15) <= 1 && force_big < 0 && (sx_tiles < 256 || (p.K < 1024 && p.M < 20000))));
  if (p.x3) {
#ifdef BT_DEV
    // development: ablations of the x3 kernel (results are garbage): BT_G3_ABL = 1 no LDS-DMA after the prologue, 4 no MFMAs
    if (abl == 8 && p.epi == G3_FF1) { launch_cfg<G3_FF1, G3CfgSX, true, 8>(p, s); return (int)hipGetLastError(); }
    if (abl == 1 || abl == 4) {
      const bool b = big || rows192;
      if (p.epi == G3_FF1) { if (abl == 1) { if (b) launch_cfg<G3_FF1, G3CfgBX, true, 1>(p, s); else launch_cfg<G3_FF1, G3CfgSX, true, 1>(p, s); }
                             else { if (b) launch_cfg<G3_FF1, G3CfgBX, true, 4>(p, s); else launch_cfg<G3_FF1, G3CfgSX, true, 4>(p, s); } }
      else if (p.epi == G3_RESID) { if (abl == 1) { if (b) launch_cfg<G3_RESID, G3CfgBX, true, 1>(p, s); else launch_cfg<G3_RESID, G3CfgSX, true, 1>(p, s); }
                                    else { if (b) launch_cfg<G3_RESID, G3CfgBX, true, 4>(p, s); else launch_cfg<G3_RESID, G3CfgSX, true, 4>(p, s); } }
      else { if (abl == 1) launch_cfg<G3_QKV, G3CfgSX, true, 1>(p, s); else launch_cfg<G3_QKV, G3CfgSX, true, 4>(p, s); }
      return (int)hipGetLastError();
    }
#endif
    if (f8) {
      if (p.epi == G3_QKV) launch_cfg<G3_QKV, G3CfgSX, 2>(p, s);
      else if (p.epi == G3_FF1) { if (big) launch_cfg<G3_FF1, G3CfgBX, 2>(p, s); else launch_cfg<G3_FF1, G3CfgSX, 2>(p, s); }
      else if (hx) launch_cfg<G3_RESID, G3CfgHX, 2>(p, s);
      else if (rows192) launch_cfg<G3_RESID, G3CfgTX, 2>(p, s);
      else if (big) launch_cfg<G3_RESID, G3CfgBX, 2>(p, s);
      else launch_cfg<G3_RESID, G3CfgSX, 2>(p, s);
      return (int)hipGetLastError();
    }
    switch (p.epi) {
      case G3_FF1:
        if (mx) launch_cfg<G3_FF1, G3CfgMX, true>(p, s);
        else if (big) launch_cfg<G3_FF1, G3CfgBX, true>(p, s); else launch_cfg<G3_FF1, G3CfgSX, true>(p, s);
        break;
      case G3_RESID:
        if (mx) launch_cfg<G3_RESID, G3CfgMX, true>(p, s);
        else if (hx) launch_cfg<G3_RESID, G3CfgHX, true>(p, s);
        else if (rows192) launch_cfg<G3_RESID, G3CfgTX, true>(p, s);
        else if (big) launch_cfg<G3_RESID, G3CfgBX, true>(p, s);
        else launch_cfg<G3_RESID, G3CfgSX, true>(p, s);
        break;
      case G3_QKV: launch_cfg<G3_QKV, G3CfgSX, true>(p, s); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  switch (p.epi) {
    case G3_FF1:
      if (big) launch_cfg<G3_FF1, CfgB>(p, s);
#ifdef BT_DEV
      else if (abl == 8) launch_cfg<G3_FF1, CfgS, false, 8>(p, s);
#endif
      else launch_cfg<G3_FF1, CfgS>(p, s);
      break;
    case G3_RESID:
#ifdef BT_DEV
      if (big && abl == 8) { launch_cfg<G3_RESID, CfgB, false, 8>(p, s); break; }
      if (!big && abl == 8) { launch_cfg<G3_RESID, CfgS, false, 8>(p, s); break; }
#endif
      if (rows192) launch_cfg<G3_RESID, G3CfgT>(p, s);
      else if (big) launch_cfg<G3_RESID, CfgB>(p, s);
      else launch_cfg<G3_RESID, CfgS>(p, s);
      break;
    case G3_QKV: launch_cfg<G3_QKV, CfgS>(p, s); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
