// bf16 GEMM for the main transformer layers in BT_PREC_BF16:  C[M,N] = epilogue(A[M,K] . W[N,K]^T), A = bf16
// activations (shadow of the residual stream / attention output / FF hidden), W = bf16 weights.
//
// Engine (differences from gemm2.hip):
//   * 128 x 128 x 32 tiles, 4 waves as 2 x 2 (64 x 64 each = 2 x 2 MFMA 32x32 tiles);
//   * operands go global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, no staging VGPRs) into a
//     3-stage ring (48 KB -> 3 workgroups per CU): two k-steps are in flight while one is multiplied,
//     ONE raw s_barrier per k-step, counted s_waitcnt vmcnt(4) so the prefetch survives the barrier;
//   * LDS rows are 64 B; the 16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3) (the XOR goes
//     on the per-lane SOURCE address, the LDS image stays lane-linear), which makes every ds_read_b128
//     fragment read conflict free without padding;
//   * no LDS staging in the epilogue.  The product is formed TRANSPOSED (C^T = W . A^T) so that a lane owns
//     ONE output row (token) and its registers hold 4-feature runs of that row: RMSNorm factor, RoPE angle,
//     bias runs, residual runs are all lane-local, and stores are 8/16-byte row pieces (bf16 pairs are
//     widened to 16 B with v_permlane32_swap).  Only the V columns of the QKV projection use the normal
//     orientation (lane = feature), which is exactly the V^T fragment layout of attn2.hip.
//   * RMSNorm: the producer of the residual stream (EPI_RESID here, gemm2's fp32 epilogues for
//     frontend.linear) writes per-row partial sums of squares per 64 columns; the consumers (QKV, FF1)
//     add the partials -- no pass over A for the statistics (LDS-DMA data never visits registers).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NST = 3;                        // ring stages
constexpr int OP_BYTES = BM * BK * 2;         // one operand tile (8 KB)
constexpr int ST_BYTES = 2 * OP_BYTES;        // A + W
constexpr unsigned OOB = 0x80000000u;         // voffset of a lane that must read zeros (beyond num_records)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

DEVI unsigned pk2(float a, float b) {
  const bf16x2 t = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, t);
}

// 16 values of one lane (features crow(r, g) of its row) -> two 16-byte stores of 8 consecutive features:
// after the half exchange lane g = 0 holds features 16k .. 16k+7, lane g = 1 features 16k+8 .. 16k+15.
DEVI void store_row_bf16(bf16* row32, const float (&v)[16], int g) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    unsigned x0 = pk2(v[8 * k], v[8 * k + 1]), x1 = pk2(v[8 * k + 2], v[8 * k + 3]);
    unsigned y0 = pk2(v[8 * k + 4], v[8 * k + 5]), y1 = pk2(v[8 * k + 6], v[8 * k + 7]);
    auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    *reinterpret_cast<u32x4*>(row32 + 16 * k + 8 * g) = u32x4{r0[0], r1[0], r0[1], r1[1]};
  }
}

// ABL (development, BT_G3_ABL): bit 0 = no LDS-DMA after the prologue, bit 1 = FF1 epilogue without GELU,
// bit 2 = no fragment reads / MFMAs in the loop (staging + barriers only)
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 3) void gemm3_kernel(const Gemm3P p, int n_tiles, int total_tiles, int per_xcd) {
  __shared__ __attribute__((aligned(16))) char smem[NST * ST_BYTES];
  // XCD-aware tile order: the n-tiles sharing one 128-row A panel run on the same XCD (block b -> XCD b % 8)
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * per_xcd + (bid >> 3);
  if (tile >= total_tiles) return;
  const int m_tile = tile / n_tiles, n_tile = tile - m_tile * n_tiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, lr = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = p.K / BK;
  const int Lv = p.nblk * 32;  // QKV: rows are addressed per sequence, padded to whole 32-token blocks

  // QKV column tile kind: 0 = q, 1 = k (lane = token, RoPE), 2 = v (lane = feature), 3 = gates
  int kind = 0;
  if constexpr (EPI == G3_QKV) kind = n0 < 3 * p.inner ? n0 / p.inner : 3;
  const bool normal = EPI == G3_QKV && kind == 2;

  // ---- staging: per-lane source offsets (bytes) of the two 4 KB pieces of each operand ---------------------
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (unsigned)((long)p.M * p.lda * 2), 0x00020000);
  const rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (unsigned)((long)n_tiles * BN * p.K * 2), 0x00020000);
  unsigned voffA[2], voffW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = i * 64 + wave * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    long row = (long)m0 + r;
    bool ok = row < p.M;
    if constexpr (EPI == G3_QKV) {
      const int seq = (int)(row / Lv), t = (int)(row - (long)seq * Lv);
      ok = seq < p.n_seq && t < p.L;
      row = (long)seq * p.L + t;
    }
    voffA[i] = ok ? (unsigned)(row * p.lda * 2 + c * 16) : OOB;
    voffW[i] = (unsigned)((long)(n0 + r) * p.K * 2 + c * 16);
  }
  auto issue = [&](int kt, int stage) {
    char* st = smem + stage * ST_BYTES + wave * 1024;
    const int so = kt * (BK * 2);
    // (the instruction's immediate offset would also move the LDS address: keep it 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)st, 16, voffA[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(st + 4096), 16, voffA[1], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(st + OP_BYTES), 16, voffW[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(st + OP_BYTES + 4096), 16, voffW[1], so, 0, 0);
  };

  // ---- fragment addresses: P = rows that become accumulator ROWS (registers), Q = rows that become LANES ----
  // transposed (default): P = W tile, Q = A tile  -> lane = token;   normal (V columns): P = A, Q = W.
  const int pofs = (normal ? wm * 64 * 64 : OP_BYTES + wn * 64 * 64) + lr * 64;
  const int qofs = (normal ? OP_BYTES + wn * 64 * 64 : wm * 64 * 64) + lr * 64;
  const int sw = (lr >> 2) & 3;
  const int kc0 = ((0 + g) ^ sw) * 16, kc1 = ((2 + g) ^ sw) * 16;  // k16 step 0 / 1: chunk 2 m + g

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int stage = 0, stage2 = 2;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk && !(ABL & 1)) issue(kt + 2, stage2);
    const char* st = smem + stage * ST_BYTES;
#pragma unroll
    for (int m = 0; m < ((ABL & 4) ? (kt == 0 ? 1 : 0) : 2); ++m) {
      const int kc = m == 0 ? kc0 : kc1;
      bf16x8 fp[2], fq[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fp[a] = *reinterpret_cast<const bf16x8*>(st + pofs + a * 32 * 64 + kc);
#pragma unroll
      for (int b = 0; b < 2; ++b) fq[b] = *reinterpret_cast<const bf16x8*>(st + qofs + b * 32 * 64 + kc);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[a], fq[b], acc[a][b], 0, 0, 0);
    }
    stage = stage == NST - 1 ? 0 : stage + 1;
    stage2 = stage2 == NST - 1 ? 0 : stage2 + 1;
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // token rows of this wave: block j (32 rows), this lane's token = row0 + 32 j + lr  (lane = token view)
  const int row0 = m0 + wm * 64;
  long trow[2];    // real row index (A / x / ssq addressing), -1 if the row does not exist
  int tseq[2], tblk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
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
  float rs[2] = {1.f, 1.f};
  if (p.ssq_in) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
      if (trow[j] >= 0)
        for (int q = 0; q < p.ssq_parts; ++q) s += p.ssq_in[(long)q * p.M + trow[j]];
      rs[j] = trow[j] >= 0 ? sqrtf((float)p.K) / fmaxf(sqrtf(s), 1e-12f) : 0.f;
    }
  }

  if constexpr (EPI == G3_FF1) {
    bf16* out = reinterpret_cast<bf16*>(p.out);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int nb = n0 + wn * 64 + a * 32;  // first feature of this 32-block
      f32x4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * q + 4 * g);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float u = fmaf(acc[a][b][r], rs[b], bq[r >> 2][r & 3]);
          v[r] = (ABL & 2) ? u : gelu_erf(u);
        }
        if (trow[b] >= 0) store_row_bf16(out + trow[b] * p.ldo + nb, v, g);
      }
    }
  } else if constexpr (EPI == G3_RESID) {
    bf16* xb = reinterpret_cast<bf16*>(p.xb);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float ssq = 0.f;
      const bool ok = trow[b] >= 0;
      float* xr = p.x + (ok ? trow[b] : 0) * p.ldx;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int nb = n0 + wn * 64 + a * 32;
        f32x4 xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = ok ? *reinterpret_cast<const f32x4*>(xr + nb + 8 * q + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 bb = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) bb = *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * q + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float o = acc[a][b][4 * q + i] + bb[i] + xv[q][i];
            v[4 * q + i] = o;
            ssq = fmaf(o, o, ssq);
          }
          if (ok) *reinterpret_cast<f32x4*>(xr + nb + 8 * q + 4 * g) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        }
        if (xb && ok) store_row_bf16(xb + trow[b] * p.ldx + nb, v, g);
      }
      ssq += __shfl_xor(ssq, 32);
      if (p.ssq_out && ok && g == 0) p.ssq_out[(long)(n0 / 64 + wn) * p.M + trow[b]] = ssq;
    }
  } else {  // G3_QKV
    if (kind < 2) {  // q / k: RoPE, fragment-major [quarter][token][8 dims]
      bf16* dst = reinterpret_cast<bf16*>(kind == 0 ? p.qf : p.kf);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (tseq[b] >= p.n_seq) continue;  // wave-uniform
        const int pos = tblk[b] * 32 + lr;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int head = (n0 - kind * p.inner + wn * 64 + a * 32) >> 5;
          bf16* blk = dst + (((long)tseq[b] * p.heads + head) * p.nbp + tblk[b]) * 1024;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rope + ((long)pos * 16 + 4 * q + 2 * g) * 2);
            const float e0 = acc[a][b][4 * q] * rs[b], o0 = acc[a][b][4 * q + 1] * rs[b];
            const float e1 = acc[a][b][4 * q + 2] * rs[b], o1 = acc[a][b][4 * q + 3] * rs[b];
            const u32x2 w = {pk2(e0 * cs[0] - o0 * cs[1], o0 * cs[0] + e0 * cs[1]),
                             pk2(e1 * cs[2] - o1 * cs[3], o1 * cs[2] + e1 * cs[3])};
            *reinterpret_cast<u32x2*>(blk + (q * 32 + lr) * 8 + 4 * g) = w;
          }
        }
      }
    } else if (kind == 2) {  // v: lane = feature (dim), registers = tokens -> V^T fragments
      bf16* dst = reinterpret_cast<bf16*>(p.vf);
#pragma unroll
      for (int a = 0; a < 2; ++a) {  // token block a of this wave (accumulator rows)
        if (tseq[a] >= p.n_seq) continue;
        float sk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sk[r] = __shfl(rs[a], crow(r, g));
#pragma unroll
        for (int b = 0; b < 2; ++b) {  // feature block b (lanes)
          const int head = (n0 - 2 * p.inner + wn * 64 + b * 32) >> 5;
          bf16* blk = dst + (((long)tseq[a] * p.heads + head) * p.nbp + tblk[a]) * 1024;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              w[i] = pk2(acc[a][b][8 * s + 2 * i] * sk[8 * s + 2 * i], acc[a][b][8 * s + 2 * i + 1] * sk[8 * s + 2 * i + 1]);
            *reinterpret_cast<u32x4*>(blk + (s * 64 + lane) * 8) = w;
          }
        }
      }
    } else {  // gates: features 3 inner + h
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (tseq[b] >= p.n_seq) continue;
        const int t = tblk[b] * 32 + lr;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int h = n0 - 3 * p.inner + wn * 64 + a * 32 + crow(r, g);
            if (h < p.heads)
              p.gates[((long)tseq[b] * p.heads + h) * p.nbp * 32 + t] = sigmoidf(fmaf(acc[a][b][r], rs[b], p.b_gates[h]));
          }
      }
    }
  }
}

}  // namespace

bool gemm3_supported(const Gemm3P& p) {
  if (p.M <= 0 || p.K % BK != 0 || p.K < 2 * BK || p.lda % 8 != 0) return false;
  if ((long)p.M * p.lda * 2 >= 0x7fffffffL || (long)(p.N + 127) / 128 * 128 * p.K * 2 >= 0x7fffffffL) return false;
  if (p.epi == G3_QKV) return p.inner % 128 == 0 && p.inner == p.heads * 32 && p.L > 0 && p.L <= 1536;
  if (p.epi == G3_FF1) return p.N % 128 == 0 && p.ldo % 8 == 0;
  if (p.epi == G3_RESID) return p.N % 128 == 0 && p.ldx % 8 == 0;
  return false;
}

int launch_gemm3(const Gemm3P& p, hipStream_t s) {
  if (!gemm3_supported(p)) return -2;
  const int n_tiles = (p.N + BN - 1) / BN;
  const long rows = p.epi == G3_QKV ? (long)p.n_seq * p.nblk * 32 : (long)p.M;
  const long m_tiles = (rows + BM - 1) / BM;
  const long total = m_tiles * n_tiles;
  if (total > 0x3fffffffL) return -3;
  long per = (total + 7) / 8;
  per = (per + n_tiles - 1) / n_tiles * n_tiles;
  dim3 grid((unsigned)(per * 8)), block(256);
  switch (p.epi) {
    case G3_FF1: {
      static const int abl = getenv("BT_G3_ABL") ? atoi(getenv("BT_G3_ABL")) : 0;
      if (abl == 1) hipLaunchKernelGGL((gemm3_kernel<G3_FF1, 1>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      else if (abl == 2) hipLaunchKernelGGL((gemm3_kernel<G3_FF1, 2>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      else if (abl == 3) hipLaunchKernelGGL((gemm3_kernel<G3_FF1, 3>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      else if (abl == 4) hipLaunchKernelGGL((gemm3_kernel<G3_FF1, 4>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      else if (abl == 6) hipLaunchKernelGGL((gemm3_kernel<G3_FF1, 6>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      else hipLaunchKernelGGL((gemm3_kernel<G3_FF1>), grid, block, 0, s, p, n_tiles, (int)total, (int)per);
      break;
    }
    case G3_RESID: hipLaunchKernelGGL((gemm3_kernel<G3_RESID>), grid, block, 0, s, p, n_tiles, (int)total, (int)per); break;
    case G3_QKV: hipLaunchKernelGGL((gemm3_kernel<G3_QKV>), grid, block, 0, s, p, n_tiles, (int)total, (int)per); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
