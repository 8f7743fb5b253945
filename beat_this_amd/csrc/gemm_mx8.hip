// BASELINE config 5 as BASELINE.json writes it ("fp8 MFMA weights ... CDNA4 fp8 attention/FFN path"), at OPERATOR level and
// report-only:  C[M,N] (fp32) = A[M,K] . W[N,K]^T  with BOTH operands in the OCP MX e4m3 format -- one byte per
// element, blocks of 32 along k sharing a power-of-two E8M0 scale -- on v_mfma_scale_f32_32x32x64_f8f6f4, which multiplies
// 32 x 32 x 64 per issue at twice the fp16 pipe rate and applies the two block scales itself.  The shapes are the main layers'
// GEMMs (roformer.py:38-61 FeedForward, :99-132 to_qkv / to_out).  Nothing in the forward calls this kernel: it exists so that
// the question "what would an fp8-operand GEMM path buy on this chip at these shapes" is answered by a measurement next to the
// fp16 GEMM of the half path (tools/mx8_probe.py -> profiles/r06_cfg5_mx8.txt) and by the error table of the same arithmetic on
// the oracle (tools/flip_soak.py sim --schemes mxfp8,halfsim), instead of by an estimate.
//
// Engine = gemm3.hip's 128 x 128 configuration with one byte per k value: rows of 64 B = 64 k per k-step (ONE scaled MFMA per
// 32 x 32 tile pair and step), 3-stage LDS-DMA ring of 16 KB stages (three workgroups per CU), the same chunk swizzle, the
// transposed product (lane = token), results leaving through LDS as whole 256-byte row pieces.  Scales: [rows][K / 32] bytes
// in memory (the OCP layout); a lane keeps the scale bytes of its four operand rows for the whole K in registers (K <= 2048:
// 16 dwords per row), pre-shifted by its lane half so that the MFMA's op_sel picks byte 0 (even k-step) or 2 (odd k-step).
#include "common.h"
#include "kernels.h"

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr unsigned OOB = 0x80000000u;

template <int NK>   // k-steps of 64 (K = 64 NK)
__global__ __launch_bounds__(256, NK >= 32 ? 2 : 3) void gemm_mx8_kernel(const GemmMx8P p, int n_tiles, int total_tiles, int per_xcd) {
  constexpr int BM = 128, BN = 128, ROWB = 64, NST = 3;
  constexpr int A_BYTES = BM * ROWB, ST_BYTES = (BM + BN) * ROWB;
  constexpr int SD = (2 * NK + 3) / 4;   // scale dwords per row
  __shared__ __attribute__((aligned(16))) char smem[NST * ST_BYTES > 4 * 8192 ? NST * ST_BYTES : 4 * 8192];
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * per_xcd + (bid >> 3);   // XCD-aware order: the n-tiles of one A panel on one XCD
  if (tile >= total_tiles) return;
  const int m_tile = tile / n_tiles, n_tile = tile - m_tile * n_tiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, lr = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = 64 * NK;

  // ---- staging offsets: two 4 KB pieces per operand and k-step (16 rows x 64 B per wave-instruction) ----------------------
  unsigned voffA[2], voffW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (i * 4 + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    voffA[i] = m0 + r < p.M ? (unsigned)((long)(m0 + r) * K + c * 16) : OOB;
    voffW[i] = (unsigned)((long)(n0 + r) * K + c * 16);
  }
  const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (unsigned)((long)p.M * K), 0x00020000);
  const rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (unsigned)((long)n_tiles * BN * K), 0x00020000);
  auto issue = [&](int kt, int stage) {
    char* st = smem + stage * ST_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(st + i * 4096), 16, voffA[i], kt * ROWB, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(st + A_BYTES + i * 4096), 16, voffW[i], kt * ROWB, 0, 0);
  };

  // ---- block scales of this lane's rows: P = W rows (accumulator rows), Q = A rows (lanes = tokens) ------------------------
  // (requested before the ring starts and waited for with vmcnt(0) behind its prologue: ordinary loads and LDS-DMA do not
  // return in order, gemm3.hip)
  unsigned sp[2][SD], sq[2][SD];
  const unsigned char* SWp = reinterpret_cast<const unsigned char*>(p.SW);
  const unsigned char* SAp = reinterpret_cast<const unsigned char*>(p.SA);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const long prow = n0 + wn * 64 + a * 32 + lr;
    const long qrow = (long)m0 + wm * 64 + a * 32 + lr;
#pragma unroll
    for (int d = 0; d < SD; ++d) {
      sp[a][d] = *reinterpret_cast<const unsigned*>(SWp + prow * (2 * NK) + 4 * d);
      sq[a][d] = qrow < p.M ? *reinterpret_cast<const unsigned*>(SAp + qrow * (2 * NK) + 4 * d) : 0x7f7f7f7fu;
    }
  }
  issue(0, 0);
  if (NK > 1) issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int d = 0; d < SD; ++d) { sp[a][d] >>= 8 * g; sq[a][d] >>= 8 * g; }   // lane half g owns k-block 2 kt + g of a step

  const int pofs = A_BYTES + wn * 64 * ROWB + lr * ROWB, qofs = wm * 64 * ROWB + lr * ROWB;
  const int sw = (lr >> 2) & 3;
  // Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4, measured (tools/mx8_debug.py: a scale set in ONE k-block): the issue is
  // two K = 32 halves; a lane's first four operand registers hold k = 16 g .. 16 g + 15 of the FIRST MX block of the step, its last
  // four the same 16-run of the SECOND block (k = 32 + 16 g ..) -- not 32 consecutive k -- while the scale byte a lane supplies is
  // the one of block g.  So lane half g reads the 16-byte chunks g and 2 + g of its 64-byte row.  (With unit or per-row scales any
  // k labelling the two operands share gives the right product, which is why the hl8 cross-term kernel of gemm3.hip never noticed.)
  const int c0 = (g ^ sw) * 16, c1 = ((2 + g) ^ sw) * 16;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    // tile kt has landed in every wave; one younger tile may stay in flight across the barrier (4 LDS-DMA instructions per step)
    if (kt + 1 < NK) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < NK) issue(kt + 2, (kt + 2) % NST);
    const char* st = smem + (kt % NST) * ST_BYTES;
    i32x8 fp[2], fq[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(st + pofs + a * 32 * ROWB + c0);
      const u32x4 hi = *reinterpret_cast<const u32x4*>(st + pofs + a * 32 * ROWB + c1);
      fp[a] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      const u32x4 ql = *reinterpret_cast<const u32x4*>(st + qofs + a * 32 * ROWB + c0);
      const u32x4 qh = *reinterpret_cast<const u32x4*>(st + qofs + a * 32 * ROWB + c1);
      fq[a] = i32x8{(int)ql[0], (int)ql[1], (int)ql[2], (int)ql[3], (int)qh[0], (int)qh[1], (int)qh[2], (int)qh[3]};
    }
    const int d = kt >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (kt & 1)   // (op_sel is an immediate: byte 2 of the pre-shifted dword on odd steps, byte 0 on even ones)
          acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp[a], fq[b], acc[a][b], 0, 0, 2, (int)sp[a][d], 2, (int)sq[b][d]);
        else
          acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp[a], fq[b], acc[a][b], 0, 0, 0, (int)sp[a][d], 0, (int)sq[b][d]);
      }
    // (the loop is unrolled so that op_sel and the scale dword are immediates.  The MFMA intrinsic is a pure function to LLVM:
    // without a use per step it SINKS the chains of the second token block below the epilogue of the first -- all their
    // fragments stay live, 255 VGPRs / 100 spills.  The empty asm statements are that use.)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: fp32 rows through LDS, whole 256-byte pieces per wave-instruction (gemm3.hip, RESID epilogue) --------------
  __syncthreads();
  char* wst = smem + wave * 8192;
  const int r4 = lane >> 4, cp = lane & 15;
  const int nb0 = n0 + wn * 64, row0 = m0 + wm * 64;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(wst + lr * 256 + (((8 * a + 2 * q + g) ^ (lr & 15)) << 4)) =
            f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int r = ps * 4 + r4;
      const long row = (long)row0 + 32 * b + r;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wst + r * 256 + (cp << 4));
      if (row < p.M && nb0 < p.N) *reinterpret_cast<f32x4*>(p.out + row * p.ldo + nb0 + ((cp ^ (r & 15)) << 2)) = v;
    }
  }
}

template <int NK>
void launch_nk(const GemmMx8P& p, hipStream_t s) {
  const int n_tiles = (p.N + 127) / 128;
  const long m_tiles = ((long)p.M + 127) / 128;
  const long total = m_tiles * n_tiles;
  long per = (total + 7) / 8;
  per = (per + n_tiles - 1) / n_tiles * n_tiles;
  hipLaunchKernelGGL((gemm_mx8_kernel<NK>), dim3((unsigned)(per * 8)), dim3(256), 0, s, p, n_tiles, (int)total, (int)per);
}

}  // namespace

bool gemm_mx8_supported(const GemmMx8P& p) {
  return p.M > 0 && p.N > 0 && p.N % 64 == 0 && (p.K == 512 || p.K == 1024 || p.K == 2048) && p.ldo % 4 == 0 &&
         (long)p.M * p.K < 0x7fffffffL && (long)(p.N + 127) / 128 * 128 * p.K < 0x7fffffffL;
}

int launch_gemm_mx8(const GemmMx8P& p, hipStream_t s) {
  if (!gemm_mx8_supported(p) || !p.A || !p.SA || !p.W || !p.SW || !p.out) return -2;
  switch (p.K) {
    case 512: launch_nk<8>(p, s); break;
    case 1024: launch_nk<16>(p, s); break;
    case 2048: launch_nk<32>(p, s); break;
    default: return -2;
  }
  return (int)hipGetLastError();
}
