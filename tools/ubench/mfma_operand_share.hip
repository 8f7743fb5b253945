// Micro-benchmark: does the matrix pipe's energy per MFMA depend on whether CONSECUTIVE MFMAs share an operand register?  Under the
// package power limit the sustained rate of a pure v_mfma_f32_32x32x16_f16 stream is an energy measurement (mfma_power.hip: 2.45
// PFLOP/s on zeros, 1.6 on changing random operands).  Cases, all on random fp16 operands, 2 waves per SIMD, ~0.4 s each:
//   A  every MFMA reads an A and a B operand different from its predecessor's      (what the x3 score / GEMM loops issue today)
//   B  consecutive MFMAs share the A operand (B alternates)                         (hi . hi next to hi . lo of the same row block)
//   C  consecutive MFMAs share both operands (four accumulators)
//   D  as A, with every second operand a "lo part": magnitudes 2^-11 of the others (what the hi + lo split feeds the pipe)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_operand_share.hip -o /tmp/mfma_operand_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 hf;
typedef __attribute__((ext_vector_type(8))) hf hfx8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void k(const hfx8* __restrict__ src, float* out, long long* ticks, int iters) {
  const int tid = threadIdx.x;
  hfx8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(tid * 8 + i) & 4095];
    b[i] = src[(tid * 8 + 4 + i + blockIdx.x) & 4095];
    if (MODE == 3 && (i & 1)) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[i][e] = a[i][e] * (hf)0.00048828125f; b[i][e] = b[i][e] * (hf)0.00048828125f; }
    }
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int ia = MODE == 1 ? u : MODE == 2 ? u : (u + v) & 3;          // B, C: the A operand stays for four MFMAs
        const int ib = MODE == 2 ? u : (u + 3 * v) & 3;                      // C: the B operand too
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[v], 0, 0, 0);
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v][0] *= 0.5f;
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[v][r];
  if (s == 123.456f) out[0] = s;
  if (tid == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, const hfx8* d_src, float* d_out, long long* d_ticks) {
  const int threads = 512, blocks = 256;
  int iters = 20000;
  for (int pass = 0; pass < 2; ++pass) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_src, d_out, d_ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (pass == 0) { iters = (int)(iters * 400.0 / ms); continue; }
    long long h[512];
    hipMemcpy(h, d_ticks, sizeof h, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < 256; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    const double flop = 16.0 * iters * 2.0 * 32 * 32 * 16 * (double)blocks * (threads / 64);
    printf("%-66s %7.1f TFLOP/s, %.3f GHz effective clock\n", name, flop / ms / 1e9, c / w / 10.0);
  }
}

int main() {
  hfx8* d_src; float* d_out; long long* d_ticks;
  hipMalloc(&d_src, 4096 * sizeof(hfx8)); hipMalloc(&d_out, 64); hipMalloc(&d_ticks, 512 * sizeof(long long));
  hfx8 h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i)
    for (int e = 0; e < 8; ++e) h[i][e] = (hf)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  hipMemcpy(d_src, h, sizeof h, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("A  no operand shared between consecutive MFMAs", d_src, d_out, d_ticks);
    run<1>("B  consecutive MFMAs share the A operand", d_src, d_out, d_ticks);
    run<2>("C  consecutive MFMAs share both operands", d_src, d_out, d_ticks);
    run<3>("D  as A, every second operand a lo part (2^-11 magnitudes)", d_src, d_out, d_ticks);
  }
  return 0;
}
