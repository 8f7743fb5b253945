// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU / MFMA instructions the
// attention kernels are made of, at 1 / 2 / 4 waves per SIMD, alone and mixed with MFMA in other waves.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ void k(long long* out, int iters, int mix_mfma_waves) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0f + a0 * 1e-9f;
  f32x16 acc = {0}, acc2 = {0};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(a0 * 1e-3f); fb[i] = (__bf16)1.0f; }
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = OP == 100 || wave < mix_mfma_waves;
  __syncthreads();
  long long t0 = clock64();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (OP == 0) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
      if (OP == 1) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
      if (OP == 2) { REP16(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4)); asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4));) }
      if (OP == 3) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %4, %5\n v_cvt_pk_bf16_f32 %2, %4, %5\n v_cvt_pk_bf16_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));) }
      if (OP == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
      if (OP == 5) { REP16(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));) }
      if (OP == 6) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4)); asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4));) }
      if (OP == 7) { REP16(asm volatile("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
      if (OP == 8) { REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
      if (OP == 9) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
    }
  }
  long long t1 = clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0;
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  if (s == 123.456f) out[1000] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd, int mix, long long* d_out) {
  const int threads = 256 * waves_per_simd;  // one block per CU: waves_per_simd waves on each of the 4 SIMDs
  const int iters = 200;
  const int iters_l = iters * 20;   // ~ms-long launches so the clock settles; wall clock via HIP events
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d_out, iters_l, mix);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d_out, iters_l, mix);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  {
    std::vector<long long> hh(256 * 16);
    hipMemcpy(hh.data(), d_out, hh.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) mx = hh[b * 16 + w] > mx ? hh[b * 16 + w] : mx;
    printf("   [%s w=%d mix=%d] wall %.3f ms, max wave ticks %lld => %.3f GHz tick rate\n", name, waves_per_simd, mix, ms, mx, mx / (ms * 1e6));
  }
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d_out, iters, mix);
  hipDeviceSynchronize();
  std::vector<long long> h(256 * 16);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  const int nw = threads / 64;
  double tv = 0, tm = 0; int nv = 0, nm = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) {
    if (OP == 100 || w < mix) { tm += h[b * 16 + w]; ++nm; } else { tv += h[b * 16 + w]; ++nv; }
  }
  // instructions per wave: VALU 64 per iter (pk variants: 64 too), MFMA 16 per iter
  if (nv) printf("%-28s waves/SIMD=%d mix=%d  VALU wave: %.2f clk/instr (wave-level), => %.2f clk/instr/SIMD\n", name, waves_per_simd, mix,
                 tv / nv / (iters * 64.0), tv / nv / (iters * 64.0) / ((nw - mix) / 4.0 > 1 ? (nw - mix) / 4.0 : 1.0));
  if (nm) printf("%-28s waves/SIMD=%d mix=%d  MFMA wave: %.2f clk/mfma (wave-level)\n", name, waves_per_simd, mix, tm / nm / (iters * 16.0));
}

int main() {
  long long* d_out;
  hipMalloc(&d_out, 8 * 4096);
  for (int w : {1, 2, 4}) {
    run<0>("v_exp_f32", w, 0, d_out);
    run<1>("v_add_f32", w, 0, d_out);
    run<2>("v_pk_add_f32", w, 0, d_out);
    run<3>("v_cvt_pk_bf16_f32", w, 0, d_out);
    run<4>("v_fma_f32", w, 0, d_out);
    run<5>("v_max3_f32", w, 0, d_out);
    run<6>("v_pk_fma_f32", w, 0, d_out);
    run<7>("exp+add alternating", w, 0, d_out);
    run<8>("v_exp_f16", w, 0, d_out);
    run<9>("v_mul_f32", w, 0, d_out);
    run<100>("mfma_32x32x16_bf16", w, 0, d_out);
  }
  // mixes: 8 waves per CU (2 per SIMD): 4 MFMA waves (one per SIMD) + 4 VALU waves
  run<0>("v_exp_f32 | mfma", 2, 4, d_out);
  run<1>("v_add_f32 | mfma", 2, 4, d_out);
  run<3>("v_cvt_pk | mfma", 2, 4, d_out);
  run<0>("v_exp_f32 x3 | mfma x1", 4, 4, d_out);
  run<1>("v_add_f32 x3 | mfma x1", 4, 4, d_out);
  return 0;
}
