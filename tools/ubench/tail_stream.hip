// Micro-benchmark behind DESIGN.md's "the fused tail is bound by its weight stream": one workgroup per CU (136 KB of LDS),
// 4 waves, each workgroup streams the same L2-resident 4.7 MB buffer through an LDS ring by LDS-DMA, one barrier per
// step, optionally with the tail kernel's 64 MFMAs per wave and step in between.
//   order 0: every workgroup walks the buffer in the same order (what tail.hip does)
//   order 1: workgroup b starts at step (b * 29) % steps (rotation: CUs hit different lines at any time)
//   ring  0: 2 stages x 64 KB (tail.hip, C = 512)      ring 1: 4 stages x 32 KB (same bytes, 3 steps in flight)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int STEP_KB, int NST, int ORDER, int MFMAS>
__global__ __launch_bounds__(256, 1) void k(const char* src, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  constexpr int STEP_B = STEP_KB * 1024, CH = STEP_B / 4096;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (unsigned)(steps * STEP_B), 0x00020000);
  const int rot = ORDER ? (blockIdx.x * 29) % steps : 0;
  auto issue = [&](int s) {
    if (s >= steps) return;
    char* st = smem + (s % NST) * STEP_B + wave * 1024;
    const int so = ((s + rot) % steps) * STEP_B;
#pragma unroll
    for (int i = 0; i < CH; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(st + i * 4096), 16, tid * 16, so + i * 4096, 0, 0);
  };
  for (int s = 0; s < NST - 1; ++s) issue(s);
  f16v acc[4] = {};
  h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int s = 0; s < steps; ++s) {
    if (NST == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * CH) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(s + NST - 1);
    const char* st = smem + (s % NST) * STEP_B;
#pragma unroll 4
    for (int m = 0; m < MFMAS; ++m) {
      const h8 w = *reinterpret_cast<const h8*>(st + ((m * 1024 + lane * 16) & (STEP_B - 1)));
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b, acc[m & 3], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i][0];
  if (t == 12345.f) sink[0] = t + a[0];
}

// register-staged variant: the next step's 64 KB come in by global_load_dwordx4 -> VGPR -> ds_write_b128, PAIRS pieces
// (1 KB per wave each) loaded at one tile iteration and written to LDS at the next, between the MFMAs
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MFMAS, int PAIRS>
__global__ __launch_bounds__(256, 1) void kreg(const char* src, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int STEP_B = 65536, CH = 16;
  f16v acc[4] = {};
  h8 b = {1, 1, 1, 1, 1, 1, 1, 1};
  // prologue: step 0
  for (int i = 0; i < CH; ++i) *reinterpret_cast<f4*>(smem + i * 4096 + tid * 16) = *reinterpret_cast<const f4*>(src + i * 4096 + tid * 16);
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* st = smem + (s & 1) * STEP_B;
    char* nx = smem + ((s + 1) & 1) * STEP_B + tid * 16;
    const char* gs = src + (size_t)((s + 1) % steps) * STEP_B + tid * 16;
    f4 r[PAIRS];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t % PAIRS == 0) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) r[q] = *reinterpret_cast<const f4*>(gs + (t + q) * 4096);
      }
#pragma unroll
      for (int m = 0; m < MFMAS / 16; ++m) {
        const h8 w = *reinterpret_cast<const h8*>(st + (((t * (MFMAS / 16) + m) * 1024 + lane * 16) & (STEP_B - 1)));
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b, acc[m & 3], 0, 0, 0);
      }
      if (t % PAIRS == PAIRS - 1) {
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) *reinterpret_cast<f4*>(nx + (t - (PAIRS - 1) + q) * 4096) = r[q];
      }
    }
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i][0];
  if (t == 12345.f) sink[0] = t;
}
template <int MFMAS, int PAIRS>
void runreg(const char* name, const char* d, float* sink, int wgs) {
  const int steps = 74;
  const size_t smem = 2 * 65536 + 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kreg<MFMAS, PAIRS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kreg<MFMAS, PAIRS>), dim3(wgs), dim3(256), smem, 0, d, steps, sink);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kreg<MFMAS, PAIRS>), dim3(wgs), dim3(256), smem, 0, d, steps, sink);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double gb = (double)wgs * steps * 65536.0 / 1e9;
  printf("%-64s WGs=%4d: %7.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", name, wgs, ms * 1e3, gb / ms, gb / ms * 1000 / (wgs < 256 ? wgs : 256));
}

template <int STEP_KB, int NST, int ORDER, int MFMAS>
void run(const char* name, const char* d, float* sink, int wgs) {
  const int steps = 4736 / STEP_KB;  // 4.7 MB
  const size_t smem = (size_t)NST * STEP_KB * 1024 + 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<STEP_KB, NST, ORDER, MFMAS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<STEP_KB, NST, ORDER, MFMAS>), dim3(wgs), dim3(256), smem, 0, d, steps, sink);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<STEP_KB, NST, ORDER, MFMAS>), dim3(wgs), dim3(256), smem, 0, d, steps, sink);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double gb = (double)wgs * steps * STEP_KB * 1024.0 / 1e9;
  printf("%-64s WGs=%4d: %7.1f us  %6.2f TB/s  %6.1f GB/s per CU   (MFMA-only time of %d per step: %.1f us at 2.4 GHz)\n", name, wgs,
         ms * 1e3, gb / ms, gb / ms * 1000 / (wgs < 256 ? wgs : 256), MFMAS, steps * MFMAS * 32 / 2.4e3);
}

int main() {
  char* d; hipMalloc(&d, 8u << 20); hipMemset(d, 0, 8u << 20);
  float* sink; hipMalloc(&sink, 4);
  for (int wgs : {188, 256}) {
    run<64, 2, 0, 0>("2 x 64 KB, same order, no compute", d, sink, wgs);
    run<64, 2, 1, 0>("2 x 64 KB, rotated,    no compute", d, sink, wgs);
    run<32, 4, 0, 0>("4 x 32 KB, same order, no compute", d, sink, wgs);
    run<32, 4, 1, 0>("4 x 32 KB, rotated,    no compute", d, sink, wgs);
    run<64, 2, 0, 64>("2 x 64 KB, same order, 64 MFMAs / wave / step", d, sink, wgs);
    run<64, 2, 1, 64>("2 x 64 KB, rotated,    64 MFMAs / wave / step", d, sink, wgs);
    run<32, 4, 0, 32>("4 x 32 KB, same order, 32 MFMAs / wave / step", d, sink, wgs);
    run<32, 4, 1, 32>("4 x 32 KB, rotated,    32 MFMAs / wave / step", d, sink, wgs);
    runreg<0, 2>("register-staged 2 x 64 KB (2 pieces / iteration), no compute", d, sink, wgs);
    runreg<64, 1>("register-staged 2 x 64 KB (1 piece / iteration), 64 MFMAs", d, sink, wgs);
    runreg<64, 2>("register-staged 2 x 64 KB (2 pieces / 2 iterations), 64 MFMAs", d, sink, wgs);
    runreg<64, 4>("register-staged 2 x 64 KB (4 pieces / 4 iterations), 64 MFMAs", d, sink, wgs);
  }
  return 0;
}
