// Decomposition of the fused tail's step time (follow-up to tail_stream.hip): one workgroup per CU, 4 waves, 2 x 64 KB ring,
// 74 steps, 64 MFMAs per wave and step.  Template switches:
//   DMA   0 none   1 burst behind the barrier (tail.hip)   2 staggered: wave w issues its 16 instructions one per MFMA in
//         slots [16w, 16w+16)   3 spread: every wave one instruction every 4th MFMA   4 one wave issues all 64
//   READS 0 MFMA operands from registers   1 one ds_read_b128 per MFMA (tail.hip)
//   MF    0 no MFMAs (reads are summed on the VALU)   1 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int DMA, int READS, int MF>
__global__ __launch_bounds__(256, 1) void k(const char* src, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  constexpr int STEP_B = 65536;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (unsigned)(steps * STEP_B), 0x00020000);
  // instruction j (0..63) of a step covers bytes [j KB, j KB + 1 KB)
  auto one = [&](int s, int j) {
    if (s >= steps) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (s & 1) * STEP_B + j * 1024), 16, lane * 16, s * STEP_B + j * 1024, 0, 0);
  };
  if (DMA) for (int j = 0; j < 64; ++j) if (DMA == 4 ? wave == 0 : (j & 3) == wave) one(0, j);
  f16v acc[4] = {};
  h8 b = {1, 1, 1, 1, 1, 1, 1, 1}, wreg = {1, 2, 3, 4, 5, 6, 7, 8};
  float vs = 0;
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (DMA == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) one(s + 1, i * 4 + wave);
    }
    if (DMA == 4 && wave == 0) {
#pragma unroll
      for (int i = 0; i < 64; ++i) one(s + 1, i);
    }
    const char* st = smem + (s & 1) * STEP_B;
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (DMA == 2 && (m >> 4) == wave) one(s + 1, (m & 15) * 4 + wave);
      if (DMA == 3 && (m & 3) == 0) one(s + 1, (m >> 2) * 4 + wave);
      h8 w = wreg;
      if (READS) w = *reinterpret_cast<const h8*>(st + m * 1024 + lane * 16);
      if (MF) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b, acc[m & 3], 0, 0, 0);
      else vs += (float)w[0];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = vs;
  for (int i = 0; i < 4; ++i) t += acc[i][0];
  if (t == 12345.f) sink[0] = t;
}

template <int DMA, int READS, int MF>
void run(const char* name, const char* d, float* sink, int wgs) {
  const int steps = 74;
  const size_t smem = 2 * 65536 + 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DMA, READS, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<DMA, READS, MF>), dim3(wgs), dim3(256), smem, 0, d, steps, sink);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<DMA, READS, MF>), dim3(wgs), dim3(256), smem, 0, d, steps, sink);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-70s WGs=%4d: %7.1f us  (%5.0f ns per step)\n", name, wgs, ms * 1e3, ms * 1e6 / steps);
}

int main() {
  char* d; hipMalloc(&d, 8u << 20); hipMemset(d, 0, 8u << 20);
  float* sink; hipMalloc(&sink, 4);
  for (int wgs : {256, 188}) {
    run<0, 0, 1>("MFMA only (register operands)", d, sink, wgs);
    run<0, 1, 1>("MFMA + ds_read_b128 per MFMA, no DMA", d, sink, wgs);
    run<0, 1, 0>("ds_read only (VALU consume), no DMA", d, sink, wgs);
    run<1, 0, 0>("DMA burst only", d, sink, wgs);
    run<4, 0, 0>("DMA by one wave only", d, sink, wgs);
    run<1, 1, 0>("DMA burst + ds_reads, no MFMA", d, sink, wgs);
    run<1, 0, 1>("DMA burst + MFMA (register operands)", d, sink, wgs);
    run<1, 1, 1>("DMA burst + ds_reads + MFMA   (tail.hip)", d, sink, wgs);
    run<2, 1, 1>("DMA staggered by wave + ds_reads + MFMA", d, sink, wgs);
    run<3, 1, 1>("DMA spread (all waves same slots) + ds_reads + MFMA", d, sink, wgs);
    run<4, 1, 1>("DMA by one wave + ds_reads + MFMA", d, sink, wgs);
    run<2, 0, 1>("DMA staggered + MFMA (register operands)", d, sink, wgs);
  }
  return 0;
}
