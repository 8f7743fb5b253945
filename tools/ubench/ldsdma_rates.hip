// Micro-benchmark: LDS-DMA (buffer_load ... lds) fill rate from an L2-resident buffer, GEMM-like access patterns.
//   pattern 0: tile = 128 rows x 64 B, rows 1 KB apart (row-major [N][K=512] bf16 operand, BK = 32)
//   pattern 1: tile = 128 rows x 128 B, rows 1 KB apart (BK = 64)
//   pattern 2: tile = contiguous 8 KB            (pre-tiled operand)
//   pattern 3: tile = 128 rows x 64 B, rows 4 KB apart (K = 2048)
// Every workgroup (256 threads) streams `steps` tiles into a 4-deep LDS ring with counted vmcnt; nothing is computed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int PAT>
__global__ __launch_bounds__(256) void k(const char* src, unsigned bytes, int steps, int* sink, int big) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 16384];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, bytes, 0x00020000);
  // each WG works on a different 128-row panel (like different n-tiles), panel = blockIdx % 16
  const unsigned panel = (big ? blockIdx.x : (blockIdx.x % 16)) * 128;
  unsigned voff[4];
  int npc;  // pieces (1 KB per wave) per thread per tile
  if (PAT == 0 || PAT == 3) {
    npc = 2;  // 8 KB tile
    const unsigned stride = PAT == 0 ? 1024 : 4096;
    for (int i = 0; i < 2; ++i) { const int r = (i * 4 + wave) * 16 + lane / 4; voff[i] = (panel + r) * stride + (lane % 4) * 16; }
  } else if (PAT == 1) {
    npc = 4;  // 16 KB tile
    for (int i = 0; i < 4; ++i) { const int r = (i * 4 + wave) * 8 + lane / 8; voff[i] = (panel + r) * 1024 + (lane % 8) * 16; }
  } else {
    npc = 2;
    for (int i = 0; i < 2; ++i) voff[i] = (big ? blockIdx.x * 524288u : (blockIdx.x % 16) * 131072u) + (i * 4 + wave) * 1024 + lane * 16;
  }
  const int kstep = PAT == 1 ? 128 : (PAT == 2 ? 8192 : 64);
  const int wrap = PAT == 2 ? (big ? 64 : 16) : (PAT == 3 ? 64 : (PAT == 1 ? 8 : 16));  // k-steps per row before wrapping
  auto issue = [&](int s) {
    char* st = smem + (s & 3) * 16384 + wave * 1024;
    const int so = (s % wrap) * kstep;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < npc) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(st + i * 4096), 16, voff[i], so, 0, 0);
  };
  issue(0); issue(1); issue(2);
  for (int s = 0; s < steps; ++s) {
    if (PAT == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(s + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[tid] == 123 && steps < 0) sink[0] = 1;
}

template <int PAT>
void run(const char* name, const char* d, unsigned bytes, int* sink, int wgs, int big = 0) {
  const int steps = big ? 64 : 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 0, 0, d, bytes, steps, sink, big);
  if (big) hipMemsetAsync(const_cast<char*>(d) + (1u << 30), 2, 1u << 30, 0);  // flush the 256 MB last-level cache
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 0, 0, d, bytes, steps, sink, big);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tile = PAT == 1 ? 16384.0 : 8192.0;
  const double gb = (double)wgs * (steps + 3) * tile / 1e9;
  printf("%-52s WGs=%5d: %.3f ms  %.2f TB/s  (%.1f GB/s per CU)\n", name, wgs, ms, gb / ms, gb / ms * 1000 / 256);
}

int main() {
  const unsigned bytes = 16u << 20;  // 16 MB working set: L2 / MALL resident
  char* d; hipMalloc(&d, 2ull << 30); hipMemset(d, 1, 2ull << 30);
  int* sink; hipMalloc(&sink, 4);
  for (int wgs : {256, 512, 768}) {
    run<0>("rows 64 B @ 1 KB stride (row-major, BK=32)", d, bytes, sink, wgs);
    run<1>("rows 128 B @ 1 KB stride (row-major, BK=64)", d, bytes, sink, wgs);
    run<3>("rows 64 B @ 4 KB stride (K=2048, BK=32)", d, bytes, sink, wgs);
    run<2>("contiguous 8 KB tiles (pre-tiled)", d, bytes, sink, wgs);
  }
  // HBM-sourced: every workgroup streams its OWN 128-row x 4 KB panel (768 x 512 KB = 393 MB, cache flushed before)
  run<3>("HBM: rows 64 B @ 4 KB stride, own panel per WG", d, 1u << 30, sink, 768, 1);
  run<2>("HBM: contiguous 8 KB tiles, own 512 KB per WG", d, 1u << 30, sink, 768, 1);
  return 0;
}
