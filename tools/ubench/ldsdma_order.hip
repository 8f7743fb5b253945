// Do LDS-DMA loads of one wave retire in issue order?  (A counted `s_waitcnt vmcnt(N)` in front of a ring stage relies on
// it.)  One workgroup per CU, 4 waves; per round every wave issues K LDS-DMA loads (1 KB each) of K different 1 KB
// blocks of an L2-resident buffer -- the same blocks in every CU at the same time, as the weight rings do -- into K
// cleared LDS slots, waits `vmcnt(K - CHECK)`, and reads the first CHECK slots at once: a slot that still holds the
// cleared pattern was counted as retired before it landed.   SPREAD = 1 puts ~300 cycles of MFMAs between the issues.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int K, int CHECK, int SPREAD>
__global__ __launch_bounds__(256, 1) void k(const unsigned* src, int rounds, int blocks, unsigned* bad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  char* mine = smem + wave * K * 1024;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (unsigned)(blocks * 1024), 0x00020000);
  unsigned nbad = 0;
  f16v acc = {};
  h8 a = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < K; ++i) *reinterpret_cast<u4*>(mine + i * 1024 + lane * 16) = u4{0xdeadbeefu, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int b0 = (r * 37 + wave * K * 5) % (blocks - K * 8);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(mine + i * 1024), 16, lane * 16, (b0 + i * 7) * 1024, 0, 0);
      if (SPREAD) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K - CHECK) : "memory");
    u4 v[CHECK];
#pragma unroll
    for (int i = 0; i < CHECK; ++i) v[i] = *reinterpret_cast<const u4*>(mine + i * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < CHECK; ++i) {
      const unsigned want = (unsigned)((b0 + i * 7) * 256 + lane * 4);  // src[w] = w
      if (v[i][0] != want) ++nbad;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (acc[0] == 12345.f) bad[1] = 1;
}

template <int K, int CHECK, int SPREAD>
void run(const unsigned* d, unsigned* bad, int blocks) {
  hipMemset(bad, 0, 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<K, CHECK, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int rounds = 2000;
  hipLaunchKernelGGL((k<K, CHECK, SPREAD>), dim3(256), dim3(256), 4 * K * 1024, 0, d, rounds, blocks, bad);
  hipDeviceSynchronize();
  unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("K=%2d loads in flight, first %2d checked after vmcnt(%2d), %s: %u stale lane-reads of %ld\n", K, CHECK, K - CHECK,
         SPREAD ? "issues spread between MFMAs" : "burst issue", h[0], 256L * 256 * rounds * CHECK);
}

int main() {
  const int blocks = 4096;  // 4 MB
  unsigned* d; hipMalloc(&d, blocks * 1024);
  unsigned* h = new unsigned[blocks * 256];
  for (int i = 0; i < blocks * 256; ++i) h[i] = (unsigned)i;
  hipMemcpy(d, h, blocks * 1024, hipMemcpyHostToDevice);
  unsigned* bad; hipMalloc(&bad, 8);
  run<16, 16, 0>(d, bad, blocks);   // control: vmcnt(0)
  run<16, 1, 0>(d, bad, blocks);
  run<16, 8, 0>(d, bad, blocks);
  run<16, 15, 0>(d, bad, blocks);
  run<24, 8, 0>(d, bad, blocks);
  run<32, 16, 0>(d, bad, blocks);
  run<16, 1, 1>(d, bad, blocks);
  run<16, 8, 1>(d, bad, blocks);
  run<16, 15, 1>(d, bad, blocks);
  run<24, 8, 1>(d, bad, blocks);
  run<32, 16, 1>(d, bad, blocks);
  return 0;
}
