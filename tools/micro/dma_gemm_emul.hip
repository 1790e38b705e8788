// The persistent NT GEMM's operand stream WITHOUT the arithmetic: is the K loop bound by the latency of its LDS-DMA pieces?
// 256 blocks x 8 waves, block -> (m tile, n tile) of a (M = 32768, N = 1280) problem exactly as the kernel deals them (XCD x owns 32
// consecutive tiles: 6.4 A panels x 5 B panels per XCD), K tiles of 64: per K tile a block fetches 256 rows x 128 B of A and of B =
// 64 pieces of 1 KiB, 8 per wave, two per "phase" followed by s_waitcnt vmcnt(8) and a barrier — 64 KiB in flight per CU, as in the
// kernel (the LDS holds two stages). PF > 0: one lane group of every wave also touches the lines of K tile t + PF with plain
// 4-byte loads whose results are discarded (an L2 prefetch: does a longer lookahead than the LDS allows help?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PF>
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, int K, int nk, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, wg = xcd * 32 + idx;
  const int mt = wg / 5, nt = wg % 5;
  const unsigned short* a0 = A + (size_t)(mt * 256) * K;
  const unsigned short* b0 = B + (size_t)(nt * 256) * K;
  // piece (wave, j): rows 32 j' ... : A pieces j = 0..3 -> rows (wave * 4 + j) * 8 + (lane >> 3), chunk lane & 7; B likewise
  float acc = 0.f;
  for (int t = 0; t < nk; ++t) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const unsigned short* base = ph < 2 ? a0 : b0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = (ph & 1) * 2 + i;
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const unsigned short* p = base + (size_t)row * K + t * 64 + (lane & 7) * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(smem + ((t & 1) * 64 + ph * 16 + wave * 2 + i) * 1024), 16, 0, 0);
      }
      if (PF > 0 && ph == 3 && t + PF < nk) {
        // 64 lanes x one line each: rows wave * 32 + (lane >> 1), A for even lanes, B for odd
        const int row = wave * 32 + (lane >> 1);
        const unsigned short* q = ((lane & 1) ? b0 : a0) + (size_t)row * K + (t + PF) * 64;
        acc += (float)__builtin_nontemporal_load((const unsigned*)q);
      }
      if (PF > 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // (the prefetch load sits in the same in-order counter)
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 12345.f) sink[0] = acc;
}
int main() {
  const int M = 32768, N = 1280, K = 20480;
  unsigned short *A, *B;
  float* sink;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 4);
  hipMemset(A, 1, (size_t)M * K * 2); hipMemset(B, 1, (size_t)N * K * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nk = K / 64;
#define RUN(P) hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<P>, dim3(256), dim3(512), 131072, 0, A, B, K, nk, sink)
  for (int rnd = 0; rnd < 2; ++rnd) for (int pf : {0, 2, 4, 8, 16}) {
    hipEventRecord(e0);
    switch (pf) { case 0: RUN(0); break; case 2: RUN(2); break; case 4: RUN(4); break; case 8: RUN(8); break; default: RUN(16); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("prefetch %2d K tiles ahead: %.3f ms, %.3f us per K tile (the GEMM: 1.5-2.5), %.1f B/clk/CU at 2.1 GHz\n", pf, ms, ms * 1e3 / nk, 65536.0 / (ms * 1e-3 / nk) / 2.1e9);
  }
  return 0;
}
