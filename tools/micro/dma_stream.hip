// LDS-DMA in the K loop's own regime: every wave issues 2 pieces per step and waits with vmcnt(8) (64 KiB in flight per CU, as the
// persistent GEMMs keep), no compute. Block b streams a 64-row x 512-byte window down a [K][ld] bf16 matrix (its own column block:
// the TN kernel's B operand; ld = 34560 or 1280), 512 steps. Piece shape 0: 4 rows x 256 B (TN today), 1: 8 rows x 128 B (NT's).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ src, int ld, int ncolblk, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* base = src + (size_t)(blockIdx.x % ncolblk) * 256;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row, colb;  // row of the 64-row window, byte offset inside the 512-byte window row
      if (SHAPE == 0) { row = 32 * i + 4 * wave + (lane >> 4); colb = (lane & 15) * 16 + 256 * (s & 1); }
      else { row = 8 * wave + (lane >> 3); colb = i * 128 + (lane & 7) * 16 + 256 * (s & 1); }
      const unsigned short* p = base + (size_t)(s * 32 + row % 64) * ld + colb / 2;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + ((s & 3) * 16 + wave * 2 + i) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main() {
  const int K = 32768;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ld : {34560, 1280}) {
    unsigned short* src;
    hipMalloc(&src, (size_t)K * ld * 2);
    hipMemset(src, 1, (size_t)K * ld * 2);
    const int ncol = ld / 256;
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rnd = 0; rnd < 3; ++rnd) for (int shape = 0; shape < 2; ++shape) {
      const int steps = 1000;
      hipEventRecord(e0);
      if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, src, ld, ncol, steps);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, src, ld, ncol, steps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("ld %5d shape %s: %.3f ms, %.2f us per step (16 KiB per CU), %.1f B/clk/CU at 2.1 GHz, %.2f TB/s chip\n", ld, shape ? "8 rows x 128 B" : "4 rows x 256 B", ms,
             ms * 1e3 / steps, 16384.0 / (ms * 1e-3 / steps) / 2.1e9, 256.0 * 16384 * steps / ms / 1e9);
    }
    hipFree(src);
  }
  return 0;
}
