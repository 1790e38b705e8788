// Does the lane -> source-address map of an LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave instruction) change its ISSUE cost?
// Round 5: the TN weight-gradient kernel loses 26-34 % to its 8 pieces per K tile (tools/dev/tn_ablate.py), the NT kernel 15 % to the
// same number of pieces. Their pieces differ in shape: NT = 8 rows x 128 B, TN = 4 rows x 256 B, both with an XOR swizzle on the
// source side that permutes the 16-byte chunks of a row among the lanes. 8 waves per block, one block per CU, every wave issues
// NP pieces back to back from an L2-warm table (row stride 2560 B) and stamps s_memtime around the issue and around the drain.
//   map 0: lane-linear 1 KiB                         map 1: 8 rows x 128 B, chunks in lane order
//   map 2: 8 x 128 B, chunk ^ (row & 7)              map 3: 4 rows x 256 B, chunks in lane order
//   map 4: 4 x 256 B, chunk ^ ((row & 7) << 1) (TN)  map 5: 4 x 256 B, chunk ^ ((row & 3) << 2) (quads intact, moved)
//   map 6: 4 x 256 B, chunk rotated by 2 (row & 7)   map 7: 4 x 256 B, flash2's c ^ (((r&3)<<2) | ((r>>2)&3))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NP = 32;
template <int MAP>
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ src, int ld, unsigned long long* out, int rows_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row, chunk;
  if (MAP == 0) { row = 0; chunk = lane; }
  else if (MAP <= 2) { row = lane >> 3; chunk = lane & 7; if (MAP == 2) chunk ^= (row & 7); }
  else {
    row = lane >> 4; chunk = lane & 15;
  }
  long long t_issue = 0, t_drain = 0;
  for (int rep = 0; rep < 8; ++rep) {
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int r = (blockIdx.x * 8 + wave) * 64 + i * (MAP == 0 ? 1 : (MAP <= 2 ? 8 : 4)) + row;
      r = r % rows_total;
      int c = chunk;
      if (MAP == 4) c = chunk ^ ((r & 7) << 1);
      if (MAP == 5) c = chunk ^ ((r & 3) << 2);
      if (MAP == 6) c = (chunk + 2 * (r & 7)) & 15;
      if (MAP == 7) c = chunk ^ (((r & 3) << 2) | ((r >> 2) & 3));
      const unsigned short* p = src + (size_t)r * ld + c * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + wave * 16384 + (i & 15) * 1024), 16, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    if (rep >= 2) { t_issue += t1 - t0; t_drain += t2 - t0; }
  }
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = t_issue / 6;
    out[(blockIdx.x * 8 + wave) * 2 + 1] = t_drain / 6;
  }
}
int main() {
  const int rows = 65536, ld = 1280;
  unsigned short* src;
  hipMalloc(&src, (size_t)rows * ld * 2);
  hipMemset(src, 1, (size_t)rows * ld * 2);
  unsigned long long* out;
  hipMalloc(&out, 256 * 8 * 2 * 8);
  unsigned long long* h = (unsigned long long*)malloc(256 * 8 * 2 * 8);
  for (int grid : {256, 1}) for (int map = 0; map < 8; ++map) {
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(grid), dim3(512), 131072, 0, src, ld, out, rows)
    for (int w = 0; w < 2; ++w) {
      switch (map) { case 0: hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(0); break;
        case 1: hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(1); break;
        case 2: hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(2); break;
        case 3: hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(3); break;
        case 4: hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(4); break;
        case 5: hipFuncSetAttribute((const void*)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(5); break;
        case 6: hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(6); break;
        default: hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); RUN(7); break; }
      hipDeviceSynchronize();
    }
    hipMemcpy(h, out, grid * 8 * 2 * 8, hipMemcpyDeviceToHost);
    double si = 0, sd = 0;
    for (int i = 0; i < grid * 8; ++i) { si += h[2 * i]; sd += h[2 * i + 1]; }
    printf("grid %3d map %d: issue %7.1f cycles per piece, issue -> all landed %7.1f per piece (%d pieces per wave, 8 waves per CU)\n", grid, map,
           si / (grid * 8) / NP, sd / (grid * 8) / NP, NP);
  }
  return 0;
}
