// Does the lane -> address map of a 16-byte-per-lane store matter? 8 waves per block, each storing a 128 x 64 bf16 sub-tile of a 256 x 256 tile
// (16 instructions of 1 KB), tiles of a [M][N] matrix, persistent over tiles.  map 0: lane = chunk * 16 + row (the GEMM epilogue's);
// map 1: lane = row * 4 + chunk (a quad covers 64 contiguous bytes).
// Round 3 adds the two maps an un-swapped MFMA operand order would give a re-tiled kernel ("what comes next"):
// map 2: 8 bytes per lane, lane = rowgroup * 16 + colchunk: 16 lanes = one 128-byte row segment, a quad = 32 contiguous bytes (wave tile 128 x 64);
// map 3: 16 bytes per lane, 16 lanes = one 256-byte row segment, a quad = 64 contiguous bytes, 4 rows per instruction (wave tile 64 x 128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int MAP>
__global__ __launch_bounds__(512) void k(unsigned short* C, int ldc, int tiles_m, int tiles_n, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  if constexpr (MAP == 2) {  // wave tile 128 rows x 64 contiguous columns: per instruction 4 rows x 128 B, 8 B per lane; 32 instructions per tile
    u32x2 v = u32x2{(unsigned)lane, 1u};
    for (int r = 0; r < reps; ++r)
      for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          unsigned short* p = C + (size_t)(m0 + wr * 128 + i * 4 + (lane >> 4)) * ldc + n0 + wc * 64 + (lane & 15) * 4;
          *(u32x2*)p = v;
          v.x += 1;
        }
      }
    return;
  }
  if constexpr (MAP == 3) {  // wave tile 64 rows x 128 contiguous columns (waves 4 x 2): per instruction 4 rows x 256 B, 16 B per lane; 16 instructions
    u32x4 v = u32x4{(unsigned)lane, 1u, 2u, 3u};
    const int wr4 = wave >> 1, wc2 = wave & 1;
    for (int r = 0; r < reps; ++r)
      for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          unsigned short* p = C + (size_t)(m0 + wr4 * 64 + i * 4 + (lane >> 4)) * ldc + n0 + wc2 * 128 + (lane & 15) * 8;
          *(u32x4*)p = v;
          v.x += 1;
        }
      }
    return;
  }
  const int row = MAP == 0 ? (lane & 15) : (lane >> 2), chunk = MAP == 0 ? (lane >> 4) : (lane & 3);
  u32x4 v = u32x4{(unsigned)lane, 1u, 2u, 3u};
  for (int r = 0; r < reps; ++r)
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
      const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            unsigned short* p = C + (size_t)(m0 + h * 128 + wr * 64 + i * 16 + row) * ldc + n0 + wc * 32 + g * 128 + chunk * 8;
            *(u32x4*)p = v;
            v.x += 1;
          }
    }
}
int main() {
  const int M = 32768, N = 5120;
  unsigned short* C;
  hipMalloc(&C, (size_t)M * N * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 8}) for (int rnd = 0; rnd < 2; ++rnd) for (int map = 0; map < 4; ++map) {
    const int reps = grid == 256 ? 10 : 1;
    hipEventRecord(e0);
    if (map == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, C, N, M / 256, N / 256, reps);
    else if (map == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, C, N, M / 256, N / 256, reps);
    else if (map == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, C, N, M / 256, N / 256, reps);
    else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, C, N, M / 256, N / 256, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_block = (double)(M / 256) * (N / 256) / grid * reps;
    printf("grid %d map %d: %.3f ms, %.2f us per tile per block, %.2f TB/s\n", grid, map, ms, ms * 1e3 / tiles_per_block, (double)M * N * 2 * reps / ms / 1e9);
  }
  return 0;
}
