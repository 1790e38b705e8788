// Does the lane -> address map of a 16-byte-per-lane store matter? 8 waves per block, each storing a 128 x 64 bf16 sub-tile of a 256 x 256 tile
// (16 instructions of 1 KB), tiles of a [M][N] matrix, persistent over tiles.  map 0: lane = chunk * 16 + row (the GEMM epilogue's);
// map 1: lane = row * 4 + chunk (a quad covers 64 contiguous bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MAP>
__global__ __launch_bounds__(512) void k(unsigned short* C, int ldc, int tiles_m, int tiles_n, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
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
  for (int grid : {256, 8}) for (int rnd = 0; rnd < 2; ++rnd) for (int map = 0; map < 2; ++map) {
    const int reps = grid == 256 ? 10 : 1;
    hipEventRecord(e0);
    if (map == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, C, N, M / 256, N / 256, reps);
    else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, C, N, M / 256, N / 256, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_block = (double)(M / 256) * (N / 256) / grid * reps;
    printf("grid %d map %d: %.3f ms, %.2f us per tile per block, %.2f TB/s\n", grid, map, ms, ms * 1e3 / tiles_per_block, (double)M * N * 2 * reps / ms / 1e9);
  }
  return 0;
}
