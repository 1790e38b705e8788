// Per-CU store-path microbenchmark: one 512-thread block per CU writes 256 x 256 bf16 tiles (128 KB) of a row-major matrix
// with different lane -> address patterns. Answers: is the GEMM epilogue bound by bytes or by the number of write requests?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bw tools/micro/store_bw.hip && /tmp/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// SEG = contiguous bytes written per row by one wave instruction (16-byte lanes): 64 -> 16 rows x 64 B, 128 -> 8 rows x 128 B,
// 256 -> 4 rows x 256 B, 512 -> 2 rows x 512 B (the whole tile row)
template <int SEG>
__global__ __launch_bounds__(512) void store_kernel(uint16_t* C, int ldc_bytes, int tiles_n, int tiles_per_wg, int nwg) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int LPR = SEG / 16;        // lanes per row segment
  constexpr int RPI = 64 / LPR;        // rows per instruction
  const u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * nwg;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    char* base = (char*)C + (int64_t)tm * 256 * ldc_bytes + tn * 512;
    // wave w owns rows [32w, 32w+32) x 512 B = 16 KB = 16 instructions
    constexpr int SEGS = 512 / SEG;    // segments per row
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int unit = i;              // each instruction covers RPI rows of one segment column
      const int rblk = unit / SEGS, sc = unit % SEGS;  // RPI*SEGS*... : 32 rows = (32/RPI) row blocks; 16 = (32/RPI)*SEGS
      const int row = wave * 32 + rblk * RPI + lane / LPR;
      const int col = sc * SEG + (lane % LPR) * 16;
      *(u32x4*)(base + (int64_t)row * ldc_bytes + col) = v;
    }
  }
}
template <int SEG>
static void run(uint16_t* C, int M, int N, int reps, int nwg = 256) {
  const int tiles_n = N / 256, tpw = 10; M = nwg * tpw * 256 / tiles_n; if (M < 256) M = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<SEG><<<nwg, 512>>>(C, N * 2, tiles_n, tpw, nwg);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) store_kernel<SEG><<<nwg, 512>>>(C, N * 2, tiles_n, tpw, nwg);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("wgs=%3d N=%5d seg=%3d B: %8.1f us per launch, %6.2f us per tile-round, %6.2f TB/s\n", nwg, N, SEG, us, us / tpw, (double)M * N * 2 / us / 1e6);
}
int main() {
  const int M = 32768;
  uint16_t* C;
  hipMalloc(&C, (size_t)M * 5120 * 2);
  for (int nwg : {8, 16, 32, 64, 128, 256}) {
    run<64>(C, M, 5120, 20, nwg);
    run<512>(C, M, 5120, 20, nwg);
  }
  return 0;
}
