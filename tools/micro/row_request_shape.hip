// Does the REQUEST SHAPE of a weight-streaming load matter on gfx950? Both kernels stream a [N, K] bf16 matrix once with 16-byte loads, 16
// rows per workgroup of four waves (each wave a quarter of K), 8 loads in flight per lane:
//   A  the MFMA B-operand shape: lane -> (row lane & 15, 16 bytes at 64 u + 16 (lane >> 4)): 64 contiguous bytes per row per instruction
//   B  lane -> (row lane >> 3 [+ 8], 16 bytes at 128 u + 16 (lane & 7)): 128 contiguous bytes per row per instruction, two instructions per 16 rows
//   C  the same bytes per instruction as B with the lanes of a row STRIDED: lane -> (row lane & 7 [+ 8], chunk 4 ((lane >> 3) & 1) + (lane >> 4)) —
//      one lane-pair exchange away from the MFMA operand layout
// hipcc --offload-arch=gfx950 -O3 tools/micro/row_request_shape.hip -o tools/micro/row_request_shape && tools/micro/row_request_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int SHAPE>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* __restrict__ W, unsigned* __restrict__ sink, int N, int K) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16, kw = K / 4, k_lo = wave * kw;
  unsigned acc = 0;
  if (SHAPE == 0) {
    const unsigned short* row = W + (size_t)(n0 + (lane & 15)) * K + k_lo + 8 * (lane >> 4);
    for (int k = 0; k < kw; k += 256) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (k + 32 * u < kw) ? __builtin_nontemporal_load((const u32x4*)(row + k + 32 * u)) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  } else {
    const unsigned short* row0 = SHAPE == 1 ? W + (size_t)(n0 + (lane >> 3)) * K + k_lo + 8 * (lane & 7)
                                            : W + (size_t)(n0 + (lane & 7)) * K + k_lo + 8 * (4 * ((lane >> 3) & 1) + (lane >> 4));
    const unsigned short* row1 = row0 + (size_t)8 * K;
    for (int k = 0; k < kw; k += 256) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[2 * u] = (k + 64 * u < kw) ? __builtin_nontemporal_load((const u32x4*)(row0 + k + 64 * u)) : u32x4{0, 0, 0, 0};
        v[2 * u + 1] = (k + 64 * u < kw) ? __builtin_nontemporal_load((const u32x4*)(row1 + k + 64 * u)) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
int main() {
  const int shapes[3][2] = {{12288, 4096}, {4096, 4096}, {4096, 11008}};
  for (auto& sh : shapes) {
    const int N = sh[0], K = sh[1];
    std::vector<unsigned short*> ws(6);
    for (auto& w : ws) { hipMalloc(&w, (size_t)N * K * 2); hipMemset(w, 1, (size_t)N * K * 2); }
    unsigned* sink; hipMalloc(&sink, N * 4);
    for (int shape = 0; shape < 3; ++shape) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (auto w : ws) { if (shape == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(N / 16), dim3(256), 0, 0, w, sink, N, K); else if (shape == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(N / 16), dim3(256), 0, 0, w, sink, N, K); else hipLaunchKernelGGL(stream_kernel<2>, dim3(N / 16), dim3(256), 0, 0, w, sink, N, K); }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) for (auto w : ws) { if (shape == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(N / 16), dim3(256), 0, 0, w, sink, N, K); else if (shape == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(N / 16), dim3(256), 0, 0, w, sink, N, K); else hipLaunchKernelGGL(stream_kernel<2>, dim3(N / 16), dim3(256), 0, 0, w, sink, N, K); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms / 30 * 1e3;
      printf("N=%d K=%d shape %s: %6.1f us  %7.1f GB/s\n", N, K, shape == 0 ? "A (64 B per row per instruction)" : shape == 1 ? "B (128 B per row per instruction)" : "C (128 B per row, strided lanes)", us, (double)N * K * 2 / us / 1e3);
    }
    for (auto w : ws) hipFree(w);
    hipFree(sink);
  }
  return 0;
}
