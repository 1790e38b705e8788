// Is the legacy 16-deep MFMA (v_mfma_f32_16x16x16_bf16, 2-register operands) rate-neutral against v_mfma_f32_16x16x32_bf16 on gfx950?
// Why it matters: a transposed LDS read (ds_read_b64_tr_b16) delivers HALF of a 16x16x32 operand; the TN weight-gradient GEMM and the
// attention backward kernels join two of them into a 4-register tuple with register moves (0.61 non-MFMA VALU instructions per MFMA in
// the TN kernel against 0.28 in the NT kernel, profiles/r02_pmc_gemm_nt_vs_tn.json). Two 16-deep MFMAs on the halves need no join.
// Register-only loops, one block of 512 threads per CU (2 waves per SIMD), 8 independent accumulators per wave:
//   a: 16x16x32                       b: 2 x 16x16x16 per 32-deep step (same FLOPs)
//   c: 16x16x32 + 4 v_mov per MFMA    d: 2 x 16x16x16 (+ 0 v_mov)      e: 16x16x32 + 2 v_mov per MFMA
// prints us per launch and TFLOP/s.   hipcc --offload-arch=gfx950 -O3 mfma_k16.hip -o mfma_k16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
  s16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[(threadIdx.x + 512 * i) & 4095];
  for (int i = 0; i < 2; ++i) b[i] = src[(threadIdx.x + 512 * i + 2048) & 4095];
  f32x4 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  int filler = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (MODE == 0 || MODE == 2 || MODE == 4) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
          if (MODE == 2) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0" : "+v"(filler));
          if (MODE == 4) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %0, %0" : "+v"(filler));
        } else {
          const s16x4 alo = s16x4{a[i][0], a[i][1], a[i][2], a[i][3]}, ahi = s16x4{a[i][4], a[i][5], a[i][6], a[i][7]};
          const s16x4 blo = s16x4{b[j][0], b[j][1], b[j][2], b[j][3]}, bhi = s16x4{b[j][4], b[j][5], b[j][6], b[j][7]};
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(alo, blo, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ahi, bhi, acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = (float)filler * 1e-30f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  std::vector<unsigned short> h(4096 * 8);
  srand(1);
  for (auto& v : h) {
    float x = -6.f;
    for (int i = 0; i < 12; ++i) x += (float)rand() / RAND_MAX;
    unsigned u; memcpy(&u, &x, 4);
    v = (unsigned short)(u >> 16);
  }
  s16x8* src; float* out;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const double fl = (double)iters * 8 * 16384.0 * 8 * 256;
  const char* names[5] = {"a: 16x16x32", "b: 2 x 16x16x16", "c: 16x16x32 + 4 v_mov", "d: 2 x 16x16x16 (again)", "e: 16x16x32 + 2 v_mov"};
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
      float best = 1e30f;
      for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, out, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, out, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, out, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, src, out, iters); break;
          default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, src, out, iters); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%-28s %8.1f us  %7.1f TFLOP/s\n", names[mode], best * 1e3, fl / best / 1e9);
      fflush(stdout);
    }
  return 0;
}
