// FP8 (OCP e4m3fn) GEMM on the block-scaled MFMA of gfx950, for the frozen linear layers of the CLIP tower and the LLaMA stack at
// inference (BASELINE config 5: "fp8 MFMA ViT+LLaMA path"):
//     C[M, N] (bf16) = act( (Aq[M, K] . Bq[N, K]^T) * sa[m] * sb[n] + bias[n] ) + residual[m, n]
// Aq / Bq hold e4m3 codes; sa (per activation row) and sb (per weight row = output channel) are fp32 de-quantisation scales
// (amax / 448), applied in the epilogue. The product runs on v_mfma_scale_f32_16x16x128_f8f6f4 — the K = 128 form that reaches the
// fp8 rate (2x bf16 per clock; the un-scaled K = 32 fp8 MFMA only runs at the bf16 rate) — with unit E8M0 block scales: per-row /
// per-channel scaling in fp32 is finer than a power-of-two per 32 elements and needs no scale operand traffic.
// Structure: 256 x 256 output tile per workgroup of 8 waves (2 x 4, 128 x 64 each = 32 accumulator tiles; at the fp8 rate a 128 x 128
// tile would need ~40 TB/s of L2 -> LDS traffic, four times what the chip delivers: the tile, not the MFMA, sets the ceiling), K
// tiles of 128 bytes, A / B tiles double-buffered in LDS (128 KB) by LDS-DMA (global_load_lds_dwordx4) with the 16-byte chunk of a
// row XOR-swizzled by (row & 7) on the SOURCE address, one counted vmcnt + two barriers per K tile, one workgroup per CU. The
// products are formed transposed (B fragment as the MFMA's A operand) so that a lane owns four consecutive output columns. Operand map (checked on hardware with
// exact integer data, tests/test_kernels_gpu.py): lane l holds row l & 15, bytes k = 32 (l >> 4) .. + 31 of the 128-deep step.
#include "common.h"
#include <string.h>

namespace {

constexpr int F8_BM = 256, F8_BN = 256, F8_BK = 128;  // BK in bytes = fp8 elements
constexpr int F8_NT = 512;
constexpr int F8_TILE = F8_BM * F8_BK;                // 32 KB per operand tile

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// XOR key of the 16-byte chunk position inside a 128-byte LDS row. Two rows share a 256-byte bank row, a ds_read_b128 lane group holds
// rows {0-3, 12-15} at chunk c and rows {4-11} at chunk c + 2 (microarch guide, LDS table): an exhaustive search over linear keys gives
// this one as conflict-free for both 16-byte halves of a fragment (`row & 7`, the usual key, is 2-way here).
__device__ __forceinline__ int f8_key(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ void f8_stage(char* lds_tile, const unsigned char* __restrict__ src, int ld, int row0, int rows_valid, int k0, int wave,
                                         int lane) {
  // tile = 2048 chunks of 16 bytes (8 waves x 4 instructions x 64 lanes): chunk d -> (row = d >> 3, position = d & 7) holds source chunk position ^ f8_key(row)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = (wave * 4 + i) * 64 + lane;
    const int row = d >> 3, pos = d & 7;
    const int gr = min(row0 + row, rows_valid - 1);
    const unsigned char* g = src + (int64_t)gr * ld + k0 + ((pos ^ f8_key(row)) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds_tile + (wave * 4 + i) * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ i32x8_t f8_frag(const char* tile, int row, int kg) {
  const int sw = f8_key(row);
  const u32x4_t lo = *(const u32x4_t*)(tile + row * F8_BK + (((2 * kg) ^ sw) << 4));
  const u32x4_t hi = *(const u32x4_t*)(tile + row * F8_BK + (((2 * kg + 1) ^ sw) << 4));
  return i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}

__global__ __launch_bounds__(F8_NT, 1) void gemm_fp8_kernel(const grove_gemm_fp8_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: 128 rows x 64 columns each
  // XCD-aware tile order: consecutive tiles of one XCD share the B column block
  const int tiles_m = (p.M + F8_BM - 1) / F8_BM;
  const int n_tiles = gridDim.x;
  const int per = n_tiles >> 3, rem = n_tiles & 7;
  const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
  const int tile = x * per + min(x, rem) + s;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * F8_BM, n0 = tn * F8_BN;
  const unsigned char* A = (const unsigned char*)p.A;
  const unsigned char* B = (const unsigned char*)p.B;
  f32x4_t acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / F8_BK;
  f8_stage(smem, A, p.lda, m0, p.M, 0, wave, lane);
  f8_stage(smem + F8_TILE, B, p.ldb, n0, p.N, 0, wave, lane);
  const int one = 127;  // E8M0 2^0
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * F8_TILE;
    if (kt + 1 < nk) {
      char* nxt = smem + ((kt + 1) & 1) * 2 * F8_TILE;
      f8_stage(nxt, A, p.lda, m0, p.M, (kt + 1) * F8_BK, wave, lane);
      f8_stage(nxt + F8_TILE, B, p.ldb, n0, p.N, (kt + 1) * F8_BK, wave, lane);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this tile's 8 loads have landed; the next tile's stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    i32x8_t bf[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bf[ni] = f8_frag(cur + F8_TILE, wn * 64 + ni * 16 + fr, kg);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const i32x8_t af = f8_frag(cur, wm * 128 + mi * 16 + fr, kg);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)  // transposed product: D[row = n][col = m]
        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ni], af, acc[mi][ni], 0, 0, 0, one, 0, one);
    }
    __syncthreads();  // everyone is done with `cur` before the next iteration's DMA overwrites it
  }
  // epilogue: lane holds C[m = fr][n = 4 kg + r] of every 16 x 16 tile: four consecutive columns -> 8-byte accesses
  const float* sa = p.scale_a;
  const float* sb = p.scale_b;
  const bool vec = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (!p.residual || p.ldr % 4 == 0) && (((uintptr_t)p.C | (uintptr_t)p.residual | (uintptr_t)p.bias) & 7) == 0;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wm * 128 + mi * 16 + fr;
    if (m >= p.M) continue;
    const float sam = sa[m];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + 4 * kg;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = min(n + r, p.N - 1);
        const float b = p.bias ? bf2f(((const bf16_raw*)p.bias)[nn]) : 0.f;
        v[r] = act_apply(p.act, acc[mi][ni][r] * (sam * sb[nn]) + b);
      }
      bf16_raw* c = (bf16_raw*)p.C + (int64_t)m * p.ldc + n;
      if (vec) {
        if (p.residual) {
          const u32x2_t rr = *(const u32x2_t*)((const bf16_raw*)p.residual + (int64_t)m * p.ldr + n);
          v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
        }
        *(u32x2_t*)c = u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.N) break;
          float o = v[r];
          if (p.residual) o += bf2f(((const bf16_raw*)p.residual)[(int64_t)m * p.ldr + n + r]);
          c[r] = f2bf(o);
        }
      }
    }
  }
}

// bf16 rows -> e4m3 codes + one fp32 scale per row (amax / 448; a zero row gets scale 1). One wave per row.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_raw* __restrict__ x, unsigned char* __restrict__ q, float* __restrict__ scale,
                                                             int rows, int K, int ld_x, int ld_q) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_raw* xr = x + (int64_t)row * ld_x;
  float amax = 0.f;
  for (int c = lane * 8; c < K; c += 512) {
    const u32x4_t u = *(const u32x4_t*)(xr + c);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf_lo(u.x)), fabsf(bf_hi(u.x))), fmaxf(fabsf(bf_lo(u.y)), fabsf(bf_hi(u.y)))));
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf_lo(u.z)), fabsf(bf_hi(u.z))), fmaxf(fabsf(bf_lo(u.w)), fabsf(bf_hi(u.w)))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / sc;
  if (lane == 0) scale[row] = sc;
  unsigned char* qr = q + (int64_t)row * ld_q;
  for (int c = lane * 8; c < K; c += 512) {
    const u32x4_t u = *(const u32x4_t*)(xr + c);
    auto cl = [inv](float v) { return fminf(fmaxf(v * inv, -448.f), 448.f); };  // e4m3fn has no infinity: keep the row maximum at 448
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.x)), cl(bf_hi(u.x)), w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.y)), cl(bf_hi(u.y)), w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.z)), cl(bf_hi(u.z)), w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.w)), cl(bf_hi(u.w)), w1, true);
    *(u32x2_t*)(qr + c) = u32x2_t{(unsigned)w0, (unsigned)w1};
  }
}

// The same with an activation in front: q = e4m3(act(x) / scale), scale = amax(act(x)) / 448 — SAM's mlp.lin1 -> GELU -> lin2 under fp8_policy
// "sam_mlp" (image_encoder.py:243-259, common.py:21-26). At the fp8 rate a K = 1280 output tile's matrix work is 10 K tiles, and a GELU in
// that GEMM's epilogue costs more than half of it again (measured 381 vs 243 us for the launch: nothing hides the epilogue of a persistent
// block); this pass is memory-bound and reads every element anyway. One wave per row; the row, rounded to bf16 after the activation (the
// value the unfused path would have stored and re-read), stays in registers between the amax sweep and the conversion: one read, one write.
template <int CH>  // 512-element chunks per row held in registers (K <= 512 CH)
__global__ __launch_bounds__(256) void quant_fp8_rows_act_kernel(const bf16_raw* __restrict__ x, unsigned char* __restrict__ q, float* __restrict__ scale,
                                                                 int rows, int K, int ld_x, int ld_q, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_raw* xr = x + (int64_t)row * ld_x;
  u32x4_t keep[CH];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = lane * 8 + i * 512;
    u32x4_t u = u32x4_t{0u, 0u, 0u, 0u};
    if (c < K) {
      u = *(const u32x4_t*)(xr + c);
      u.x = pack2bf(act_apply(act, bf_lo(u.x)), act_apply(act, bf_hi(u.x)));
      u.y = pack2bf(act_apply(act, bf_lo(u.y)), act_apply(act, bf_hi(u.y)));
      u.z = pack2bf(act_apply(act, bf_lo(u.z)), act_apply(act, bf_hi(u.z)));
      u.w = pack2bf(act_apply(act, bf_lo(u.w)), act_apply(act, bf_hi(u.w)));
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf_lo(u.x)), fabsf(bf_hi(u.x))), fmaxf(fabsf(bf_lo(u.y)), fabsf(bf_hi(u.y)))));
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf_lo(u.z)), fabsf(bf_hi(u.z))), fmaxf(fabsf(bf_lo(u.w)), fabsf(bf_hi(u.w)))));
    }
    keep[i] = u;
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / sc;
  if (lane == 0) scale[row] = sc;
  unsigned char* qr = q + (int64_t)row * ld_q;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = lane * 8 + i * 512;
    if (c >= K) continue;
    const u32x4_t u = keep[i];
    auto cl = [inv](float v) { return fminf(fmaxf(v * inv, -448.f), 448.f); };
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.x)), cl(bf_hi(u.x)), w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.y)), cl(bf_hi(u.y)), w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.z)), cl(bf_hi(u.z)), w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(bf_lo(u.w)), cl(bf_hi(u.w)), w1, true);
    *(u32x2_t*)(qr + c) = u32x2_t{(unsigned)w0, (unsigned)w1};
  }
}

}  // namespace

int grove_gemm_fp8_pipelined(const grove_gemm_fp8_params* q, hipStream_t s);  // gemm.hip
static int g_fp8_pipelined = 1;  // 0 = always the two-barrier kernel of this file (A/B arm: grove_gemm_fp8_set_pipelined)
extern "C" int grove_gemm_fp8_set_pipelined(int on) {
  g_fp8_pipelined = on ? 1 : 0;
  return GROVE_OK;
}

void grove_gemm_ctx_set(const grove_gemm_workspace* w, int mode, grove_gemm_plan* plan, void* host_image, size_t host_bytes);
void grove_gemm_ctx_clear();

static int gemm_fp8_run(const grove_gemm_fp8_params* p, int mode, void* stream) {
  GROVE_CHECK(p && p->M > 0 && p->N > 0 && p->K > 0 && (mode || (p->A && p->B && p->C && p->scale_a && p->scale_b)), GROVE_E_SHAPE, "gemm_fp8: bad arguments");
  GROVE_CHECK(p->K % F8_BK == 0, GROVE_E_SHAPE, "gemm_fp8: K=%d must be a multiple of %d", p->K, F8_BK);
  GROVE_CHECK(p->lda % 16 == 0 && p->ldb % 16 == 0 && ((uintptr_t)p->A & 15) == 0 && ((uintptr_t)p->B & 15) == 0, GROVE_E_ALIGN,
              "gemm_fp8: operand rows must be 16-byte aligned");
  if (g_fp8_pipelined) {  // the FP8 instances of the persistent pipelined kernel (gemm.hip) when the problem fits them
    const int rc = grove_gemm_fp8_pipelined(p, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  if (mode) return GROVE_OK;  // plan / image request for a problem the two-barrier kernel runs: no workspace (sizes stay 0)
  const int tiles = ((p->M + F8_BM - 1) / F8_BM) * ((p->N + F8_BN - 1) / F8_BN);
  const size_t lds = 4 * F8_TILE;
  hipFuncSetAttribute((const void*)gemm_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gemm_fp8_kernel, dim3(tiles), dim3(F8_NT), lds, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_gemm_fp8(const grove_gemm_fp8_params* p, const grove_gemm_workspace* w, void* stream) {
  grove_gemm_ctx_set(w, 0, nullptr, nullptr, 0);
  const int rc = gemm_fp8_run(p, 0, stream);
  grove_gemm_ctx_clear();
  return rc;
}

extern "C" int grove_gemm_fp8_make_plan(const grove_gemm_fp8_params* p, grove_gemm_plan* out) {
  GROVE_CHECK(out != nullptr, GROVE_E_SHAPE, "gemm_fp8_plan: null output");
  memset(out, 0, sizeof(*out));
  grove_gemm_ctx_set(nullptr, 1, out, nullptr, 0);
  const int rc = gemm_fp8_run(p, 1, nullptr);
  grove_gemm_ctx_clear();
  return rc;
}

extern "C" int grove_gemm_fp8_plan_image(const grove_gemm_fp8_params* p, void* host_image, size_t bytes) {
  grove_gemm_ctx_set(nullptr, 2, nullptr, host_image, bytes);
  const int rc = gemm_fp8_run(p, 2, nullptr);
  grove_gemm_ctx_clear();
  return rc;
}

extern "C" int grove_quant_fp8_rows(const void* x, void* q, float* scale, int32_t rows, int32_t K, int32_t ld_x, int32_t ld_q, void* stream) {
  GROVE_CHECK(x && q && scale && rows > 0 && K > 0, GROVE_E_SHAPE, "quant_fp8_rows: bad arguments");
  GROVE_CHECK(K % 8 == 0 && ld_x % 8 == 0 && ld_q % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0, GROVE_E_ALIGN,
              "quant_fp8_rows: K and the leading dims must be multiples of 8");
  hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x, (unsigned char*)q, scale, rows, K,
                     ld_x, ld_q);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_quant_fp8_rows_act(const void* x, void* q, float* scale, int32_t rows, int32_t K, int32_t ld_x, int32_t ld_q, int32_t act, void* stream) {
  GROVE_CHECK(x && q && scale && rows > 0 && K > 0, GROVE_E_SHAPE, "quant_fp8_rows_act: bad arguments");
  GROVE_CHECK(K % 8 == 0 && ld_x % 8 == 0 && ld_q % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0, GROVE_E_ALIGN,
              "quant_fp8_rows_act: K and the leading dims must be multiples of 8");
  GROVE_CHECK(K <= 8192, GROVE_E_SHAPE, "quant_fp8_rows_act: K=%d > 8192 (the row is held in registers)", K);
  GROVE_CHECK(act >= GROVE_ACT_NONE && act <= GROVE_ACT_SIGMOID, GROVE_E_SHAPE, "quant_fp8_rows_act: act %d", act);
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bf16_raw* xs = (const bf16_raw*)x;
  unsigned char* qs = (unsigned char*)q;
  if (K <= 2048) hipLaunchKernelGGL(quant_fp8_rows_act_kernel<4>, grid, block, 0, s, xs, qs, scale, rows, K, ld_x, ld_q, act);
  else if (K <= 5120) hipLaunchKernelGGL(quant_fp8_rows_act_kernel<10>, grid, block, 0, s, xs, qs, scale, rows, K, ld_x, ld_q, act);
  else hipLaunchKernelGGL(quant_fp8_rows_act_kernel<16>, grid, block, 0, s, xs, qs, scale, rows, K, ld_x, ld_q, act);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
