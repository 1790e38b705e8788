// "TN" bf16 GEMM for weight gradients on gfx950:  C[m, n] += scale * sum_k A[k, m] * B[row_b(k, n), n']
//
// Both operands are K-major (row k = one token): A = dY [tokens, M], B = X [tokens, N] — exactly what the
// backward pass has in HBM, so dW = dY^T X needs no transposed copies. Tiles are staged as plain [64 k][128]
// bf16 images with a 288-byte row stride and consumed with ds_read_b64_tr_b16 (the transposed read validated
// in flash_attn.hip: conflict-free at that stride); both MFMA operands use the same permuted k-order.
// B rows may be gathered per tap (b_idx[tap*K + k], -1 = zero row, tap = n / (N / b_taps)): the weight gradient
// of the implicit-GEMM Conv3d/Conv2d (SAM / CLIP adapters, neck) in ONE launch, no im2col in HBM.
// C is fp32 and always accumulated; K is split over blockIdx.z (fp32 atomics) when the tile grid cannot fill
// the chip. 128 x 128 x 64 tile, 4 waves (2 x 2), register-staged prefetch of the next K tile.
#include "common.h"

namespace {

constexpr int TBM = 128, TBN = 128, TBK = 64, TNT = 256;
constexpr int TROWB = 128 * 2 + 32;           // LDS row stride (bytes)
constexpr int TTILEB = TBK * TROWB;           // one operand tile
constexpr int TLPT = (TBK * 16) / TNT;        // 16-byte chunks per thread per operand tile (= 4)

__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int row0, int c0, int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const int qq = fr >> 2, pp = fr & 3;
  const char* a0 = tile + (row0 + 4 * g + qq) * TROWB + (c0 + 4 * pp) * 2;
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 16 * TROWB));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(TNT) void gemm_tn_kernel(const grove_gemm_tn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  const int tiles_m = (p.M + TBM - 1) / TBM;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;  // consecutive blocks share the B (gathered) panel
  const int m0 = tm * TBM, n0 = tn * TBN;
  const bf16_raw* __restrict__ A = (const bf16_raw*)p.A;
  const bf16_raw* __restrict__ B = (const bf16_raw*)p.B;
  const int n_per_tap = p.N / p.b_taps;
  const int tap = n0 / n_per_tap;
  const int nb0 = n0 - tap * n_per_tap;  // column inside the gathered row
  const int32_t* bidx = p.b_idx ? p.b_idx + (int64_t)tap * p.K : nullptr;

  const int nkt = (p.K + TBK - 1) / TBK;
  const int nsplit = gridDim.z;
  const int per = (nkt + nsplit - 1) / nsplit;
  const int kt0 = blockIdx.z * per;
  const int kt1 = min(nkt, kt0 + per);
  if (kt0 >= kt1) return;

  const int st_c = tid & 15;   // chunk (8 columns) inside the 128-wide tile row
  const int st_r = tid >> 4;   // row, + 16 per pass
  const bool a_col_ok = m0 + st_c * 8 < p.M;
  const bool b_col_ok = n0 + st_c * 8 < p.N && nb0 + st_c * 8 < n_per_tap;
  u32x4_t ra[TLPT], rb[TLPT];
  // gathered-row indices are fetched ONE tile ahead of the loads that use them, so the dependent
  // index -> row load chain never sits on the critical path of a K tile
  int rowidx[TLPT];
  auto fetch_idx = [&](int kt) {
#pragma unroll
    for (int i = 0; i < TLPT; ++i) {
      const int k = kt * TBK + st_r + 16 * i;
      rowidx[i] = k < p.K ? (bidx ? bidx[k] : k) : -1;
    }
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int i = 0; i < TLPT; ++i) {
      const int k = kt * TBK + st_r + 16 * i;
      ra[i] = u32x4_t{0u, 0u, 0u, 0u};
      rb[i] = u32x4_t{0u, 0u, 0u, 0u};
      if (k < p.K && a_col_ok) ra[i] = *(const u32x4_t*)(A + (int64_t)k * p.lda + m0 + st_c * 8);
      const int row = rowidx[i];
      if (row >= 0 && b_col_ok) rb[i] = *(const u32x4_t*)(B + (int64_t)row * p.ldb + nb0 + st_c * 8);
    }
    fetch_idx(kt + 1);
  };
  auto commit = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < TLPT; ++i) {
      const int off = (st_r + 16 * i) * TROWB + st_c * 16;
      *(u32x4_t*)(buf + off) = ra[i];
      *(u32x4_t*)(buf + TTILEB + off) = rb[i];
    }
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const char* buf) {
    const char* As = buf;
    const char* Bs = buf + TTILEB;
#pragma unroll
    for (int ks = 0; ks < TBK / 32; ++ks) {
      bf16x8_t af[4], bfg[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(As, ks * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfg[j] = tr_frag(Bs, ks * 32, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  char* buf0 = smem;
  char* buf1 = smem + 2 * TTILEB;
  fetch_idx(kt0);
  issue(kt0);
  commit(buf0);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    char* cur = ((kt - kt0) & 1) ? buf1 : buf0;
    char* nxt = ((kt - kt0) & 1) ? buf0 : buf1;
    const bool more = kt + 1 < kt1;
    if (more) issue(kt + 1);
    compute(cur);
    if (more) commit(nxt);
    __syncthreads();
  }
  float scale = p.alpha;
  if (p.scale_ptr) scale *= p.scale_tanh ? tanhf(*p.scale_ptr) : *p.scale_ptr;
  // acc[i][j] holds D[n = 4g + r][m = fr] of m-tile i, n-tile j
  float* C = p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      float* c = C + (int64_t)m * p.ldc + n;
      if (nsplit > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) atomicAdd(c + e, acc[i][j][e] * scale);
      } else if (n + 3 < p.N && (p.ldc & 3) == 0) {
        const f32x4_t old = *(const f32x4_t*)c;
        *(f32x4_t*)c = old + acc[i][j] * scale;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) c[e] += acc[i][j][e] * scale;
      }
    }
  }
}

}  // namespace

extern "C" int grove_gemm_tn_bf16(const grove_gemm_tn_params* pp, void* stream) {
  GROVE_CHECK(pp && pp->M > 0 && pp->N > 0 && pp->K > 0, GROVE_E_SHAPE, "gemm_tn: bad shape");
  grove_gemm_tn_params p = *pp;
  if (p.b_taps <= 0) p.b_taps = 1;
  GROVE_CHECK(p.M % 8 == 0 && p.N % 8 == 0, GROVE_E_SHAPE, "gemm_tn: M=%d and N=%d must be multiples of 8", p.M, p.N);
  GROVE_CHECK(p.N % p.b_taps == 0 && (p.N / p.b_taps) % TBN == 0 || p.b_taps == 1, GROVE_E_SHAPE, "gemm_tn: N/b_taps must be a multiple of %d", TBN);
  GROVE_CHECK(p.lda % 8 == 0 && p.ldb % 8 == 0, GROVE_E_ALIGN, "gemm_tn: lda/ldb must be multiples of 8");
  GROVE_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && ((uintptr_t)p.C & 15) == 0, GROVE_E_ALIGN, "gemm_tn: pointers must be 16-byte aligned");
  const int tiles = ((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
  const int nkt = (p.K + TBK - 1) / TBK;
  int split = p.split_k;
  if (split <= 0) {
    split = 1;
    if (tiles < 256 && nkt >= 8) {
      split = (768 + tiles - 1) / tiles;
      if (split > nkt / 2) split = nkt / 2;
      if (split > 256) split = 256;
      if (split < 1) split = 1;
    }
  }
  static bool attr = false;
  const size_t lds = 4 * TTILEB;
  if (!attr) {
    hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, 1, split), dim3(TNT), lds, (hipStream_t)stream, p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
