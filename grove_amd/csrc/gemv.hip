// Weight-streaming GEMV for the greedy-decode step (HF generate with use_cache=True feeds ONE token per
// sequence after step 0, llava_llama.py:144-180): y[b, n] = epi( sum_k x[b, k] * W[n, k] ), 1 <= b <= 8 rows.
//
// HBM-bound: every weight byte is read exactly once per token (13.2 GB per token for the 7B decoder), so the kernel is a
// pure stream — no LDS staging of W, no MFMA. A wave owns RW consecutive output rows; lane l reads the 16-byte chunk
// k = 8 (l + 64 i) .. +7 of each row (a wave instruction = 1 KB contiguous), multiplies with the matching chunk of every x
// row (x sits in LDS, read with ds_read_b128 — every lane a different chunk, conflict-free) and accumulates in fp32;
// one cross-lane reduction per (row, b) at the end. RW x UNROLL = 8 independent 16-byte loads in flight per lane.
#include "common.h"

namespace {
constexpr int GV_THREADS = 256;

// x -> LDS as bf16 rows of XS elements (XS >= K): plain copy, the folded RMSNorm (bf16 or fp32 stream input) or the SwiGLU of a fused
// gate|up row — the prologue shared by the VALU kernel (1-2 sequences) and the matrix-core kernel (3-8 sequences)
template <int MX>
// (rows M .. MX - 1 of the LDS image repeat row M - 1: x has M rows, the instances 1 / 2 / 4 / 8; their results are never stored)
__device__ __forceinline__ void gemv_stage_x(const grove_gemv_params& p, bf16_raw* xs, const int XS, float* red, const int tid) {
  const int K = p.K;
  const bf16_raw* __restrict__ X = (const bf16_raw*)p.x;
  if (p.x_mode == GROVE_GEMV_X_SWIGLU) {
    // x' = silu(gate) * up of a fused [M, 2K] gate|up row (HF LlamaMLP), rounded to bf16 like grove_swiglu_fwd
    for (int c = tid; c < MX * (K >> 3); c += GV_THREADS) {
      const int b = c / (K >> 3), kc = c - b * (K >> 3);
      const u32x4_t gv = *(const u32x4_t*)(X + (int64_t)min(b, p.M - 1) * p.ldx + kc * 8);
      const u32x4_t uv = *(const u32x4_t*)(X + (int64_t)min(b, p.M - 1) * p.ldx + K + kc * 8);
      const float g[8] = {bf_lo(gv.x), bf_hi(gv.x), bf_lo(gv.y), bf_hi(gv.y), bf_lo(gv.z), bf_hi(gv.z), bf_lo(gv.w), bf_hi(gv.w)};
      const float u[8] = {bf_lo(uv.x), bf_hi(uv.x), bf_lo(uv.y), bf_hi(uv.y), bf_lo(uv.z), bf_hi(uv.z), bf_lo(uv.w), bf_hi(uv.w)};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = g[e] * fast_sigmoid(g[e]) * u[e];
      *(u32x4_t*)(xs + b * XS + kc * 8) = u32x4_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
    __syncthreads();
  } else if (p.x_f32) {
    // x is the FP32 residual stream of the decode step (round 3: 32 layers of bf16 rounding of the stream put the generated rows'
    // hidden state at 1.2 % rms from the fp32 oracle against 0.5 % for the prefill rows — tools/decode_precision_probe.py): the
    // statistics and the normalisation run on the fp32 values, the MFMA-free products on their bf16 rounding as everywhere else
    const float* __restrict__ XF = (const float*)p.x;
    const bf16_raw* nw = (const bf16_raw*)p.norm_weight;
    constexpr int XV = 4;  // float4 chunks a thread keeps: K <= 256 * 4 * XV = 4096 (the 7B hidden size) in ONE pass over global memory (16-byte loads)
    const int nq = K >> 2;  // (K % 8 == 0)
    for (int b = 0; b < MX; ++b) {
      const f32x4_t* xr = (const f32x4_t*)(XF + (int64_t)min(b, p.M - 1) * p.ldx);
      if (nq <= GV_THREADS * XV) {
        f32x4_t v[XV];
#pragma unroll
        for (int i = 0; i < XV; ++i) {
          const int q = tid + i * GV_THREADS;
          v[i] = q < nq ? xr[q] : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        float rstd = 1.f;
        if (p.x_mode == GROVE_GEMV_X_RMSNORM) {
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < XV; ++i) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
          rstd = rsqrtf(block_sum<GV_THREADS>(ss, red) / (float)K + p.eps);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
          const int q = tid + i * GV_THREADS;
          if (q >= nq) continue;
          float o[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
          if (p.x_mode == GROVE_GEMV_X_RMSNORM) {
            const u32x2_t w2 = *(const u32x2_t*)(nw + q * 4);
            o[0] *= rstd * bf_lo(w2.x), o[1] *= rstd * bf_hi(w2.x), o[2] *= rstd * bf_lo(w2.y), o[3] *= rstd * bf_hi(w2.y);
          }
          *(u32x2_t*)(xs + b * XS + q * 4) = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        }
      } else {  // long rows: two passes over global memory
        float rstd = 1.f;
        if (p.x_mode == GROVE_GEMV_X_RMSNORM) {
          float ss = 0.f;
          for (int q = tid; q < nq; q += GV_THREADS) {
            const f32x4_t a = xr[q];
            ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
          }
          rstd = rsqrtf(block_sum<GV_THREADS>(ss, red) / (float)K + p.eps);
        }
        for (int q = tid; q < nq; q += GV_THREADS) {
          const f32x4_t a = xr[q];
          float o[4] = {a[0], a[1], a[2], a[3]};
          if (p.x_mode == GROVE_GEMV_X_RMSNORM) {
            const u32x2_t w2 = *(const u32x2_t*)(nw + q * 4);
            o[0] *= rstd * bf_lo(w2.x), o[1] *= rstd * bf_hi(w2.x), o[2] *= rstd * bf_lo(w2.y), o[3] *= rstd * bf_hi(w2.y);
          }
          *(u32x2_t*)(xs + b * XS + q * 4) = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        }
      }
    }
    __syncthreads();
  } else {
    for (int c = tid; c < MX * (K >> 3); c += GV_THREADS) {
      const int b = c / (K >> 3), kc = c - b * (K >> 3);
      *(u32x4_t*)(xs + b * XS + kc * 8) = *(const u32x4_t*)(X + (int64_t)min(b, p.M - 1) * p.ldx + kc * 8);
    }
    __syncthreads();
    if (p.x_mode == GROVE_GEMV_X_RMSNORM) {
      // x' = bf16(x * rsqrt(mean(x^2) + eps) * w): every block normalises its own copy of the rows (K elements: free)
      const bf16_raw* nw = (const bf16_raw*)p.norm_weight;
      for (int b = 0; b < MX; ++b) {
        float ss = 0.f;
        for (int k = tid; k < K; k += GV_THREADS) {
          const float v = bf2f(xs[b * XS + k]);
          ss += v * v;
        }
        const float rstd = rsqrtf(block_sum<GV_THREADS>(ss, red) / (float)K + p.eps);
        for (int k = tid; k < K; k += GV_THREADS) xs[b * XS + k] = f2bf(bf2f(xs[b * XS + k]) * rstd * bf2f(nw[k]));
        __syncthreads();
      }
    }
  }
}

template <int MX, int GV_RW>  // GV_RW: output rows per wave (4; 2 when N is too small to fill the chip with 4)
__global__ __launch_bounds__(GV_THREADS) void gemv_kernel(const grove_gemv_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_raw* xs = (bf16_raw*)smem;  // [MX][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  const bf16_raw* __restrict__ X = (const bf16_raw*)p.x;
  __shared__ float red[GV_THREADS / 64];
  // The weight stream starts BEFORE the x prologue (round 3): the first trip of this wave's rows (4 chunks x GV_RW rows = 16 / 8
  // loads of 16 B per lane) does not depend on x, and the prologue — x into LDS, the folded RMSNorm's block reduction or the SwiGLU —
  // is 2-3 us during which the HBM pipe of this CU would otherwise sit idle (129 such kernels per generated token).
  const int n0 = (blockIdx.x * (GV_THREADS / 64) + wave) * GV_RW;
  const bf16_raw* __restrict__ W = (const bf16_raw*)p.W;
  const bf16_raw* wrow[GV_RW];
#pragma unroll
  for (int r = 0; r < GV_RW; ++r) wrow[r] = W + (int64_t)min(n0 + r, p.N - 1) * p.ldw;
  const bool prefetched = K >= 2048;
  u32x4_t pre[4][GV_RW];
  if (prefetched) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < GV_RW; ++r) pre[q][r] = __builtin_nontemporal_load((const u32x4_t*)(wrow[r] + lane * 8 + q * 512));
  }
  gemv_stage_x<MX>(p, xs, K, red, tid);
  if (n0 >= p.N) return;
  float acc[GV_RW][MX];
#pragma unroll
  for (int r = 0; r < GV_RW; ++r)
#pragma unroll
    for (int b = 0; b < MX; ++b) acc[r][b] = 0.f;

  auto mac = [&](const u32x4_t (&wv)[GV_RW], int k) {  // k: this lane's chunk start
#pragma unroll
    for (int b = 0; b < MX; ++b) {
      const u32x4_t xv = *(const u32x4_t*)(xs + b * K + k);
      const float x0 = bf_lo(xv.x), x1 = bf_hi(xv.x), x2 = bf_lo(xv.y), x3 = bf_hi(xv.y);
      const float x4 = bf_lo(xv.z), x5 = bf_hi(xv.z), x6 = bf_lo(xv.w), x7 = bf_hi(xv.w);
#pragma unroll
      for (int r = 0; r < GV_RW; ++r) {
        float a = acc[r][b];
        a = fmaf(bf_lo(wv[r].x), x0, a); a = fmaf(bf_hi(wv[r].x), x1, a);
        a = fmaf(bf_lo(wv[r].y), x2, a); a = fmaf(bf_hi(wv[r].y), x3, a);
        a = fmaf(bf_lo(wv[r].z), x4, a); a = fmaf(bf_hi(wv[r].z), x5, a);
        a = fmaf(bf_lo(wv[r].w), x6, a); a = fmaf(bf_hi(wv[r].w), x7, a);
        acc[r][b] = a;
      }
    }
  };
  auto step = [&](int k) {
    u32x4_t wv[GV_RW];
#pragma unroll
    for (int r = 0; r < GV_RW; ++r) wv[r] = __builtin_nontemporal_load((const u32x4_t*)(wrow[r] + k));
    mac(wv, k);
  };
  int k = lane * 8;
  if (prefetched) {  // the trip that was issued before the prologue
#pragma unroll
    for (int q = 0; q < 4; ++q) mac(pre[q], k + q * 512);
    k += 2048;
  }
  for (; k + 1536 < K; k += 2048) {  // four chunks per trip: 4 * GV_RW 16-byte loads in flight per lane
    step(k);
    step(k + 512);
    step(k + 1024);
    step(k + 1536);
  }
  for (; k < K; k += 512) step(k);

#pragma unroll
  for (int r = 0; r < GV_RW; ++r)
#pragma unroll
    for (int b = 0; b < MX; ++b) acc[r][b] = wave_sum(acc[r][b]);
  if (p.act == GROVE_ACT_SWIGLU_PAIR) {
    // W rows interleaved [4 gate, 4 up] per 8 (ops.swiglu_interleave; GV_RW = 4): wave 2i holds gate rows, wave 2i + 1 the matching up
    // rows. The pair meets in LDS and the even wave writes silu(gate) * up — HF LlamaMLP's activation — as N / 2 outputs, rounded
    // through bf16 exactly like grove_swiglu_fwd, so the down projection's GEMV reads its input as is (its 512 blocks no longer
    // each recompute the SwiGLU of all 11008 elements in their prologue).
    __shared__ float pair_s[GV_THREADS / 64][GV_RW][MX];
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < GV_RW; ++r)
#pragma unroll
        for (int b = 0; b < MX; ++b) pair_s[wave][r][b] = acc[r][b];
    }
    __syncthreads();
    if (lane == 0 && (wave & 1) == 0) {
#pragma unroll
      for (int r = 0; r < GV_RW; ++r) {
        const int n = n0 + r;          // a gate row of the interleaved matrix: rows 8q .. 8q + 3 -> output column 4q + r
        if (n >= p.N) continue;
        const int col = (n >> 3) * 4 + (n & 3);
#pragma unroll
        for (int b = 0; b < MX; ++b) {
          if (b >= p.M) continue;  // (x is padded to MX rows by the caller; y and the residual have M)
          const float gt = bf2f(f2bf(pair_s[wave][r][b])), up = bf2f(f2bf(pair_s[wave + 1][r][b]));
          const float v = gt * fast_sigmoid(gt) * up;
          if (p.y_dtype == GROVE_BF16) ((bf16_raw*)p.y)[(int64_t)b * p.ldy + col] = f2bf(v);
          else ((float*)p.y)[(int64_t)b * p.ldy + col] = v;
        }
      }
    }
    return;
  }
  if (lane == 0) {
    const bf16_raw* bias = (const bf16_raw*)p.bias;
#pragma unroll
    for (int r = 0; r < GV_RW; ++r) {
      const int n = n0 + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int b = 0; b < MX; ++b) {
        if (b >= p.M) continue;  // y and the residual have M rows (3, 5, 6, 7 sequences run the 4- / 8-row instance on a padded x)
        float v = acc[r][b];
        if (bias) v += bf2f(bias[n]);
        v = act_apply(p.act, v);
        if (p.residual) v += p.res_f32 ? ((const float*)p.residual)[(int64_t)b * p.ldr + n] : bf2f(((const bf16_raw*)p.residual)[(int64_t)b * p.ldr + n]);
        if (p.y_dtype == GROVE_BF16) ((bf16_raw*)p.y)[(int64_t)b * p.ldy + n] = f2bf(v);
        else ((float*)p.y)[(int64_t)b * p.ldy + n] = v;
      }
    }
  }
}

// ---- 3..8 sequences (round 5: the clip-batched decode of infer_iground): the same weight stream on the MATRIX cores.
// The VALU kernel above does MX fused multiply-adds per weight element: at MX = 8 it is compute-bound (11.4 ms per token step at
// LLaMA-7B size against 3.1 ms for the 13.2 GB weight stream). Here a workgroup owns 16 output rows and its four waves a quarter of K
// each; per 32-deep k-step a wave issues ONE v_mfma_f32_16x16x32_bf16 with A = x (rows = sequences, 8..13 of the 16 rows are zero
// padding: the matrix pipe is idle otherwise) and B = the weight rows loaded straight from HBM in fragment form (lane = weight row
// n0 + (lane & 15), 16 bytes at k + 8 (lane >> 4): no LDS staging of W, 8 k-steps = 8 independent 16-byte loads in flight per lane).
// x comes from LDS (staged by gemv_stage_x: folded RMSNorm / SwiGLU / fp32 stream input; rows XS = K + 32 elements apart) or, in
// plain mode, straight from global memory (8 x 11008 bf16 = 176 KB would not fit; it is L2-resident). The four partial 16 x 16
// tiles meet in LDS; thread (m, n) of the workgroup runs the epilogue of element (sequence m, row n0 + n).
// PAIR (round 6): the REQUEST SHAPE of the weight loads. The MFMA B operand wants lane (n = lane & 15, g = lane >> 4) to hold row n, k = 8 g .. 8 g + 7
// of a 32-deep k-step: loaded that way a wave instruction asks for 64 contiguous bytes of each of 16 rows. A stream of such requests
// runs at 4.4 TB/s where 128 contiguous bytes per row per instruction run at 5.3-5.4 (tools/micro/row_request_shape.hip, the down_proj
// matrix; lanes of a row need not be neighbours). So a PAIR of k-steps is loaded as two instructions of 8 rows x 128 bytes — lane
// (g, b = (lane >> 3) & 1, r = lane & 7) takes chunk 4 b + g of row r (R0) and of row 8 + r (R1) — and one exchange between lanes l and
// l ^ 8 (DPP row_ror:8, no LDS) puts them into operand order: k-step 0 = b ? partner's R1 : R0, k-step 1 = b ? R1 : partner's R0.
__device__ __forceinline__ unsigned swap8(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }
__device__ __forceinline__ void pair_to_operands(u32x4_t& r0, u32x4_t& r1, bool b) {
  const u32x4_t send = b ? r0 : r1;
  const u32x4_t got = u32x4_t{swap8(send.x), swap8(send.y), swap8(send.z), swap8(send.w)};
  const u32x4_t s0 = b ? got : r0, s1 = b ? r1 : got;
  r0 = s0, r1 = s1;
}

template <bool X_LDS, int MXS, int NG = 1, bool PAIR = false>  // MXS: rows of x staged in LDS; NG: groups of 16 output rows per workgroup (2: the x fragments feed two MFMAs — half the x traffic per weight byte)
__global__ __launch_bounds__(GV_THREADS) void gemv_mfma_kernel(const grove_gemv_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_raw* xs = (bf16_raw*)smem;
  __shared__ float red[GV_THREADS / 64];
  __shared__ float part[NG][GV_THREADS / 64][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K, XS = K + 32;
  const int fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (16 * NG);
  const int kw = K >> 2;                 // this wave's K range (K % 128 == 0)
  const int k_lo = wave * kw;
  const bf16_raw* __restrict__ wrow[NG];
  const bf16_raw* __restrict__ wrow8[NG];  // PAIR: rows 8 + r
  const bool pb = (lane >> 3) & 1;
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    if constexpr (PAIR) {
      wrow[q] = (const bf16_raw*)p.W + (int64_t)min(n0 + 16 * q + (lane & 7), p.N - 1) * p.ldw + k_lo + (4 * (int)pb + g) * 8;
      wrow8[q] = (const bf16_raw*)p.W + (int64_t)min(n0 + 16 * q + 8 + (lane & 7), p.N - 1) * p.ldw + k_lo + (4 * (int)pb + g) * 8;
    } else {
      wrow[q] = (const bf16_raw*)p.W + (int64_t)min(n0 + 16 * q + fr, p.N - 1) * p.ldw + k_lo + g * 8;
      wrow8[q] = wrow[q];
    }
  }
  constexpr int U = NG == 1 ? 8 : 4;     // k-steps per trip: NG * U = 8 independent 16-byte weight loads in flight per lane beside the next trip's 8
  // weight load u of a trip starting at k (PAIR: loads 2 j and 2 j + 1 are the two 8-row halves of k-steps 2 j, 2 j + 1)
  auto wload = [&](int q, int k, int u) -> u32x4_t {
    if constexpr (PAIR) {
      const int ko = k + 64 * (u >> 1);
      return ko < kw ? __builtin_nontemporal_load((const u32x4_t*)(((u & 1) ? wrow8[q] : wrow[q]) + ko)) : u32x4_t{0u, 0u, 0u, 0u};
    } else {
      return (k + u * 32 < kw) ? __builtin_nontemporal_load((const u32x4_t*)(wrow[q] + k + u * 32)) : u32x4_t{0u, 0u, 0u, 0u};
    }
  };
  // the first trip of the weight stream starts before the x prologue (it does not depend on x)
  u32x4_t wv[NG][U];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int u = 0; u < U; ++u) wv[q][u] = wload(q, 0, u);
  if constexpr (X_LDS) gemv_stage_x<MXS>(p, xs, XS, red, tid);
  const bool row_ok = fr < p.M;
  const bf16_raw* __restrict__ xg = (const bf16_raw*)p.x + (int64_t)min(fr, p.M - 1) * p.ldx + k_lo + g * 8;
  const bf16_raw* xl = xs + min(fr, MXS - 1) * XS + k_lo + g * 8;
  f32x4_t acc[NG];
#pragma unroll
  for (int q = 0; q < NG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const u32x4_t z4 = u32x4_t{0u, 0u, 0u, 0u};
  for (int k = 0; k < kw; k += 32 * U) {
    u32x4_t xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = k + u * 32 < kw;
      if constexpr (X_LDS) xv[u] = (in && row_ok) ? *(const u32x4_t*)(xl + k + u * 32) : z4;
      else xv[u] = (in && row_ok) ? *(const u32x4_t*)(xg + k + u * 32) : z4;
    }
    u32x4_t wn[NG][U];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int u = 0; u < U; ++u) wn[q][u] = wload(q, k + 32 * U, u);
    if constexpr (PAIR) {
#pragma unroll
      for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int j = 0; j < U / 2; ++j) pair_to_operands(wv[q][2 * j], wv[q][2 * j + 1], pb);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NG; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, xv[u]), __builtin_bit_cast(bf16x8_t, wv[q][u]), acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int u = 0; u < U; ++u) wv[q][u] = wn[q][u];
  }
  // acc[q][r] = partial y[sequence 4 g + r][row n0 + 16 q + fr]
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[q][wave][4 * g + r][fr] = acc[q][r];
  __syncthreads();
  const int m = tid >> 4, n = tid & 15;
  if (m >= p.M) return;
  // DEFERRED RMSNorm (round 6, plain x only): x is bf16(stream * norm weight) written by the producer launch (xs_out below) and the
  // row's rsqrt(mean(stream^2) + eps) multiplies the PRODUCT here: rstd = rsqrt(sum of the producer's per-workgroup partial sums / K + eps),
  // summed in a fixed order (16 lanes x blocks / 16 terms each, then the 16 lanes) — the 64 norm launches of a decode step are gone.
  float rstd = 1.f;
  if (!X_LDS && p.ssq_in) {
    float sacc = 0.f;
    for (int b = n; b < p.ssq_in_blocks; b += 16) sacc += p.ssq_in[(int64_t)b * 8 + m];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 16);
    rstd = rsqrtf(sacc / (float)p.K + p.eps);
  }
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const int nn = n0 + 16 * q + n;
    const bool live = nn < p.N;
    float v = live ? ((part[q][0][m][n] + part[q][1][m][n]) + (part[q][2][m][n] + part[q][3][m][n])) * rstd : 0.f;
    if (p.act == GROVE_ACT_SWIGLU_PAIR) {
      // W rows interleaved [4 gate, 4 up] per 8 (ops.swiglu_interleave): row n (n & 4 == 0) is a gate row, row n + 4 its up row
      if (!live || (n & 4)) continue;
      const float u_ = ((part[q][0][m][n + 4] + part[q][1][m][n + 4]) + (part[q][2][m][n + 4] + part[q][3][m][n + 4])) * rstd;
      const float gt = bf2f(f2bf(v)), up = bf2f(f2bf(u_));
      const float o = gt * fast_sigmoid(gt) * up;
      const int col = (nn >> 3) * 4 + (n & 3);
      if (p.y_dtype == GROVE_BF16) ((bf16_raw*)p.y)[(int64_t)m * p.ldy + col] = f2bf(o);
      else ((float*)p.y)[(int64_t)m * p.ldy + col] = o;
      continue;
    }
    if (live) {
      if (p.bias) v += bf2f(((const bf16_raw*)p.bias)[nn]);
      v = act_apply(p.act, v);
      if (p.residual) v += p.res_f32 ? ((const float*)p.residual)[(int64_t)m * p.ldr + nn] : bf2f(((const bf16_raw*)p.residual)[(int64_t)m * p.ldr + nn]);
      if (p.y_dtype == GROVE_BF16) ((bf16_raw*)p.y)[(int64_t)m * p.ldy + nn] = f2bf(v);
      else ((float*)p.y)[(int64_t)m * p.ldy + nn] = v;
    }
    if (!X_LDS && p.ssq_out) {
      // the producer side of the deferred norm: the NEXT norm's input is the value just stored (the residual stream); its weighted bf16
      // rounding goes to xs_out, and this workgroup's sum of squares over its 16 columns of row m to ssq_out[16-row group][m]
      if (live && p.xs_out) ((bf16_raw*)p.xs_out)[(int64_t)m * p.ld_xs + nn] = f2bf(v * bf2f(((const bf16_raw*)p.xs_weight)[nn]));
      float sq = live ? v * v : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 16);
      if (n == 0) p.ssq_out[(int64_t)((n0 >> 4) + q) * 8 + m] = sq;
    }
  }
}

static int g_gemv_rw1 = 1;  // one output row per wave for N <= 4096 at M = 1 (o_proj 10.9 -> 10.5 us, down 20.4 -> 19.5; bit 2 of grove_gemv_set_mfma clears it)
static int g_gemv_mfma = 1;
static int g_gemv_pair = 1;    // 128-byte-per-row weight requests + lane-pair exchange in the plain-x matrix-core launches (grove_gemv_set_mfma bit 3 clears it: the A/B arm; same results bit for bit)
static int g_gemv_rows32 = 1;  // (grove_gemv_set_mfma bit 1 clears it: 16-row workgroups everywhere, the A/B arm)  // 0 = the VALU kernel for every M (A/B arm: grove_gemv_set_mfma)

template <int MX, int RW>
int launch_gemv_rw(const grove_gemv_params& p, hipStream_t s) {
  const size_t lds = (size_t)MX * p.K * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemv_kernel<MX, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);  // + the static reduction scratch
    attr_set = true;
  }
  const int rows_per_block = (GV_THREADS / 64) * RW;
  hipLaunchKernelGGL((gemv_kernel<MX, RW>), dim3((p.N + rows_per_block - 1) / rows_per_block), dim3(GV_THREADS), lds, s, p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
template <int MX>
int launch_gemv(const grove_gemv_params& p, hipStream_t s) {
  // 4 rows per wave amortise the x reads; below ~3 blocks per CU take 2 rows per wave for more loads in flight
  if (p.act == GROVE_ACT_SWIGLU_PAIR) return launch_gemv_rw<MX, 4>(p, s);  // (the pairing is laid out for 4 rows per wave)
  if (MX == 1 && p.N <= 4096 && g_gemv_rw1) return launch_gemv_rw<MX, 1>(p, s);  // one row per wave for the narrow projections: twice the workgroups
  return p.N >= 12288 ? launch_gemv_rw<MX, 4>(p, s) : launch_gemv_rw<MX, 2>(p, s);
}
}  // namespace

extern "C" int grove_gemv_set_mfma(int32_t on) {
  g_gemv_mfma = on != 0;
  g_gemv_rows32 = (on & 2) == 0;
  g_gemv_rw1 = (on & 4) == 0;
  g_gemv_pair = (on & 8) == 0;
  return GROVE_OK;
}

// 3..8 sequences: the matrix-core kernel (x in LDS for the folded prologues — MXS = 4 rows for M <= 4, else 8 — straight from global
// memory in plain mode). ONE predicate for the dispatcher and for callers that plan around it (grove_gemv_uses_mfma).
static bool gemv_takes_mfma(const grove_gemv_params& p) {
  const bool plain_x = p.x_mode == GROVE_GEMV_X_PLAIN && !p.x_f32;
  const size_t mxs = p.M <= 4 ? 4 : 8;
  return g_gemv_mfma && p.M >= (p.force_mfma ? 1 : 3) && p.K % 128 == 0 && (p.act != GROVE_ACT_SWIGLU_PAIR || p.N % 16 == 0) && (plain_x || mxs * (p.K + 32) * 2 <= 150 * 1024);
}
extern "C" int grove_gemv_uses_mfma(const grove_gemv_params* pp) { return pp && gemv_takes_mfma(*pp) ? 1 : 0; }

extern "C" int grove_gemv_bf16(const grove_gemv_params* pp, void* stream) {
  GROVE_CHECK(pp != nullptr, GROVE_E_SHAPE, "gemv: null params");
  const grove_gemv_params& p = *pp;
  GROVE_CHECK(p.M >= 1 && p.M <= 8, GROVE_E_SHAPE, "gemv: M=%d must be 1..8 (use grove_gemm_bf16 beyond)", p.M);
  GROVE_CHECK(p.N > 0 && p.K > 0 && p.K % 8 == 0, GROVE_E_SHAPE, "gemv: N=%d K=%d (K must be a multiple of 8)", p.N, p.K);
  const bool plain_x = p.x_mode == GROVE_GEMV_X_PLAIN && !p.x_f32;
  const bool mfma = gemv_takes_mfma(p);
  GROVE_CHECK(!(p.ssq_in || p.ssq_out) || (mfma && plain_x), GROVE_E_SHAPE,
              "gemv: the deferred RMSNorm (ssq_in / ssq_out) exists in the matrix-core kernel with a plain bf16 x only (K %% 128 == 0, M >= 3 or force_mfma)");
  GROVE_CHECK(!p.ssq_out || (p.act != GROVE_ACT_SWIGLU_PAIR && (!p.xs_out || (p.xs_weight && p.ld_xs >= p.N))), GROVE_E_SHAPE,
              "gemv: ssq_out / xs_out describe the value stored to y (no SWIGLU_PAIR); xs_out needs xs_weight and ld_xs >= N");
  GROVE_CHECK(!p.ssq_in || p.ssq_in_blocks > 0, GROVE_E_SHAPE, "gemv: ssq_in needs ssq_in_blocks (the producer's N / 16, rounded up)");
  GROVE_CHECK(mfma || (size_t)(p.M <= 2 ? p.M : p.M <= 4 ? 4 : 8) * p.K * 2 <= 159 * 1024, GROVE_E_SHAPE, "gemv: M*K=%d*%d does not fit the LDS", p.M, p.K);
  GROVE_CHECK(p.ldx % 8 == 0 && p.ldw % 8 == 0, GROVE_E_ALIGN, "gemv: ldx=%d ldw=%d must be multiples of 8", p.ldx, p.ldw);
  GROVE_CHECK(!p.x_f32 || p.x_mode != GROVE_GEMV_X_SWIGLU, GROVE_E_DTYPE, "gemv: x_mode swiglu reads a bf16 gate|up row");
  GROVE_CHECK(!p.x_f32 || (p.ldx % 4 == 0 && ((uintptr_t)p.x & 15) == 0), GROVE_E_ALIGN, "gemv: fp32 x rows must be 16-byte aligned");
  GROVE_CHECK(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.W & 15) == 0, GROVE_E_ALIGN, "gemv: x/W must be 16-byte aligned");
  GROVE_CHECK(p.y_dtype == GROVE_BF16 || p.y_dtype == GROVE_F32, GROVE_E_DTYPE, "gemv: bad y_dtype %d", p.y_dtype);
  GROVE_CHECK(p.x_mode >= GROVE_GEMV_X_PLAIN && p.x_mode <= GROVE_GEMV_X_SWIGLU, GROVE_E_SHAPE, "gemv: bad x_mode %d", p.x_mode);
  GROVE_CHECK(p.x_mode != GROVE_GEMV_X_RMSNORM || p.norm_weight, GROVE_E_SHAPE, "gemv: x_mode rmsnorm needs norm_weight");
  GROVE_CHECK(p.act != GROVE_ACT_SWIGLU_PAIR || (p.N % 16 == 0 && !p.bias && !p.residual), GROVE_E_SHAPE,
              "gemv: act SWIGLU_PAIR needs N %% 16 == 0 (whole wave pairs), no bias, no residual");
  hipStream_t s = (hipStream_t)stream;
  if (mfma) {
    const dim3 grid((p.N + 15) / 16);
    // the widest projections (gate | up, lm_head: >= 2.7 workgroups of 32 rows per CU): two row groups per workgroup share the x fragments
    // (M = 8, tools/bench_gemv.py 8: N = 22016 44.3 -> 41.2 us, N = 32008 59.9 -> 54.1; N = 12288 = 1.5 workgroups per CU: 24.1 -> 27.9, so not there)
    const bool pair = plain_x && g_gemv_pair && p.K % 256 == 0;  // (a wave's quarter of K in whole pairs of k-steps)
    if (plain_x && g_gemv_rows32 && p.N >= 16384) {
      if (pair) hipLaunchKernelGGL((gemv_mfma_kernel<false, 8, 2, true>), dim3((p.N + 31) / 32), dim3(GV_THREADS), 0, s, p);
      else hipLaunchKernelGGL((gemv_mfma_kernel<false, 8, 2>), dim3((p.N + 31) / 32), dim3(GV_THREADS), 0, s, p);
    } else if (plain_x) {
      if (pair) hipLaunchKernelGGL((gemv_mfma_kernel<false, 8, 1, true>), grid, dim3(GV_THREADS), 0, s, p);
      else hipLaunchKernelGGL((gemv_mfma_kernel<false, 8>), grid, dim3(GV_THREADS), 0, s, p);
    } else {
      static bool attr_set = false;
      if (!attr_set) {
        hipFuncSetAttribute((const void*)gemv_mfma_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void*)gemv_mfma_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
      }
      if (p.M <= 4) hipLaunchKernelGGL((gemv_mfma_kernel<true, 4>), grid, dim3(GV_THREADS), (size_t)4 * (p.K + 32) * 2, s, p);
      else hipLaunchKernelGGL((gemv_mfma_kernel<true, 8>), grid, dim3(GV_THREADS), (size_t)8 * (p.K + 32) * 2, s, p);
    }
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  switch (p.M) {
    case 1: return launch_gemv<1>(p, s);
    case 2: return launch_gemv<2>(p, s);
    case 3: case 4: return launch_gemv<4>(p, s);
    default: return launch_gemv<8>(p, s);
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Fused decode attention: RoPE(q, k) + cache append + one-query attention. One block per (sequence, head).
// HBM-bound on the cache read (2 * S * H * hd * 2 B per layer per sequence) — and, with one block per head, LATENCY-bound: a CU
// streams what it keeps in flight per memory round trip. Round 3: DA_U = 16 rows in flight per lane (64 KB per block per trip
// instead of 16 KB: the 650-key cache of a head is three trips instead of eleven), the first trip of key rows issued BEFORE the
// position is known (rows below S_max are allocated and zero beyond the sequence, so the speculative read is safe and simply
// masked), the first trip of value rows issued before the softmax reductions, and the new token's own key / value taken from
// LDS instead of being read back from the cache row the kernel has just written (no fence between the append and the reads:
// the cached rows 0..t-1 do not depend on it).
// Scores: the hd/8 lanes of a group cover one key row (16 B each, contiguous), 256 / (hd/8) rows per pass; the partial dot
// products are summed across the group's lanes. Values: the same lane map; the partial sums are reduced through LDS.
// ----------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int DA_THREADS = 256;
constexpr int DA_MAXS = 4096;  // scores kept in LDS
constexpr int DA_U = 8;        // cache rows in flight per lane and trip

// One block = one (sequence, head, split). A CU pulls ~10 B / clock from HBM whatever it keeps in flight (6 TB/s over 256 CUs), so
// one block per head — 32 blocks for the 7B decoder at B = 1 — streams the 333 KB of a head's keys and values in ~15 us however the
// loads are arranged (measured: 16 -> 19 us per launch with 4, 8 or 16 rows in flight per lane, position-major or head-major cache).
// Round 3: the keys of a head are dealt to n_split blocks by passes of G rows (pass i -> split i % n_split), every block keeps its own
// softmax statistics (max m, sum l) and un-normalised output, and decode_attn_merge_kernel combines the n_split partial results —
// flash-decoding with the merge as a second, tiny launch (deterministic; no tickets, fences or atomics).
template <int HD>
__global__ __launch_bounds__(DA_THREADS) void decode_attn_kernel(const grove_decode_attn_params p) {
  __shared__ float q_s[HD];
  __shared__ float kn_s[HD];   // the new token's rotated key / value (fp32 of the bf16 values the cache row holds)
  __shared__ float vn_s[HD];
  __shared__ float sc[DA_MAXS + 64];  // (+ the new token's slot behind a full last pass)
  __shared__ float red[DA_THREADS / 64];
  __shared__ float part[16][HD + 1];
  const int tid = threadIdx.x;
  const int NS = p.n_split > 1 ? p.n_split : 1;
  const int bh = blockIdx.x / NS, sp = blockIdx.x - bh * NS;
  const int b = bh / p.H, h = bh - b * p.H;
  const int HH = p.H * HD;
  bf16_raw* qkv = (bf16_raw*)p.qkv + (int64_t)b * p.ld_qkv;
  // cache layout [B, 2 (keys, values), H, S_max, hd]: the rows of one head are contiguous
  bf16_raw* kc = (bf16_raw*)p.cache + (((int64_t)b * 2) * p.H + h) * p.S_max * HD;      // key row j at kc + j * HD
  bf16_raw* vc = (bf16_raw*)p.cache + (((int64_t)b * 2 + 1) * p.H + h) * p.S_max * HD;  // value row j at vc + j * HD
  constexpr int CPR = HD / 8;           // 16-byte chunks per row
  constexpr int G = DA_THREADS / CPR;   // rows per pass (16 for hd 128, 32 for hd 64)
  const int g = tid / CPR, c = tid - g * CPR;
  const bf16_raw* kbase = kc + c * 8;
  const bf16_raw* vbase = vc + c * 8;
  const int t = p.pos[b];  // the new token's position; cached keys 0..t-1 + the new one are visible
  const int npass = (t + G - 1) / G;                 // passes over the cached rows 0..t-1; this block takes sp, sp + NS, ...
  const int mine = npass > sp ? (npass - sp + NS - 1) / NS : 0;
  auto row_of = [&](int i) { return (sp + NS * i) * G + g; };  // row of my i-th pass for this lane group
  // first trip of key rows before the rotation (they do not depend on it)
  u32x4_t kv[DA_U];
#pragma unroll
  for (int u = 0; u < DA_U; ++u) {
    const int j = min(row_of(min(u, max(mine - 1, 0))), max(t - 1, 0));  // (clamped inside the sequence; masked below)
    kv[u] = *(const u32x4_t*)(kbase + (int64_t)j * HD);
  }
  // rotate q (-> LDS, fp32 of the bf16-rounded value, as the unfused path stores it); split 0 also rotates k, appends k | v to
  // the cache and keeps them in LDS (the new token's own score and value never come back from memory)
  if (tid < HD / 2) {
    const float inv_freq = powf(p.theta, -2.f * (float)tid / (float)HD);
    float sn, cs;
    sincosf((float)t * inv_freq, &sn, &cs);
    const bf16_raw* q = qkv + h * HD;
    const float q1 = bf2f(q[tid]), q2 = bf2f(q[tid + HD / 2]);
    q_s[tid] = bf2f(f2bf(q1 * cs - q2 * sn));
    q_s[tid + HD / 2] = bf2f(f2bf(q2 * cs + q1 * sn));
    if (sp == 0) {
      const bf16_raw* k = qkv + HH + h * HD;
      const float k1 = bf2f(k[tid]), k2 = bf2f(k[tid + HD / 2]);
      const bf16_raw r1 = f2bf(k1 * cs - k2 * sn), r2 = f2bf(k2 * cs + k1 * sn);
      kc[(int64_t)t * HD + tid] = r1;
      kc[(int64_t)t * HD + tid + HD / 2] = r2;
      kn_s[tid] = bf2f(r1);
      kn_s[tid + HD / 2] = bf2f(r2);
    }
  } else if (sp == 0 && tid < HD / 2 + HD / 8) {
    const int cc = tid - HD / 2;
    const u32x4_t vv = *(const u32x4_t*)(qkv + 2 * HH + h * HD + cc * 8);
    *(u32x4_t*)(vc + (int64_t)t * HD + cc * 8) = vv;
    const float f[8] = {bf_lo(vv.x), bf_hi(vv.x), bf_lo(vv.y), bf_hi(vv.y), bf_lo(vv.z), bf_hi(vv.z), bf_lo(vv.w), bf_hi(vv.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e) vn_s[cc * 8 + e] = f[e];
  }
  __syncthreads();
  float mx = -INFINITY;
  float qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] = q_s[c * 8 + e];
  auto dot8 = [&](const u32x4_t k4) {
    float s = bf_lo(k4.x) * qv[0];
    s = fmaf(bf_hi(k4.x), qv[1], s);
    s = fmaf(bf_lo(k4.y), qv[2], s); s = fmaf(bf_hi(k4.y), qv[3], s);
    s = fmaf(bf_lo(k4.z), qv[4], s); s = fmaf(bf_hi(k4.z), qv[5], s);
    s = fmaf(bf_lo(k4.w), qv[6], s); s = fmaf(bf_hi(k4.w), qv[7], s);
    return s;
  };
  auto group_sum = [&](float s) {  // sum over the group's lanes (CPR = 4, 8 or 16 consecutive lanes)
#pragma unroll
    for (int off = CPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
  };
  // scores of my rows: sc[i * G + g] for my pass i (local numbering)
  for (int i0 = 0; i0 < mine; i0 += DA_U) {
    if (i0 > 0) {
#pragma unroll
      for (int u = 0; u < DA_U; ++u) {
        const int j = min(row_of(min(i0 + u, mine - 1)), t - 1);
        kv[u] = *(const u32x4_t*)(kbase + (int64_t)j * HD);
      }
    }
#pragma unroll
    for (int u = 0; u < DA_U; ++u) {
      const int i = i0 + u;
      const float s = group_sum(dot8(kv[u])) * p.alpha;
      if (i < mine && row_of(i) < t) {
        if (c == 0) sc[i * G + g] = s;
        mx = fmaxf(mx, s);
      } else if (i < mine && c == 0) {
        sc[i * G + g] = -INFINITY;  // a row of my last pass beyond the sequence
      }
    }
  }
  // first trip of the value rows goes out now: the softmax reductions below run under its latency
  u32x4_t vv[DA_U];
#pragma unroll
  for (int u = 0; u < DA_U; ++u) {
    const int j = min(row_of(min(u, max(mine - 1, 0))), max(t - 1, 0));
    vv[u] = *(const u32x4_t*)(vbase + (int64_t)j * HD);
  }
  const int nloc = mine * G;   // local score slots (the new token's score goes to slot nloc of split 0)
  if (sp == 0 && g == 0) {     // the new token itself, from LDS
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(kn_s[c * 8 + e], qv[e], s);
    s = group_sum(s) * p.alpha;
    if (c == 0) sc[nloc] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_max<DA_THREADS>(mx, red);
  const int nsc = nloc + (sp == 0 ? 1 : 0);
  const float m_use = mx == -INFINITY ? 0.f : mx;  // a split without rows: every exp below is exp(-inf) = 0
  float sum = 0.f;
  for (int j = tid; j < nsc; j += DA_THREADS) {
    const float e = __expf(sc[j] - m_use);
    sc[j] = e;
    sum += e;
  }
  sum = block_sum<DA_THREADS>(sum, red);  // (its barriers also publish sc[])
  // o[d] = sum_j p_j v_j[d] over my rows: lane group g takes row g of each of my passes; chunk c = tid % (HD/8)
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto pv = [&](float pj, const u32x4_t v4) {
    o[0] = fmaf(pj, bf_lo(v4.x), o[0]); o[1] = fmaf(pj, bf_hi(v4.x), o[1]);
    o[2] = fmaf(pj, bf_lo(v4.y), o[2]); o[3] = fmaf(pj, bf_hi(v4.y), o[3]);
    o[4] = fmaf(pj, bf_lo(v4.z), o[4]); o[5] = fmaf(pj, bf_hi(v4.z), o[5]);
    o[6] = fmaf(pj, bf_lo(v4.w), o[6]); o[7] = fmaf(pj, bf_hi(v4.w), o[7]);
  };
  for (int i0 = 0; i0 < mine; i0 += DA_U) {
    if (i0 > 0) {
#pragma unroll
      for (int u = 0; u < DA_U; ++u) {
        const int j = min(row_of(min(i0 + u, mine - 1)), t - 1);
        vv[u] = *(const u32x4_t*)(vbase + (int64_t)j * HD);
      }
    }
#pragma unroll
    for (int u = 0; u < DA_U; ++u) {
      const int i = i0 + u;
      pv(i < mine ? sc[i * G + g] : 0.f, vv[u]);  // (rows beyond the sequence carry exp(-inf) = 0)
    }
  }
  if (sp == 0 && g == 0) {  // the new token's value, from LDS
    const float pj = sc[nloc];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, vn_s[c * 8 + e], o[e]);
  }
  // reduce the G row groups, 16 at a time, through the LDS scratch
  for (int base = 0; base < G; base += 16) {
    __syncthreads();
    if (g >= base && g < base + 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) part[g - base][c * 8 + e] = o[e];
    }
    __syncthreads();
    if (tid < HD) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += part[r][tid];
      if (base == 0) q_s[tid] = a;  // q is dead: reuse as the accumulator
      else q_s[tid] += a;
    }
  }
  __syncthreads();
  if (NS == 1) {
    if (tid < HD) ((bf16_raw*)p.out)[(int64_t)b * HH + h * HD + tid] = f2bf(q_s[tid] / sum);
  } else {  // partial result {m, l, o[HD]} of this split
    float* pr = (float*)p.partial + ((int64_t)bh * NS + sp) * (HD + 2);
    if (tid < HD) pr[2 + tid] = q_s[tid];
    if (tid == 0) {
      pr[0] = mx;
      pr[1] = sum;
    }
  }
}

// out[b, h, :] = sum_s exp(m_s - M) o_s / sum_s exp(m_s - M) l_s  over the n_split partial results of a head.
// All loads of a group of 8 splits are issued before the first use (independent addresses): a loop that loads, uses, loads pays one
// memory round trip per split — 5.5 us per launch for this 32-block kernel in the first version, most of the 12 us of the pair.
template <int HD>
__global__ __launch_bounds__(HD) void decode_attn_merge_kernel(const grove_decode_attn_params p) {
  const int bh = blockIdx.x, d = threadIdx.x;
  const int NS = p.n_split;
  const float* pr = (const float*)p.partial + (int64_t)bh * NS * (HD + 2);
  float M = -INFINITY, L = 0.f, o = 0.f;
  for (int s0 = 0; s0 < NS; s0 += 8) {
    float m[8], l[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* q = pr + (int64_t)min(s0 + u, NS - 1) * (HD + 2);
      m[u] = q[0], l[u] = q[1], v[u] = q[2 + d];
    }
    float Mg = M;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < NS) Mg = fmaxf(Mg, m[u]);
    const float corr = M == -INFINITY ? 0.f : __expf(M - Mg);  // (Mg is finite from the first group on: split 0 holds the new token)
    L *= corr, o *= corr;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w = (s0 + u < NS && m[u] != -INFINITY) ? __expf(m[u] - Mg) : 0.f;
      L = fmaf(w, l[u], L);
      o = fmaf(w, v[u], o);
    }
    M = Mg;
  }
  const int b = bh / p.H, h = bh - b * p.H;
  ((bf16_raw*)p.out)[(int64_t)b * p.H * HD + h * HD + d] = f2bf(o / L);
}
}  // namespace

// ----------------------------------------------------------------------------------------------------------------------
// One launch for everything HF's greedy loop does between two decoder steps (GenerationMixin greedy search as GROVE.py:418-422 drives
// it): argmax over the [V] logits of every sequence (first maximum), finished rows emit pad, a row finishes at eos, the token becomes
// the next input, the position advances, and the step's id / hidden row are filed under the step number (= pos - pos0: no counter
// shared between blocks). One block per sequence. Replaces nine small torch launches per generated token inside the captured step.
// ----------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int GP_THREADS = 1024;
__global__ __launch_bounds__(GP_THREADS) void greedy_pick_kernel(const grove_greedy_pick_params p) {
  __shared__ float vmax[GP_THREADS / 64];
  __shared__ int imax[GP_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lg = p.logits + (int64_t)b * p.ld_logits;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < p.V; i += GP_THREADS) {
    const float v = lg[i];
    if (v > best || (v == best && i < bi)) best = v, bi = i;  // (NaN never wins: torch.argmax would return it; logits are finite here)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
  }
  if (lane == 0) vmax[wave] = best, imax[wave] = bi;
  __syncthreads();
  const int step = p.pos[b] - p.pos0;  // (read by every thread before thread 0 advances it below: barrier first)
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < GP_THREADS / 64; ++w)
      if (vmax[w] > best || (vmax[w] == best && imax[w] < bi)) best = vmax[w], bi = imax[w];
    int nxt = p.finished[b] ? p.pad : bi;
    if (nxt == p.eos) p.finished[b] = 1;
    p.tok[b] = nxt;
    p.pos[b] = p.pos0 + step + 1;
    if (step >= 0 && step < p.max_steps) p.ids_out[(int64_t)b * p.ld_ids + step] = (int64_t)nxt;
  }
  if (step < 0 || step >= p.max_steps) return;
  if (p.hidden && p.hid_out) {
    const bf16_raw* src = (const bf16_raw*)p.hidden + (int64_t)b * p.H;
    bf16_raw* dst = (bf16_raw*)p.hid_out + ((int64_t)step * p.B + b) * p.H;
    for (int i = tid; i < p.H; i += GP_THREADS) dst[i] = src[i];
  }
  if (p.hidden_f32 && p.hid_out_f32) {
    const float* src = p.hidden_f32 + (int64_t)b * p.H;
    float* dst = p.hid_out_f32 + ((int64_t)step * p.B + b) * p.H;
    for (int i = tid; i < p.H; i += GP_THREADS) dst[i] = src[i];
  }
}
}  // namespace

extern "C" int grove_greedy_pick(const grove_greedy_pick_params* pp, void* stream) {
  GROVE_CHECK(pp && pp->logits && pp->finished && pp->tok && pp->pos && pp->ids_out && pp->B > 0 && pp->V > 0 && pp->max_steps > 0, GROVE_E_SHAPE,
              "greedy_pick: bad arguments");
  hipLaunchKernelGGL(greedy_pick_kernel, dim3(pp->B), dim3(GP_THREADS), 0, (hipStream_t)stream, *pp);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_decode_attn(const grove_decode_attn_params* pp, void* stream) {
  GROVE_CHECK(pp != nullptr, GROVE_E_SHAPE, "decode_attn: null params");
  const grove_decode_attn_params& p = *pp;
  GROVE_CHECK(p.B > 0 && p.H > 0 && (p.hd == 64 || p.hd == 128 || p.hd == 32), GROVE_E_SHAPE, "decode_attn: B=%d H=%d hd=%d (hd must be 32, 64 or 128)", p.B, p.H, p.hd);
  GROVE_CHECK(p.S_max > 0 && p.S_max <= DA_MAXS, GROVE_E_SHAPE, "decode_attn: S_max=%d must be <= %d", p.S_max, DA_MAXS);
  GROVE_CHECK(p.ld_qkv % 8 == 0 && ((uintptr_t)p.qkv & 15) == 0 && ((uintptr_t)p.cache & 15) == 0, GROVE_E_ALIGN, "decode_attn: qkv/cache alignment");
  GROVE_CHECK(p.n_split <= 1 || (p.partial != nullptr && p.n_split <= 64), GROVE_E_WORKSPACE,
              "decode_attn: n_split=%d needs the caller's partial-result buffer (B * H * n_split * (hd + 2) floats)", p.n_split);
  hipStream_t s = (hipStream_t)stream;
  const int NS = p.n_split > 1 ? p.n_split : 1;
  const dim3 grid(p.B * p.H * NS);
  if (p.hd == 128) hipLaunchKernelGGL(decode_attn_kernel<128>, grid, dim3(DA_THREADS), 0, s, p);
  else if (p.hd == 64) hipLaunchKernelGGL(decode_attn_kernel<64>, grid, dim3(DA_THREADS), 0, s, p);
  else hipLaunchKernelGGL(decode_attn_kernel<32>, grid, dim3(DA_THREADS), 0, s, p);
  GROVE_LAUNCH_CHECK();
  if (NS > 1) {
    if (p.hd == 128) hipLaunchKernelGGL(decode_attn_merge_kernel<128>, dim3(p.B * p.H), dim3(128), 0, s, p);
    else if (p.hd == 64) hipLaunchKernelGGL(decode_attn_merge_kernel<64>, dim3(p.B * p.H), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(decode_attn_merge_kernel<32>, dim3(p.B * p.H), dim3(32), 0, s, p);
    GROVE_LAUNCH_CHECK();
  }
  return GROVE_OK;
}
