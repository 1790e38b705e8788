// LayerNorm / RMSNorm forward + backward for gfx950. HBM-bound row kernels: one wave64 per row,
// 16-byte vector loads (8 bf16 per lane per step), row held in registers (C <= 4096), fp32 stats.
#include "common.h"

namespace {

constexpr int MAXCH_ALL = 8;  // 16-byte chunks per lane -> C <= 8 * 64 * 8 = 4096

template <int MAXCH>
struct RowRegs {
  float v[MAXCH][8];
};

template <int MAXCH>
__device__ __forceinline__ void load_row(const bf16_raw* __restrict__ x, int C, int lane, RowRegs<MAXCH>& r) {
  const int nch = C >> 3;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < nch) {
      const u32x4_t u = *(const u32x4_t*)(x + ch * 8);
      r.v[i][0] = bf_lo(u.x); r.v[i][1] = bf_hi(u.x); r.v[i][2] = bf_lo(u.y); r.v[i][3] = bf_hi(u.y);
      r.v[i][4] = bf_lo(u.z); r.v[i][5] = bf_hi(u.z); r.v[i][6] = bf_lo(u.w); r.v[i][7] = bf_hi(u.w);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[i][e] = 0.f;
    }
  }
}

__device__ __forceinline__ void store_chunk_bf16(bf16_raw* y, const float* o) {
  *(u32x4_t*)y = u32x4_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
}

// RES: the fp32 residual-stream form. The row that is normalised is v = res[row] (+ x[row], the bf16 output of the branch
// that just finished); v is written back to res (fp32: the stream is never rounded to bf16 between blocks — at 24-32 layers
// that rounding, not the MFMA arithmetic, was most of the distance to the fp32 oracle) and, optionally, its bf16 rounding to
// res_bf16 (what a backward pass or a consumer that needs the stream as a GEMM operand reads). y == NULL: stream update only.
template <bool RMS, int MAXCH, bool RES>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const grove_norm_params p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  RowRegs<MAXCH> r;
  const int nch = p.C >> 3;
  if constexpr (RES) {
    float* res = p.res + (int64_t)row * p.ld_res;
    if (p.x) {
      load_row((const bf16_raw*)p.x + (int64_t)row * p.ld_x, p.C, lane, r);
    } else {
#pragma unroll
      for (int i = 0; i < MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[i][e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + i * 64;
      if (ch < nch) {
        const f32x4_t a = *(const f32x4_t*)(res + ch * 8), b = *(const f32x4_t*)(res + ch * 8 + 4);
        r.v[i][0] += a[0]; r.v[i][1] += a[1]; r.v[i][2] += a[2]; r.v[i][3] += a[3];
        r.v[i][4] += b[0]; r.v[i][5] += b[1]; r.v[i][6] += b[2]; r.v[i][7] += b[3];
        if (p.x) {
          *(f32x4_t*)(res + ch * 8) = f32x4_t{r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]};
          *(f32x4_t*)(res + ch * 8 + 4) = f32x4_t{r.v[i][4], r.v[i][5], r.v[i][6], r.v[i][7]};
        }
        if (p.res_bf16) store_chunk_bf16((bf16_raw*)p.res_bf16 + (int64_t)row * p.C + ch * 8, r.v[i]);
      }
    }
    if (!p.y) return;
  } else {
    load_row((const bf16_raw*)p.x + (int64_t)row * p.ld_x, p.C, lane, r);
  }
  float mean = 0.f;
  if constexpr (!RMS) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += r.v[i][e];
    mean = wave_sum(s) / (float)p.C;
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (lane + i * 64 < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = r.v[i][e] - mean;
        ss += d * d;
      }
    }
  }
  const float var = wave_sum(ss) / (float)p.C;
  const float rstd = rsqrtf(var + p.eps);
  if (lane == 0) {
    if (p.mean) p.mean[row] = mean;
    if (p.rstd) p.rstd[row] = rstd;
  }
  int orow = row;
  if (p.out_idx) {
    orow = p.out_idx[row];
    if (orow < 0) return;
  }
  const bf16_raw* w = (const bf16_raw*)p.weight;
  const bf16_raw* b = (const bf16_raw*)p.bias;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < nch) {
      const u32x4_t wu = *(const u32x4_t*)(w + ch * 8);
      float wv[8] = {bf_lo(wu.x), bf_hi(wu.x), bf_lo(wu.y), bf_hi(wu.y), bf_lo(wu.z), bf_hi(wu.z), bf_lo(wu.w), bf_hi(wu.w)};
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (!RMS && b) {
        const u32x4_t bu = *(const u32x4_t*)(b + ch * 8);
        bv[0] = bf_lo(bu.x); bv[1] = bf_hi(bu.x); bv[2] = bf_lo(bu.y); bv[3] = bf_hi(bu.y);
        bv[4] = bf_lo(bu.z); bv[5] = bf_hi(bu.z); bv[6] = bf_lo(bu.w); bv[7] = bf_hi(bu.w);
      }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (r.v[i][e] - mean) * rstd * wv[e] + bv[e];
      if (p.y_dtype == GROVE_BF16) {
        store_chunk_bf16((bf16_raw*)p.y + (int64_t)orow * p.ld_y + ch * 8, o);
      } else {
        float* y = (float*)p.y + (int64_t)orow * p.ld_y + ch * 8;
        *(f32x4_t*)y = f32x4_t{o[0], o[1], o[2], o[3]};
        *(f32x4_t*)(y + 4) = f32x4_t{o[4], o[5], o[6], o[7]};
      }
    }
  }
}

// Backward. Each wave walks ROWS_PER_WAVE rows, keeping per-lane partial dweight/dbias in
// registers; the 4 waves of a block are combined through LDS and added to global with one
// float atomic per (block, channel).
// Rows per wave: 8 when per-lane dweight/dbias partials are carried across rows (fewer LDS/global combines); 1 otherwise —
// a frozen norm's backward is a pure stream and 2812 rows x 8 per wave would leave most of the chip idle.
template <bool need_dw>
constexpr int rows_per_wave() { return need_dw ? 8 : 1; }

template <bool RMS, int MAXCH, bool need_dw>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const grove_norm_bwd_params p, unsigned* det, const int rows_per_wave_rt) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;  // [2][C] when dweight requested
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nch = p.C >> 3;
  const bf16_raw* w = (const bf16_raw*)p.weight;
  float wv[MAXCH][8];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < nch) {
      const u32x4_t wu = *(const u32x4_t*)(w + ch * 8);
      wv[i][0] = bf_lo(wu.x); wv[i][1] = bf_hi(wu.x); wv[i][2] = bf_lo(wu.y); wv[i][3] = bf_hi(wu.y);
      wv[i][4] = bf_lo(wu.z); wv[i][5] = bf_hi(wu.z); wv[i][6] = bf_lo(wu.w); wv[i][7] = bf_hi(wu.w);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[i][e] = 0.f;
    }
  }
  float dw[MAXCH][8], db[MAXCH][8];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { dw[i][e] = 0.f; db[i][e] = 0.f; }

  // (need_dw: 8 rows per wave, more on long inputs — every block ends in one global atomic per channel and parameter, and 3,072 blocks
  // adding onto the same 512 addresses were most of the 111 us of the box decoder's key-side norms, 98,304 rows of 256)
  const int ROWS_PER_WAVE = rows_per_wave_rt;
  const int row_base = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int row = row_base + rr;
    if (row >= p.rows) break;
    int drow = row;
    if (p.in_idx) drow = p.in_idx[row];
    RowRegs<MAXCH> x, dy;
    load_row((const bf16_raw*)p.x + (int64_t)row * p.ld_x, p.C, lane, x);
    if (drow >= 0) {
      load_row((const bf16_raw*)p.dy + (int64_t)drow * p.ld_dy, p.C, lane, dy);
    } else {
#pragma unroll
      for (int i = 0; i < MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dy.v[i][e] = 0.f;
    }
    float mean = 0.f, rstd;
    if constexpr (RMS) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += x.v[i][e] * x.v[i][e];
      rstd = rsqrtf(wave_sum(ss) / (float)p.C + p.eps);
    } else {
      mean = p.mean[row];
      rstd = p.rstd[row];
    }
    // xhat, g = dy * w
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (x.v[i][e] - mean) * rstd;
          const float g = dy.v[i][e] * wv[i][e];
          s1 += g;
          s2 += g * xh;
          if (need_dw) {
            dw[i][e] += dy.v[i][e] * xh;
            db[i][e] += dy.v[i][e];
          }
          x.v[i][e] = xh;
          dy.v[i][e] = g;
        }
      }
    }
    s2 = wave_sum(s2) / (float)p.C;
    if constexpr (RMS) s1 = 0.f;
    else s1 = wave_sum(s1) / (float)p.C;
    bf16_raw* dx = (bf16_raw*)p.dx + (int64_t)row * p.ld_dx;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + i * 64;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (dy.v[i][e] - s1 - x.v[i][e] * s2);
        if (p.accumulate) {
          const u32x4_t u = *(const u32x4_t*)(dx + ch * 8);
          o[0] += bf_lo(u.x); o[1] += bf_hi(u.x); o[2] += bf_lo(u.y); o[3] += bf_hi(u.y);
          o[4] += bf_lo(u.z); o[5] += bf_hi(u.z); o[6] += bf_lo(u.w); o[7] += bf_hi(u.w);
        }
        store_chunk_bf16(dx + ch * 8, o);
      }
    }
  }
  if (need_dw) {
    // combine the 4 waves through LDS, then one atomic per channel per block
    for (int c = threadIdx.x; c < 2 * p.C; c += 256) red[c] = 0.f;
    __syncthreads();
    if (det) {  // deterministic mode: the four waves add to the LDS row one after the other
      for (int wv_turn = 0; wv_turn < 4; ++wv_turn) {
        if (wave == wv_turn) {
#pragma unroll
          for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                red[ch * 8 + e] += dw[i][e];
                red[p.C + ch * 8 + e] += db[i][e];
              }
            }
          }
        }
        __syncthreads();
      }
    } else {
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            atomicAdd(&red[ch * 8 + e], dw[i][e]);
            atomicAdd(&red[p.C + ch * 8 + e], db[i][e]);
          }
        }
      }
      __syncthreads();
    }
    det_block_enter(det);
    for (int c = threadIdx.x; c < p.C; c += 256) {
      atomicAdd(&p.dweight[c], red[c]);
      if (p.dbias) atomicAdd(&p.dbias[c], red[p.C + c]);
    }
    det_block_leave(det);
  }
}

// The backward of a FROZEN norm (no dweight): one row per wave, a pure stream. The row operands stay PACKED (bf16 pairs as
// loaded) and are unpacked where they are used, twice: 3 x 4 x MAXCH registers instead of 3 x 8 x MAXCH floats. The float
// form needed 276 registers at C = 4096 — one wave per SIMD, so the 2812 rows of a LLaMA norm went through the chip in three
// latency-bound rounds (50 us for 69 MB). The arithmetic and its order are those of norm_bwd_kernel.
template <bool RMS, int MAXCH>
__global__ __launch_bounds__(256) void norm_bwd_stream_kernel(const grove_norm_bwd_params p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int nch = p.C >> 3;
  int drow = row;
  if (p.in_idx) drow = p.in_idx[row];  // (in flight beside the x / weight loads; only the dy loads wait for it)
  const bf16_raw* xr = (const bf16_raw*)p.x + (int64_t)row * p.ld_x;
  const bf16_raw* wr = (const bf16_raw*)p.weight;
  const u32x4_t zero = u32x4_t{0u, 0u, 0u, 0u};
  u32x4_t xp[MAXCH], gp[MAXCH], wp[MAXCH];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + i * 64;
    xp[i] = ch < nch ? *(const u32x4_t*)(xr + ch * 8) : zero;
    wp[i] = ch < nch ? *(const u32x4_t*)(wr + ch * 8) : zero;
  }
  float mean = 0.f, rstd = 0.f;
  if constexpr (!RMS) {
    mean = p.mean[row];
    rstd = p.rstd[row];
  }
  const bf16_raw* dyr = (const bf16_raw*)p.dy + (int64_t)max(drow, 0) * p.ld_dy;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + i * 64;
    gp[i] = (ch < nch && drow >= 0) ? *(const u32x4_t*)(dyr + ch * 8) : zero;
  }
  auto elem = [](const u32x4_t& u, int e) {
    const unsigned w = e < 2 ? u.x : e < 4 ? u.y : e < 6 ? u.z : u.w;
    return (e & 1) ? bf_hi(w) : bf_lo(w);
  };
  if constexpr (RMS) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = elem(xp[i], e);
        ss += x * x;
      }
    rstd = rsqrtf(wave_sum(ss) / (float)p.C + p.eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) asm volatile("" : "+v"(xp[i]));
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (lane + i * 64 < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (elem(xp[i], e) - mean) * rstd;
        const float g = elem(gp[i], e) * elem(wp[i], e);
        s1 += g;
        s2 += g * xh;
      }
    }
  }
  s2 = wave_sum(s2) / (float)p.C;
  if constexpr (RMS) s1 = 0.f;
  else s1 = wave_sum(s1) / (float)p.C;
  // (opaque to the optimizer from here on: otherwise it keeps the unpacked xh / g of the first pass alive for the second)
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) asm volatile("" : "+v"(xp[i]), "+v"(gp[i]), "+v"(wp[i]));
  bf16_raw* dx = (bf16_raw*)p.dx + (int64_t)row * p.ld_dx;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (elem(xp[i], e) - mean) * rstd;
        const float g = elem(gp[i], e) * elem(wp[i], e);
        o[e] = rstd * (g - s1 - xh * s2);
      }
      if (p.accumulate) {
        const u32x4_t u = *(const u32x4_t*)(dx + ch * 8);
        o[0] += bf_lo(u.x); o[1] += bf_hi(u.x); o[2] += bf_lo(u.y); o[3] += bf_hi(u.y);
        o[4] += bf_lo(u.z); o[5] += bf_hi(u.z); o[6] += bf_lo(u.w); o[7] += bf_hi(u.w);
      }
      store_chunk_bf16(dx + ch * 8, o);
    }
  }
}

int check_fwd(const grove_norm_params* p, const char* name) {
  GROVE_CHECK(p && p->rows > 0 && p->C > 0, GROVE_E_SHAPE, "%s: bad shape", name);
  GROVE_CHECK(p->C % 8 == 0 && p->C <= MAXCH_ALL * 512, GROVE_E_SHAPE, "%s: C=%d must be a multiple of 8 and <= %d", name, p->C, MAXCH_ALL * 512);
  GROVE_CHECK(p->ld_x % 8 == 0 && p->ld_y % 8 == 0, GROVE_E_ALIGN, "%s: ld_x/ld_y must be multiples of 8", name);
  GROVE_CHECK(((uintptr_t)p->x & 15) == 0 && ((uintptr_t)p->y & 15) == 0 && ((uintptr_t)p->weight & 15) == 0, GROVE_E_ALIGN,
              "%s: pointers must be 16-byte aligned", name);
  GROVE_CHECK(p->res || (p->x && p->y && p->weight), GROVE_E_SHAPE, "%s: x, y and weight required", name);
  GROVE_CHECK(!p->res || (p->ld_res % 8 == 0 && ((uintptr_t)p->res & 15) == 0 && ((uintptr_t)p->res_bf16 & 15) == 0 && (!p->y || p->weight)),
              GROVE_E_ALIGN, "%s: residual stream must be 16-byte aligned with ld_res %% 8 == 0 (and y needs weight)", name);
  return GROVE_OK;
}

}  // namespace

#define NORM_FWD_DISPATCH_R(RMS, RES)                                                                      \
  do {                                                                                                     \
    dim3 g_((p->rows + 3) / 4), b_(256);                                                                   \
    hipStream_t s_ = (hipStream_t)stream;                                                                  \
    if (p->C <= 512) hipLaunchKernelGGL((norm_fwd_kernel<RMS, 1, RES>), g_, b_, 0, s_, *p);                \
    else if (p->C <= 1024) hipLaunchKernelGGL((norm_fwd_kernel<RMS, 2, RES>), g_, b_, 0, s_, *p);          \
    else if (p->C <= 2048) hipLaunchKernelGGL((norm_fwd_kernel<RMS, 4, RES>), g_, b_, 0, s_, *p);          \
    else hipLaunchKernelGGL((norm_fwd_kernel<RMS, 8, RES>), g_, b_, 0, s_, *p);                            \
  } while (0)
#define NORM_FWD_DISPATCH(RMS)                 \
  do {                                         \
    if (p->res) NORM_FWD_DISPATCH_R(RMS, true); \
    else NORM_FWD_DISPATCH_R(RMS, false);      \
  } while (0)

extern "C" int grove_layernorm_fwd(const grove_norm_params* p, void* stream) {
  int rc = check_fwd(p, "layernorm_fwd");
  if (rc) return rc;
  NORM_FWD_DISPATCH(false);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_rmsnorm_fwd(const grove_norm_params* p, void* stream) {
  int rc = check_fwd(p, "rmsnorm_fwd");
  if (rc) return rc;
  NORM_FWD_DISPATCH(true);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

static int norm_bwd_launch(const grove_norm_bwd_params* p, bool rms, void* stream) {
  GROVE_CHECK(p && p->rows > 0 && p->C > 0, GROVE_E_SHAPE, "norm_bwd: bad shape");
  GROVE_CHECK(p->C % 8 == 0 && p->C <= MAXCH_ALL * 512, GROVE_E_SHAPE, "norm_bwd: C=%d unsupported", p->C);
  GROVE_CHECK(p->ld_x % 8 == 0 && p->ld_dy % 8 == 0 && p->ld_dx % 8 == 0, GROVE_E_ALIGN, "norm_bwd: leading dims must be multiples of 8");
  GROVE_CHECK(rms || (p->mean && p->rstd), GROVE_E_SHAPE, "layernorm_bwd: saved mean/rstd required");
  int rpw = p->dweight ? rows_per_wave<true>() : rows_per_wave<false>();
  if (p->dweight) {  // about 768 blocks on long inputs (at most 64 rows per wave); short ones (the decoder's 576 token rows: 8 rows per wave were
                     // 18 blocks of eight serial rows, 54 us) spread over up to 256 blocks first
    const int want = (p->rows + 4 * 768 - 1) / (4 * 768);
    const int few = p->rows / 1024;
    rpw = want > 64 ? 64 : want > rpw ? want : few < 1 ? 1 : few < rpw ? few : rpw;
  }
  const int rows_per_block = 4 * rpw;
  const size_t lds = p->dweight ? (size_t)2 * p->C * sizeof(float) : 0;
  dim3 grid((p->rows + rows_per_block - 1) / rows_per_block);
  hipStream_t s_ = (hipStream_t)stream;
  const bool dw = p->dweight != nullptr;
  GROVE_CHECK(!(dw && p->C > 2048), GROVE_E_SHAPE, "norm_bwd: dweight supported for C <= 2048 only");
  unsigned* det = dw ? grove_det_ticket() : nullptr;
#define NB(RMS, NCH, DW) hipLaunchKernelGGL((norm_bwd_kernel<RMS, NCH, DW>), grid, dim3(256), lds, s_, *p, det, (DW) ? rpw : 1)
#define NB_C(RMS, DW)                 \
  do {                                \
    if (p->C <= 512) NB(RMS, 1, DW);  \
    else if (p->C <= 1024) NB(RMS, 2, DW); \
    else NB(RMS, 4, DW);              \
  } while (0)
#define NBS(RMS, NCH) hipLaunchKernelGGL((norm_bwd_stream_kernel<RMS, NCH>), grid, dim3(256), 0, s_, *p)
#define NBS_C(RMS)                     \
  do {                                 \
    if (p->C <= 512) NBS(RMS, 1);      \
    else if (p->C <= 1024) NBS(RMS, 2); \
    else if (p->C <= 2048) NBS(RMS, 4); \
    else NBS(RMS, 8);                  \
  } while (0)
  if (!dw) {
    if (rms) NBS_C(true); else NBS_C(false);
  } else if (p->C > 2048) {
    if (rms) NB(true, 8, false); else NB(false, 8, false);
  } else if (rms) {
    if (dw) NB_C(true, true); else NB_C(true, false);
  } else {
    if (dw) NB_C(false, true); else NB_C(false, false);
  }
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_layernorm_bwd(const grove_norm_bwd_params* p, void* stream) { return norm_bwd_launch(p, false, stream); }
extern "C" int grove_rmsnorm_bwd(const grove_norm_bwd_params* p, void* stream) { return norm_bwd_launch(p, true, stream); }
