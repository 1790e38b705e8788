// Stem im2col, CLIP token pooling, cross entropy, the wavefront-reduced kernels of the SAM
// two-way decoder / box heads, GROVE box losses and the fused AdamW step (gfx950).
#include "common.h"

namespace {

// ---------------------------------------------------------------- patch im2col
// img [B, C, T, H, W] -> col [(b t py px), ld_col], k = (c, dy, dx)
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_raw* __restrict__ img, bf16_raw* __restrict__ col, int B, int C, int T, int H, int W, int P,
                                                     int ld_col) {
  const int nh = H / P, nw = W / P;
  const int64_t rows = (int64_t)B * T * nh * nw;
  const int K = C * P * P;
  const int64_t n = rows * ld_col;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t row = t / ld_col;
    const int k = (int)(t - row * ld_col);
    bf16_raw v = 0;
    if (k < K) {
      const int c = k / (P * P), rem = k - c * P * P, dy = rem / P, dx = rem - dy * P;
      const int px = (int)(row % nw);
      const int64_t r2 = row / nw;
      const int py = (int)(r2 % nh);
      const int64_t bt = r2 / nh;
      const int tt = (int)(bt % T), b = (int)(bt / T);
      v = img[((((int64_t)b * C + c) * T + tt) * H + (py * P + dy)) * W + (px * P + dx)];
    }
    col[t] = v;
  }
}

// ---------------------------------------------------------------- CLIP adaptive 3-D pool
// x [G*8, 577, C] (CLS at token 0 skipped) -> y [G, 8*8*9, C]; 24 -> 8 rows (exact 3-bins),
// 24 -> 9 cols (start = floor(i*24/9), end = ceil((i+1)*24/9)).
__global__ __launch_bounds__(256) void clip_pool_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y, int G, int C) {
  const int cpv = C >> 3;
  const int64_t n = (int64_t)G * 576 * cpv;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t % cpv) * 8;
    const int64_t tok = t / cpv;  // g*576 + (tt*8 + oh)*9 + ow
    const int ow = (int)(tok % 9);
    const int oh = (int)((tok / 9) % 8);
    const int tt = (int)((tok / 72) % 8);
    const int g = (int)(tok / 576);
    const int w0 = (ow * 24) / 9, w1 = ((ow + 1) * 24 + 8) / 9;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_raw* base = x + ((int64_t)(g * 8 + tt) * 577 + 1) * C + c;
    for (int h = oh * 3; h < oh * 3 + 3; ++h)
      for (int w = w0; w < w1; ++w) {
        const u32x4_t u = *(const u32x4_t*)(base + (int64_t)(h * 24 + w) * C);
        acc[0] += bf_lo(u.x); acc[1] += bf_hi(u.x); acc[2] += bf_lo(u.y); acc[3] += bf_hi(u.y);
        acc[4] += bf_lo(u.z); acc[5] += bf_hi(u.z); acc[6] += bf_lo(u.w); acc[7] += bf_hi(u.w);
      }
    const float inv = 1.f / (float)(3 * (w1 - w0));
    *(u32x4_t*)(y + tok * C + c) = u32x4_t{pack2bf(acc[0] * inv, acc[1] * inv), pack2bf(acc[2] * inv, acc[3] * inv),
                                            pack2bf(acc[4] * inv, acc[5] * inv), pack2bf(acc[6] * inv, acc[7] * inv)};
  }
}

// ---------------------------------------------------------------- cross entropy
// one block per row; fp32 log-sum-exp; writes dlogits = (softmax - onehot) * grad_scale
__global__ __launch_bounds__(256) void ce_kernel(const bf16_raw* __restrict__ logits, const int32_t* __restrict__ labels, float* __restrict__ loss_sum,
                                                 bf16_raw* __restrict__ dlogits, const float* __restrict__ grad_scale, int V, int ld,
                                                 unsigned* det) {
  __shared__ float scratch[4];
  const int r = blockIdx.x;
  const bf16_raw* x = logits + (int64_t)r * ld;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < V; j += 256) mx = fmaxf(mx, bf2f(x[j]));
  mx = block_max<256>(mx, scratch);
  float s = 0.f;
  for (int j = threadIdx.x; j < V; j += 256) s += __expf(bf2f(x[j]) - mx);
  s = block_sum<256>(s, scratch);
  const int lab = labels[r];
  const float lse = mx + __logf(s);
  if (threadIdx.x == 0) {
    det_wait(det);
    atomicAdd(loss_sum, lse - bf2f(x[lab]));
    det_pass(det);
  }
  if (dlogits) {
    const float gs = grad_scale ? *grad_scale : 1.f;
    const float inv = 1.f / s;
    bf16_raw* d = dlogits + (int64_t)r * ld;
    for (int j = threadIdx.x; j < V; j += 256) {
      const float pj = __expf(bf2f(x[j]) - mx) * inv;
      d[j] = f2bf((pj - (j == lab ? 1.f : 0.f)) * gs);
    }
  }
}

// ---------------------------------------------------------------- small attention (decoder)
// "few queries": one wave per (instance, head), loops the <= 8 queries; lanes own keys.
constexpr int MAXD = 32;
constexpr int MAXQ = 8;
// element i of a bf16 or f32 array (the generic kernels serve the fp32 token path of the decoder: q_f32 / kv_f32 / o_f32)
__device__ __forceinline__ float ldx(const void* base, int64_t i, int f32) { return f32 ? ((const float*)base)[i] : bf2f(((const bf16_raw*)base)[i]); }
__device__ __forceinline__ void stx(void* base, int64_t i, float v, int f32) {
  if (f32) ((float*)base)[i] = v;
  else ((bf16_raw*)base)[i] = f2bf(v);
}

__global__ __launch_bounds__(64) void attn_fewq_fwd_kernel(const grove_small_attn_params p) {
  const int inst = blockIdx.x / p.heads, h = blockIdx.x - inst * p.heads;
  const int lane = threadIdx.x;
  const int d = p.d;
  const float scale = rsqrtf((float)d);
  const int qf = p.q_f32, kf = p.kv_f32, of = p.o_f32;
  const int64_t q0 = (int64_t)inst * p.Lq * p.ld_q + h * d, k0 = (int64_t)inst * p.Lk * p.ld_k + h * d;
  const int64_t v0 = (int64_t)inst * p.Lk * p.ld_v + h * d, o0 = (int64_t)inst * p.Lq * p.ld_o + h * d;
  for (int qi = 0; qi < p.Lq; ++qi) {
    float qv[MAXD];
#pragma unroll
    for (int c = 0; c < MAXD; ++c) qv[c] = c < d ? ldx(p.q, q0 + (int64_t)qi * p.ld_q + c, qf) * scale : 0.f;
    // pass 1: max
    float mx = -INFINITY;
    for (int j = lane; j < p.Lk; j += 64) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < MAXD; ++c)
        if (c < d) s += qv[c] * ldx(p.k, k0 + (int64_t)j * p.ld_k + c, kf);
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float l = 0.f;
    float acc[MAXD];
#pragma unroll
    for (int c = 0; c < MAXD; ++c) acc[c] = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < MAXD; ++c)
        if (c < d) s += qv[c] * ldx(p.k, k0 + (int64_t)j * p.ld_k + c, kf);
      const float e = __expf(s - mx);
      l += e;
#pragma unroll
      for (int c = 0; c < MAXD; ++c)
        if (c < d) acc[c] += e * ldx(p.v, v0 + (int64_t)j * p.ld_v + c, kf);
    }
    l = wave_sum(l);
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < MAXD; ++c) {
      if (c < d) {
        const float r = wave_sum(acc[c]) * inv;
        if (lane == 0) stx(p.o, o0 + (int64_t)qi * p.ld_o + c, r, of);
      }
    }
  }
}

// backward, few queries: lanes own keys -> dk/dv need no atomics; dq wave-reduced.
// dq, dk, dv are f32 [inst, L, heads*d] dense (ld = heads*d).
__global__ __launch_bounds__(64) void attn_fewq_bwd_kernel(const grove_small_attn_params p) {
  __shared__ float qs[MAXQ][MAXD], dos[MAXQ][MAXD], ms[MAXQ], ls[MAXQ], deltas[MAXQ];
  const int inst = blockIdx.x / p.heads, h = blockIdx.x - inst * p.heads;
  const int lane = threadIdx.x;
  const int d = p.d;
  const int HD = p.heads * d;
  const float scale = rsqrtf((float)d);
  const bf16_raw* q = (const bf16_raw*)p.q + (int64_t)inst * p.Lq * p.ld_q + h * d;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * d;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * d;
  const bf16_raw* o = (const bf16_raw*)p.o + (int64_t)inst * p.Lq * p.ld_o + h * d;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + (int64_t)inst * p.Lq * p.ld_o + h * d;
  for (int t = lane; t < p.Lq * d; t += 64) {
    const int qi = t / d, c = t - qi * d;
    qs[qi][c] = bf2f(q[(int64_t)qi * p.ld_q + c]);
    dos[qi][c] = bf2f(dO[(int64_t)qi * p.ld_o + c]);
  }
  __syncthreads();
  // softmax statistics + delta per query
  for (int qi = 0; qi < p.Lq; ++qi) {
    float mx = -INFINITY;
    for (int j = lane; j < p.Lk; j += 64) {
      float s = 0.f;
      for (int c = 0; c < d; ++c) s += qs[qi][c] * bf2f(k[(int64_t)j * p.ld_k + c]);
      mx = fmaxf(mx, s * scale);
    }
    mx = wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      float s = 0.f;
      for (int c = 0; c < d; ++c) s += qs[qi][c] * bf2f(k[(int64_t)j * p.ld_k + c]);
      l += __expf(s * scale - mx);
    }
    l = wave_sum(l);
    float dl = 0.f;
    for (int c = lane; c < d; c += 64) dl += dos[qi][c] * bf2f(o[(int64_t)qi * p.ld_o + c]);
    dl = wave_sum(dl);
    if (lane == 0) { ms[qi] = mx; ls[qi] = l; deltas[qi] = dl; }
  }
  __syncthreads();
  float dq[MAXQ][MAXD];
#pragma unroll
  for (int qi = 0; qi < MAXQ; ++qi)
#pragma unroll
    for (int c = 0; c < MAXD; ++c) dq[qi][c] = 0.f;
  float* dk = (float*)p.dk + (int64_t)inst * p.Lk * HD + h * d;
  float* dv = (float*)p.dv + (int64_t)inst * p.Lk * HD + h * d;
  for (int j = lane; j < p.Lk; j += 64) {
    float kv[MAXD], vv[MAXD], dkk[MAXD], dvv[MAXD];
#pragma unroll
    for (int c = 0; c < MAXD; ++c) {
      kv[c] = c < d ? bf2f(k[(int64_t)j * p.ld_k + c]) : 0.f;
      vv[c] = c < d ? bf2f(v[(int64_t)j * p.ld_v + c]) : 0.f;
      dkk[c] = 0.f;
      dvv[c] = 0.f;
    }
#pragma unroll
    for (int qi = 0; qi < MAXQ; ++qi) {
      if (qi < p.Lq) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < MAXD; ++c) {
          if (c < d) {
            s += qs[qi][c] * kv[c];
            dp += dos[qi][c] * vv[c];
          }
        }
        const float pr = __expf(s * scale - ms[qi]) / ls[qi];
        const float ds = pr * (dp - deltas[qi]) * scale;
#pragma unroll
        for (int c = 0; c < MAXD; ++c) {
          if (c < d) {
            dkk[c] += ds * qs[qi][c];
            dvv[c] += pr * dos[qi][c];
            dq[qi][c] += ds * kv[c];
          }
        }
      }
    }
    for (int c = 0; c < d; ++c) {
      dk[(int64_t)j * HD + c] = dkk[c];
      dv[(int64_t)j * HD + c] = dvv[c];
    }
  }
  float* dqo = (float*)p.dq + (int64_t)inst * p.Lq * HD + h * d;
#pragma unroll
  for (int qi = 0; qi < MAXQ; ++qi) {
    if (qi < p.Lq) {
#pragma unroll
      for (int c = 0; c < MAXD; ++c) {
        if (c < d) {
          const float r = wave_sum(dq[qi][c]);
          if (lane == 0) dqo[(int64_t)qi * HD + c] = r;
        }
      }
    }
  }
}

// "few keys" (Lk <= 8): thread per (instance, head, query); a wave = 64 consecutive queries of
// one (instance, head) so that k/v loads are wave-uniform broadcasts.
constexpr int MAXK = 8;
// ---- vectorised "few queries" kernels for head dim 16 (the decoder's token -> image cross attention: 6 queries x 1024 keys,
// internal dim 128 = 8 heads x 16). One block of 256 threads per (instance, head); a thread owns keys tid, tid + 256, ... and
// reads each K / V row ONCE with two 16-byte loads (the generic kernels above read every element 2 x Lq times with 2-byte
// loads: 0.5-0.7 ms per launch for 0.3 GFLOP); all Lq queries are evaluated against a key while it sits in registers, with a
// per-thread online softmax that is merged across the block at the end.
constexpr int FQ_T = 256;
// wave-wide sum / max on the VALU alone (DPP quad permutes and row mirrors, then the two cross-row swaps of gfx950): the merges below
// reduce ~100 values per thread, and __shfl_xor's ds_bpermute put 600 LDS-pipe instructions per wave in front of every store
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_v(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_f<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_f<0x141>(v);   // row_half_mirror: lanes i <-> 7 - i
  v += dpp_f<0x140>(v);   // row_mirror: lanes i <-> 15 - i
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max_v(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ void load_row16(const bf16_raw* p, float (&x)[16]) {
  const u32x4_t a = *(const u32x4_t*)p, b = *(const u32x4_t*)(p + 8);
  x[0] = bf_lo(a.x); x[1] = bf_hi(a.x); x[2] = bf_lo(a.y); x[3] = bf_hi(a.y); x[4] = bf_lo(a.z); x[5] = bf_hi(a.z); x[6] = bf_lo(a.w); x[7] = bf_hi(a.w);
  x[8] = bf_lo(b.x); x[9] = bf_hi(b.x); x[10] = bf_lo(b.y); x[11] = bf_hi(b.y); x[12] = bf_lo(b.z); x[13] = bf_hi(b.z); x[14] = bf_lo(b.w); x[15] = bf_hi(b.w);
}

// a gradient row piece of N consecutive elements at element offset `off` of an fp32 or (grove_small_attn_params.grad_bf16) bf16 array
template <int N>
__device__ __forceinline__ void store_grad(void* base, int64_t off, const float (&v)[N], bool as_bf16) {
  if (as_bf16) {
    bf16_raw* o = (bf16_raw*)base + off;
#pragma unroll
    for (int c = 0; c < N; c += 8)
      *(u32x4_t*)(o + c) = u32x4_t{pack2bf(v[c], v[c + 1]), pack2bf(v[c + 2], v[c + 3]), pack2bf(v[c + 4], v[c + 5]), pack2bf(v[c + 6], v[c + 7])};
  } else {
    float* o = (float*)base + off;
#pragma unroll
    for (int c = 0; c < N; c += 4) *(f32x4_t*)(o + c) = f32x4_t{v[c], v[c + 1], v[c + 2], v[c + 3]};
  }
}

template <int NQ>  // compile-time bound on Lq (6 on the path, 8 = MAXQ otherwise): sizes the per-thread accumulators
__global__ __launch_bounds__(FQ_T) void attn_fewq16_fwd_kernel(const grove_small_attn_params p) {
  __shared__ float qs[MAXQ][16];
  __shared__ float wm[FQ_T / 64][MAXQ], wred[FQ_T / 64][MAXQ][17];
  const int inst = blockIdx.x / p.heads, h = blockIdx.x - inst * p.heads;
  const int tid = threadIdx.x;
  const bf16_raw* q = (const bf16_raw*)p.q + (int64_t)inst * p.Lq * p.ld_q + h * 16;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * 16;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * 16;
  if (tid < p.Lq * 16) qs[tid >> 4][tid & 15] = bf2f(q[(int64_t)(tid >> 4) * p.ld_q + (tid & 15)]) * 0.25f;  // 16^-1/2
  __syncthreads();
  float m[NQ], l[NQ], acc[NQ][16];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    m[qi] = -INFINITY;
    l[qi] = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[qi][c] = 0.f;
  }
  for (int j = tid; j < p.Lk; j += FQ_T) {
    float kr[16], vr[16];
    load_row16(k + (int64_t)j * p.ld_k, kr);
    load_row16(v + (int64_t)j * p.ld_v, vr);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      if (qi < p.Lq) {
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sc = fmaf(qs[qi][c], kr[c], sc);
        const float mn = fmaxf(m[qi], sc);
        const float corr = __expf(m[qi] - mn), e = __expf(sc - mn);
        l[qi] = l[qi] * corr + e;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[qi][c] = fmaf(e, vr[c], acc[qi][c] * corr);
        m[qi] = mn;
      }
    }
  }
  // merge of the 256 per-thread softmax states: the wave reductions of ALL values first, then one trip through LDS per stage (two barriers
  // in all — one block_max / block_sum per value was 108 reductions x 2 barriers, most of this launch's 93 us).
  bf16_raw* o = (bf16_raw*)p.o + (int64_t)inst * p.Lq * p.ld_o + h * 16;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float mw = wave_max_v(m[qi]);
    if (lane == 0) wm[wave][qi] = mw;
  }
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    if (qi < p.Lq) {
      float M = wm[0][qi];
#pragma unroll
      for (int i = 1; i < FQ_T / 64; ++i) M = fmaxf(M, wm[i][qi]);
      const float w = m[qi] == -INFINITY ? 0.f : __expf(m[qi] - M);
      const float lw = wave_sum_v(l[qi] * w);
      if (lane == 0) wred[wave][qi][16] = lw;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float r = wave_sum_v(acc[qi][c] * w);
        if (lane == 0) wred[wave][qi][c] = r;
      }
    }
  }
  __syncthreads();
  if (tid < p.Lq * 16) {
    const int qi = tid >> 4, c = tid & 15;
    float L = 0.f, r = 0.f;
#pragma unroll
    for (int i = 0; i < FQ_T / 64; ++i) { L += wred[i][qi][16]; r += wred[i][qi][c]; }
    o[(int64_t)qi * p.ld_o + c] = f2bf(r * (1.f / L));
  }
}

template <int NQ>
__global__ __launch_bounds__(FQ_T) void attn_fewq16_bwd_kernel(const grove_small_attn_params p) {
  __shared__ float qs[MAXQ][16], dos[MAXQ][16], Ms[MAXQ], Ls[MAXQ], deltas[MAXQ];
  __shared__ float wm[FQ_T / 64][MAXQ], wred[FQ_T / 64][MAXQ][17];
  const int inst = blockIdx.x / p.heads, h = blockIdx.x - inst * p.heads;
  const int tid = threadIdx.x;
  const int HD = p.heads * 16;
  const bf16_raw* q = (const bf16_raw*)p.q + (int64_t)inst * p.Lq * p.ld_q + h * 16;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * 16;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * 16;
  const bf16_raw* o = (const bf16_raw*)p.o + (int64_t)inst * p.Lq * p.ld_o + h * 16;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + (int64_t)inst * p.Lq * p.ld_o + h * 16;
  if (tid < p.Lq * 16) {
    const int qi = tid >> 4, c = tid & 15;
    qs[qi][c] = bf2f(q[(int64_t)qi * p.ld_q + c]);
    dos[qi][c] = bf2f(dO[(int64_t)qi * p.ld_o + c]);
  }
  if (tid < p.Lq) {
    float dl = 0.f;
    for (int c = 0; c < 16; ++c) dl += bf2f(dO[(int64_t)tid * p.ld_o + c]) * bf2f(o[(int64_t)tid * p.ld_o + c]);
    deltas[tid] = dl;
  }
  __syncthreads();
  // pass 1: softmax statistics of every query (scores scaled by 16^-1/2 = 0.25)
  float m[NQ], l[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) { m[qi] = -INFINITY; l[qi] = 0.f; }
  for (int j = tid; j < p.Lk; j += FQ_T) {
    float kk[16];
    load_row16(k + (int64_t)j * p.ld_k, kk);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      if (qi < p.Lq) {
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sc = fmaf(qs[qi][c], kk[c], sc);
        sc *= 0.25f;
        const float mn = fmaxf(m[qi], sc);
        l[qi] = l[qi] * __expf(m[qi] - mn) + __expf(sc - mn);
        m[qi] = mn;
      }
    }
  }
  // (block-wide merges batched as in the forward: wave reductions of every value, then one LDS stage)
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float mw = wave_max_v(m[qi]);
    if (lane == 0) wm[wave][qi] = mw;
  }
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    if (qi < p.Lq) {
      float M = wm[0][qi];
#pragma unroll
      for (int i = 1; i < FQ_T / 64; ++i) M = fmaxf(M, wm[i][qi]);
      const float lw = wave_sum_v(m[qi] == -INFINITY ? 0.f : l[qi] * __expf(m[qi] - M));
      if (lane == 0) wred[wave][qi][16] = lw;
      if (tid == 0) Ms[qi] = M;
    }
  }
  __syncthreads();
  if (tid < p.Lq) {
    float L = 0.f;
#pragma unroll
    for (int i = 0; i < FQ_T / 64; ++i) L += wred[i][tid][16];
    Ls[tid] = L;
  }
  __syncthreads();
  // pass 2 (K re-read: an L2 hit): dK, dV rows of my keys (written once, 16-byte stores), dQ partials reduced over the block at the end
  float dq[NQ][16];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
    for (int c = 0; c < 16; ++c) dq[qi][c] = 0.f;
  const int64_t kv0 = (int64_t)inst * p.Lk * HD + h * 16;
  const bool gbf = p.grad_bf16 != 0;
  {
    for (int j = tid; j < p.Lk; j += FQ_T) {
      float kk[16], vv[16], dkk[16], dvv[16];
      load_row16(k + (int64_t)j * p.ld_k, kk);
      load_row16(v + (int64_t)j * p.ld_v, vv);
#pragma unroll
      for (int c = 0; c < 16; ++c) { dkk[c] = 0.f; dvv[c] = 0.f; }
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        if (qi < p.Lq) {
          float sc = 0.f, dp = 0.f;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            sc = fmaf(qs[qi][c], kk[c], sc);
            dp = fmaf(dos[qi][c], vv[c], dp);
          }
          const float pr = __expf(sc * 0.25f - Ms[qi]) / Ls[qi];
          const float ds = pr * (dp - deltas[qi]) * 0.25f;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            dkk[c] = fmaf(ds, qs[qi][c], dkk[c]);
            dvv[c] = fmaf(pr, dos[qi][c], dvv[c]);
            dq[qi][c] = fmaf(ds, kk[c], dq[qi][c]);
          }
        }
      }
      store_grad<16>(p.dk, kv0 + (int64_t)j * HD, dkk, gbf);
      store_grad<16>(p.dv, kv0 + (int64_t)j * HD, dvv, gbf);
    }
  }
  const int64_t dq0 = (int64_t)inst * p.Lq * HD + h * 16;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    if (qi < p.Lq) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float r = wave_sum_v(dq[qi][c]);
        if (lane == 0) wred[wave][qi][c] = r;
      }
    }
  }
  __syncthreads();
  if (tid < p.Lq * 16) {
    const int qi = tid >> 4, c = tid & 15;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < FQ_T / 64; ++i) r += wred[i][qi][c];
    if (gbf) ((bf16_raw*)p.dq)[dq0 + (int64_t)qi * HD + c] = f2bf(r);
    else ((float*)p.dq)[dq0 + (int64_t)qi * HD + c] = r;
  }
}

// ---- "tiny" attention: Lq, Lk <= 8 at head dim 32 — the self attention of the box decoder's 6 tokens per instance (transformer.py:
// 153-160; 8 heads x 32). The generic few-keys kernels gave it a 64-thread block per (instance, head) with 6 live lanes and 2-byte loads:
// 55 us forward, 159 us backward (two memsets, a wave reduction + two atomics per gradient element) for 0.3 MFLOP per instance. Here a
// lane is (pair, role): eight (instance, head) pairs per wave, role = the query (forward, dq) and then the key (dk, dv: plain stores,
// every element written once — no atomics, no memset, nothing run-to-run); rows travel as 16-byte loads.
__device__ __forceinline__ void load_row32(const bf16_raw* p, float (&x)[32]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x4_t a = *(const u32x4_t*)(p + 8 * i);
    x[8 * i + 0] = bf_lo(a.x); x[8 * i + 1] = bf_hi(a.x); x[8 * i + 2] = bf_lo(a.y); x[8 * i + 3] = bf_hi(a.y);
    x[8 * i + 4] = bf_lo(a.z); x[8 * i + 5] = bf_hi(a.z); x[8 * i + 6] = bf_lo(a.w); x[8 * i + 7] = bf_hi(a.w);
  }
}
__device__ __forceinline__ float dot32(const float (&a)[32], const float (&b)[32]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) s = fmaf(a[c], b[c], s);
  return s;
}

__global__ __launch_bounds__(64) void attn_tiny32_fwd_kernel(const grove_small_attn_params p) {
  const int pair = blockIdx.x * 8 + (threadIdx.x >> 3), role = threadIdx.x & 7;
  if (pair >= p.inst * p.heads || role >= p.Lq) return;
  const int inst = pair / p.heads, h = pair - inst * p.heads;
  const float scale = rsqrtf(32.f);
  float qv[32], row[32], s[MAXK];
  load_row32((const bf16_raw*)p.q + ((int64_t)inst * p.Lq + role) * p.ld_q + h * 32, qv);
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * 32;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * 32;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] = -INFINITY;
    if (j < p.Lk) {
      load_row32(k + (int64_t)j * p.ld_k, row);
      s[j] = dot32(qv, row) * scale;
      mx = fmaxf(mx, s[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] = j < p.Lk ? __expf(s[j] - mx) : 0.f;
    l += s[j];
  }
  const float inv = 1.f / l;
  float o[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < p.Lk) {
      load_row32(v + (int64_t)j * p.ld_v, row);
#pragma unroll
      for (int c = 0; c < 32; ++c) o[c] = fmaf(s[j], row[c], o[c]);
    }
  }
  bf16_raw* orow = (bf16_raw*)p.o + ((int64_t)inst * p.Lq + role) * p.ld_o + h * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *(u32x4_t*)(orow + 8 * i) = u32x4_t{pack2bf(o[8 * i] * inv, o[8 * i + 1] * inv), pack2bf(o[8 * i + 2] * inv, o[8 * i + 3] * inv),
                                        pack2bf(o[8 * i + 4] * inv, o[8 * i + 5] * inv), pack2bf(o[8 * i + 6] * inv, o[8 * i + 7] * inv)};
}

__global__ __launch_bounds__(64) void attn_tiny32_bwd_kernel(const grove_small_attn_params p) {
  __shared__ float Ps[8][MAXQ][MAXK + 1], dSs[8][MAXQ][MAXK + 1];
  const int pl = threadIdx.x >> 3, role = threadIdx.x & 7;
  const int pair = blockIdx.x * 8 + pl;
  const bool live = pair < p.inst * p.heads;
  const int inst = live ? pair / p.heads : 0, h = live ? pair - inst * p.heads : 0;
  const int HD = p.heads * 32;
  const float scale = rsqrtf(32.f);
  const bf16_raw* q = (const bf16_raw*)p.q + (int64_t)inst * p.Lq * p.ld_q + h * 32;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + (int64_t)inst * p.Lq * p.ld_o + h * 32;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * 32;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * 32;
  if (live && role < p.Lq) {  // this lane as QUERY `role`: P, dS of its row, dq
    float qv[32], dov[32], row[32], s[MAXK], dp[MAXK];
    load_row32(q + (int64_t)role * p.ld_q, qv);
    load_row32(dO + (int64_t)role * p.ld_o, dov);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
      s[j] = -INFINITY;
      dp[j] = 0.f;
      if (j < p.Lk) {
        load_row32(k + (int64_t)j * p.ld_k, row);
        s[j] = dot32(qv, row) * scale;
        mx = fmaxf(mx, s[j]);
        load_row32(v + (int64_t)j * p.ld_v, row);
        dp[j] = dot32(dov, row);
      }
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
      s[j] = j < p.Lk ? __expf(s[j] - mx) : 0.f;
      l += s[j];
    }
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
      s[j] /= l;
      delta += s[j] * dp[j];
    }
    float dq[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) dq[c] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
      if (j < p.Lk) {
        const float ds = s[j] * (dp[j] - delta) * scale;
        Ps[pl][role][j] = s[j];
        dSs[pl][role][j] = ds;
        load_row32(k + (int64_t)j * p.ld_k, row);
#pragma unroll
        for (int c = 0; c < 32; ++c) dq[c] = fmaf(ds, row[c], dq[c]);
      }
    }
    store_grad<32>(p.dq, ((int64_t)inst * p.Lq + role) * HD + h * 32, dq, p.grad_bf16 != 0);
  }
  __syncthreads();
  if (live && role < p.Lk) {  // this lane as KEY `role`: dk = sum_q dS[q][role] q_q, dv = sum_q P[q][role] dO_q
    float dk[32], dv[32], row[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
#pragma unroll
    for (int qi = 0; qi < MAXQ; ++qi) {
      if (qi < p.Lq) {
        const float a = dSs[pl][qi][role], b = Ps[pl][qi][role];
        load_row32(q + (int64_t)qi * p.ld_q, row);
#pragma unroll
        for (int c = 0; c < 32; ++c) dk[c] = fmaf(a, row[c], dk[c]);
        load_row32(dO + (int64_t)qi * p.ld_o, row);
#pragma unroll
        for (int c = 0; c < 32; ++c) dv[c] = fmaf(b, row[c], dv[c]);
      }
    }
    store_grad<32>(p.dk, ((int64_t)inst * p.Lk + role) * HD + h * 32, dk, p.grad_bf16 != 0);
    store_grad<32>(p.dv, ((int64_t)inst * p.Lk + role) * HD + h * 32, dv, p.grad_bf16 != 0);
  }
}

// does the tiny kernel pair take this problem? (bf16 everywhere, Lq, Lk <= 8, head dim 32, 16-byte rows)
static bool tiny32_applicable(const grove_small_attn_params* p, bool bwd) {
  if (p->q_f32 || p->kv_f32 || p->o_f32 || p->d != 32 || p->Lq > MAXQ || p->Lk > MAXK) return false;
  if (p->ld_q % 8 || p->ld_k % 8 || p->ld_v % 8 || p->ld_o % 8) return false;
  uintptr_t a = (uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->o;
  if (bwd) a |= (uintptr_t)p->d_o | (uintptr_t)p->dq | (uintptr_t)p->dk | (uintptr_t)p->dv;
  return (a & 15) == 0;
}

__global__ __launch_bounds__(64) void attn_fewk_fwd_kernel(const grove_small_attn_params p) {
  const int qblocks = (p.Lq + 63) / 64;
  const int qb = blockIdx.x % qblocks;
  const int ih = blockIdx.x / qblocks;
  const int inst = ih / p.heads, h = ih - inst * p.heads;
  const int qi = qb * 64 + threadIdx.x;
  if (qi >= p.Lq) return;
  const int d = p.d;
  const float scale = rsqrtf((float)d);
  const int qf = p.q_f32, kf = p.kv_f32, of = p.o_f32;
  const int64_t q0 = ((int64_t)inst * p.Lq + qi) * p.ld_q + h * d, k0 = (int64_t)inst * p.Lk * p.ld_k + h * d;
  const int64_t v0 = (int64_t)inst * p.Lk * p.ld_v + h * d, o0 = ((int64_t)inst * p.Lq + qi) * p.ld_o + h * d;
  float qv[MAXD];
#pragma unroll
  for (int c = 0; c < MAXD; ++c) qv[c] = c < d ? ldx(p.q, q0 + c, qf) * scale : 0.f;
  float s[MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] = -INFINITY;
    if (j < p.Lk) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < MAXD; ++c)
        if (c < d) a += qv[c] * ldx(p.k, k0 + (int64_t)j * p.ld_k + c, kf);
      s[j] = a;
      mx = fmaxf(mx, a);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] = j < p.Lk ? __expf(s[j] - mx) : 0.f;
    l += s[j];
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int c = 0; c < MAXD; ++c) {
    if (c < d) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < MAXK; ++j)
        if (j < p.Lk) a += s[j] * ldx(p.v, v0 + (int64_t)j * p.ld_v + c, kf);
      stx(p.o, o0 + c, a * inv, of);
    }
  }
}

__global__ __launch_bounds__(64) void attn_fewk_bwd_kernel(const grove_small_attn_params p, unsigned* det) {
  const int qblocks = (p.Lq + 63) / 64;
  const int qb = blockIdx.x % qblocks;
  const int ih = blockIdx.x / qblocks;
  const int inst = ih / p.heads, h = ih - inst * p.heads;
  const int qi = qb * 64 + threadIdx.x;
  const bool active = qi < p.Lq;
  const int qc = active ? qi : p.Lq - 1;
  const int d = p.d;
  const int HD = p.heads * d;
  const float scale = rsqrtf((float)d);
  const bf16_raw* q = (const bf16_raw*)p.q + ((int64_t)inst * p.Lq + qc) * p.ld_q + h * d;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + ((int64_t)inst * p.Lq + qc) * p.ld_o + h * d;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * d;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * d;
  float qv[MAXD], dov[MAXD];
#pragma unroll
  for (int c = 0; c < MAXD; ++c) {
    qv[c] = c < d ? bf2f(q[c]) : 0.f;
    dov[c] = (c < d && active) ? bf2f(dO[c]) : 0.f;
  }
  float s[MAXK], dp[MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] = -INFINITY;
    dp[j] = 0.f;
    if (j < p.Lk) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int c = 0; c < MAXD; ++c) {
        if (c < d) {
          a += qv[c] * bf2f(k[(int64_t)j * p.ld_k + c]);
          b += dov[c] * bf2f(v[(int64_t)j * p.ld_v + c]);
        }
      }
      s[j] = a * scale;
      dp[j] = b;
      mx = fmaxf(mx, s[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] = j < p.Lk ? __expf(s[j] - mx) : 0.f;
    l += s[j];
  }
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    s[j] /= l;
    delta += s[j] * dp[j];
  }
  float dqv[MAXD];
#pragma unroll
  for (int c = 0; c < MAXD; ++c) dqv[c] = 0.f;
  float* dk = (float*)p.dk + (int64_t)inst * p.Lk * HD + h * d;
  float* dv = (float*)p.dv + (int64_t)inst * p.Lk * HD + h * d;
  det_block_enter(det);
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < p.Lk) {
      const float ds = active ? s[j] * (dp[j] - delta) * scale : 0.f;
      const float pj = active ? s[j] : 0.f;
#pragma unroll
      for (int c = 0; c < MAXD; ++c) {
        if (c < d) {
          dqv[c] += ds * bf2f(k[(int64_t)j * p.ld_k + c]);
          const float a = wave_sum(ds * qv[c]);
          const float b = wave_sum(pj * dov[c]);
          if (threadIdx.x == 0) {
            atomicAdd(&dk[(int64_t)j * HD + c], a);
            atomicAdd(&dv[(int64_t)j * HD + c], b);
          }
        }
      }
    }
  }
  det_block_leave(det);
  if (active) {
    float* dqo = (float*)p.dq + ((int64_t)inst * p.Lq + qi) * HD + h * d;
    for (int c = 0; c < d; ++c) dqo[c] = dqv[c];
  }
}

// ---------------------------------------------------------------- box / objectness heads (fp32)
// one block of 256 threads (4 waves) per instance; each output neuron is a wave-reduced dot.
__global__ __launch_bounds__(256) void box_head_fwd_kernel(const grove_box_head_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* xs = (float*)smem_raw;  // [D]
  float* hs = xs + p.D;          // [D]
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = p.D;
  for (int c = threadIdx.x; c < D; c += 256) xs[c] = p.x[(int64_t)n * D + c];
  __syncthreads();
  const bf16_raw* W1 = (const bf16_raw*)p.W1;
  const bf16_raw* b1 = (const bf16_raw*)p.b1;
  if (D == 256 && (((uintptr_t)W1) & 15) == 0) {
    // the path's size (transformer_dim 256): a row of W1 is 32 lanes x 16 bytes, so a wave takes two rows per load and four loads per trip
    // (eight rows in flight), reduced over the half wave on the VALU — the loop below did one row per wave per trip with 2-byte loads and a
    // six-step ds_bpermute reduction: 64 serial trips, 110 us for 96 instances
    const int half = lane >> 5, l5 = lane & 31;
    float xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = xs[l5 * 8 + e];
    for (int j0 = wave * 64; j0 < wave * 64 + 64; j0 += 8) {
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const u32x4_t w = *(const u32x4_t*)(W1 + (int64_t)(j0 + 2 * u + half) * 256 + l5 * 8);
        a[u] = bf_lo(w.x) * xv[0] + bf_hi(w.x) * xv[1] + bf_lo(w.y) * xv[2] + bf_hi(w.y) * xv[3] + bf_lo(w.z) * xv[4] + bf_hi(w.z) * xv[5] +
               bf_lo(w.w) * xv[6] + bf_hi(w.w) * xv[7];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v = a[u];
        v += dpp_f<0xB1>(v);
        v += dpp_f<0x4E>(v);
        v += dpp_f<0x141>(v);
        v += dpp_f<0x140>(v);
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // the sum over this lane's 32-lane half
        const int j = j0 + 2 * u + half;
        v = fmaxf(v + bf2f(b1[j]), 0.f);
        if (l5 == 0) {
          hs[j] = v;
          if (p.hidden) p.hidden[(int64_t)n * D + j] = v;
        }
      }
    }
  } else {
    for (int j = wave; j < D; j += 4) {
      float a = 0.f;
      for (int c = lane; c < D; c += 64) a += bf2f(W1[(int64_t)j * D + c]) * xs[c];
      a = wave_sum(a) + bf2f(b1[j]);
      a = fmaxf(a, 0.f);
      if (lane == 0) {
        hs[j] = a;
        if (p.hidden) p.hidden[(int64_t)n * D + j] = a;
      }
    }
  }
  __syncthreads();
  const bf16_raw* W2 = (const bf16_raw*)p.W2;
  const bf16_raw* b2 = (const bf16_raw*)p.b2;
  {
    const int j = wave;  // 4 box outputs, one per wave
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += bf2f(W2[(int64_t)j * D + c]) * hs[c];
    a = wave_sum(a) + bf2f(b2[j]);
    if (lane == 0) p.box[(int64_t)n * 4 + j] = 1.f / (1.f + __expf(-a));
  }
  if (wave == 0 && p.obj) {
    const bf16_raw* Wo = (const bf16_raw*)p.Wo;
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += bf2f(Wo[c]) * xs[c];
    a = wave_sum(a) + bf2f(((const bf16_raw*)p.bo)[0]);
    if (lane == 0) p.obj[n] = a;
  }
}


// backward of the heads: one block per instance; weight gradients accumulated with f32 atomics.
__global__ __launch_bounds__(256) void box_head_bwd_kernel(const grove_box_head_bwd_params p, unsigned* det) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* xs = (float*)smem_raw;  // [D]
  float* hs = xs + p.D;          // [D] hidden (post-relu)
  float* dh = hs + p.D;          // [D]
  __shared__ float dz2[4];
  const int n = blockIdx.x;
  const int D = p.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < D; c += 256) {
    xs[c] = p.x[(int64_t)n * D + c];
    hs[c] = p.hidden[(int64_t)n * D + c];
  }
  if (threadIdx.x < 4) {
    const float b = p.box[(int64_t)n * 4 + threadIdx.x];
    dz2[threadIdx.x] = p.dbox[(int64_t)n * 4 + threadIdx.x] * b * (1.f - b);
  }
  __syncthreads();
  const float dobj = p.dobj ? p.dobj[n] : 0.f;
  const bf16_raw* W1 = (const bf16_raw*)p.W1;
  const bf16_raw* W2 = (const bf16_raw*)p.W2;
  const bf16_raw* Wo = (const bf16_raw*)p.Wo;
  det_block_enter(det);  // (each address gets one add per block: the instances take turns)
  // dW2, db2, dh
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a += bf2f(W2[(int64_t)j * D + c]) * dz2[j];
      atomicAdd(&p.dW2[(int64_t)j * D + c], dz2[j] * hs[c]);
    }
    a = hs[c] > 0.f ? a : 0.f;
    dh[c] = a;
    atomicAdd(&p.db1[c], a);
    if (p.dWo) atomicAdd(&p.dWo[c], dobj * xs[c]);
  }
  if (threadIdx.x < 4) atomicAdd(&p.db2[threadIdx.x], dz2[threadIdx.x]);
  if (threadIdx.x == 0 && p.dbo) atomicAdd(p.dbo, dobj);
  __syncthreads();
  // dW1[j, c] += dh[j] * x[c];   dx[c] = sum_j W1[j, c] * dh[j] + Wo[c] * dobj
  for (int j = wave; j < D; j += 4) {
    const float g = dh[j];
    if (g != 0.f)
      for (int c = lane; c < D; c += 64) atomicAdd(&p.dW1[(int64_t)j * D + c], g * xs[c]);
  }
  det_block_leave(det);
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = p.dobj ? bf2f(Wo[c]) * dobj : 0.f;
    for (int j = 0; j < D; ++j) a += bf2f(W1[(int64_t)j * D + c]) * dh[j];
    p.dx[(int64_t)n * D + c] = a;
  }
}

// ---------------------------------------------------------------- box losses (forward-mode AD)
struct Dual {
  float v, d[4];
};
__device__ __forceinline__ Dual dconst(float v) { return Dual{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual dvar(float v, int i) {
  Dual r = dconst(v);
  r.d[i] = 1.f;
  return r;
}
__device__ __forceinline__ Dual operator+(Dual a, Dual b) {
  Dual r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator-(Dual a, Dual b) {
  Dual r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator*(Dual a, Dual b) {
  Dual r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  Dual r; r.v = a.v / b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
  return r;
}
__device__ __forceinline__ Dual dmax(Dual a, Dual b) { return a.v >= b.v ? a : b; }  // torch.max: grad to first on ties
__device__ __forceinline__ Dual dmin(Dual a, Dual b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ Dual dscale(Dual a, float s) {
  Dual r; r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s;
  return r;
}

__global__ __launch_bounds__(256) void box_losses_kernel(const float* __restrict__ pb, const float* __restrict__ ol, const float* __restrict__ gb,
                                                         const float* __restrict__ vis, float* __restrict__ sums, float* __restrict__ dbox,
                                                         float* __restrict__ dobj, int N, float wb, float wo, unsigned* det) {
  __shared__ float scratch[4];
  float giou = 0.f, l1 = 0.f, bce = 0.f;
  const float eps = 1e-7f;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
    if (vis[n] > 0.5f) {
      const Dual cx = dvar(pb[n * 4 + 0], 0), cy = dvar(pb[n * 4 + 1], 1), w = dvar(pb[n * 4 + 2], 2), h = dvar(pb[n * 4 + 3], 3);
      const Dual x1 = cx - dscale(w, 0.5f), y1 = cy - dscale(h, 0.5f), x2 = cx + dscale(w, 0.5f), y2 = cy + dscale(h, 0.5f);
      const float gcx = gb[n * 4 + 0], gcy = gb[n * 4 + 1], gw = gb[n * 4 + 2], gh = gb[n * 4 + 3];
      const Dual x1g = dconst(gcx - gw / 2), y1g = dconst(gcy - gh / 2), x2g = dconst(gcx + gw / 2), y2g = dconst(gcy + gh / 2);
      const Dual xk1 = dmax(x1, x1g), yk1 = dmax(y1, y1g), xk2 = dmin(x2, x2g), yk2 = dmin(y2, y2g);
      Dual inter = dconst(0.f);
      if (yk2.v > yk1.v && xk2.v > xk1.v) inter = (xk2 - xk1) * (yk2 - yk1);
      const Dual uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter;
      const Dual iou = inter / (uni + dconst(eps));
      const Dual xc1 = dmin(x1, x1g), yc1 = dmin(y1, y1g), xc2 = dmax(x2, x2g), yc2 = dmax(y2, y2g);
      const Dual areac = (xc2 - xc1) * (yc2 - yc1);
      const Dual miou = iou - (areac - uni) / (areac + dconst(eps));
      giou += 1.f - miou.v;
#pragma unroll
      for (int i = 0; i < 4; ++i) g4[i] = -miou.d[i];
      const float pv[4] = {cx.v, cy.v, w.v, h.v};
      const float gv[4] = {gcx, gcy, gw, gh};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float df = pv[i] - gv[i];
        l1 += fabsf(df);
        g4[i] += df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
      }
    }
    if (dbox) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dbox[n * 4 + i] = g4[i] * wb;
    }
    if (ol) {
      const float z = ol[n], y = vis[n];
      // BCE with logits: max(z,0) - z*y + log(1 + exp(-|z|))
      bce += fmaxf(z, 0.f) - z * y + log1pf(__expf(-fabsf(z)));
      if (dobj) dobj[n] = (1.f / (1.f + __expf(-z)) - y) * wo;
    }
  }
  giou = block_sum<256>(giou, scratch);
  l1 = block_sum<256>(l1, scratch);
  bce = block_sum<256>(bce, scratch);
  if (threadIdx.x == 0) {
    det_wait(det);
    atomicAdd(sums + 0, giou);
    atomicAdd(sums + 1, l1);
    atomicAdd(sums + 2, bce);
    det_pass(det);
  }
}

// ---------------------------------------------------------------- AdamW
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, bf16_raw* __restrict__ model, const float* __restrict__ grad,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float gs, float bc1, float bc2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g = grad[i] * gs;
    float w = master[i];
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    w -= lr * wd * w;  // decoupled weight decay (torch.optim.AdamW / DeepSpeed FusedAdam adam_w_mode)
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    master[i] = w;
    if (model) model[i] = f2bf(w);
  }
}

// The same update over ALL trainable tensors in one launch: the fp32 state is one flat buffer (segment s occupies
// [seg_off[s], seg_off[s] + seg_len[s]), starts 16-byte aligned), only the bf16 model weights live in separate tensors.
// A block takes 1024 consecutive flat elements; the few elements of an alignment gap between two segments hold zeros in every
// state buffer and stay zero. Element-wise identical arithmetic to adamw_kernel.
__global__ __launch_bounds__(256) void adamw_multi_kernel(float* __restrict__ master, const float* __restrict__ grad, float* __restrict__ m,
                                                          float* __restrict__ v, const int64_t* __restrict__ seg_off,
                                                          const int64_t* __restrict__ seg_len, bf16_raw* const* __restrict__ model, int nseg,
                                                          int64_t total, float lr, float b1, float b2, float eps, float wd, float gs, float bc1,
                                                          float bc2, const float* __restrict__ sumsq, float clip, float* __restrict__ norm_out) {
  if (sumsq) {  // global-norm clipping from a DEVICE scalar (no host round trip): norm of the scaled gradient, DeepSpeed's clip_coef
    const float norm = sqrtf(*sumsq) * gs;
    if (clip > 0.f && norm > clip) gs *= clip / (norm + 1e-6f);
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  }
  for (int64_t c0 = (int64_t)blockIdx.x * 1024; c0 < total; c0 += (int64_t)gridDim.x * 1024) {
    // segment of the chunk's first element: last s with seg_off[s] <= c0 (uniform per block, scalar loads)
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg_off[mid] <= c0) lo = mid; else hi = mid - 1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = c0 + j * 256 + threadIdx.x;
      if (i >= total) break;
      int sg = lo;
      while (sg + 1 < nseg && seg_off[sg + 1] <= i) ++sg;
      const float g = grad[i] * gs;
      float w = master[i];
      const float mi = b1 * m[i] + (1.f - b1) * g;
      const float vi = b2 * v[i] + (1.f - b2) * g * g;
      m[i] = mi;
      v[i] = vi;
      w -= lr * wd * w;
      w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
      master[i] = w;
      const int64_t k = i - seg_off[sg];
      if (k < seg_len[sg] && model[sg]) model[sg][k] = f2bf(w);
    }
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, unsigned* det) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) {
    det_wait(det);
    atomicAdd(out, s);
    det_pass(det);
  }
}

inline dim3 cap_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

}  // namespace

extern "C" int grove_im2col_patch(const void* img, void* col, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W, int32_t P, int32_t ld_col,
                                  void* stream) {
  GROVE_CHECK(B > 0 && C > 0 && T > 0 && H % P == 0 && W % P == 0 && ld_col >= C * P * P, GROVE_E_SHAPE, "im2col_patch: bad shape");
  const int64_t n = (int64_t)B * T * (H / P) * (W / P) * ld_col;
  hipLaunchKernelGGL(im2col_kernel, cap_grid(n), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)img, (bf16_raw*)col, B, C, T, H, W, P, ld_col);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_clip_pool(const void* x, void* y, int32_t G, int32_t C, void* stream) {
  GROVE_CHECK(G > 0 && C > 0 && C % 8 == 0, GROVE_E_SHAPE, "clip_pool: bad shape");
  hipLaunchKernelGGL(clip_pool_kernel, cap_grid((int64_t)G * 576 * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x, (bf16_raw*)y, G, C);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_cross_entropy(const void* logits, const int32_t* labels, float* loss_sum, void* dlogits, const float* grad_scale, int32_t R,
                                   int32_t V, int32_t ld, void* stream) {
  GROVE_CHECK(R > 0 && V > 0 && ld >= V, GROVE_E_SHAPE, "cross_entropy: bad shape");
  hipLaunchKernelGGL(ce_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)logits, labels, loss_sum, (bf16_raw*)dlogits, grad_scale, V,
                     ld, grove_det_ticket());
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

// ---- vectorised "few keys" kernels for head dim 16 (the decoder's image -> token cross attention: 1024 queries x 6 keys).
// A block of 256 threads = 256 queries of one (instance, head); the keys' K / V rows sit in LDS; q / dO rows are read with
// 16-byte loads. Backward: every thread keeps its dS, P in LDS next to its q, dO row, then 2 x Lk x 16 threads each reduce one
// (key, column) of dK / dV over the block's queries — ONE atomic per (key, column) per block instead of a cross-lane
// reduction plus an atomic per wave (the generic kernel above).
constexpr int FK_T = 256;
__global__ __launch_bounds__(FK_T) void attn_fewk16_fwd_kernel(const grove_small_attn_params p) {
  __shared__ float ks[MAXK][16], vs[MAXK][16];
  const int qblocks = (p.Lq + FK_T - 1) / FK_T;
  const int qb = blockIdx.x % qblocks, ih = blockIdx.x / qblocks;
  const int inst = ih / p.heads, h = ih - inst * p.heads;
  const int tid = threadIdx.x;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * 16;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * 16;
  if (tid < p.Lk * 16) {
    ks[tid >> 4][tid & 15] = bf2f(k[(int64_t)(tid >> 4) * p.ld_k + (tid & 15)]);
    vs[tid >> 4][tid & 15] = bf2f(v[(int64_t)(tid >> 4) * p.ld_v + (tid & 15)]);
  }
  __syncthreads();
  const int qi = qb * FK_T + tid;
  if (qi >= p.Lq) return;
  float qv[16];
  load_row16((const bf16_raw*)p.q + ((int64_t)inst * p.Lq + qi) * p.ld_q + h * 16, qv);
  float sc[MAXK], mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = -INFINITY;
    if (j < p.Lk) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) a = fmaf(qv[c], ks[j][c], a);
      sc[j] = a * 0.25f;
      mx = fmaxf(mx, sc[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = j < p.Lk ? __expf(sc[j] - mx) : 0.f;
    l += sc[j];
  }
  const float inv = 1.f / l;
  float o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j)
    if (j < p.Lk) {
#pragma unroll
      for (int c = 0; c < 16; ++c) o[c] = fmaf(sc[j], vs[j][c], o[c]);
    }
  bf16_raw* op = (bf16_raw*)p.o + ((int64_t)inst * p.Lq + qi) * p.ld_o + h * 16;
  *(u32x4_t*)op = u32x4_t{pack2bf(o[0] * inv, o[1] * inv), pack2bf(o[2] * inv, o[3] * inv), pack2bf(o[4] * inv, o[5] * inv), pack2bf(o[6] * inv, o[7] * inv)};
  *(u32x4_t*)(op + 8) = u32x4_t{pack2bf(o[8] * inv, o[9] * inv), pack2bf(o[10] * inv, o[11] * inv), pack2bf(o[12] * inv, o[13] * inv), pack2bf(o[14] * inv, o[15] * inv)};
}

__global__ __launch_bounds__(FK_T) void attn_fewk16_bwd_kernel(const grove_small_attn_params p, unsigned* det) {
  __shared__ float ks[MAXK][16], vs[MAXK][16];
  __shared__ float qs[FK_T][17], dos[FK_T][17], dss[FK_T][MAXK + 1], pss[FK_T][MAXK + 1];
  const int qblocks = (p.Lq + FK_T - 1) / FK_T;
  const int qb = blockIdx.x % qblocks, ih = blockIdx.x / qblocks;
  const int inst = ih / p.heads, h = ih - inst * p.heads;
  const int tid = threadIdx.x;
  const int HD = p.heads * 16;
  const bf16_raw* k = (const bf16_raw*)p.k + (int64_t)inst * p.Lk * p.ld_k + h * 16;
  const bf16_raw* v = (const bf16_raw*)p.v + (int64_t)inst * p.Lk * p.ld_v + h * 16;
  if (tid < p.Lk * 16) {
    ks[tid >> 4][tid & 15] = bf2f(k[(int64_t)(tid >> 4) * p.ld_k + (tid & 15)]);
    vs[tid >> 4][tid & 15] = bf2f(v[(int64_t)(tid >> 4) * p.ld_v + (tid & 15)]);
  }
  __syncthreads();
  const int qi = qb * FK_T + tid;
  const bool active = qi < p.Lq;
  float qv[16], dov[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { qv[c] = 0.f; dov[c] = 0.f; }
  if (active) {
    load_row16((const bf16_raw*)p.q + ((int64_t)inst * p.Lq + qi) * p.ld_q + h * 16, qv);
    load_row16((const bf16_raw*)p.d_o + ((int64_t)inst * p.Lq + qi) * p.ld_o + h * 16, dov);
  }
  float sc[MAXK], dp[MAXK], mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = -INFINITY;
    dp[j] = 0.f;
    if (j < p.Lk) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        a = fmaf(qv[c], ks[j][c], a);
        b = fmaf(dov[c], vs[j][c], b);
      }
      sc[j] = a * 0.25f;
      dp[j] = b;
      mx = fmaxf(mx, sc[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = j < p.Lk ? __expf(sc[j] - mx) : 0.f;
    l += sc[j];
  }
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] /= l;
    delta += sc[j] * dp[j];
  }
  float dqv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) dqv[c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    const float ds = (active && j < p.Lk) ? sc[j] * (dp[j] - delta) * 0.25f : 0.f;
    dss[tid][j] = ds;
    pss[tid][j] = (active && j < p.Lk) ? sc[j] : 0.f;
    if (j < p.Lk) {
#pragma unroll
      for (int c = 0; c < 16; ++c) dqv[c] = fmaf(ds, ks[j][c], dqv[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) { qs[tid][c] = qv[c]; dos[tid][c] = dov[c]; }
  if (active) {
    float* dqo = (float*)p.dq + ((int64_t)inst * p.Lq + qi) * HD + h * 16;
#pragma unroll
    for (int c = 0; c < 16; c += 4) *(f32x4_t*)(dqo + c) = f32x4_t{dqv[c], dqv[c + 1], dqv[c + 2], dqv[c + 3]};
  }
  __syncthreads();
  // dK[j][c] = sum_q dS[q][j] q[q][c],  dV[j][c] = sum_q P[q][j] dO[q][c] over this block's queries
  det_block_enter(det);
  for (int t = tid; t < 2 * p.Lk * 16; t += FK_T) {
    const int which = t / (p.Lk * 16), r = t - which * p.Lk * 16;
    const int j = r >> 4, c = r & 15;
    float a = 0.f;
    if (which == 0) {
      for (int qq = 0; qq < FK_T; ++qq) a = fmaf(dss[qq][j], qs[qq][c], a);
      atomicAdd((float*)p.dk + ((int64_t)inst * p.Lk + j) * HD + h * 16 + c, a);
    } else {
      for (int qq = 0; qq < FK_T; ++qq) a = fmaf(pss[qq][j], dos[qq][c], a);
      atomicAdd((float*)p.dv + ((int64_t)inst * p.Lk + j) * HD + h * 16 + c, a);
    }
  }
  det_block_leave(det);
}


// ---------------------------------------------------------------- exact-fp32 small GEMM (decoder token path, text_hidden_fcs)
// C[M, N] (f32) = act(A[M, K] (f32) . W[N, K]^T (bf16 weights, exact in f32) + bias) + residual (f32), on v_mfma_f32_16x16x4_f32:
// products and sums in fp32 — no bf16 rounding of the activations. One wave per 16 x 16 output tile, four column tiles per
// block; a lane loads 16 bytes of its A row and 8 bytes of its W row per 16-deep K chunk and issues four MFMAs whose k-slot
// is (lane >> 4) * 4 + j for both operands. For M of a few hundred rows (6 tokens per box instance): the exact unit is 1/16 of
// the bf16 rate, which is irrelevant at ~1 GFLOP per decoder pass.
typedef __attribute__((ext_vector_type(4))) float f32x4v;
__global__ __launch_bounds__(256) void gemm_f32_kernel(const grove_gemm_f32_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 16;
  if (n0 >= p.N) return;
  const int am = min(m0 + i, p.M - 1), wn = min(n0 + i, p.N - 1);
  const float* a = p.A + (int64_t)am * p.lda + kq * 4;
  const bf16_raw* w = (const bf16_raw*)p.W + (int64_t)wn * p.ldw + kq * 4;
  f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    const f32x4_t av = *(const f32x4_t*)(a + k0);
    const u32x2_t wv = *(const u32x2_t*)(w + k0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bf_lo(wv.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bf_hi(wv.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bf_lo(wv.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bf_hi(wv.y), acc, 0, 0, 0);
  }
  // lane holds C[m0 + 4 kq + r][n0 + i]
  const int n = n0 + i;
  if (n >= p.N) return;
  const float b = p.bias ? bf2f(((const bf16_raw*)p.bias)[n]) : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * kq + r;
    if (m >= p.M) continue;
    float v = act_apply(p.act, acc[r] + b);
    if (p.residual) v += p.residual[(int64_t)m * p.ldr + n];
    p.C[(int64_t)m * p.ldc + n] = v;
    if (p.C_bf16) ((bf16_raw*)p.C_bf16)[(int64_t)m * p.ldc + n] = f2bf(v);
  }
}

static int g_small_attn_tiny = 1;  // 0: the generic few-keys kernels also for the 6 x 6 self attention (A/B arm and second implementation for the tests)
extern "C" int grove_small_attn_set_tiny(int32_t on) {
  g_small_attn_tiny = on != 0;
  return GROVE_OK;
}

static int small_attn_check(const grove_small_attn_params* p, const char* name) {
  GROVE_CHECK(p && p->inst > 0 && p->heads > 0 && p->Lq > 0 && p->Lk > 0, GROVE_E_SHAPE, "%s: bad shape", name);
  GROVE_CHECK(p->d > 0 && p->d <= MAXD, GROVE_E_SHAPE, "%s: head dim %d > %d", name, p->d, MAXD);
  GROVE_CHECK(p->Lk <= MAXK || p->Lq <= MAXQ, GROVE_E_SHAPE, "%s: needs Lq <= %d or Lk <= %d", name, MAXQ, MAXK);
  return GROVE_OK;
}

extern "C" int grove_gemm_f32(const grove_gemm_f32_params* p, void* stream) {
  GROVE_CHECK(p && p->M > 0 && p->N > 0 && p->K > 0 && p->A && p->W && p->C, GROVE_E_SHAPE, "gemm_f32: bad shape");
  GROVE_CHECK(p->K % 16 == 0 && p->lda % 4 == 0 && p->ldw % 4 == 0 && ((uintptr_t)p->A & 15) == 0 && ((uintptr_t)p->W & 7) == 0, GROVE_E_ALIGN,
              "gemm_f32: K must be a multiple of 16, rows 16-byte (A) / 8-byte (W) aligned");
  dim3 grid((p->N + 63) / 64, (p->M + 15) / 16);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_small_attn_fwd(const grove_small_attn_params* p, void* stream) {
  int rc = small_attn_check(p, "small_attn_fwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (p->q_f32 || p->kv_f32 || p->o_f32) {  // the decoder's fp32 token path: the generic kernels read / write either dtype
    if (p->Lk <= MAXK) hipLaunchKernelGGL(attn_fewk_fwd_kernel, dim3(p->inst * p->heads * ((p->Lq + 63) / 64)), dim3(64), 0, s, *p);
    else hipLaunchKernelGGL(attn_fewq_fwd_kernel, dim3(p->inst * p->heads), dim3(64), 0, s, *p);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  if (tiny32_applicable(p, false) && g_small_attn_tiny) {
    hipLaunchKernelGGL(attn_tiny32_fwd_kernel, dim3((p->inst * p->heads + 7) / 8), dim3(64), 0, s, *p);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  const bool vec16 = p->d == 16 && p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && p->ld_o % 8 == 0 &&
                     (((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->o) & 15) == 0;
  if (p->Lk <= MAXK && vec16 && p->Lq >= 64) {
    hipLaunchKernelGGL(attn_fewk16_fwd_kernel, dim3(p->inst * p->heads * ((p->Lq + FK_T - 1) / FK_T)), dim3(FK_T), 0, s, *p);
  } else if (p->Lk <= MAXK) {
    const int qblocks = (p->Lq + 63) / 64;
    hipLaunchKernelGGL(attn_fewk_fwd_kernel, dim3(p->inst * p->heads * qblocks), dim3(64), 0, s, *p);
  } else if (p->d == 16 && p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && (((uintptr_t)p->k | (uintptr_t)p->v) & 15) == 0) {
    if (p->Lq <= 6) hipLaunchKernelGGL(attn_fewq16_fwd_kernel<6>, dim3(p->inst * p->heads), dim3(FQ_T), 0, s, *p);
    else hipLaunchKernelGGL(attn_fewq16_fwd_kernel<MAXQ>, dim3(p->inst * p->heads), dim3(FQ_T), 0, s, *p);
  } else {
    hipLaunchKernelGGL(attn_fewq_fwd_kernel, dim3(p->inst * p->heads), dim3(64), 0, s, *p);
  }
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

// which backward kernel family takes the problem: 0 tiny32, 1 few keys (atomics), 2 few queries at head dim 16, 3 generic few queries
static int small_attn_bwd_family(const grove_small_attn_params* p) {
  if (tiny32_applicable(p, true) && g_small_attn_tiny) return 0;
  if (p->Lk <= MAXK) return 1;
  if (p->d == 16 && p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && (((uintptr_t)p->k | (uintptr_t)p->v) & 15) == 0 &&
      (((uintptr_t)p->dk | (uintptr_t)p->dv) & 15) == 0)
    return 2;
  return 3;
}
extern "C" int grove_small_attn_bwd_stores_bf16(const grove_small_attn_params* p) {
  if (!p || p->q_f32 || p->kv_f32 || p->o_f32 || p->inst <= 0 || (p->heads * p->d) % 8 != 0) return 0;
  grove_small_attn_params q = *p;  // (asked before the gradient arrays exist: judge the shape with aligned stand-ins)
  q.dq = q.dk = q.dv = (void*)16;
  if (!q.d_o) q.d_o = q.o;
  const int f = small_attn_bwd_family(&q);
  return (f == 0 || f == 2) ? 1 : 0;
}

extern "C" int grove_small_attn_bwd(const grove_small_attn_params* p, void* stream) {
  int rc = small_attn_check(p, "small_attn_bwd");
  if (rc) return rc;
  {
    const int f = small_attn_bwd_family(p);
    GROVE_CHECK(!p->grad_bf16 || ((f == 0 || f == 2) && (p->heads * p->d) % 8 == 0), GROVE_E_DTYPE,
                "small_attn_bwd: grad_bf16 only where every gradient element is stored once (ask grove_small_attn_bwd_stores_bf16)");
  }
  GROVE_CHECK(p->d_o && p->dq && p->dk && p->dv && p->o, GROVE_E_SHAPE, "small_attn_bwd: d_o/dq/dk/dv/o required");
  GROVE_CHECK(!(p->q_f32 || p->kv_f32 || p->o_f32), GROVE_E_DTYPE, "small_attn_bwd: bf16 operands only (the fp32 token path is inference-only)");
  hipStream_t s = (hipStream_t)stream;
  if (tiny32_applicable(p, true) && g_small_attn_tiny) {
    hipLaunchKernelGGL(attn_tiny32_bwd_kernel, dim3((p->inst * p->heads + 7) / 8), dim3(64), 0, s, *p);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  if (p->Lk <= MAXK) {
    // dk/dv accumulate with atomics -> zero them first
    const size_t bytes = (size_t)p->inst * p->Lk * p->heads * p->d * sizeof(float);
    hipError_t e = hipMemsetAsync(p->dk, 0, bytes, s);
    if (e == hipSuccess) e = hipMemsetAsync(p->dv, 0, bytes, s);
    GROVE_CHECK(e == hipSuccess, GROVE_E_HIP, "small_attn_bwd: memset failed");
    const bool vec16 = p->d == 16 && p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && p->ld_o % 8 == 0 &&
                       (((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->d_o | (uintptr_t)p->dq) & 15) == 0;
    if (vec16 && p->Lq >= 64) {
      hipLaunchKernelGGL(attn_fewk16_bwd_kernel, dim3(p->inst * p->heads * ((p->Lq + FK_T - 1) / FK_T)), dim3(FK_T), 0, s, *p, grove_det_ticket());
    } else {
      const int qblocks = (p->Lq + 63) / 64;
      hipLaunchKernelGGL(attn_fewk_bwd_kernel, dim3(p->inst * p->heads * qblocks), dim3(64), 0, s, *p, grove_det_ticket());
    }
  } else if (p->d == 16 && p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && (((uintptr_t)p->k | (uintptr_t)p->v) & 15) == 0 &&
             (((uintptr_t)p->dk | (uintptr_t)p->dv) & 15) == 0) {
    if (p->Lq <= 6) hipLaunchKernelGGL(attn_fewq16_bwd_kernel<6>, dim3(p->inst * p->heads), dim3(FQ_T), 0, s, *p);
    else hipLaunchKernelGGL(attn_fewq16_bwd_kernel<MAXQ>, dim3(p->inst * p->heads), dim3(FQ_T), 0, s, *p);
  } else {
    hipLaunchKernelGGL(attn_fewq_bwd_kernel, dim3(p->inst * p->heads), dim3(64), 0, s, *p);
  }
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_box_head_fwd(const grove_box_head_params* p, void* stream) {
  GROVE_CHECK(p && p->N > 0 && p->D > 0 && p->D <= 4096, GROVE_E_SHAPE, "box_head_fwd: bad shape");
  hipLaunchKernelGGL(box_head_fwd_kernel, dim3(p->N), dim3(256), (size_t)2 * p->D * sizeof(float), (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}


extern "C" int grove_box_head_bwd(const grove_box_head_bwd_params* p, void* stream) {
  GROVE_CHECK(p && p->N > 0 && p->D > 0 && p->D <= 4096, GROVE_E_SHAPE, "box_head_bwd: bad shape");
  GROVE_CHECK(p->dW1 && p->db1 && p->dW2 && p->db2 && p->dx && p->hidden && p->box && p->dbox, GROVE_E_SHAPE, "box_head_bwd: missing buffers");
  hipLaunchKernelGGL(box_head_bwd_kernel, dim3(p->N), dim3(256), (size_t)3 * p->D * sizeof(float), (hipStream_t)stream, *p, grove_det_ticket());
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_box_losses(const float* pred_box, const float* obj_logit, const float* gt_box, const float* visible, float* sums, float* dbox,
                                float* dobj, int32_t N, float w_box_over_ngt, float w_obj_over_n, void* stream) {
  GROVE_CHECK(N > 0 && pred_box && gt_box && visible && sums, GROVE_E_SHAPE, "box_losses: bad args");
  hipLaunchKernelGGL(box_losses_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred_box, obj_logit, gt_box, visible, sums, dbox, dobj,
                     N, w_box_over_ngt, w_obj_over_n, grove_det_ticket());
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_adamw_step(float* master, void* model_bf16, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, float grad_scale, int32_t step, void* stream) {
  GROVE_CHECK(n > 0 && step >= 1, GROVE_E_SHAPE, "adamw: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, cap_grid(n), dim3(256), 0, (hipStream_t)stream, master, (bf16_raw*)model_bf16, grad, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, grad_scale, bc1, bc2);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_adamw_step_multi(float* master, const float* grad, float* m, float* v, const int64_t* seg_off, const int64_t* seg_len,
                                      void* const* model_bf16, int32_t nseg, int64_t total, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, float grad_scale, int32_t step, const float* sumsq, float clip, float* norm_out,
                                      void* stream) {
  GROVE_CHECK(nseg > 0 && total > 0 && step >= 1 && seg_off && seg_len && model_bf16, GROVE_E_SHAPE, "adamw_multi: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  int64_t g = (total + 1023) / 1024;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, master, grad, m, v, seg_off, seg_len,
                     (bf16_raw* const*)model_bf16, nseg, total, lr, beta1, beta2, eps, weight_decay, grad_scale, bc1, bc2, sumsq, clip,
                     norm_out);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_sumsq_f32(const float* x, float* out, int64_t n, void* stream) {
  GROVE_CHECK(n > 0, GROVE_E_SHAPE, "sumsq: bad size");
  hipLaunchKernelGGL(sumsq_kernel, cap_grid(n), dim3(256), 0, (hipStream_t)stream, x, out, n, grove_det_ticket());
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
