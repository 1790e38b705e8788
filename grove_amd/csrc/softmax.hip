// Row softmax (fwd/bwd) over materialised attention scores, decomposed rel-pos terms of SAM
// attention, and rotary embedding. HBM-bound kernels: one wave64 per row, 16-byte accesses.
#include "common.h"

namespace {

// ---------------------------------------------------------------- softmax forward
// One wave per (batch, query) row. Lane l owns float4 chunks l, l+64, ... (NCH of them).
template <int NCH>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const grove_softmax_params p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nrows = (int64_t)p.batch * p.Lq;
  if (row >= nrows) return;
  const int b = (int)(row / p.Lq);
  const int i = (int)(row - (int64_t)b * p.Lq);
  const float* __restrict__ s = p.scores + row * p.ld_s;
  int lim = p.Lk;  // keys [0, lim) are visible
  if (p.causal) lim = min(lim, i + (p.Lk - p.Lq) + 1);
  if (p.kv_len) lim = min(lim, p.kv_len[b / p.heads]);
  const float* rel = p.rel ? p.rel + row * (p.rel_kh + p.rel_kw) : nullptr;

  float v[NCH][4];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j0 = (lane + c * 64) * 4;
    if (j0 < p.ld_s && j0 < p.Lk + 3) {
      const f32x4_t x = *(const f32x4_t*)(s + j0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        float t = x[e];
        if (j < lim) {
          if (rel) t += rel[j / p.rel_kw] + rel[p.rel_kh + j % p.rel_kw];
          mx = fmaxf(mx, t);
        } else {
          t = -INFINITY;
        }
        v[c][e] = t;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[c][e] = -INFINITY;
    }
  }
  mx = wave_max(mx);
  if (mx == -INFINITY) mx = 0.f;  // fully masked row -> all zeros
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = __expf(v[c][e] - mx);
      v[c][e] = t;
      sum += t;
    }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  bf16_raw* o = (bf16_raw*)p.probs + row * p.ld_p;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j0 = (lane + c * 64) * 4;
    if (j0 < p.ld_p) *(u32x2_t*)(o + j0) = u32x2_t{pack2bf(v[c][0] * inv, v[c][1] * inv), pack2bf(v[c][2] * inv, v[c][3] * inv)};
  }
}

// ---------------------------------------------------------------- softmax backward
template <int NCH>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const grove_softmax_bwd_params p) {
  // drel: each wave parks its (unscaled) dS row in LDS, then lane t < kh sums stripe t over kw and lane
  // kh + t sums column t over kh: (kh + kw) short reads instead of two LDS atomics per element.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rowbuf = (float*)smem_raw;  // [4][NCH * 256] when drel is requested
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nrows = (int64_t)p.batch * p.Lq;
  const bool active = row < nrows;
  const int nrel = p.rel_kh + p.rel_kw;
  float g[NCH][4], pr[NCH][4];
  float dot = 0.f;
  if (active) {
    const float* __restrict__ dp = p.dprobs + row * p.ld_s;
    const bf16_raw* __restrict__ pb = (const bf16_raw*)p.probs + row * p.ld_p;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j0 = (lane + c * 64) * 4;
      if (j0 < p.ld_s && j0 < p.ld_p && j0 < p.Lk) {
        const f32x4_t x = *(const f32x4_t*)(dp + j0);
        const u32x2_t u = *(const u32x2_t*)(pb + j0);
        pr[c][0] = bf_lo(u.x); pr[c][1] = bf_hi(u.x); pr[c][2] = bf_lo(u.y); pr[c][3] = bf_hi(u.y);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[c][e] = (j0 + e < p.Lk) ? x[e] : 0.f;
          if (j0 + e >= p.Lk) pr[c][e] = 0.f;
          dot += g[c][e] * pr[c][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[c][e] = 0.f; pr[c][e] = 0.f; }
      }
    }
  }
  dot = wave_sum(dot);
  float* myrow = rowbuf + wave * (NCH * 256);
  if (active) {
    bf16_raw* ds = (bf16_raw*)p.dscores + row * p.ld_p;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j0 = (lane + c * 64) * 4;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pr[c][e] * (g[c][e] - dot);
      if (p.drel) *(f32x4_t*)(myrow + j0) = f32x4_t{o[0], o[1], o[2], o[3]};
      if (j0 < p.ld_p) *(u32x2_t*)(ds + j0) = u32x2_t{pack2bf(o[0] * p.scale, o[1] * p.scale), pack2bf(o[2] * p.scale, o[3] * p.scale)};
    }
  }
  if (p.drel) {
    __syncthreads();
    if (active && lane < nrel) {
      float acc = 0.f;
      if (lane < p.rel_kh) {
        const float* r = myrow + lane * p.rel_kw;
        for (int w = 0; w < p.rel_kw; ++w) acc += r[w];
      } else {
        const float* r = myrow + (lane - p.rel_kh);
        for (int h = 0; h < p.rel_kh; ++h) acc += r[h * p.rel_kw];
      }
      p.drel[row * nrel + lane] = acc;
    }
  }
}

// ---------------------------------------------------------------- decomposed rel-pos
// grid: one block per (b, q) row; threads loop over (head, k) outputs.
__global__ __launch_bounds__(256) void relpos_fwd_kernel(const grove_relpos_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* qs = (float*)smem_raw;  // [heads * hd]
  const int L = p.qh * p.qw;
  const int row = blockIdx.x;  // b * L + q
  const int b = row / L, q = row - b * L;
  const int qy = q / p.qw, qx = q - qy * p.qw;
  const bf16_raw* qrow = (const bf16_raw*)p.q + (int64_t)row * p.ld_q;
  for (int t = threadIdx.x; t < p.heads * p.hd; t += blockDim.x) {
    const int h = t / p.hd, c = t - h * p.hd;
    qs[t] = bf2f(qrow[h * p.hd_stride + c]);
  }
  __syncthreads();
  const int nrel = p.kh + p.kw;
  for (int t = threadIdx.x; t < p.heads * nrel; t += blockDim.x) {
    const int h = t / nrel, k = t - h * nrel;
    const float* R = k < p.kh ? p.Rh + ((int64_t)qy * p.kh + k) * p.hd : p.Rw + ((int64_t)qx * p.kw + (k - p.kh)) * p.hd;
    const float* qv = qs + h * p.hd;
    float acc = 0.f;
    for (int c = 0; c < p.hd; ++c) acc += qv[c] * R[c];
    p.rel[(((int64_t)b * p.heads + h) * L + q) * nrel + k] = acc;
  }
}

__global__ __launch_bounds__(256) void relpos_bwd_kernel(const grove_relpos_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* dr = (float*)smem_raw;  // [heads * nrel]
  const int L = p.qh * p.qw;
  const int row = blockIdx.x;
  const int b = row / L, q = row - b * L;
  const int qy = q / p.qw, qx = q - qy * p.qw;
  const int nrel = p.kh + p.kw;
  for (int t = threadIdx.x; t < p.heads * nrel; t += blockDim.x) {
    const int h = t / nrel, k = t - h * nrel;
    dr[t] = p.rel[(((int64_t)b * p.heads + h) * L + q) * nrel + k];
  }
  __syncthreads();
  bf16_raw* dq = (bf16_raw*)p.dq + (int64_t)row * p.ld_q;
  for (int t = threadIdx.x; t < p.heads * p.hd; t += blockDim.x) {
    const int h = t / p.hd, c = t - h * p.hd;
    const float* d = dr + h * nrel;
    float acc = 0.f;
    for (int k = 0; k < p.kh; ++k) acc += d[k] * p.Rh[((int64_t)qy * p.kh + k) * p.hd + c];
    for (int k = 0; k < p.kw; ++k) acc += d[p.kh + k] * p.Rw[((int64_t)qx * p.kw + k) * p.hd + c];
    bf16_raw* dst = dq + h * p.hd_stride + c;
    *dst = f2bf(bf2f(*dst) + acc);
  }
}

// ---------------------------------------------------------------- rotary embedding
// thread = (row, group of 8 heads, 16-byte chunk c of the first half): the 8 angles pos * theta^(-2i/hd), i = 8c .. 8c+7, are
// evaluated once and reused for the group's heads; every access is 16 bytes (x[i] and its partner x[i + hd/2] live in two
// chunks of the same head). hd % 16 == 0; any hd with hd/2 % 8 != 0 takes the scalar kernel below.
constexpr int ROPE_HG = 8;
__global__ __launch_bounds__(256) void rope_vec_kernel(const grove_rope_params p) {
  const int half = p.hd >> 1, cph = half >> 3;  // chunks per half head
  const int ngroups = (p.nheads + ROPE_HG - 1) / ROPE_HG;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)p.rows * ngroups * cph) return;
  const int c = (int)(t % cph);
  const int hg = (int)((t / cph) % ngroups);
  const int row = (int)(t / ((int64_t)cph * ngroups));
  const float pos = (float)p.pos[row];
  float cs[8], sn[8];
  if (p.table) {  // cos | sin of this position from the caller's table (two 32-byte reads instead of 8 x (powf + sincosf))
    const float* t = p.table + (int64_t)p.pos[row] * p.hd + c * 8;
    const f32x4_t c0 = *(const f32x4_t*)t, c1 = *(const f32x4_t*)(t + 4), s0 = *(const f32x4_t*)(t + half), s1 = *(const f32x4_t*)(t + half + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cs[j] = c0[j], cs[4 + j] = c1[j];
      sn[j] = p.inverse ? -s0[j] : s0[j], sn[4 + j] = p.inverse ? -s1[j] : s1[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float inv_freq = powf(p.theta, -2.f * (float)(c * 8 + j) / (float)p.hd);
      sincosf(pos * inv_freq, &sn[j], &cs[j]);
      if (p.inverse) sn[j] = -sn[j];
    }
  }
  const int h0 = hg * ROPE_HG, h1 = min(p.nheads, h0 + ROPE_HG);
  bf16_raw* x = (bf16_raw*)p.x + (int64_t)row * p.ld + p.col0 + (int64_t)h0 * p.hd + c * 8;
  for (int h = h0; h < h1; ++h, x += p.hd) {
    const u32x4_t a = *(const u32x4_t*)x, b2 = *(const u32x4_t*)(x + half);
    const float x1[8] = {bf_lo(a.x), bf_hi(a.x), bf_lo(a.y), bf_hi(a.y), bf_lo(a.z), bf_hi(a.z), bf_lo(a.w), bf_hi(a.w)};
    const float x2[8] = {bf_lo(b2.x), bf_hi(b2.x), bf_lo(b2.y), bf_hi(b2.y), bf_lo(b2.z), bf_hi(b2.z), bf_lo(b2.w), bf_hi(b2.w)};
    float o1[8], o2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o1[j] = x1[j] * cs[j] - x2[j] * sn[j];
      o2[j] = x2[j] * cs[j] + x1[j] * sn[j];
    }
    *(u32x4_t*)x = u32x4_t{pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]), pack2bf(o1[4], o1[5]), pack2bf(o1[6], o1[7])};
    *(u32x4_t*)(x + half) = u32x4_t{pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]), pack2bf(o2[4], o2[5]), pack2bf(o2[6], o2[7])};
  }
}

// scalar form — thread = (row, i < hd/2); loops over heads. angle = pos * theta^(-2i/hd).
__global__ __launch_bounds__(256) void rope_kernel(const grove_rope_params p) {
  const int half = p.hd >> 1;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)p.rows * half) return;
  const int row = (int)(t / half), i = (int)(t - (int64_t)row * half);
  const float inv_freq = powf(p.theta, -2.f * (float)i / (float)p.hd);
  const float ang = (float)p.pos[row] * inv_freq;
  float sn, cs;
  sincosf(ang, &sn, &cs);
  if (p.inverse) sn = -sn;
  bf16_raw* x = (bf16_raw*)p.x + (int64_t)row * p.ld + p.col0 + i;
  for (int h = 0; h < p.nheads; ++h) {
    const float x1 = bf2f(x[0]), x2 = bf2f(x[half]);
    x[0] = f2bf(x1 * cs - x2 * sn);
    x[half] = f2bf(x2 * cs + x1 * sn);
    x += p.hd;
  }
}

}  // namespace

extern "C" int grove_softmax_fwd(const grove_softmax_params* p, void* stream) {
  GROVE_CHECK(p && p->batch > 0 && p->Lq > 0 && p->Lk > 0, GROVE_E_SHAPE, "softmax_fwd: bad shape");
  GROVE_CHECK(p->ld_s % 4 == 0 && p->ld_p % 4 == 0 && p->ld_s >= p->Lk && p->ld_p >= p->Lk, GROVE_E_ALIGN,
              "softmax_fwd: ld_s=%d ld_p=%d must be multiples of 4 and >= Lk=%d", p->ld_s, p->ld_p, p->Lk);
  GROVE_CHECK(p->ld_p <= 2048 && p->ld_s <= 2048, GROVE_E_SHAPE, "softmax_fwd: rows longer than 2048 unsupported");
  GROVE_CHECK(p->heads > 0, GROVE_E_SHAPE, "softmax_fwd: heads must be > 0");
  GROVE_CHECK(!p->rel || (p->rel_kh * p->rel_kw == p->Lk && p->rel_kh + p->rel_kw <= 64), GROVE_E_SHAPE, "softmax_fwd: rel dims mismatch");
  const int64_t nrows = (int64_t)p->batch * p->Lq;
  dim3 grid((unsigned)((nrows + 3) / 4));
  const int w = p->ld_p > p->ld_s ? p->ld_p : p->ld_s;
  const int nch = (w + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 1) hipLaunchKernelGGL((softmax_fwd_kernel<1>), grid, dim3(256), 0, s, *p);
  else if (nch <= 2) hipLaunchKernelGGL((softmax_fwd_kernel<2>), grid, dim3(256), 0, s, *p);
  else if (nch <= 3) hipLaunchKernelGGL((softmax_fwd_kernel<3>), grid, dim3(256), 0, s, *p);
  else if (nch <= 4) hipLaunchKernelGGL((softmax_fwd_kernel<4>), grid, dim3(256), 0, s, *p);
  else hipLaunchKernelGGL((softmax_fwd_kernel<8>), grid, dim3(256), 0, s, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_softmax_bwd(const grove_softmax_bwd_params* p, void* stream) {
  GROVE_CHECK(p && p->batch > 0 && p->Lq > 0 && p->Lk > 0, GROVE_E_SHAPE, "softmax_bwd: bad shape");
  GROVE_CHECK(p->ld_s % 4 == 0 && p->ld_p % 4 == 0 && p->ld_s >= p->Lk && p->ld_p >= p->Lk, GROVE_E_ALIGN, "softmax_bwd: bad leading dims");
  GROVE_CHECK(p->ld_p <= 2048 && p->ld_s <= 2048, GROVE_E_SHAPE, "softmax_bwd: rows longer than 2048 unsupported");
  GROVE_CHECK(!p->drel || (p->rel_kh * p->rel_kw == p->Lk && p->rel_kh + p->rel_kw <= 64), GROVE_E_SHAPE, "softmax_bwd: rel dims mismatch");
  const int64_t nrows = (int64_t)p->batch * p->Lq;
  dim3 grid((unsigned)((nrows + 3) / 4));
  const int w = p->ld_p > p->ld_s ? p->ld_p : p->ld_s;
  const int nch = (w + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 1) hipLaunchKernelGGL((softmax_bwd_kernel<1>), grid, dim3(256), p->drel ? (size_t)4 * 1 * 256 * sizeof(float) : 0, s, *p);
  else if (nch <= 2) hipLaunchKernelGGL((softmax_bwd_kernel<2>), grid, dim3(256), p->drel ? (size_t)4 * 2 * 256 * sizeof(float) : 0, s, *p);
  else if (nch <= 3) hipLaunchKernelGGL((softmax_bwd_kernel<3>), grid, dim3(256), p->drel ? (size_t)4 * 3 * 256 * sizeof(float) : 0, s, *p);
  else if (nch <= 4) hipLaunchKernelGGL((softmax_bwd_kernel<4>), grid, dim3(256), p->drel ? (size_t)4 * 4 * 256 * sizeof(float) : 0, s, *p);
  else hipLaunchKernelGGL((softmax_bwd_kernel<8>), grid, dim3(256), p->drel ? (size_t)4 * 8 * 256 * sizeof(float) : 0, s, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_relpos_fwd(const grove_relpos_params* p, void* stream) {
  GROVE_CHECK(p && p->batch > 0 && p->heads > 0 && p->hd > 0, GROVE_E_SHAPE, "relpos_fwd: bad shape");
  const int rows = p->batch * p->qh * p->qw;
  hipLaunchKernelGGL(relpos_fwd_kernel, dim3(rows), dim3(256), (size_t)p->heads * p->hd * sizeof(float), (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_relpos_bwd(const grove_relpos_params* p, void* stream) {
  GROVE_CHECK(p && p->batch > 0 && p->heads > 0 && p->hd > 0 && p->dq, GROVE_E_SHAPE, "relpos_bwd: bad shape");
  const int rows = p->batch * p->qh * p->qw;
  hipLaunchKernelGGL(relpos_bwd_kernel, dim3(rows), dim3(256), (size_t)p->heads * (p->kh + p->kw) * sizeof(float), (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_rope_inplace(const grove_rope_params* p, void* stream) {
  GROVE_CHECK(p && p->rows > 0 && p->nheads > 0 && p->hd > 0 && (p->hd & 1) == 0, GROVE_E_SHAPE, "rope: bad shape");
  const bool vec = (p->hd % 16 == 0) && (p->ld % 8 == 0) && (p->col0 % 8 == 0) && (((uintptr_t)p->x & 15) == 0);
  GROVE_CHECK(!p->table || (vec && ((uintptr_t)p->table & 15) == 0), GROVE_E_ALIGN, "rope: the cos | sin table needs the vector kernel (hd % 16 == 0, aligned x) and 16-byte alignment");
  if (vec) {
    const int64_t nv = (int64_t)p->rows * ((p->nheads + ROPE_HG - 1) / ROPE_HG) * (p->hd / 16);
    hipLaunchKernelGGL(rope_vec_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  const int64_t n = (int64_t)p->rows * (p->hd / 2);
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
