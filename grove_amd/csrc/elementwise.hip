// Elementwise / data-movement kernels for gfx950. All HBM-bound: 16-byte accesses per lane,
// grid capped at 2048 blocks with a grid-stride loop (cdna_hip_programming.md Guideline 11/13).
#include "common.h"

namespace {

constexpr int EB = 256;
inline dim3 grid_for(int64_t nvec) {
  int64_t g = (nvec + EB - 1) / EB;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

__device__ __forceinline__ void unpack8(const u32x4_t u, float* f) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  return u32x4_t{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
}

// y[r, i] = silu(gu[r, i]) * gu[r, I + i]
__global__ __launch_bounds__(EB) void swiglu_fwd_kernel(const bf16_raw* __restrict__ gu, bf16_raw* __restrict__ y, int rows, int I) {
  const int ipv = I >> 3;
  const int64_t n = (int64_t)rows * ipv;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < n; t += (int64_t)gridDim.x * EB) {
    const int r = (int)(t / ipv), c = (int)(t - (int64_t)r * ipv) * 8;
    float g[8], u[8], o[8];
    unpack8(*(const u32x4_t*)(gu + (int64_t)r * 2 * I + c), g);
    unpack8(*(const u32x4_t*)(gu + (int64_t)r * 2 * I + I + c), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = g[e] * fast_sigmoid(g[e]) * u[e];
    *(u32x4_t*)(y + (int64_t)r * I + c) = pack8(o);
  }
}

__global__ __launch_bounds__(EB) void swiglu_bwd_kernel(const bf16_raw* __restrict__ gu, const bf16_raw* __restrict__ dy,
                                                        bf16_raw* __restrict__ dgu, int rows, int I) {
  const int ipv = I >> 3;
  const int64_t n = (int64_t)rows * ipv;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < n; t += (int64_t)gridDim.x * EB) {
    const int r = (int)(t / ipv), c = (int)(t - (int64_t)r * ipv) * 8;
    float g[8], u[8], d[8], dg[8], du[8];
    unpack8(*(const u32x4_t*)(gu + (int64_t)r * 2 * I + c), g);
    unpack8(*(const u32x4_t*)(gu + (int64_t)r * 2 * I + I + c), u);
    unpack8(*(const u32x4_t*)(dy + (int64_t)r * I + c), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) swiglu_bwd_elem(d[e], g[e], u[e], dg[e], du[e]);
    *(u32x4_t*)(dgu + (int64_t)r * 2 * I + c) = pack8(dg);
    *(u32x4_t*)(dgu + (int64_t)r * 2 * I + I + c) = pack8(du);
  }
}

__global__ __launch_bounds__(EB) void act_bwd_kernel(const bf16_raw* __restrict__ pre, const bf16_raw* __restrict__ dy,
                                                     bf16_raw* __restrict__ dx, int64_t nvec, int act) {
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < nvec; t += (int64_t)gridDim.x * EB) {
    float x[8], d[8], o[8];
    unpack8(*(const u32x4_t*)(pre + t * 8), x);
    unpack8(*(const u32x4_t*)(dy + t * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = d[e] * act_grad(act, x[e]);
    *(u32x4_t*)(dx + t * 8) = pack8(o);
  }
}

__global__ __launch_bounds__(EB) void act_fwd_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y, int64_t nvec, int act) {
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < nvec; t += (int64_t)gridDim.x * EB) {
    float v[8];
    unpack8(*(const u32x4_t*)(x + t * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_apply(act, v[e]);
    *(u32x4_t*)(y + t * 8) = pack8(v);
  }
}

// bilinear resize, align_corners = False (torch F.interpolate): fp32 planes [P, h, w] -> [P, H, W]; the source may be a crop
// (rows < h_use, columns < w_use) of a plane stored with pitch w / plane stride h * w.
__global__ __launch_bounds__(EB) void resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int h, int w, int h_use,
                                                             int w_use, int H, int W) {
  const float sy = (float)h_use / (float)H, sx = (float)w_use / (float)W;
  const int64_t n = (int64_t)P * H * W;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < n; t += (int64_t)gridDim.x * EB) {
    const int X = (int)(t % W), Y = (int)((t / W) % H), pl = (int)(t / ((int64_t)W * H));
    float fy = ((float)Y + 0.5f) * sy - 0.5f, fx = ((float)X + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = min((int)fy, h_use - 1), x0 = min((int)fx, w_use - 1);
    const int y1 = min(y0 + 1, h_use - 1), x1 = min(x0 + 1, w_use - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* s = src + (int64_t)pl * h * w;
    const float v00 = s[(int64_t)y0 * w + x0], v01 = s[(int64_t)y0 * w + x1], v10 = s[(int64_t)y1 * w + x0], v11 = s[(int64_t)y1 * w + x1];
    dst[t] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

__global__ __launch_bounds__(EB) void add_kernel(const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b, bf16_raw* __restrict__ y, int64_t nvec) {
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < nvec; t += (int64_t)gridDim.x * EB) {
    float x[8], z[8];
    unpack8(*(const u32x4_t*)(a + t * 8), x);
    if (b) {
      unpack8(*(const u32x4_t*)(b + t * 8), z);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += z[e];
    }
    *(u32x4_t*)(y + t * 8) = pack8(x);
  }
}

__global__ __launch_bounds__(EB) void add_bcast_kernel(const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b, bf16_raw* __restrict__ y,
                                                       int rows, int C, int period) {
  const int cpv = C >> 3;
  const int64_t n = (int64_t)rows * cpv;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < n; t += (int64_t)gridDim.x * EB) {
    const int r = (int)(t / cpv), c = (int)(t - (int64_t)r * cpv) * 8;
    float x[8], z[8];
    unpack8(*(const u32x4_t*)(a + (int64_t)r * C + c), x);
    unpack8(*(const u32x4_t*)(b + (int64_t)(r % period) * C + c), z);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += z[e];
    *(u32x4_t*)(y + (int64_t)r * C + c) = pack8(x);
  }
}

// 2^tpr_shift threads walk one row (rows per block = EB >> tpr_shift): the row indices are read once per row, no division per
// chunk, and a thread's chunks are independent loads — the flat form (index loads + a 64-bit division per 16 bytes, six
// dependent rounds per thread) filled SAM's 6432 pad rows of 7.5 KB at 1.4 TB/s.
__global__ __launch_bounds__(EB) void copy_rows_kernel(const grove_rows_params p, const int tpr_shift) {
  const int cpv = p.C >> 3;
  const int tpr = 1 << tpr_shift, rpb = EB >> tpr_shift;
  const int lr = threadIdx.x >> tpr_shift, lc = threadIdx.x & (tpr - 1);
  const bf16_raw* __restrict__ src = (const bf16_raw*)p.src;
  bf16_raw* __restrict__ dst = (bf16_raw*)p.dst;
  for (int r = blockIdx.x * rpb + lr; r < p.rows; r += gridDim.x * rpb) {
    const int sr = p.idx_src ? p.idx_src[r] : r;
    const int dr = p.idx_dst ? p.idx_dst[r] : r;
    if (dr < 0) continue;
    const bf16_raw* sp = src + (int64_t)max(sr, 0) * p.ld_src;
    bf16_raw* dp = dst + (int64_t)dr * p.ld_dst;
    for (int c = lc; c < cpv; c += tpr) {
      u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
      if (sr >= 0) v = *(const u32x4_t*)(sp + c * 8);
      if (p.accumulate) {
        float x[8], z[8];
        unpack8(v, x);
        unpack8(*(const u32x4_t*)(dp + c * 8), z);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += z[e];
        v = pack8(x);
      }
      *(u32x4_t*)(dp + c * 8) = v;
    }
  }
}

__global__ __launch_bounds__(EB) void scatter_add_kernel(const bf16_raw* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ idx,
                                                         int rows, int C, int ld_src, int ld_dst) {
  const int cp2 = C >> 1;
  const int64_t n = (int64_t)rows * cp2;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < n; t += (int64_t)gridDim.x * EB) {
    const int r = (int)(t / cp2), c = (int)(t - (int64_t)r * cp2) * 2;
    const int d = idx ? idx[r] : r;
    if (d < 0) continue;
    const unsigned u = *(const unsigned*)(src + (int64_t)r * ld_src + c);
    atomicAdd(dst + (int64_t)d * ld_dst + c, bf_lo(u));
    atomicAdd(dst + (int64_t)d * ld_dst + c + 1, bf_hi(u));
  }
}

// the same with fp32 source rows (the fp32-wire form of the sparse embedding-row gradient exchange)
__global__ __launch_bounds__(EB) void scatter_add_f32src_kernel(const float* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ idx,
                                                                int rows, int C, int ld_src, int ld_dst) {
  const int64_t n = (int64_t)rows * C;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < n; t += (int64_t)gridDim.x * EB) {
    const int r = (int)(t / C), c = (int)(t - (int64_t)r * C);
    const int d = idx ? idx[r] : r;
    if (d < 0) continue;
    atomicAdd(dst + (int64_t)d * ld_dst + c, src[(int64_t)r * ld_src + c]);
  }
}

// Deterministic mode: one thread per destination column walks the source rows IN ORDER (plain read-modify-write: the only writer
// of its column) — the serial definition of index_add. rows x C / 256 threads is a fraction of the chip; it is a debug mode.
template <typename SRC>
__global__ __launch_bounds__(EB) void scatter_add_det_kernel(const SRC* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ idx, int rows,
                                                             int C, int ld_src, int ld_dst) {
  const int c = blockIdx.x * EB + threadIdx.x;
  if (c >= C) return;
  for (int r = 0; r < rows; ++r) {
    const int d = idx ? idx[r] : r;
    if (d < 0) continue;
    float v;
    if constexpr (sizeof(SRC) == 2) v = bf2f(src[(int64_t)r * ld_src + c]);
    else v = src[(int64_t)r * ld_src + c];
    dst[(int64_t)d * ld_dst + c] += v;
  }
}

// column sums: block (64 column-pairs x 4 row lanes); each thread sums 2 adjacent columns over a
// row slice, LDS-combine the 4 row lanes, one atomic per column per block.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_raw* __restrict__ x, float* __restrict__ out, int rows, int C, int ld, int rows_per_block) {
  __shared__ float red[4][128];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 2;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    // four independent loads in flight per thread (round 6: the loop was one dependent 4-byte load per trip — 23 us for the box decoder's
    // [8192, 256] bias gradients, 40 launches a step)
    int r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      const unsigned u0 = *(const unsigned*)(x + (int64_t)r * ld + c), u1 = *(const unsigned*)(x + (int64_t)(r + 4) * ld + c);
      const unsigned u2 = *(const unsigned*)(x + (int64_t)(r + 8) * ld + c), u3 = *(const unsigned*)(x + (int64_t)(r + 12) * ld + c);
      s0 += (bf_lo(u0) + bf_lo(u1)) + (bf_lo(u2) + bf_lo(u3));
      s1 += (bf_hi(u0) + bf_hi(u1)) + (bf_hi(u2) + bf_hi(u3));
    }
    for (; r < r1; r += 4) {
      const unsigned u = *(const unsigned*)(x + (int64_t)r * ld + c);
      s0 += bf_lo(u);
      s1 += bf_hi(u);
    }
  }
  red[ry][cx * 2] = s0;
  red[ry][cx * 2 + 1] = s1;
  __syncthreads();
  if (ry == 0 && c < C) {
    s0 = red[0][cx * 2] + red[1][cx * 2] + red[2][cx * 2] + red[3][cx * 2];
    s1 = red[0][cx * 2 + 1] + red[1][cx * 2 + 1] + red[2][cx * 2 + 1] + red[3][cx * 2 + 1];
    atomicAdd(out + c, s0);
    if (c + 1 < C) atomicAdd(out + c + 1, s1);
  }
}

__global__ __launch_bounds__(EB) void cast_f2b_kernel(const float* __restrict__ x, bf16_raw* __restrict__ y, int64_t n) {
  const int64_t nvec = n >> 2;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < nvec; t += (int64_t)gridDim.x * EB) {
    const f32x4_t v = *(const f32x4_t*)(x + t * 4);
    *(u32x2_t*)(y + t * 4) = u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    y[i] = f2bf(x[i]);
  }
}
__global__ __launch_bounds__(EB) void cast_b2f_kernel(const bf16_raw* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t nvec = n >> 2;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < nvec; t += (int64_t)gridDim.x * EB) {
    const u32x2_t u = *(const u32x2_t*)(x + t * 4);
    *(f32x4_t*)(y + t * 4) = f32x4_t{bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    y[i] = bf2f(x[i]);
  }
}


// out += f(scale) * sum_i a[i] * b[i]   (bf16 inputs, fp32 accumulate); mode 1: f = 1 - tanh(s)^2
__global__ __launch_bounds__(EB) void dot_kernel(const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b, float* __restrict__ out, int64_t nvec,
                                                 const float* __restrict__ scale_ptr, int mode, unsigned* det) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < nvec; t += (int64_t)gridDim.x * EB) {
    float x[8], y[8];
    unpack8(*(const u32x4_t*)(a + t * 8), x);
    unpack8(*(const u32x4_t*)(b + t * 8), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e] * y[e];
  }
  s = block_sum<EB>(s, scratch);
  if (threadIdx.x == 0) {
    float f = 1.f;
    if (scale_ptr) {
      const float v = *scale_ptr;
      if (mode == 1) { const float t = tanhf(v); f = 1.f - t * t; }
      else if (mode == 2) f = tanhf(v);
      else f = v;
    }
    det_wait(det);
    atomicAdd(out, s * f);
    det_pass(det);
  }
}

// y[i] += f(scale) * x[i]  (fp32); mode 2: f = tanh(*scale_ptr), mode 0: f = *scale_ptr (or 1)
__global__ __launch_bounds__(EB) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, const float* __restrict__ scale_ptr, int mode) {
  float f = 1.f;
  if (scale_ptr) f = mode == 2 ? tanhf(*scale_ptr) : *scale_ptr;
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) y[i] += f * x[i];
}

// Batched 2-D transpose through a 64x64 LDS tile (+1 pad): coalesced on both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const grove_transpose_params p) {
  __shared__ bf16_raw tile[64][66];
  const int bz = blockIdx.z;
  const int b1 = bz / p.batch2, b2 = bz - b1 * p.batch2;
  const bf16_raw* __restrict__ in = (const bf16_raw*)p.in + (int64_t)b1 * p.s_in1 + (int64_t)b2 * p.s_in2;
  bf16_raw* __restrict__ out = (bf16_raw*)p.out + (int64_t)b1 * p.s_out1 + (int64_t)b2 * p.s_out2;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;  // input row / col origin
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < p.rows && c < p.cols) ? in[(int64_t)r * p.ld_in + c] : (bf16_raw)0;
  }
  __syncthreads();
  // output row = input col, output col = input row; columns [rows, pad_to) are zero filled
  for (int i = ty; i < 64; i += 4) {
    const int oc = r0 + tx, orow = c0 + i;
    if (orow < p.cols && oc < p.pad_to) out[(int64_t)orow * p.ld_out + oc] = tile[tx][i];
  }
}

// Many small 2-D transposes in ONE launch (the box decoder's ~40 weight transposes per backward were ~40 launch-bound kernels of 8-15 us
// on a serial chain): a device array of items, block -> (item, 64 x 64 tile) through the items' running tile counts.
__global__ __launch_bounds__(256) void transpose_many_kernel(const grove_transpose_item* __restrict__ items, int n_items) {
  __shared__ bf16_raw tile[64][66];
  int it = 0;
  while (it + 1 < n_items && (int)blockIdx.x >= items[it + 1].tile0) ++it;
  const grove_transpose_item m = items[it];
  const int t = blockIdx.x - m.tile0;
  const int tiles_x = (m.cols + 63) / 64;
  const int r0 = (t / tiles_x) * 64, c0 = (t % tiles_x) * 64;
  const bf16_raw* __restrict__ in = (const bf16_raw*)m.src;
  bf16_raw* __restrict__ out = (bf16_raw*)m.dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < m.rows && c < m.cols) ? in[(int64_t)r * m.ld_src + c] : (bf16_raw)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int oc = r0 + tx, orow = c0 + i;
    if (orow < m.cols && oc < m.rows) out[(int64_t)orow * m.ld_dst + oc] = tile[tx][i];
  }
}

}  // namespace

extern "C" int grove_transpose_many(const grove_transpose_item* items_dev, int32_t n_items, int32_t total_tiles, void* stream) {
  GROVE_CHECK(items_dev && n_items > 0 && total_tiles > 0, GROVE_E_SHAPE, "transpose_many: bad arguments");
  hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

#define CHECK_VEC8(n, name) GROVE_CHECK((n) % 8 == 0, GROVE_E_ALIGN, "%s: size must be a multiple of 8 elements", name)

extern "C" int grove_swiglu_fwd(const void* gu, void* y, int32_t rows, int32_t I, void* stream) {
  GROVE_CHECK(rows > 0 && I > 0, GROVE_E_SHAPE, "swiglu_fwd: bad shape");
  CHECK_VEC8(I, "swiglu_fwd");
  hipLaunchKernelGGL(swiglu_fwd_kernel, grid_for((int64_t)rows * (I / 8)), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)gu, (bf16_raw*)y, rows, I);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_swiglu_bwd(const void* gu, const void* dy, void* dgu, int32_t rows, int32_t I, void* stream) {
  GROVE_CHECK(rows > 0 && I > 0, GROVE_E_SHAPE, "swiglu_bwd: bad shape");
  CHECK_VEC8(I, "swiglu_bwd");
  hipLaunchKernelGGL(swiglu_bwd_kernel, grid_for((int64_t)rows * (I / 8)), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)gu, (const bf16_raw*)dy,
                     (bf16_raw*)dgu, rows, I);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_act_bwd(const void* pre, const void* dy, void* dx, int64_t n, int32_t act, void* stream) {
  GROVE_CHECK(n > 0, GROVE_E_SHAPE, "act_bwd: bad size");
  CHECK_VEC8(n, "act_bwd");
  hipLaunchKernelGGL(act_bwd_kernel, grid_for(n / 8), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)pre, (const bf16_raw*)dy, (bf16_raw*)dx, n / 8, act);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_act_fwd(const void* x, void* y, int64_t n, int32_t act, void* stream) {
  GROVE_CHECK(n > 0, GROVE_E_SHAPE, "act_fwd: bad size");
  CHECK_VEC8(n, "act_fwd");
  hipLaunchKernelGGL(act_fwd_kernel, grid_for(n / 8), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)x, (bf16_raw*)y, n / 8, act);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_resize_bilinear_f32(const float* src, float* dst, int32_t planes, int32_t h, int32_t w, int32_t h_use, int32_t w_use, int32_t H,
                                         int32_t W, void* stream) {
  GROVE_CHECK(src && dst && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0 && h_use > 0 && h_use <= h && w_use > 0 && w_use <= w, GROVE_E_SHAPE,
              "resize_bilinear: bad shape");
  hipLaunchKernelGGL(resize_bilinear_kernel, grid_for((int64_t)planes * H * W), dim3(EB), 0, (hipStream_t)stream, src, dst, planes, h, w, h_use, w_use, H,
                     W);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
  GROVE_CHECK(n > 0, GROVE_E_SHAPE, "add: bad size");
  CHECK_VEC8(n, "add");
  hipLaunchKernelGGL(add_kernel, grid_for(n / 8), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)a, (const bf16_raw*)b, (bf16_raw*)y, n / 8);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_add_bcast_rows(const void* a, const void* b, void* y, int32_t rows, int32_t C, int32_t period, void* stream) {
  GROVE_CHECK(rows > 0 && C > 0 && period > 0, GROVE_E_SHAPE, "add_bcast_rows: bad shape");
  CHECK_VEC8(C, "add_bcast_rows");
  hipLaunchKernelGGL(add_bcast_kernel, grid_for((int64_t)rows * (C / 8)), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)a, (const bf16_raw*)b,
                     (bf16_raw*)y, rows, C, period);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_copy_rows(const grove_rows_params* p, void* stream) {
  GROVE_CHECK(p && p->rows > 0 && p->C > 0, GROVE_E_SHAPE, "copy_rows: bad shape");
  GROVE_CHECK(p->C % 8 == 0 && p->ld_src % 8 == 0 && p->ld_dst % 8 == 0, GROVE_E_ALIGN, "copy_rows: C/ld must be multiples of 8");
  int tpr_shift = 0;  // threads per row: the power of two >= chunks per row / 2 (two chunks per thread when the row is long), at most the block
  while ((1 << tpr_shift) < EB && (2 << tpr_shift) < p->C / 8 + 1) ++tpr_shift;
  const int rpb = EB >> tpr_shift;
  const int64_t blocks = ((int64_t)p->rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(EB), 0, (hipStream_t)stream, *p, tpr_shift);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_scatter_add_f32(const void* src, float* dst, const int32_t* idx, int32_t rows, int32_t C, int32_t ld_src, int32_t ld_dst,
                                     void* stream) {
  GROVE_CHECK(rows > 0 && C > 0 && C % 2 == 0 && ld_src % 2 == 0, GROVE_E_SHAPE, "scatter_add: bad shape");
  if (grove_det_on()) {
    hipLaunchKernelGGL(scatter_add_det_kernel<bf16_raw>, dim3((C + EB - 1) / EB), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)src, dst, idx, rows, C,
                       ld_src, ld_dst);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  hipLaunchKernelGGL(scatter_add_kernel, grid_for((int64_t)rows * (C / 2)), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)src, dst, idx, rows, C,
                     ld_src, ld_dst);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
// dst[(s * R + r), :] += sum over the segment's members i in [ptr[s], ptr[s + 1]) of src[(i * R + r), :]: a scatter-add whose index is
// "member i belongs to segment s, members of a segment are consecutive" turned around — every output element has ONE owner, no atomics
__global__ __launch_bounds__(EB) void segment_sum_rows_kernel(const bf16_raw* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ ptr,
                                                              int nseg, int R, int C8, int ld_src, int ld_dst) {
  const int64_t total = (int64_t)nseg * R * C8;
  for (int64_t t = (int64_t)blockIdx.x * EB + threadIdx.x; t < total; t += (int64_t)gridDim.x * EB) {
    const int ch = (int)(t % C8);
    const int64_t row = t / C8;          // s * R + r
    const int sgm = (int)(row / R), r = (int)(row - (int64_t)sgm * R);
    const int lo = ptr[sgm], hi = ptr[sgm + 1];
    if (lo >= hi) continue;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = lo; i < hi; ++i) {
      const u32x4_t u = *(const u32x4_t*)(src + ((int64_t)i * R + r) * ld_src + ch * 8);
      acc[0] += bf_lo(u.x); acc[1] += bf_hi(u.x); acc[2] += bf_lo(u.y); acc[3] += bf_hi(u.y);
      acc[4] += bf_lo(u.z); acc[5] += bf_hi(u.z); acc[6] += bf_lo(u.w); acc[7] += bf_hi(u.w);
    }
    float* d = dst + row * ld_dst + ch * 8;
    f32x4_t a = *(const f32x4_t*)d, b = *(const f32x4_t*)(d + 4);
    a[0] += acc[0]; a[1] += acc[1]; a[2] += acc[2]; a[3] += acc[3];
    b[0] += acc[4]; b[1] += acc[5]; b[2] += acc[6]; b[3] += acc[7];
    *(f32x4_t*)d = a;
    *(f32x4_t*)(d + 4) = b;
  }
}
extern "C" int grove_segment_sum_rows(const void* src, float* dst, const int32_t* seg_ptr, int32_t nseg, int32_t rows_per_seg, int32_t C, int32_t ld_src,
                                      int32_t ld_dst, void* stream) {
  GROVE_CHECK(src && dst && seg_ptr && nseg > 0 && rows_per_seg > 0 && C > 0, GROVE_E_SHAPE, "segment_sum_rows: bad shape");
  GROVE_CHECK(C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, GROVE_E_ALIGN,
              "segment_sum_rows: C / ld_src multiples of 8, ld_dst of 4, 16-byte aligned operands");
  hipLaunchKernelGGL(segment_sum_rows_kernel, grid_for((int64_t)nseg * rows_per_seg * (C / 8)), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)src, dst,
                     seg_ptr, nseg, rows_per_seg, C / 8, ld_src, ld_dst);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_scatter_add_rows_f32(const float* src, float* dst, const int32_t* idx, int32_t rows, int32_t C, int32_t ld_src, int32_t ld_dst,
                                          void* stream) {
  GROVE_CHECK(rows > 0 && C > 0, GROVE_E_SHAPE, "scatter_add_rows_f32: bad shape");
  if (grove_det_on()) {
    hipLaunchKernelGGL(scatter_add_det_kernel<float>, dim3((C + EB - 1) / EB), dim3(EB), 0, (hipStream_t)stream, src, dst, idx, rows, C, ld_src, ld_dst);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  hipLaunchKernelGGL(scatter_add_f32src_kernel, grid_for((int64_t)rows * C), dim3(EB), 0, (hipStream_t)stream, src, dst, idx, rows, C, ld_src, ld_dst);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_colsum_f32(const void* x, float* out, int32_t rows, int32_t C, int32_t ld, int32_t accumulate, void* stream) {
  GROVE_CHECK(rows > 0 && C > 0 && ld % 2 == 0, GROVE_E_SHAPE, "colsum: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)C * sizeof(float), s);
    GROVE_CHECK(e == hipSuccess, GROVE_E_HIP, "colsum: memset failed");
  }
  // deterministic mode: one block per column group, so one add per column; else enough row slices to fill the chip (narrow matrices:
  // C = 256 is two column groups — at 256 rows per block a [8192, 256] sum ran on 64 of the 256 CUs)
  const int col_groups = (C + 127) / 128;
  int rpb = grove_det_on() ? rows : 256;
  if (!grove_det_on())
    while (rpb > 32 && (long)col_groups * ((rows + rpb - 1) / rpb) < 512) rpb >>= 1;
  dim3 grid((C + 127) / 128, (rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, (const bf16_raw*)x, out, rows, C, ld, rpb);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  GROVE_CHECK(n > 0, GROVE_E_SHAPE, "cast: bad size");
  hipLaunchKernelGGL(cast_f2b_kernel, grid_for(n / 4 + 1), dim3(EB), 0, (hipStream_t)stream, x, (bf16_raw*)y, n);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
  GROVE_CHECK(n > 0, GROVE_E_SHAPE, "cast: bad size");
  hipLaunchKernelGGL(cast_b2f_kernel, grid_for(n / 4 + 1), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)x, y, n);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_dot_bf16(const void* a, const void* b, float* out, int64_t n, const float* scale_ptr, int32_t mode, void* stream) {
  GROVE_CHECK(n > 0 && out, GROVE_E_SHAPE, "dot: bad args");
  CHECK_VEC8(n, "dot");
  hipLaunchKernelGGL(dot_kernel, grid_for(n / 8), dim3(EB), 0, (hipStream_t)stream, (const bf16_raw*)a, (const bf16_raw*)b, out, n / 8, scale_ptr, mode,
                     grove_det_ticket());
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_axpy_f32(float* y, const float* x, int64_t n, const float* scale_ptr, int32_t mode, void* stream) {
  GROVE_CHECK(n > 0 && y && x, GROVE_E_SHAPE, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, y, x, n, scale_ptr, mode);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
extern "C" int grove_transpose_bf16(const grove_transpose_params* pp, void* stream) {
  GROVE_CHECK(pp && pp->rows > 0 && pp->cols > 0, GROVE_E_SHAPE, "transpose: bad shape");
  grove_transpose_params p = *pp;
  if (p.batch1 <= 0) p.batch1 = 1;
  if (p.batch2 <= 0) p.batch2 = 1;
  if (p.pad_to < p.rows) p.pad_to = p.rows;
  GROVE_CHECK(p.ld_out >= p.pad_to && p.ld_in >= p.cols, GROVE_E_SHAPE, "transpose: leading dims too small");
  dim3 grid((p.cols + 63) / 64, (p.pad_to + 63) / 64, p.batch1 * p.batch2);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
