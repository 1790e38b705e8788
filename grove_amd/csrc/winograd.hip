// Winograd F(2 x 2 x 2, 3 x 3 x 3) transforms for the Conv3d adapters on gfx950 (MI355X).
//
// The SAM adapters (image_encoder.py:43-59: Conv3d(C, C, 3, padding = 1) over [B, C, T = 8, 32, 32]) are 12.8 % of the training
// step as 27-tap implicit GEMMs. In the minimal-filtering form an output tile of 2 x 2 x 2 positions needs 4 x 4 x 4 = 64
// element-wise products of TRANSFORMED operands instead of 8 x 27 = 216, and those 64 "transform points" are independent
// [tiles, C] x [C, C] matrix products — which the pipelined GEMMs run as one grouped (forward / dgrad) or K-batched (weight
// gradient) launch. The kernels here are the memory-bound ends of that pipeline; all arithmetic is fp32, operands are rounded
// to bf16 once, where the MFMA reads them:
//
//   V  = (B^T (x) B^T (x) B^T) d     d  = a 4 x 4 x 4 input tile (zero padded at the volume's faces), tiles overlap by 2
//   U  = (G (x) G (x) G) g           g  = the 3 x 3 x 3 taps of one (co, ci)
//   Y  = (A^T (x) A^T (x) A^T) M     M  = sum_ci U (.) V per transform point
//   dM = (A (x) A (x) A) dY          the adjoint of the output transform on a 2 x 2 x 2 tile of the output gradient
//   dg = (G^T (x) G^T (x) G^T) dU    dU = sum_tiles dM (.) V per transform point
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// Layouts. Tokens are rows [(g, t, y, x)][C] (frame-major, groups of T frames: the layout of the towers). Tile
// (g, tt, ty, tx) covers outputs (2 tt + {0,1}, 2 ty + {0,1}, 2 tx + {0,1}); tile id = ((g T/2 + tt) H/2 + ty) W/2 + tx.
// Transformed tensors are POINT-MAJOR: [64 points][tiles][C] with point = (a 4 + b) 4 + c over (t, y, x) — each point's
// [tiles, C] matrix is contiguous, i.e. one group of the grouped GEMM / one K range of the batched weight-gradient GEMM.
// Every kernel: one thread per (tile or weight row, PAIR of channels) — a lane moves one dword per position, a wave 256
// contiguous bytes — with the whole 64-point tile of that pair in registers (128 fp32).
#include "common.h"

namespace {

struct tile_pos {
  int g, tt, ty, tx;
};
__device__ __forceinline__ tile_pos tile_of(int tile, int T2, int H2, int W2) {
  tile_pos r;
  r.tx = tile % W2;
  int q = tile / W2;
  r.ty = q % H2;
  q /= H2;
  r.tt = q % T2;
  r.g = q / T2;
  return r;
}

// row of token (g, t, y, x): dense frames of H W rows, or frames of frame_rows rows whose tokens start at row_offset (CLIP's CLS row)
__device__ __forceinline__ int64_t token_row(const grove_wino3d_params& p, int g, int t, int y, int x) {
  const int fr = p.frame_rows ? p.frame_rows : p.H * p.W;
  return (int64_t)(g * p.T + t) * fr + p.row_offset + y * p.W + x;
}

// one 1-D transform of a line of 4 (in place)
__device__ __forceinline__ void bt4(float& d0, float& d1, float& d2, float& d3) {
  const float a = d0 - d2, b = d1 + d2, c = d2 - d1, e = d1 - d3;
  d0 = a, d1 = b, d2 = c, d3 = e;
}

// V = B^T-transform of the overlapping, zero-padded 4 x 4 x 4 input tiles (MODE 0), or dM = A-transform of the disjoint
// 2 x 2 x 2 output-gradient tiles (MODE 1). grid (channel-pair blocks, tiles).
template <int MODE>
__global__ __launch_bounds__(320) void wino3d_tokens_kernel(const grove_wino3d_params p) {
  const int cp = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * cp >= p.C) return;
  const int T2 = p.T >> 1, H2 = p.H >> 1, W2 = p.W >> 1;
  const int tile = blockIdx.y + gridDim.y * blockIdx.z;
  const int tiles = p.groups * T2 * H2 * W2;
  if (tile >= tiles) return;
  const tile_pos tp = tile_of(tile, T2, H2, W2);
  const unsigned* __restrict__ src = (const unsigned*)p.src + cp;
  const int lds = p.ld_src >> 1;  // dwords per token row
  float v[4][4][4][2];
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 2 * tp.tt - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = 2 * tp.ty - 1 + j;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = 2 * tp.tx - 1 + k;
          unsigned u = 0;  // (the tests are block-uniform: scalar branches)
          if ((unsigned)t < (unsigned)p.T && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
            u = src[token_row(p, tp.g, t, y, x) * lds];
          v[i][j][k][0] = bf_lo(u), v[i][j][k][1] = bf_hi(u);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) bt4(v[i][j][0][c], v[i][j][1][c], v[i][j][2][c], v[i][j][3][c]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) bt4(v[i][0][k][c], v[i][1][k][c], v[i][2][k][c], v[i][3][k][c]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) bt4(v[0][j][k][c], v[1][j][k][c], v[2][j][k][c], v[3][j][k][c]);
    }
  } else {
    float d[2][2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const unsigned u = src[token_row(p, tp.g, 2 * tp.tt + i, 2 * tp.ty + j, 2 * tp.tx + k) * lds];
          d[i][j][k][0] = bf_lo(u), d[i][j][k][1] = bf_hi(u);
        }
    // A = [1 0; 1 1; 1 -1; 0 -1] along x, then y, then t
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float e[2][2][4], f[2][4][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float a = d[i][j][0][c], b = d[i][j][1][c];
          e[i][j][0] = a, e[i][j][1] = a + b, e[i][j][2] = a - b, e[i][j][3] = -b;
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = e[i][0][k], b = e[i][1][k];
          f[i][0][k] = a, f[i][1][k] = a + b, f[i][2][k] = a - b, f[i][3][k] = -b;
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = f[0][j][k], b = f[1][j][k];
          v[0][j][k][c] = a, v[1][j][k][c] = a + b, v[2][j][k][c] = a - b, v[3][j][k][c] = -b;
        }
    }
  }
  unsigned* __restrict__ dst = (unsigned*)p.dst + (int64_t)tile * (p.ld_dst >> 1) + cp;
  const int64_t ps = (int64_t)(p.tiles_ld ? p.tiles_ld : tiles) * (p.ld_dst >> 1);  // dwords per transform point
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[((i * 4 + j) * 4 + k) * ps] = pack2bf(v[i][j][k][0], v[i][j][k][1]);
}

// U[point][co][ci] = G-transform of w[co][tap][ci] (tap = (kt 3 + kh) 3 + kw: the packed layout of the implicit-GEMM weights).
// grid (channel-pair blocks, Co).
__global__ __launch_bounds__(320) void wino3d_weight_kernel(const grove_wino3d_params p) {
  const int cp = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * cp >= p.C) return;
  const int co = blockIdx.y;
  const unsigned* __restrict__ src = (const unsigned*)p.src + (int64_t)co * (p.ld_src >> 1) + cp;
  const int ts = p.C >> 1;  // dwords per tap
  float g[3][3][3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const unsigned u = src[((i * 3 + j) * 3 + k) * ts];
        g[i][j][k][0] = bf_lo(u), g[i][j][k][1] = bf_hi(u);
      }
  unsigned* __restrict__ dst = (unsigned*)p.dst + (int64_t)co * (p.ld_dst >> 1) + cp;
  const int64_t ps = (int64_t)p.rows * (p.ld_dst >> 1);
  float o[4][4][4][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float e[3][3][4], f[3][4][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float a = g[i][j][0][c], b = g[i][j][1][c], d = g[i][j][2][c];
        e[i][j][0] = a, e[i][j][1] = 0.5f * (a + b + d), e[i][j][2] = 0.5f * (a - b + d), e[i][j][3] = d;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = e[i][0][k], b = e[i][1][k], d = e[i][2][k];
        f[i][0][k] = a, f[i][1][k] = 0.5f * (a + b + d), f[i][2][k] = 0.5f * (a - b + d), f[i][3][k] = d;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = f[0][j][k], b = f[1][j][k], d = f[2][j][k];
        o[0][j][k][c] = a, o[1][j][k][c] = 0.5f * (a + b + d), o[2][j][k][c] = 0.5f * (a - b + d), o[3][j][k][c] = d;
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[((i * 4 + j) * 4 + k) * ps] = pack2bf(o[i][j][k][0], o[i][j][k][1]);
}

// y = epilogue((A^T (x) A^T (x) A^T) M): M bf16 [64][tiles][C] -> 8 token rows per tile.
//   v = y + bias; aux = bf16(v) (the pre-activation the backward reads); v = act(v) * scale + residual.   grid (pair blocks, tiles).
__global__ __launch_bounds__(320) void wino3d_output_kernel(const grove_wino3d_params p) {
  const int cp = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * cp >= p.C) return;
  const int T2 = p.T >> 1, H2 = p.H >> 1, W2 = p.W >> 1;
  const int tile = blockIdx.y + gridDim.y * blockIdx.z;
  const int tiles = p.groups * T2 * H2 * W2;
  if (tile >= tiles) return;
  const tile_pos tp = tile_of(tile, T2, H2, W2);
  const unsigned* __restrict__ src = (const unsigned*)p.src + (int64_t)tile * (p.ld_src >> 1) + cp;
  const int64_t ps = (int64_t)(p.tiles_ld ? p.tiles_ld : tiles) * (p.ld_src >> 1);
  float m[4][4][4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned u = src[((i * 4 + j) * 4 + k) * ps];
        m[i][j][k][0] = bf_lo(u), m[i][j][k][1] = bf_hi(u);
      }
  float scale = p.alpha;
  if (p.scale_ptr) scale *= p.scale_tanh ? tanhf(*p.scale_ptr) : *p.scale_ptr;
  float b0 = 0.f, b1 = 0.f;
  if (p.bias) {
    const unsigned u = ((const unsigned*)p.bias)[cp];
    b0 = bf_lo(u), b1 = bf_hi(u);
  }
  float y[2][2][2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float e[4][4][2], f[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[i][j][0] = m[i][j][0][c] + m[i][j][1][c] + m[i][j][2][c];
        e[i][j][1] = m[i][j][1][c] - m[i][j][2][c] - m[i][j][3][c];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f[i][0][k] = e[i][0][k] + e[i][1][k] + e[i][2][k];
        f[i][1][k] = e[i][1][k] - e[i][2][k] - e[i][3][k];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        y[0][j][k][c] = f[0][j][k] + f[1][j][k] + f[2][j][k];
        y[1][j][k][c] = f[1][j][k] - f[2][j][k] - f[3][j][k];
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int64_t row = token_row(p, tp.g, 2 * tp.tt + i, 2 * tp.ty + j, 2 * tp.tx + k);
        float v0 = y[i][j][k][0] + b0, v1 = y[i][j][k][1] + b1;
        if (p.aux) ((unsigned*)p.aux)[row * (p.ld_aux >> 1) + cp] = pack2bf(v0, v1);
        v0 = act_apply(p.act, v0) * scale, v1 = act_apply(p.act, v1) * scale;
        if (p.residual) {
          const unsigned u = ((const unsigned*)p.residual)[row * (p.ld_res >> 1) + cp];
          v0 += bf_lo(u), v1 += bf_hi(u);
        }
        ((unsigned*)p.dst)[row * (p.ld_dst >> 1) + cp] = pack2bf(v0, v1);
      }
}

// dW[co][tap][ci] += scale * (G^T (x) G^T (x) G^T) dU[point][co][ci]   (both fp32).   grid (pair blocks, Co).
__global__ __launch_bounds__(320) void wino3d_wgrad_kernel(const grove_wino3d_params p) {
  const int cp = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * cp >= p.C) return;
  const int co = blockIdx.y;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const f32x2* __restrict__ src = (const f32x2*)p.src + (int64_t)co * (p.ld_src >> 1) + cp;
  const int64_t ps = (int64_t)p.rows * (p.ld_src >> 1);
  float u[4][4][4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 t = src[((i * 4 + j) * 4 + k) * ps];
        u[i][j][k][0] = t.x, u[i][j][k][1] = t.y;
      }
  float scale = p.alpha;
  if (p.scale_ptr) scale *= p.scale_tanh ? tanhf(*p.scale_ptr) : *p.scale_ptr;
  f32x2* __restrict__ dst = (f32x2*)p.dst + (int64_t)co * (p.ld_dst >> 1) + cp;
  const int ts = p.C >> 1;
  float o[3][3][3][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float e[4][4][3], f[4][3][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = u[i][j][0][c], h1 = 0.5f * u[i][j][1][c], h2 = 0.5f * u[i][j][2][c], d = u[i][j][3][c];
        e[i][j][0] = a + h1 + h2, e[i][j][1] = h1 - h2, e[i][j][2] = h1 + h2 + d;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float a = e[i][0][k], h1 = 0.5f * e[i][1][k], h2 = 0.5f * e[i][2][k], d = e[i][3][k];
        f[i][0][k] = a + h1 + h2, f[i][1][k] = h1 - h2, f[i][2][k] = h1 + h2 + d;
      }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float a = f[0][j][k], h1 = 0.5f * f[1][j][k], h2 = 0.5f * f[2][j][k], d = f[3][j][k];
        o[0][j][k][c] = a + h1 + h2, o[1][j][k][c] = h1 - h2, o[2][j][k][c] = h1 + h2 + d;
      }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        f32x2* q = dst + ((i * 3 + j) * 3 + k) * ts;
        f32x2 t = *q;
        t.x += scale * o[i][j][k][0], t.y += scale * o[i][j][k][1];
        *q = t;
      }
}

inline int pair_block(int C) {  // threads per block over the C / 2 channel pairs: whole waves, as few idle lanes as possible
  const int pairs = C / 2;
  for (int b = 320; b >= 64; b -= 64)
    if (pairs % b == 0) return b;
  return pairs >= 256 ? 256 : (pairs + 63) / 64 * 64;
}
inline int check_common(const grove_wino3d_params* p, const char* who) {
  GROVE_CHECK(p && p->src && p->dst, GROVE_E_SHAPE, "%s: null params / pointers", who);
  GROVE_CHECK(p->C > 0 && p->C % 2 == 0, GROVE_E_SHAPE, "%s: C=%d must be even", who, p->C);
  GROVE_CHECK(p->ld_src % 2 == 0 && p->ld_dst % 2 == 0 && p->ld_src >= p->C && p->ld_dst >= p->C, GROVE_E_ALIGN, "%s: ld_src=%d ld_dst=%d must be even and >= C", who,
              p->ld_src, p->ld_dst);
  GROVE_CHECK(((uintptr_t)p->src & 7) == 0 && ((uintptr_t)p->dst & 7) == 0, GROVE_E_ALIGN, "%s: pointers must be 8-byte aligned", who);
  return GROVE_OK;
}
inline int check_geometry(const grove_wino3d_params* p, const char* who, long* tiles) {
  GROVE_CHECK(p->groups > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->T % 2 == 0 && p->H % 2 == 0 && p->W % 2 == 0, GROVE_E_SHAPE,
              "%s: groups=%d and even T, H, W needed (got %d, %d, %d)", who, p->groups, p->T, p->H, p->W);
  *tiles = (long)p->groups * (p->T / 2) * (p->H / 2) * (p->W / 2);
  GROVE_CHECK(p->frame_rows == 0 || (p->frame_rows >= p->row_offset + p->H * p->W && p->row_offset >= 0), GROVE_E_SHAPE,
              "%s: frame_rows=%d must hold row_offset=%d + H W tokens", who, p->frame_rows, p->row_offset);
  GROVE_CHECK(p->tiles_ld == 0 || p->tiles_ld >= *tiles, GROVE_E_SHAPE, "%s: tiles_ld=%d < %ld tiles", who, p->tiles_ld, *tiles);
  const long fr = p->frame_rows ? p->frame_rows : (long)p->H * p->W, tl = p->tiles_ld ? p->tiles_ld : *tiles;
  GROVE_CHECK((long)p->groups * p->T * fr < (1L << 31) && tl * 64 < (1L << 31), GROVE_E_SHAPE, "%s: too many rows", who);
  return GROVE_OK;
}
inline dim3 tile_grid(int C, int block, long tiles) {
  const unsigned gy = (unsigned)(tiles < 32768 ? tiles : 32768);
  return dim3((C / 2 + block - 1) / block, gy, (unsigned)((tiles + gy - 1) / gy));
}

}  // namespace

extern "C" int grove_wino3d_transform_tokens(const grove_wino3d_params* p, void* stream) {
  if (int rc = check_common(p, "wino3d_transform_tokens")) return rc;
  long tiles;
  if (int rc = check_geometry(p, "wino3d_transform_tokens", &tiles)) return rc;
  GROVE_CHECK(p->mode == 0 || p->mode == 1, GROVE_E_SHAPE, "wino3d_transform_tokens: mode %d (0 = input tiles, 1 = output-gradient tiles)", p->mode);
  const int block = pair_block(p->C);
  const dim3 grid = tile_grid(p->C, block, tiles);
  if (p->mode == 0) hipLaunchKernelGGL(wino3d_tokens_kernel<0>, grid, dim3(block), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(wino3d_tokens_kernel<1>, grid, dim3(block), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_wino3d_transform_weight(const grove_wino3d_params* p, void* stream) {
  if (int rc = check_common(p, "wino3d_transform_weight")) return rc;
  GROVE_CHECK(p->rows > 0 && p->rows < 65536 && p->ld_src >= 27 * p->C, GROVE_E_SHAPE, "wino3d_transform_weight: rows=%d (output channels), ld_src=%d >= 27 C needed",
              p->rows, p->ld_src);
  const int block = pair_block(p->C);
  hipLaunchKernelGGL(wino3d_weight_kernel, dim3((p->C / 2 + block - 1) / block, p->rows), dim3(block), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_wino3d_output(const grove_wino3d_params* p, void* stream) {
  if (int rc = check_common(p, "wino3d_output")) return rc;
  long tiles;
  if (int rc = check_geometry(p, "wino3d_output", &tiles)) return rc;
  GROVE_CHECK((!p->residual || (p->ld_res % 2 == 0 && ((uintptr_t)p->residual & 3) == 0)) && (!p->aux || (p->ld_aux % 2 == 0 && ((uintptr_t)p->aux & 3) == 0)) &&
                  (!p->bias || ((uintptr_t)p->bias & 3) == 0),
              GROVE_E_ALIGN, "wino3d_output: residual / aux / bias must be 4-byte aligned with even row strides");
  GROVE_CHECK(p->act == GROVE_ACT_NONE || p->act == GROVE_ACT_RELU, GROVE_E_SHAPE, "wino3d_output: act %d (NONE or RELU)", p->act);
  const int block = pair_block(p->C);
  hipLaunchKernelGGL(wino3d_output_kernel, tile_grid(p->C, block, tiles), dim3(block), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_wino3d_wgrad_output(const grove_wino3d_params* p, void* stream) {
  if (int rc = check_common(p, "wino3d_wgrad_output")) return rc;
  GROVE_CHECK(p->rows > 0 && p->rows < 65536 && p->ld_dst >= 27 * p->C, GROVE_E_SHAPE, "wino3d_wgrad_output: rows=%d (output channels), ld_dst=%d >= 27 C needed", p->rows,
              p->ld_dst);
  const int block = pair_block(p->C);
  hipLaunchKernelGGL(wino3d_wgrad_kernel, dim3((p->C / 2 + block - 1) / block, p->rows), dim3(block), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
