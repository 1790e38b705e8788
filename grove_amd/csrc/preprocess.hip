// On-device frame preprocessing: what the reference does per clip on the host with PIL
// (HowTo100M.py:309-313, transforms.py:27-34, CLIPImageProcessor) — uint8 RGB frames [F, H, W, 3] in HBM to the two
// encoder inputs. Bit-exact with Pillow's 8-bit resampler: the same two separable passes (horizontal, then vertical) with
// 22-bit fixed-point coefficients, half-up rounding and a uint8 clamp BETWEEN the passes; the coefficient and bound tables
// are built on the host exactly as ImagingResample's precompute_coeffs / normalize_coeffs_8bpc do (grove_amd/preprocess.py).
// HBM-bound byte shuffling: one thread per output pixel (3 channels), no LDS.
#include "common.h"

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// axis 0: out[f, y, x] = sum_k src[f, y, bounds[x].min + k] * kk[x][k]   (horizontal pass, W changes)
// axis 1: out[f, y, x] = sum_k src[f, bounds[y].min + k, x] * kk[y][k]   (vertical pass, H changes)
__global__ __launch_bounds__(256) void resample_u8_kernel(const grove_resample_params p) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)p.F * p.Hout * p.Wout;
  if (t >= n) return;
  const int x = (int)(t % p.Wout);
  const int y = (int)((t / p.Wout) % p.Hout);
  const int f = (int)(t / ((int64_t)p.Wout * p.Hout));
  const int o = p.axis == 0 ? x : y;
  const int kmin = p.bounds[2 * o], kn = p.bounds[2 * o + 1];
  const int* k = p.kk + (int64_t)o * p.ksize;
  const unsigned char* src = (const unsigned char*)p.src + (int64_t)f * p.Hin * p.Win * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  if (p.axis == 0) {
    const unsigned char* row = src + ((int64_t)y * p.Win + kmin) * 3;
    for (int i = 0; i < kn; ++i) {
      const int w = k[i];
      s0 += row[3 * i] * w; s1 += row[3 * i + 1] * w; s2 += row[3 * i + 2] * w;
    }
  } else {
    const unsigned char* col = src + ((int64_t)kmin * p.Win + x) * 3;
    for (int i = 0; i < kn; ++i) {
      const int w = k[i];
      const unsigned char* px = col + (int64_t)i * p.Win * 3;
      s0 += px[0] * w; s1 += px[1] * w; s2 += px[2] * w;
    }
  }
  unsigned char* dst = (unsigned char*)p.dst + t * 3;
  dst[0] = clip8(s0); dst[1] = clip8(s1); dst[2] = clip8(s2);
}

// u8 [F, H, W, 3] -> bf16 or f32 [3, F, Ho, Wo]: out[c, f, y, x] = (src[f, top + y, left + x, c] * rescale - mean[c]) / std[c] inside
// the source window, 0 outside (SAM's right / bottom padding is applied AFTER normalisation, HowTo100M.py:168-178).
__global__ __launch_bounds__(256) void normalize_pack_kernel(const grove_normalize_params p) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)p.F * p.Ho * p.Wo;
  if (t >= n) return;
  const int x = (int)(t % p.Wo);
  const int y = (int)((t / p.Wo) % p.Ho);
  const int f = (int)(t / ((int64_t)p.Wo * p.Ho));
  const int sy = y + p.top, sx = x + p.left;
  const bool in = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
  float v[3] = {0.f, 0.f, 0.f};
  if (in) {
    const unsigned char* px = (const unsigned char*)p.src + (((int64_t)f * p.H + sy) * p.W + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ((float)px[c] * p.rescale - p.mean[c]) / p.std[c];
  }
  const int64_t plane = (int64_t)p.F * p.Ho * p.Wo;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (p.out_dtype == GROVE_BF16) ((bf16_raw*)p.dst)[c * plane + t] = f2bf(v[c]);
    else ((float*)p.dst)[c * plane + t] = v[c];
  }
}
}  // namespace

extern "C" int grove_resample_u8(const grove_resample_params* p, void* stream) {
  GROVE_CHECK(p && p->F > 0 && p->Hin > 0 && p->Win > 0 && p->Hout > 0 && p->Wout > 0, GROVE_E_SHAPE, "resample_u8: bad shape");
  GROVE_CHECK(p->axis == 0 ? p->Hin == p->Hout : (p->axis == 1 && p->Win == p->Wout), GROVE_E_SHAPE,
              "resample_u8: axis 0 keeps H, axis 1 keeps W");
  GROVE_CHECK(p->kk && p->bounds && p->ksize > 0, GROVE_E_SHAPE, "resample_u8: coefficient tables required");
  const int64_t n = (int64_t)p->F * p->Hout * p->Wout;
  hipLaunchKernelGGL(resample_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_normalize_pack(const grove_normalize_params* p, void* stream) {
  GROVE_CHECK(p && p->F > 0 && p->H > 0 && p->W > 0 && p->Ho > 0 && p->Wo > 0, GROVE_E_SHAPE, "normalize_pack: bad shape");
  GROVE_CHECK(p->out_dtype == GROVE_BF16 || p->out_dtype == GROVE_F32, GROVE_E_DTYPE, "normalize_pack: bad out_dtype");
  const int64_t n = (int64_t)p->F * p->Ho * p->Wo;
  hipLaunchKernelGGL(normalize_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
