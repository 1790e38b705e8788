// Common device helpers for the gfx950 (CDNA4 / MI355X) kernels of grove_amd.
// Wave size is 64; all kernels here are written for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grove_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef unsigned short bf16_raw;

__device__ __forceinline__ float bf2f(bf16_raw x) { return __uint_as_float(((unsigned)x) << 16); }
__device__ __forceinline__ bf16_raw f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_raw, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 for the pair (two scalar casts + shift/or otherwise)
  typedef __attribute__((ext_vector_type(2))) float f32x2_t_;
  const bf16x2_t v = __builtin_convertvector(f32x2_t_{lo, hi}, bf16x2_t);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64); scratch: NT/64 floats in LDS.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  constexpr int NW = NT / 64;
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += scratch[i];
  return r;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* scratch) {
  v = wave_max(v);
  constexpr int NW = NT / 64;
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): one v_rcp, one v_exp and five FMAs instead of libm's
// erff polynomial ladder — the GELU epilogue is ~20 % of a K=1280 GEMM tile otherwise.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);  // v_rcp_f32 itself (1 ulp); __frcp_rn expands to an IEEE division
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}

// logistic function with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (~10 instructions): every
// sigmoid-shaped activation of the path goes through it, so fused and unfused forms stay bit-identical to each other.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// d(gate), d(up) of silu(gate) * up for one element. ONE definition with floating-point contraction OFF: the elementwise kernel
// (grove_swiglu_bwd), the GEMM epilogue (GROVE_ACT_SWIGLU_BWD) and its stream-K fix-up kernel are three compilation contexts, and
// hipcc fuses multiply-adds differently in each — with the contraction left to it the three differed in the last bf16 bit.
__device__ __forceinline__ void swiglu_bwd_elem(const float d, const float g, const float u, float& dg, float& du) {
#pragma clang fp contract(off)
  const float s = fast_sigmoid(g);
  const float gs = g * s;
  dg = (d * u) * (s + gs * (1.f - s));
  du = d * gs;
}

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case GROVE_ACT_RELU: return fmaxf(x, 0.f);
    case GROVE_ACT_GELU: return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752440f));
    case GROVE_ACT_QUICKGELU: return x * fast_sigmoid(1.702f * x);
    case GROVE_ACT_SILU: return x * fast_sigmoid(x);
    case GROVE_ACT_SIGMOID: return fast_sigmoid(x);
    default: return x;
  }
}
// d act(x) / dx
__device__ __forceinline__ float act_grad(int act, float x) {
  switch (act) {
    case GROVE_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case GROVE_ACT_GELU: {
      const float c = 0.70710678118654752440f;
      float cdf = 0.5f * (1.f + fast_erf(x * c));
      float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case GROVE_ACT_QUICKGELU: {
      float s = fast_sigmoid(1.702f * x);
      return s + 1.702f * x * s * (1.f - s);
    }
    case GROVE_ACT_SILU: {
      float s = fast_sigmoid(x);
      return s + x * s * (1.f - s);
    }
    case GROVE_ACT_SIGMOID: {
      float s = fast_sigmoid(x);
      return s * (1.f - s);
    }
    default: return 1.f;
  }
}

// act(x) and d act(x) / dx together: the erf GELU's exp(-x^2 / 2) and the sigmoids are computed once (a GEMM epilogue that
// stores the derivative for the backward pass — grove_gemm_params.aux_grad — pays two or three extra VALU ops, not a second
// transcendental chain). Same values as act_apply / act_grad up to the rounding of the shared terms.
__device__ __forceinline__ void act_apply_grad(int act, float x, float& y, float& g) {
  switch (act) {
    case GROVE_ACT_GELU: {
      const float z = x * 0.70710678118654752440f, az = fabsf(z);
      const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * az);
      const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
      const float e = __expf(-az * az);  // = exp(-x^2 / 2)
      const float cdf = 0.5f * (1.f + copysignf(1.f - poly * e, z));
      y = x * cdf;
      g = cdf + x * (0.3989422804014327f * e);
      return;
    }
    case GROVE_ACT_QUICKGELU: {
      const float s = fast_sigmoid(1.702f * x);
      y = x * s;
      g = s + 1.702f * x * s * (1.f - s);
      return;
    }
    case GROVE_ACT_SILU: {
      const float s = fast_sigmoid(x);
      y = x * s;
      g = s + x * s * (1.f - s);
      return;
    }
    default:
      y = act_apply(act, x);
      g = act_grad(act, x);
  }
}

// ---------------------------------------------------------------- deterministic mode (grove_set_deterministic)
// fp32 atomics make a sum depend on the order the blocks arrive in. In deterministic mode every kernel that adds to global memory
// is handed a TICKET (a zeroed unsigned from a device ring, capi.hip): block b waits until the ticket reads b, adds, and passes b + 1
// on (the last block puts the 0 back). Blocks are dispatched in linear-id order — per XCD in order too — so the lowest block that
// has not passed yet is always resident or next to be dispatched: the wait cannot deadlock, and only the add phase is serialised.
// With the ticket null (the default) each helper is a single uniform branch.
// A TEST / DEBUG mode, not a production one: (i) the no-deadlock argument rests on the dispatcher handing out workgroups in linear-id
// order, which gfx950 does and no specification promises; (ii) a ticketed kernel must reach det_pass / det_block_leave in EVERY block —
// no early return before it (none has one; keep it so); (iii) the mode must not be switched while launches are in flight. Kernels that
// do not add to shared global memory take no ticket (e.g. norm_bwd_stream_kernel, the frozen norms' backward: no dweight, no atomics).
__device__ __forceinline__ unsigned det_block_id() { return (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; }
__device__ __forceinline__ unsigned det_block_count() { return gridDim.x * gridDim.y * gridDim.z; }
// one thread of the block does all of its adds
__device__ __forceinline__ void det_wait(unsigned* t) {
  if (!t) return;
  const unsigned turn = det_block_id();
  while (__hip_atomic_load(t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != turn) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void det_pass(unsigned* t) {
  if (!t) return;
  const unsigned turn = det_block_id();
  __threadfence();
  __hip_atomic_store(t, turn + 1 == det_block_count() ? 0u : turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// every thread of the block adds (each address at most once per block, or always from the same thread): call from ALL threads
__device__ __forceinline__ void det_block_enter(unsigned* t) {
  if (!t) return;
  if (threadIdx.x == 0 && threadIdx.y == 0) det_wait(t);
  __syncthreads();
}
__device__ __forceinline__ void det_block_leave(unsigned* t) {
  if (!t) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) det_pass(t);
}
// host: the mode, and the ticket of one launch (nullptr when the mode is off)
bool grove_det_on();
unsigned* grove_det_ticket();

// error plumbing (host)
void grove_set_error(const char* fmt, ...);
#define GROVE_CHECK(cond, code, ...)         \
  do {                                       \
    if (!(cond)) {                           \
      grove_set_error(__VA_ARGS__);          \
      return (code);                         \
    }                                        \
  } while (0)
#define GROVE_LAUNCH_CHECK()                                          \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      grove_set_error("HIP launch failed: %s", hipGetErrorString(e_)); \
      return GROVE_E_HIP;                                             \
    }                                                                 \
  } while (0)
