// Error plumbing + version for libgrove_hip.so
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

void grove_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int grove_version(void) { return 1; }

extern "C" int grove_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return GROVE_E_SHAPE;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return GROVE_OK;
}

// struct sizes, so that a language binding can verify its mirror of include/grove_hip.h
extern "C" int grove_sizeof(const char* name) {
#define SZ(T) if (!strcmp(name, #T)) return (int)sizeof(T)
  SZ(grove_gemm_params);
  SZ(grove_transpose_params);
  SZ(grove_norm_params);
  SZ(grove_norm_bwd_params);
  SZ(grove_softmax_params);
  SZ(grove_softmax_bwd_params);
  SZ(grove_relpos_params);
  SZ(grove_rel_bias_params);
  SZ(grove_rope_params);
  SZ(grove_rows_params);
  SZ(grove_small_attn_params);
  SZ(grove_box_head_params);
  SZ(grove_box_head_bwd_params);
  SZ(grove_flash_attn_params);
  SZ(grove_gemm_tn_params);
  SZ(grove_gemv_params);
  SZ(grove_decode_attn_params);
  SZ(grove_resample_params);
  SZ(grove_normalize_params);
  SZ(grove_gemm_f32_params);
  SZ(grove_gemm_fp8_params);
  SZ(grove_greedy_pick_params);
  SZ(grove_gemm_workspace);
  SZ(grove_gemm_plan);
#undef SZ
  return -1;
}
