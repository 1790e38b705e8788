// Error plumbing + version for libgrove_hip.so
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>

static thread_local char g_err[512] = "";

void grove_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// deterministic mode: see common.h (det_wait / det_pass). The ticket ring is CALLER-OWNED like every other workspace (zeroed unsigned
// words on the device the caller launches on; one process drives one GPU): a launch takes the next ticket and its last block puts the 0
// back, so a ticket is reused only `n` ticketed launches later.
static std::atomic<unsigned*> g_det_ring{nullptr};
static std::atomic<unsigned> g_det_n{0}, g_det_next{0};
bool grove_det_on() { return g_det_ring.load(std::memory_order_relaxed) != nullptr; }
unsigned* grove_det_ticket() {
  const unsigned n = g_det_n.load(std::memory_order_acquire);  // n before the ring: a concurrent switch-off must not leave n == 0 behind a live ring
  unsigned* ring = g_det_ring.load(std::memory_order_acquire);
  if (!ring || n == 0) return nullptr;
  return ring + g_det_next.fetch_add(1, std::memory_order_relaxed) % n;
}
extern "C" int grove_set_deterministic(void* tickets, int32_t n) {
  GROVE_CHECK(!tickets || n >= 64, GROVE_E_SHAPE, "set_deterministic: the ticket ring needs at least 64 zeroed words");
  g_det_ring.store(nullptr);
  g_det_n.store(tickets ? (unsigned)n : 0u);
  g_det_ring.store((unsigned*)tickets);
  return GROVE_OK;
}
extern "C" int grove_deterministic(void) { return grove_det_on() ? 1 : 0; }

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
  SZ(grove_wino3d_params);
#undef SZ
  return -1;
}
