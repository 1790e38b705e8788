// SAM's decomposed relative-position terms as two streams on the matrix cores (image_encoder.py:420-458, add_decomposed_rel_pos).
//
//   forward   rel'[(w, h), q, bin] = sum_d Q[(w, q), h, d] * Rcat[q][bin][d]          bf16, bins = [rel_h | rel_w], pre-divided by alpha
//   backward  dQ[(w, q), h, d]    += sum_bin d rel'[(w, h), q, bin] * RcatT[q][d][bin]   bf16, in place
//
// Both are "one small matrix per query POSITION q, shared by every window and head": a wave takes one q, keeps that
// position's table (<= 12 KB) in registers as the MFMA A operand and streams windows past it — the B operand is one window's
// 16 heads at position q (a 3 KB contiguous piece of the qkv row forward; sixteen 64-byte rows of d rel' backward), loaded
// global -> registers in fragment form, no LDS. Everything is HBM traffic: Q (144 MB) + rel' (58 MB) forward at SAM-H window
// size, d rel' + 2 x dQ backward. As batched 4608 x 32 x 96 GEMMs over 196 positions these ran at 2.2 TB/s (91 / 175 us).
//
// Accumulator map (v_mfma_f32_16x16x32_bf16; lane l: fr = l & 15, g = l >> 4): D[row = 16t + 4g + j][col = fr], rows = table
// rows (bins forward, head-dim backward), cols = heads: a lane owns four consecutive table rows of ONE head per tile, and the
// table rows are dealt to tile PAIRS so that those are eight consecutive bins / dims (paired_row): 16-byte accesses.
#include "common.h"

namespace {

constexpr int RB_THREADS = 256;

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_raw* p) { return *(const bf16x8_t*)p; }
__device__ __forceinline__ bf16x8_t zero_frag() {
  const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, z);
}

// Table row of A-operand row i of tile t when tiles are PAIRED: rows (2s, 4g + j) and (2s + 1, 4g + j) are table rows
// 32s + 8g + j and 32s + 8g + 4 + j, so a lane's accumulators of a tile pair are 8 consecutive bins / dims: 16-byte accesses.
__device__ __forceinline__ int paired_row(int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }


// Bit u of the result: window w0 + u keeps position q as a QUERY (grove_rel_bias_params.q_valid, the window kernels' rule: the
// top-left vy x vx positions of the (L / kw) x kw window). Padded positions get no rel' row and no dq contribution: the
// attention kernels neither read nor write them. NULL: every window. A wave's chunk holds at most 64 windows.
__device__ __forceinline__ unsigned long long valid_windows(const grove_rel_bias_params& p, int q, int w0, int w1, int lane) {
  if (!p.q_valid) return ~0ull;
  const int qy = q / p.kw, qx = q - qy * p.kw;
  bool ok = false;
  if (w0 + lane < w1) ok = qy < p.q_valid[2 * (w0 + lane)] && qx < p.q_valid[2 * (w0 + lane) + 1];
  return __builtin_amdgcn_ballot_w64(ok);
}

// KS: 32-deep steps along the head dim (hp / 32); NP: pairs of 16-bin tiles (rel_ld / 32)
template <int KS, int NP>
__global__ __launch_bounds__(RB_THREADS) void rel_bias_fwd_kernel(const grove_rel_bias_params p, const int win_per_wave) {
  const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
  const int q = blockIdx.x;
  const int w0 = (blockIdx.y * (RB_THREADS / 64) + (threadIdx.x >> 6)) * win_per_wave;
  const int w1 = min(w0 + win_per_wave, p.nb);
  if (w0 >= w1) return;
  const bf16_raw* R = (const bf16_raw*)p.table + (int64_t)q * p.rel_ld * p.hp;
  bf16x8_t rf[2 * NP][KS];
#pragma unroll
  for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rf[t][ks] = ld_frag(R + paired_row(t, fr) * p.hp + 32 * ks + 8 * g);
  const bool head_ok = fr < p.nh;
  const int hq = min(fr, p.nh - 1);
  const bf16_raw* Q = (const bf16_raw*)p.q + (int64_t)q * p.ld_q + hq * p.hp + 8 * g;
  bf16_raw* O = (bf16_raw*)p.rel + ((int64_t)hq * p.L + q) * p.rel_ld + 8 * g;
  const int64_t q_step = (int64_t)p.L * p.ld_q, o_step = (int64_t)p.nh * p.L * p.rel_ld;
  const unsigned long long live = valid_windows(p, q, w0, w1, lane);
  constexpr int U = 4;  // windows in flight per wave (loads of all U first: the stream is latency-bound otherwise)
  for (int w = w0; w < w1; w += U) {
    if (((live >> (w - w0)) & ((1ull << U) - 1)) == 0) continue;  // (wave-uniform)
    bf16x8_t qf[U][KS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool on = (live >> (w - w0 + u)) & 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[u][ks] = on ? ld_frag(Q + min(w + u, w1 - 1) * q_step + 32 * ks) : zero_frag();
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[2 * s][ks], qf[u][ks], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[2 * s + 1][ks], qf[u][ks], a1, 0, 0, 0);
        }
        if (head_ok && w + u < w1 && ((live >> (w - w0 + u)) & 1))
          *(u32x4_t*)(O + (w + u) * o_step + 32 * s) = u32x4_t{pack2bf(a0[0], a0[1]), pack2bf(a0[2], a0[3]), pack2bf(a1[0], a1[1]), pack2bf(a1[2], a1[3])};
      }
  }
}

// KB: 32-deep steps along the bins (rel_ld / 32); NP: tile pairs = 32-wide slabs of the head dim that hold real dims (the last
// may reach into the pad dims, whose table rows are zero); NS: 1 if one more 16-wide tile of real dims follows them
template <int KB, int NP, int NS>
__global__ __launch_bounds__(RB_THREADS) void rel_bias_bwd_kernel(const grove_rel_bias_params p, const int win_per_wave) {
  const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
  const int q = blockIdx.x;
  const int w0 = (blockIdx.y * (RB_THREADS / 64) + (threadIdx.x >> 6)) * win_per_wave;
  const int w1 = min(w0 + win_per_wave, p.nb);
  if (w0 >= w1) return;
  const bf16_raw* R = (const bf16_raw*)p.table + (int64_t)q * p.hp * p.rel_ld;  // RcatT[q]: [hp][rel_ld]
  constexpr int NTILE = 2 * NP + NS;
  bf16x8_t rt[NTILE][KB];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) rt[t][kb] = ld_frag(R + (t < 2 * NP ? paired_row(t, fr) : 16 * t + fr) * p.rel_ld + 32 * kb + 8 * g);
  const bool head_ok = fr < p.nh;
  const int hq = min(fr, p.nh - 1);
  const bf16_raw* D = (const bf16_raw*)p.rel + ((int64_t)hq * p.L + q) * p.rel_ld + 8 * g;
  // dq rows: (window, position) in the layout of q, or token order through dq_map (compact heads of dq_hs columns)
  const int32_t* __restrict__ dmap = p.dq_map ? p.dq_map + q : nullptr;
  bf16_raw* DQ = (bf16_raw*)p.dq + (dmap ? (int64_t)hq * p.dq_hs : (int64_t)q * p.ld_dq + hq * p.hp);
  const int64_t d_step = (int64_t)p.nh * p.L * p.rel_ld, q_step = (int64_t)p.L * p.ld_dq;
  auto dq_off = [&](int w) -> int64_t { return dmap ? (int64_t)dmap[(int64_t)w * p.L] * p.ld_dq : w * q_step; };
  const unsigned long long live = valid_windows(p, q, w0, w1, lane);
  constexpr int U = 2;  // windows in flight per wave
  for (int w = w0; w < w1; w += U) {
    if (((live >> (w - w0)) & ((1ull << U) - 1)) == 0) continue;  // (wave-uniform)
    bf16x8_t df[U][KB];
    u32x4_t old[U][NP];
    u32x2_t old1[U];
    int64_t qoff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int wu = ((live >> (w - w0 + u)) & 1) ? min(w + u, w1 - 1) : w0 + __builtin_ctzll(live);  // (a dead window reads a live one's rows: finite, L2-hot, never stored)
      qoff[u] = dq_off(wu);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) df[u][kb] = ld_frag(D + wu * d_step + 32 * kb);
#pragma unroll
      for (int s = 0; s < NP; ++s) old[u][s] = *(const u32x4_t*)(DQ + qoff[u] + 32 * s + 8 * g);
      if (NS) old1[u] = *(const u32x2_t*)(DQ + qoff[u] + 32 * NP + 4 * g);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = head_ok && w + u < w1 && ((live >> (w - w0 + u)) & 1);
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rt[2 * s][kb], df[u][kb], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rt[2 * s + 1][kb], df[u][kb], a1, 0, 0, 0);
        }
        const u32x4_t o = old[u][s];
        const u32x4_t r = u32x4_t{pack2bf(a0[0] + bf_lo(o.x), a0[1] + bf_hi(o.x)), pack2bf(a0[2] + bf_lo(o.y), a0[3] + bf_hi(o.y)),
                                  pack2bf(a1[0] + bf_lo(o.z), a1[1] + bf_hi(o.z)), pack2bf(a1[2] + bf_lo(o.w), a1[3] + bf_hi(o.w))};
        if (ok) *(u32x4_t*)(DQ + qoff[u] + 32 * s + 8 * g) = r;
      }
      if (NS) {
        f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rt[2 * NP][kb], df[u][kb], a, 0, 0, 0);
        const u32x2_t o = old1[u];
        const u32x2_t r = u32x2_t{pack2bf(a[0] + bf_lo(o.x), a[1] + bf_hi(o.x)), pack2bf(a[2] + bf_lo(o.y), a[3] + bf_hi(o.y))};
        if (ok) *(u32x2_t*)(DQ + qoff[u] + 32 * NP + 4 * g) = r;
      }
    }
  }
}

int check(const grove_rel_bias_params* p, const char* name, bool bwd) {
  GROVE_CHECK(p && p->nb > 0 && p->nh > 0 && p->L > 0, GROVE_E_SHAPE, "%s: bad shape", name);
  GROVE_CHECK(p->nh <= 16, GROVE_E_SHAPE, "%s: nh=%d: at most 16 heads (one MFMA column tile)", name, p->nh);
  GROVE_CHECK(p->hp % 32 == 0 && p->hp <= 128 && (p->rel_ld == 32 || p->rel_ld == 64), GROVE_E_SHAPE,
              "%s: hp=%d must be a multiple of 32 up to 128, rel_ld=%d must be 32 or 64", name, p->hp, p->rel_ld);
  GROVE_CHECK(p->hd > 0 && p->hd <= p->hp, GROVE_E_SHAPE, "%s: hd=%d outside (0, hp]", name, p->hd);
  GROVE_CHECK(!p->q_valid || (p->kw > 0 && p->L % p->kw == 0), GROVE_E_SHAPE, "%s: q_valid needs kw (window width) dividing L", name);
  GROVE_CHECK(!p->dq_map || (bwd && p->q_valid && p->dq_hs >= p->hd && p->dq_hs % 8 == 0 && p->hd % 16 == 0), GROVE_E_SHAPE,
              "%s: dq_map (token-order dq) needs the backward, q_valid, and compact heads of dq_hs = %d >= hd columns in whole 16-byte pieces", name, p->dq_hs);
  GROVE_CHECK(p->table && p->rel && (bwd ? p->dq != nullptr : p->q != nullptr), GROVE_E_SHAPE, "%s: null operand", name);
  GROVE_CHECK((bwd ? p->ld_dq : p->ld_q) % 8 == 0, GROVE_E_ALIGN, "%s: leading dims must be multiples of 8", name);
  GROVE_CHECK((((uintptr_t)p->table | (uintptr_t)p->rel | (uintptr_t)(bwd ? p->dq : p->q)) & 15) == 0, GROVE_E_ALIGN, "%s: operands must be 16-byte aligned", name);
  return GROVE_OK;
}

// windows per wave: enough waves to fill the chip (~8 per SIMD), few enough that the table load (the only reuse) is amortised
inline void plan(const grove_rel_bias_params* p, int& win_per_wave, dim3& grid) {
  const long target_waves = 8192;
  long per_q = (target_waves + p->L - 1) / p->L;            // waves per position
  win_per_wave = (int)((p->nb + per_q - 1) / per_q);
  if (win_per_wave < 4) win_per_wave = p->nb < 4 ? p->nb : 4;
  if (win_per_wave > 64) win_per_wave = 64;  // (valid_windows: one ballot bit per window of a wave's chunk)
  const int waves = (p->nb + win_per_wave - 1) / win_per_wave;
  grid = dim3(p->L, (waves + RB_THREADS / 64 - 1) / (RB_THREADS / 64), 1);
}

}  // namespace

extern "C" int grove_rel_bias_fwd(const grove_rel_bias_params* p, void* stream) {
  int rc = check(p, "rel_bias_fwd", false);
  if (rc) return rc;
  int wpw;
  dim3 grid;
  plan(p, wpw, grid);
  hipStream_t s = (hipStream_t)stream;
#define RBF(KS, NP) hipLaunchKernelGGL((rel_bias_fwd_kernel<KS, NP>), grid, dim3(RB_THREADS), 0, s, *p, wpw)
  const int ks = p->hp / 32;
  if (p->rel_ld == 32) {
    if (ks == 1) RBF(1, 1); else if (ks == 2) RBF(2, 1); else if (ks == 3) RBF(3, 1); else RBF(4, 1);
  } else {
    if (ks == 1) RBF(1, 2); else if (ks == 2) RBF(2, 2); else if (ks == 3) RBF(3, 2); else RBF(4, 2);
  }
#undef RBF
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_rel_bias_bwd(const grove_rel_bias_params* p, void* stream) {
  int rc = check(p, "rel_bias_bwd", true);
  if (rc) return rc;
  int wpw;
  dim3 grid;
  plan(p, wpw, grid);
  hipStream_t s = (hipStream_t)stream;
  // real dims in 32-wide slabs, plus one 16-wide tile when the remainder is 1..16 (hd = 80: two slabs + one tile; the pad
  // columns a slab reaches into get + 0: their table rows are zero)
  const int rem = p->hd % 32;
  const int np = p->hd / 32 + (rem > 16 ? 1 : 0), ns = (rem > 0 && rem <= 16) ? 1 : 0;
#define RBB(KB, NP, NS) hipLaunchKernelGGL((rel_bias_bwd_kernel<KB, NP, NS>), grid, dim3(RB_THREADS), 0, s, *p, wpw)
#define RBB_D(KB)                                                                    \
  do {                                                                               \
    if (ns) {                                                                        \
      if (np == 0) RBB(KB, 0, 1); else if (np == 1) RBB(KB, 1, 1); else if (np == 2) RBB(KB, 2, 1); else RBB(KB, 3, 1); \
    } else {                                                                         \
      if (np == 1) RBB(KB, 1, 0); else if (np == 2) RBB(KB, 2, 0); else if (np == 3) RBB(KB, 3, 0); else RBB(KB, 4, 0); \
    }                                                                                \
  } while (0)
  if (p->rel_ld == 32) RBB_D(1); else RBB_D(2);
#undef RBB_D
#undef RBB
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
