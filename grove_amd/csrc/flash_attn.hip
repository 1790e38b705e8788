// Fused (flash-style) multi-head attention for gfx950: forward, and backward as two kernels
// (dK/dV per key block, dQ per query block) — no score matrix ever reaches HBM.
//
// Layout conventions (v_mfma_f32_16x16x32_bf16, wave64; lane l: fr = l & 15, g = l >> 4):
//   * operands live in LDS as plain row-major [row][HS] bf16 tiles with a row stride of HS*2 + 32 bytes:
//     that stride is conflict-free both for ds_read_b128 row reads (MFMA A/B fragments along the head dim)
//     and for ds_read_b64_tr_b16 transposed reads (fragments along the token dim).
//   * a 16x16 accumulator tile holds D[row = 4g + r][col = fr]. Products are oriented so that the NEXT
//     product sums over the accumulator's ROW index: the bf16-packed accumulator then is an MFMA operand
//     with no cross-lane movement (k-order inside a 32-step is permuted identically on both operands:
//     element j of lane group g is row 4g + j (j < 4) of tile 2s and row 4g + j - 4 of tile 2s + 1).
//   * forward / dQ kernels compute S^T[key][q] (lane = query: softmax statistics are per lane, reduced
//     over the 4 lane groups with two shuffles); the dK/dV kernel computes S[q][key] (lane = key).
// Masks: causal (key j visible iff j <= i + Lk - Lq), per-batch kv_len (right padding), and SAM's
// decomposed relative-position bias rel[q][j / kw] + rel[q][kh + j % kw] (image_encoder.py:420-458).
// Replaces: modeling_clip.py:279-319, image_encoder.py:310-319, HF LlamaAttention / flash-attn-2 varlen.
#include "common.h"

namespace {

constexpr int NTHR = 256;
constexpr int BKV = 64;  // keys (or queries, in the dK/dV kernel) staged per LDS tile

template <int HS>
struct Cfg {
  static constexpr int KS = HS / 32;          // 32-deep k-steps along the head dim
  static constexpr int DT = HS / 16;          // 16-wide tiles along the head dim
  static constexpr int ROWB = HS * 2 + 32;    // LDS row stride in bytes
  static constexpr int TILEB = BKV * ROWB;    // one staged tile
  static constexpr int CPR = HS / 8;          // 16-byte chunks per row
};

__device__ __forceinline__ bf16x8_t lds_row_frag(const char* tile, int rowb, int row, int kchunk) {
  return *(const bf16x8_t*)(tile + row * rowb + kchunk * 16);
}

// two transposed reads -> the 8-element fragment {rows 4g..4g+3, rows 16+4g..16+4g+3} of column (c0 + fr)
__device__ __forceinline__ bf16x8_t lds_tr_frag(const char* tile, int rowb, int row0, int c0, int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const int qq = fr >> 2, pp = fr & 3;
  const char* a0 = tile + (row0 + 4 * g + qq) * rowb + (c0 + 4 * pp) * 2;
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 16 * rowb));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t pack_frag(const f32x4_t a, const f32x4_t b) {
  const u32x4_t u = u32x4_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, u);
}

// stage `rows` rows (clamped to [0, nrows_valid-1]) of a [*, ld] bf16 matrix into an LDS tile
template <int HS>
__device__ __forceinline__ void stage_tile(char* tile, const bf16_raw* __restrict__ src, int ld, int row0, int nrows_valid, int tid) {
  using C = Cfg<HS>;
  constexpr int PER = (BKV * C::CPR) / NTHR;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int q = tid + i * NTHR;
    const int r = q / C::CPR, c = q - r * C::CPR;
    const int gr = min(row0 + r, nrows_valid - 1);
    const u32x4_t v = *(const u32x4_t*)(src + (int64_t)gr * ld + c * 8);
    *(u32x4_t*)(tile + r * C::ROWB + c * 16) = v;
  }
}

// split form (T14: issue the global loads early, write LDS late): the next tile's loads are in flight while the
// current tile feeds the MFMAs
template <int HS>
struct TileRegs {
  u32x4_t v[(BKV * (HS / 8)) / NTHR];
};
template <int HS>
__device__ __forceinline__ void load_tile(TileRegs<HS>& t, const bf16_raw* __restrict__ src, int ld, int row0, int nrows_valid, int tid) {
  using C = Cfg<HS>;
  constexpr int PER = (BKV * C::CPR) / NTHR;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int q = tid + i * NTHR;
    const int r = q / C::CPR, c = q - r * C::CPR;
    const int gr = min(row0 + r, nrows_valid - 1);
    t.v[i] = *(const u32x4_t*)(src + (int64_t)gr * ld + c * 8);
  }
}
template <int HS>
__device__ __forceinline__ void store_tile(char* tile, const TileRegs<HS>& t, int tid) {
  using C = Cfg<HS>;
  constexpr int PER = (BKV * C::CPR) / NTHR;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int q = tid + i * NTHR;
    const int r = q / C::CPR, c = q - r * C::CPR;
    *(u32x4_t*)(tile + r * C::ROWB + c * 16) = t.v[i];
  }
}

// ---- decomposed rel-pos bias on the matrix cores ------------------------------------------------------
// bias[q][j] = rel[q][kh(j)] + rel[q][KH + kw(j)] = sum_bin rel[q][bin] * E[bin][j] with the 0/1 indicator
// E[bin][j] = (bin == kh(j)) | (bin == KH + kw(j)). rel is pre-divided by alpha and rounded to bf16 (the
// reference computes rel_h / rel_w in bf16 as well), so ONE extra 32-deep MFMA per score tile adds the bias into
// the QK^T accumulator, and in backward d rel = dS . E^T is one MFMA per (32 keys, 16 bins) — no LDS traffic,
// no atomics. kbin[j] = kh(j) | kw(j) << 8 (0xFFFF beyond Lk) is a small LDS table built once per block.
__device__ __forceinline__ void build_kbin(unsigned short* tab, int Lk, int n, int kw, int tid) {
  for (int j = tid; j < n; j += NTHR) tab[j] = j < Lk ? (unsigned short)((j / kw) | ((j % kw) << 8)) : (unsigned short)0xFFFF;
}
// indicator fragment of ONE key over bins bin0 .. bin0+7
__device__ __forceinline__ bf16x8_t efrag_key(unsigned kb, int bin0, int KH) {
  const int kh = kb & 0xff, kwb = KH + (kb >> 8);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t e;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) e[jj] = ((bin0 + jj) == kh || (bin0 + jj) == kwb) ? (short)0x3F80 : (short)0;
  return __builtin_bit_cast(bf16x8_t, e);
}
// 8-bin indicator fragment with a 1.0 at position idx (none if idx is outside 0..7)
__device__ __forceinline__ bf16x8_t onehot8(const int idx) {
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t e;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) e[jj] = jj == idx ? (short)0x3F80 : (short)0;
  return __builtin_bit_cast(bf16x8_t, e);
}
// rel'[q][bin0 .. bin0+7] (bf16, already divided by alpha; rows are rel_ld wide, rel_ld % 8 == 0)
__device__ __forceinline__ bf16x8_t relfrag(const bf16_raw* relrow, int bin0, int rel_ld) {
  if (bin0 < rel_ld) return *(const bf16x8_t*)(relrow + bin0);
  const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, z);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (exp2(-inf) = 0)

__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---- shared pieces of the query-major kernels (forward, dQ) ---------------------------------------------------
constexpr int ESB = 64 * 2 + 32;        // row stride of the per-tile indicator image E[64 keys][<=64 bins]
constexpr int ETILEB = BKV * ESB;

// E[key][bin] for the 64 keys of the current tile, built cooperatively (one key x 16 bins per thread)
__device__ __forceinline__ void build_etile(char* Es, int kv0, int Lk, int kw, int KH, int nrel, int tid) {
  const int key = tid >> 2, quarter = tid & 3;
  if (quarter * 16 >= nrel) return;
  const int j = kv0 + key;
  const unsigned kb = j < Lk ? (unsigned)((j / kw) | ((j % kw) << 8)) : 0xFFFFu;
  *(bf16x8_t*)(Es + key * ESB + quarter * 32) = efrag_key(kb, quarter * 16, KH);
  *(bf16x8_t*)(Es + key * ESB + quarter * 32 + 16) = efrag_key(kb, quarter * 16 + 8, KH);
}

// x[j] *= s on a bf16x8 fragment (one-time operand pre-scaling: the softmax then runs in the exp2 domain unscaled)
__device__ __forceinline__ bf16x8_t scale_frag(bf16x8_t f, float sc) {
  const u32x4_t u = __builtin_bit_cast(u32x4_t, f);
  const u32x4_t o = u32x4_t{pack2bf(bf_lo(u.x) * sc, bf_hi(u.x) * sc), pack2bf(bf_lo(u.y) * sc, bf_hi(u.y) * sc),
                            pack2bf(bf_lo(u.z) * sc, bf_hi(u.z) * sc), pack2bf(bf_lo(u.w) * sc, bf_hi(u.w) * sc)};
  return __builtin_bit_cast(bf16x8_t, o);
}

// reductions over the 4 lane groups (lanes fr, fr+16, fr+32, fr+48) with the half-swap permutes: no LDS traffic
__device__ __forceinline__ float group_max4(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum4(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}


// Launch order of a CAUSAL problem: the work of a 128-row block grows (query-major kernels) or shrinks (key-major) with its
// position in the sequence, and 768 blocks of 2..12 key tiles on 512 block slots end with the longest ones running alone. Blocks
// are dispatched in linear blockIdx order, so the sequence block becomes the SLOWEST index, longest first (LPT order).
__device__ __forceinline__ void causal_block_order(const bool causal, const bool reverse, int& bx, int& h, int& b) {
  bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (!causal) return;
  const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int nbh = gridDim.y * gridDim.z;
  const int qi = id / nbh, bh = id - qi * nbh;
  bx = reverse ? (int)gridDim.x - 1 - qi : qi;
  h = bh % (int)gridDim.y;
  b = bh / (int)gridDim.y;
}

// ================================================================================ forward
// min 2 waves per SIMD (<= 256 registers): keeps the MFMA accumulators in arch VGPRs — with the 512-register budget
// hipcc parks them in AGPRs and pays ~350 v_accvgpr moves per tile around the softmax VALU work.
template <int HS, int NRK>
__global__ __launch_bounds__(NTHR, 2) void flash_fwd_kernel(const grove_flash_attn_params p) {
  using C = Cfg<HS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + C::TILEB;
  char* Es = smem + 2 * C::TILEB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // an SGPR: everything derived from it (key / query ranges, mask tests) is wave-uniform control flow
  const int fr = lane & 15, g = lane >> 4;
  int bx, h, b;
  causal_block_order(p.causal != 0, true, bx, h, b);
  const int qblk = bx * 128;
  const int q0 = qblk + wave * 32;
  const bf16_raw* Q = (const bf16_raw*)p.q + (int64_t)b * p.sq + h * HS;
  const bf16_raw* K = (const bf16_raw*)p.k + (int64_t)b * p.sk + h * HS;
  const bf16_raw* V = (const bf16_raw*)p.v + (int64_t)b * p.sv + h * HS;
  constexpr bool REL = NRK > 0;
  // NRK == 3 (round 4, "register E"): rel_kw == rel_kh == 32, rel_ld == 64 (SAM's global blocks) — two bias k-steps whose indicator
  // fragments never touch LDS: a 64-key tile starts on a row boundary, so a key's kw bin is ((ni & 1) * 16 + fr) for EVERY tile (two
  // constant fragments) and its kh bin is kv0 / 32 + (ni >> 1) (two fragments per tile, a compare each). The per-tile build of the
  // LDS image (integer div / mod, 16 compares and 2 LDS writes per thread, 8 LDS reads per wave: 18 % of this kernel) is gone; the
  // MFMAs and their operands' VALUES are unchanged, so the results are bit-identical to the NRK = 2 instance.
  constexpr bool FE = NRK == 3;
  constexpr int NK = FE ? 2 : NRK;
  const int nrel = REL ? p.rel_ld : 0;
  constexpr int nrk = NK;  // 32-bin k-steps of the bias MFMA (compile time: 0 without rel, 1 for rel_ld <= 32, else 2)
  const float sc = p.alpha * 1.4426950408889634f;  // scores live in the exp2 domain

  bf16x8_t qf[2][C::KS];
  bf16x8_t relf[2][NK > 0 ? NK : 1];
  bf16x8_t ew[2];
  if constexpr (FE) {
    ew[0] = onehot8(fr - g * 8);
    ew[1] = onehot8(16 + fr - g * 8);
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int qi = min(q0 + mi * 16 + fr, p.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[mi][ks] = scale_frag(*(const bf16x8_t*)(Q + (int64_t)qi * p.ld_q + ks * 32 + g * 8), sc);
    if (nrk > 0) {
      const bf16_raw* rrow = (const bf16_raw*)p.rel + ((int64_t)(b * p.H + h) * p.Lq + qi) * nrel;
#pragma unroll
      for (int k2 = 0; k2 < NK; ++k2) relf[mi][k2] = scale_frag(relfrag(rrow, k2 * 32 + g * 8, nrel), sc);
    }
  }
  f32x4_t oacc[2][C::DT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[mi][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  int kv_end = p.Lk;
  if (p.kv_len) kv_end = min(kv_end, p.kv_len[b]);
  int kv_lim = kv_end;
  if (p.causal) kv_lim = min(kv_lim, min(qblk + 127, p.Lq - 1) + (p.Lk - p.Lq) + 1);

  TileRegs<HS> kreg, vreg;
  load_tile<HS>(kreg, K, p.ld_k, 0, p.Lk, tid);
  load_tile<HS>(vreg, V, p.ld_v, 0, p.Lk, tid);
  for (int kv0 = 0; kv0 < kv_lim; kv0 += BKV) {
    __syncthreads();
    store_tile<HS>(Ks, kreg, tid);
    store_tile<HS>(Vs, vreg, tid);
    if (nrk > 0 && !FE) build_etile(Es, kv0, p.Lk, p.rel_kw, p.rel_kh, nrel, tid);
    __syncthreads();
    if (kv0 + BKV < kv_lim) {
      load_tile<HS>(kreg, K, p.ld_k, kv0 + BKV, p.Lk, tid);
      load_tile<HS>(vreg, V, p.ld_v, kv0 + BKV, p.Lk, tid);
    }
    // wave-uniform work trimming: dead query waves, key tiles past the last visible key, mask only at edges
    int lim_hi = kv_end, lim_lo = kv_end;
    if (p.causal) {
      lim_hi = min(lim_hi, q0 + 31 + (p.Lk - p.Lq) + 1);
      lim_lo = min(lim_lo, q0 + (p.Lk - p.Lq) + 1);
    }
    if (q0 >= p.Lq || kv0 >= lim_hi) continue;
    const int nv = min(4, (lim_hi - kv0 + 15) >> 4);
    const bool need_mask = kv0 + BKV > lim_lo;
    f32x4_t s[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) s[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      if (ni < nv) {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8_t kf = lds_row_frag(Ks, C::ROWB, ni * 16 + fr, ks * 4 + g);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) s[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[mi][ks], s[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int k2 = 0; k2 < NK; ++k2) {
          bf16x8_t ef;
          if constexpr (FE) ef = k2 == 0 ? onehot8((kv0 >> 5) + (ni >> 1) - g * 8) : ew[ni & 1];
          else ef = lds_row_frag(Es, ESB, ni * 16 + fr, k2 * 4 + g);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) s[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, relf[mi][k2], s[mi][ni], 0, 0, 0);
        }
      }
    }
    bf16x8_t pf[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      if (need_mask) {
        const int qi = q0 + mi * 16 + fr;
        int lim = kv_end;
        if (p.causal) lim = min(lim, qi + (p.Lk - p.Lq) + 1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = kv0 + ni * 16 + g * 4 + r;
            if (j >= lim) s[mi][ni][r] = -INFINITY;
          }
      }
      float mx = fmaxf(fmaxf(s[mi][0][0], s[mi][0][1]), fmaxf(s[mi][0][2], s[mi][0][3]));
#pragma unroll
      for (int ni = 1; ni < 4; ++ni) mx = fmaxf(mx, fmaxf(fmaxf(s[mi][ni][0], s[mi][ni][1]), fmaxf(s[mi][ni][2], s[mi][ni][3])));
      mx = group_max4(mx);
      const float m_new = fmaxf(m_run[mi], mx);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      float rs = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp2(s[mi][ni][r] - m_use);
          s[mi][ni][r] = e;
          rs += e;
        }
      rs = group_sum4(rs);
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[mi]) != 0) {  // wave-uniform: rescale only when some row's max moved
        const float corr = fast_exp2(m_run[mi] - m_use);            // m_run = -inf -> 0
        l_run[mi] *= corr;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) oacc[mi][dt] *= corr;
      }
      l_run[mi] += rs;
      m_run[mi] = m_new;
      pf[mi][0] = pack_frag(s[mi][0], s[mi][1]);
      pf[mi][1] = pack_frag(s[mi][2], s[mi][3]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (s2 * 2 < nv) {
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const bf16x8_t vf = lds_tr_frag(Vs, C::ROWB, s2 * 32, dt * 16, lane);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) oacc[mi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[mi][s2], oacc[mi][dt], 0, 0, 0);
        }
      }
    }
  }
  // epilogue: lane holds O^T[d = dt*16 + 4g + r][q = fr]
  bf16_raw* O = (bf16_raw*)p.o + (int64_t)b * p.so + h * HS;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int qi = q0 + mi * 16 + fr;
    if (qi >= p.Lq) continue;
    const float inv = l_run[mi] > 0.f ? 1.f / l_run[mi] : 0.f;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      const f32x4_t o = oacc[mi][dt] * inv;
      *(u32x2_t*)(O + (int64_t)qi * p.ld_o + dt * 16 + g * 4) = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    }
    if (p.lse && g == 0) {
      // natural-log LSE of the scaled + biased scores
      const float mm = m_run[mi] == -INFINITY ? 0.f : m_run[mi];
      p.lse[(int64_t)(b * p.H + h) * p.Lq + qi] = (mm + log2f(fmaxf(l_run[mi], 1e-30f))) * 0.6931471805599453f;
    }
  }
}

// ================================================================================ delta = rowsum(dO * O)
template <bool VEC16>  // 16-byte loads when o / d_o rows allow them (the product's layouts do), else 4-byte
__global__ __launch_bounds__(NTHR) void flash_delta_kernel(const grove_flash_attn_params p) {
  // one 16-lane group per (b, h, q) row
  const int64_t t = (int64_t)blockIdx.x * (NTHR / 16) + (threadIdx.x >> 4);
  const int l16 = threadIdx.x & 15;
  const int64_t n = (int64_t)p.B * p.H * p.Lq;
  float acc = 0.f;
  if (t < n) {
    const int qi = (int)(t % p.Lq);
    const int bh = (int)(t / p.Lq);
    const int b = bh / p.H, h = bh - b * p.H;
    const bf16_raw* O = (const bf16_raw*)p.o + (int64_t)b * p.so + (int64_t)qi * p.ld_o + h * p.hs;
    const bf16_raw* dO = (const bf16_raw*)p.d_o + (int64_t)b * p.sdo + (int64_t)qi * p.ld_do + h * p.hs;
    if constexpr (VEC16) {
      for (int c = l16 * 8; c < p.hs; c += 128) {  // a 128-wide head row is one load per operand for the 16-lane group
        const u32x4_t a = *(const u32x4_t*)(O + c), d = *(const u32x4_t*)(dO + c);
        acc += bf_lo(a.x) * bf_lo(d.x) + bf_hi(a.x) * bf_hi(d.x) + bf_lo(a.y) * bf_lo(d.y) + bf_hi(a.y) * bf_hi(d.y) +
               bf_lo(a.z) * bf_lo(d.z) + bf_hi(a.z) * bf_hi(d.z) + bf_lo(a.w) * bf_lo(d.w) + bf_hi(a.w) * bf_hi(d.w);
      }
    } else {
      for (int c = l16 * 2; c < p.hs; c += 32) {
        const unsigned a = *(const unsigned*)(O + c), d = *(const unsigned*)(dO + c);
        acc += bf_lo(a) * bf_lo(d) + bf_hi(a) * bf_hi(d);
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (t < n && l16 == 0) p.delta[t] = acc;
}

// Inverse rotate-half RoPE on a lane's accumulators (backward kernels' epilogues; grove_flash_attn_params.rope): the lane holds dims
// dt * 16 + 4 g + r (r = 0..3) of ONE row for every 16-wide tile dt, i.e. both halves (dt and dt + DT / 2) of its rotation pairs.
template <int HS>
__device__ __forceinline__ void rope_inverse_acc(f32x4_t (&a)[HS / 16], const float* __restrict__ cs_row, int g) {
  constexpr int DT = HS / 16;
#pragma unroll
  for (int dt = 0; dt < DT / 2; ++dt) {
    const f32x4_t c = *(const f32x4_t*)(cs_row + dt * 16 + g * 4), s = *(const f32x4_t*)(cs_row + HS / 2 + dt * 16 + g * 4);
    const f32x4_t y1 = a[dt], y2 = a[dt + DT / 2];
    a[dt] = y1 * c + y2 * s;
    a[dt + DT / 2] = y2 * c - y1 * s;
  }
}

// ================================================================================ backward: dK, dV
// block = 128 keys (wave = 32 keys, K/V fragments in registers); loops over 64-query tiles of Q and dO in LDS.
template <int HS, int NRK>
__global__ __launch_bounds__(NTHR, (HS <= 96 ? 2 : 1)) void flash_bwd_dkv_kernel(const grove_flash_attn_params p) {
  using C = Cfg<HS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + C::TILEB;
  float* lse_s = (float*)(smem + 2 * C::TILEB);  // [64]
  float* del_s = lse_s + BKV;                    // [64]
  char* relq = (char*)(del_s + BKV);             // bf16 [64 q][64 bins], row stride RELB (rel / alpha)
  constexpr int RELB = 64 * 2 + 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // an SGPR: everything derived from it (key / query ranges, mask tests) is wave-uniform control flow
  const int fr = lane & 15, g = lane >> 4;
  int bx, h, b;
  causal_block_order(p.causal != 0, false, bx, h, b);
  const int kblk = bx * 128;
  const int k0 = kblk + wave * 32;
  const bf16_raw* Q = (const bf16_raw*)p.q + (int64_t)b * p.sq + h * HS;
  const bf16_raw* K = (const bf16_raw*)p.k + (int64_t)b * p.sk + h * HS;
  const bf16_raw* V = (const bf16_raw*)p.v + (int64_t)b * p.sv + h * HS;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + (int64_t)b * p.sdo + h * HS;
  const float* LSE = p.lse + (int64_t)(b * p.H + h) * p.Lq;
  const float* DEL = p.delta + (int64_t)(b * p.H + h) * p.Lq;
  const int nrel = NRK > 0 ? p.rel_ld : 0;
  const bf16_raw* REL = NRK > 0 ? (const bf16_raw*)p.rel + ((int64_t)(b * p.H + h) * p.Lq) * nrel : nullptr;
  constexpr int nrk = NRK;

  // K, V fragments of this wave's 32 keys as MFMA B operands: B[k = d][col = key = fr]
  bf16x8_t kf[2][C::KS], vf[2][C::KS];
  // indicator fragments E^T[bin][key] of this wave's keys (constant for the whole kernel)
  bf16x8_t ekf[2][NRK > 0 ? NRK : 1];
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int kj = k0 + nj * 16 + fr;
    const unsigned kb = (REL && kj < p.Lk) ? (unsigned)((kj / p.rel_kw) | ((kj % p.rel_kw) << 8)) : 0xFFFFu;
#pragma unroll
    for (int k2 = 0; k2 < NRK; ++k2) ekf[nj][k2] = efrag_key(kb, k2 * 32 + g * 8, p.rel_kh);
  }
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int kj = min(k0 + nj * 16 + fr, p.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      kf[nj][ks] = scale_frag(*(const bf16x8_t*)(K + (int64_t)kj * p.ld_k + ks * 32 + g * 8), p.alpha * 1.4426950408889634f);
      vf[nj][ks] = *(const bf16x8_t*)(V + (int64_t)kj * p.ld_v + ks * 32 + g * 8);
    }
  }
  f32x4_t dk[2][C::DT], dv[2][C::DT];
#pragma unroll
  for (int nj = 0; nj < 2; ++nj)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) { dk[nj][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[nj][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  int kv_end = p.Lk;
  if (p.kv_len) kv_end = min(kv_end, p.kv_len[b]);
  // causal: queries i with i + (Lk - Lq) >= key are the only ones that see this block's first key
  int qstart = 0;
  if (p.causal) qstart = max(0, kblk - (p.Lk - p.Lq));
  qstart = (qstart / BKV) * BKV;
  const float sc = p.alpha * 1.4426950408889634f;

  // register prefetch of the next Q / dO tile only where the 256-register budget has room for it
  constexpr bool PREFETCH = HS <= 64 || HS == 128;  // (head dim 128 runs one wave per SIMD: 512 registers)
  TileRegs<PREFETCH ? HS : 32> qreg, doreg;
  if constexpr (PREFETCH) {
    load_tile<HS>(qreg, Q, p.ld_q, qstart, p.Lq, tid);
    load_tile<HS>(doreg, dO, p.ld_do, qstart, p.Lq, tid);
  }
  for (int qt0 = qstart; qt0 < p.Lq; qt0 += BKV) {
    __syncthreads();
    if constexpr (PREFETCH) {
      store_tile<HS>(Qs, qreg, tid);
      store_tile<HS>(dOs, doreg, tid);
      if (qt0 + BKV < p.Lq) {
        load_tile<HS>(qreg, Q, p.ld_q, qt0 + BKV, p.Lq, tid);
        load_tile<HS>(doreg, dO, p.ld_do, qt0 + BKV, p.Lq, tid);
      }
    } else {
      stage_tile<HS>(Qs, Q, p.ld_q, qt0, p.Lq, tid);
      stage_tile<HS>(dOs, dO, p.ld_do, qt0, p.Lq, tid);
    }
    if (tid < BKV) {
      const int qi = min(qt0 + tid, p.Lq - 1);
      lse_s[tid] = LSE[qi] * 1.4426950408889634f;
      del_s[tid] = DEL[qi];
    }
    if (REL) {
      for (int t = tid; t < BKV * 8; t += NTHR) {  // 16-byte chunks of the 64 bf16 rows
        const int r = t >> 3, c = t & 7;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (c * 8 < nrel) v = *(const u32x4_t*)(REL + (int64_t)min(qt0 + r, p.Lq - 1) * nrel + c * 8);
        *(bf16x8_t*)(relq + r * RELB + c * 16) = scale_frag(__builtin_bit_cast(bf16x8_t, v), sc);
      }
    }
    __syncthreads();
    if (k0 >= kv_end) continue;  // wave-uniform: this wave's 32 keys are all padding / masked
    // two 32-query k-steps per staged tile
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (qt0 + s2 * 32 >= p.Lq) continue;
      if (p.causal && qt0 + s2 * 32 + 31 + (p.Lk - p.Lq) < k0) continue;  // every query of this half is above the diagonal
      // S[q][key] and dP[q][key] for q tiles 2*s2, 2*s2+1 and this wave's 2 key tiles
      f32x4_t sacc[2][2], pacc[2][2];
      const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};  // (round 5: the chains start on the MFMA's zero operand — instances without rel-pos)
      constexpr bool TRIM0 = NRK == 0;
      if constexpr (!TRIM0) {
#pragma unroll
        for (int qi_ = 0; qi_ < 2; ++qi_)
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) { sacc[qi_][nj] = zero4; pacc[qi_][nj] = zero4; }
      }
#pragma unroll
      for (int qi_ = 0; qi_ < 2; ++qi_)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const int row = (s2 * 2 + qi_) * 16 + fr;
          const bf16x8_t qa = lds_row_frag(Qs, C::ROWB, row, ks * 4 + g);
          const bf16x8_t da = lds_row_frag(dOs, C::ROWB, row, ks * 4 + g);
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) {
            sacc[qi_][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[nj][ks], (TRIM0 && ks == 0) ? zero4 : sacc[qi_][nj], 0, 0, 0);
            pacc[qi_][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[nj][ks], (TRIM0 && ks == 0) ? zero4 : pacc[qi_][nj], 0, 0, 0);
          }
        }
#pragma unroll
      for (int qi_ = 0; qi_ < 2; ++qi_)
#pragma unroll
        for (int k2 = 0; k2 < NRK; ++k2) {
          const bf16x8_t ra = lds_row_frag(relq, RELB, (s2 * 2 + qi_) * 16 + fr, k2 * 4 + g);
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) sacc[qi_][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra, ekf[nj][k2], sacc[qi_][nj], 0, 0, 0);
        }
      // P and dS (lane: key = fr of tile nj; rows q = 4g + r of tile qi_)
      f32x4_t lse4[2], del4[2];
#pragma unroll
      for (int qi_ = 0; qi_ < 2; ++qi_) {
        lse4[qi_] = *(const f32x4_t*)(lse_s + (s2 * 2 + qi_) * 16 + g * 4);
        del4[qi_] = *(const f32x4_t*)(del_s + (s2 * 2 + qi_) * 16 + g * 4);
      }
      const int qlo = qt0 + s2 * 32;
      const bool need_mask = (qlo + 31 >= p.Lq) || (k0 + 31 >= kv_end) || (p.causal && k0 + 31 > qlo + (p.Lk - p.Lq));
      bf16x8_t pfr[2], dsfr[2];
      // (round 5: the mask is ONE wave-uniform branch around two straight-line bodies; inside the element loops hipcc had turned it
      // into a scalar branch per score — 180 SALU instructions and 32 branches per tile)
      constexpr bool TRIM = NRK == 0;  // (rel-pos instances: at the register limit, the single body spills less — see the dQ kernel)
      if (!TRIM || need_mask) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int j = k0 + nj * 16 + fr;
          f32x4_t pp[2], dd[2];
#pragma unroll
          for (int qi_ = 0; qi_ < 2; ++qi_)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qi = qlo + qi_ * 16 + g * 4 + r;
              int lim = kv_end;
              if (p.causal) lim = min(lim, qi + (p.Lk - p.Lq) + 1);
              float pr = fast_exp2(sacc[qi_][nj][r] - lse4[qi_][r]);
              if (need_mask && !((j < lim) && (qi < p.Lq))) pr = 0.f;
              pp[qi_][r] = pr;
              dd[qi_][r] = pr * (pacc[qi_][nj][r] - del4[qi_][r]) * p.alpha;
            }
          pfr[nj] = pack_frag(pp[0], pp[1]);
          dsfr[nj] = pack_frag(dd[0], dd[1]);
        }
      } else {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          f32x4_t pp[2], dd[2];
#pragma unroll
          for (int qi_ = 0; qi_ < 2; ++qi_)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = fast_exp2(sacc[qi_][nj][r] - lse4[qi_][r]);
              pp[qi_][r] = pr;
              dd[qi_][r] = pr * (pacc[qi_][nj][r] - del4[qi_][r]) * p.alpha;
            }
          pfr[nj] = pack_frag(pp[0], pp[1]);
          dsfr[nj] = pack_frag(dd[0], dd[1]);
        }
      }
      // dV[key][d] += sum_q P[q][key] dO[q][d];  dK[key][d] += sum_q dS[q][key] Q[q][d]
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const bf16x8_t dob = lds_tr_frag(dOs, C::ROWB, s2 * 32, dt * 16, lane);
        const bf16x8_t qb = lds_tr_frag(Qs, C::ROWB, s2 * 32, dt * 16, lane);
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          // operands swapped (D^T): the lane then owns 4 CONSECUTIVE d of one key — 8-byte stores below instead of 2-byte ones
          dv[nj][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dob, pfr[nj], dv[nj][dt], 0, 0, 0);
          dk[nj][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, dsfr[nj], dk[nj][dt], 0, 0, 0);
        }
      }
    }
  }
  // lane holds dK^T/dV^T[d = dt*16 + 4g + r][key = fr]
  bf16_raw* DK = (bf16_raw*)p.dk + (int64_t)b * p.sdk + h * HS;
  bf16_raw* DV = (bf16_raw*)p.dv + (int64_t)b * p.sdv + h * HS;
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int kj = k0 + nj * 16 + fr;
    if (kj >= p.Lk) continue;
    if (p.rope) rope_inverse_acc<HS>(dk[nj], p.rope + (int64_t)kj * HS, g);  // key j sits at position j
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      const f32x4_t a = dk[nj][dt], c = dv[nj][dt];
      *(u32x2_t*)(DK + (int64_t)kj * p.ld_dk + dt * 16 + g * 4) = u32x2_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
      *(u32x2_t*)(DV + (int64_t)kj * p.ld_dv + dt * 16 + g * 4) = u32x2_t{pack2bf(c[0], c[1]), pack2bf(c[2], c[3])};
    }
  }
}

// ================================================================================ backward: dQ (+ d rel)
// The two 16-query tiles of a wave are processed one after the other per key tile (halves the live S / dP
// accumulators) so that the kernel fits the 256-register budget that keeps MFMA results in arch VGPRs.
template <int HS, int NRK>
__global__ __launch_bounds__(NTHR, 2) void flash_bwd_dq_kernel(const grove_flash_attn_params p, const int make_delta) {
  using C = Cfg<HS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + C::TILEB;
  char* Es = smem + 2 * C::TILEB;
  constexpr bool REL = NRK > 0;
  const int nrel = REL ? p.rel_ld : 0;
  constexpr int nrk = NRK;
  const int nbt = (REL && p.drel) ? (nrel + 15) >> 4 : 0;  // 16-bin tiles of d rel (<= 2 * NRK)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // an SGPR: everything derived from it (key / query ranges, mask tests) is wave-uniform control flow
  const int fr = lane & 15, g = lane >> 4;
  int bx, h, b;
  causal_block_order(p.causal != 0, true, bx, h, b);
  const int qblk = bx * 128;
  const int q0 = qblk + wave * 32;
  const bf16_raw* Q = (const bf16_raw*)p.q + (int64_t)b * p.sq + h * HS;
  const bf16_raw* K = (const bf16_raw*)p.k + (int64_t)b * p.sk + h * HS;
  const bf16_raw* V = (const bf16_raw*)p.v + (int64_t)b * p.sv + h * HS;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + (int64_t)b * p.sdo + h * HS;
  const float sc = p.alpha * 1.4426950408889634f;

  bf16x8_t qf[2][C::KS], dof[2][C::KS];
  bf16x8_t relf[2][NRK > 0 ? NRK : 1];
  constexpr int NBT = NRK > 0 ? 2 * NRK : 1;
  f32x4_t drl[2][NBT];
  f32x4_t dq[2][C::DT];
  float lse2[2], del[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int qi = min(q0 + mi * 16 + fr, p.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      qf[mi][ks] = scale_frag(*(const bf16x8_t*)(Q + (int64_t)qi * p.ld_q + ks * 32 + g * 8), sc);
      dof[mi][ks] = *(const bf16x8_t*)(dO + (int64_t)qi * p.ld_do + ks * 32 + g * 8);
    }
    if (nrk > 0) {
      const bf16_raw* rrow = (const bf16_raw*)p.rel + ((int64_t)(b * p.H + h) * p.Lq + qi) * nrel;
#pragma unroll
      for (int k2 = 0; k2 < NRK; ++k2) relf[mi][k2] = scale_frag(relfrag(rrow, k2 * 32 + g * 8, nrel), sc);
    }
    lse2[mi] = p.lse[(int64_t)(b * p.H + h) * p.Lq + qi] * 1.4426950408889634f;
    if (make_delta) {
      // delta = rowsum(dO * O) of this block's queries, made here (the lane already holds its 8-column pieces of the dO row) and
      // left in p.delta for the dK / dV kernel, which is launched after this one: no separate pass over o and d_o
      const bf16_raw* orow = (const bf16_raw*)p.o + (int64_t)b * p.so + h * HS + (int64_t)qi * p.ld_o;
      float dp = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const u32x4_t a = *(const u32x4_t*)(orow + ks * 32 + g * 8), d = __builtin_bit_cast(u32x4_t, dof[mi][ks]);
        dp += bf_lo(a.x) * bf_lo(d.x) + bf_hi(a.x) * bf_hi(d.x) + bf_lo(a.y) * bf_lo(d.y) + bf_hi(a.y) * bf_hi(d.y) +
              bf_lo(a.z) * bf_lo(d.z) + bf_hi(a.z) * bf_hi(d.z) + bf_lo(a.w) * bf_lo(d.w) + bf_hi(a.w) * bf_hi(d.w);
      }
      dp = group_sum(dp);
      del[mi] = dp;
      if (g == 0 && q0 + mi * 16 + fr < p.Lq) p.delta[(int64_t)(b * p.H + h) * p.Lq + qi] = dp;
    } else {
      del[mi] = p.delta[(int64_t)(b * p.H + h) * p.Lq + qi];
    }
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) drl[mi][bt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dq[mi][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  int kv_end = p.Lk;
  if (p.kv_len) kv_end = min(kv_end, p.kv_len[b]);
  int kv_lim = kv_end;
  if (p.causal) kv_lim = min(kv_lim, min(qblk + 127, p.Lq - 1) + (p.Lk - p.Lq) + 1);

  // register prefetch of the next K / V tile only where the 256-register budget has room for it
  constexpr bool PREFETCH = !(HS >= 96 && NRK > 0);
  TileRegs<PREFETCH ? HS : 32> kreg, vreg;
  if constexpr (PREFETCH) {
    load_tile<HS>(kreg, K, p.ld_k, 0, p.Lk, tid);
    load_tile<HS>(vreg, V, p.ld_v, 0, p.Lk, tid);
  }
  for (int kv0 = 0; kv0 < kv_lim; kv0 += BKV) {
    __syncthreads();
    if constexpr (PREFETCH) {
      store_tile<HS>(Ks, kreg, tid);
      store_tile<HS>(Vs, vreg, tid);
    } else {
      stage_tile<HS>(Ks, K, p.ld_k, kv0, p.Lk, tid);
      stage_tile<HS>(Vs, V, p.ld_v, kv0, p.Lk, tid);
    }
    if (nrk > 0) build_etile(Es, kv0, p.Lk, p.rel_kw, p.rel_kh, nrel, tid);
    __syncthreads();
    if constexpr (PREFETCH) {
      if (kv0 + BKV < kv_lim) {
        load_tile<HS>(kreg, K, p.ld_k, kv0 + BKV, p.Lk, tid);
        load_tile<HS>(vreg, V, p.ld_v, kv0 + BKV, p.Lk, tid);
      }
    }
    int lim_hi = kv_end;
    if (p.causal) lim_hi = min(lim_hi, q0 + 31 + (p.Lk - p.Lq) + 1);
    if (q0 >= p.Lq || kv0 >= lim_hi) continue;  // wave-uniform: dead query wave / nothing visible in this key tile
    const int nv = min(4, (lim_hi - kv0 + 15) >> 4);
    // round 5: (i) the S / dP chains start on the MFMA's zero operand (the 64 v_mov per tile that cleared the accumulators are gone);
    // (ii) the key mask runs on edge tiles only (wave-uniform test; interior tiles: no compare / select per score). Same arithmetic.
    int lim_lo = kv_end;
    if (p.causal) lim_lo = min(lim_lo, q0 + (p.Lk - p.Lq) + 1);
    // (the rel-pos instances sit at the 256-register limit: there the two-path form spills more than it saves — measured 1.74 -> 1.85 ms
    // on SAM's global backward — so they keep the single masked body)
    constexpr bool TRIM = NRK == 0;
    const bool edge = !TRIM || kv0 + BKV > lim_lo;
    const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f32x4_t s[4], dp[4];
      if constexpr (!TRIM) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { s[ni] = zero4; dp[ni] = zero4; }
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        if (ni >= nv) {
          if constexpr (TRIM) { s[ni] = zero4; dp[ni] = zero4; }
          continue;
        }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8_t ka = lds_row_frag(Ks, C::ROWB, ni * 16 + fr, ks * 4 + g);
          const bf16x8_t va = lds_row_frag(Vs, C::ROWB, ni * 16 + fr, ks * 4 + g);
          s[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[mi][ks], (TRIM && ks == 0) ? zero4 : s[ni], 0, 0, 0);
          dp[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[mi][ks], (TRIM && ks == 0) ? zero4 : dp[ni], 0, 0, 0);
        }
#pragma unroll
        for (int k2 = 0; k2 < NRK; ++k2) {
          const bf16x8_t ef = lds_row_frag(Es, ESB, ni * 16 + fr, k2 * 4 + g);
          s[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, relf[mi][k2], s[ni], 0, 0, 0);
        }
      }
      if (edge) {
        const int qi = q0 + mi * 16 + fr;
        int lim = kv_end;
        if (p.causal) lim = min(lim, qi + (p.Lk - p.Lq) + 1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = kv0 + ni * 16 + g * 4 + r;
            const float pr = j < lim ? fast_exp2(s[ni][r] - lse2[mi]) : 0.f;
            s[ni][r] = pr * (dp[ni][r] - del[mi]) * p.alpha;
          }
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[ni][r] = fast_exp2(s[ni][r] - lse2[mi]) * (dp[ni][r] - del[mi]) * p.alpha;
      }
      const bf16x8_t dsf0 = pack_frag(s[0], s[1]), dsf1 = pack_frag(s[2], s[3]);
      // dQ^T[d][q] += sum_key K^T[d][key] dS^T[key][q];   d rel'^T[bin][q] += sum_key E^T[bin][key] dS^T[key][q]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 * 2 >= nv) continue;
        const bf16x8_t dsf = s2 == 0 ? dsf0 : dsf1;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const bf16x8_t kt = lds_tr_frag(Ks, C::ROWB, s2 * 32, dt * 16, lane);
          dq[mi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[mi][dt], 0, 0, 0);
        }
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
          if (bt < nbt) {
            const bf16x8_t et = lds_tr_frag(Es, ESB, s2 * 32, bt * 16, lane);
            drl[mi][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(et, dsf, drl[mi][bt], 0, 0, 0);
          }
        }
      }
    }
  }
  bf16_raw* DQ = (bf16_raw*)p.dq + (int64_t)b * p.sdq + h * HS;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int qi = q0 + mi * 16 + fr;
    if (qi >= p.Lq) continue;
    if (p.rope) rope_inverse_acc<HS>(dq[mi], p.rope + (int64_t)(qi + p.Lk - p.Lq) * HS, g);  // query i sits at position i + Lk - Lq
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      const f32x4_t o = dq[mi][dt];
      *(u32x2_t*)(DQ + (int64_t)qi * p.ld_dq + dt * 16 + g * 4) = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    }
    if (REL && p.drel) {
      // lane holds d rel'^T[bin = bt*16 + 4g + r][q = fr]; rel' = rel / alpha, and dS carried alpha: no rescale
      bf16_raw* DR = (bf16_raw*)p.drel + ((int64_t)(b * p.H + h) * p.Lq + qi) * nrel;
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        if (bt < nbt) {
          const f32x4_t o = drl[mi][bt];
          *(u32x2_t*)(DR + bt * 16 + g * 4) = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        }
      }
    }
  }
}

int check(const grove_flash_attn_params* p, const char* name) {
  GROVE_CHECK(p && p->B > 0 && p->H > 0 && p->Lq > 0 && p->Lk > 0, GROVE_E_SHAPE, "%s: bad shape", name);
  GROVE_CHECK(p->hs == 32 || p->hs == 64 || p->hs == 96 || p->hs == 128, GROVE_E_SHAPE, "%s: head dim %d not in {32,64,96,128}", name, p->hs);
  GROVE_CHECK(p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && p->ld_o % 4 == 0, GROVE_E_ALIGN, "%s: leading dims must be multiples of 8", name);
  GROVE_CHECK(((uintptr_t)p->q & 15) == 0 && ((uintptr_t)p->k & 15) == 0 && ((uintptr_t)p->v & 15) == 0, GROVE_E_ALIGN, "%s: q/k/v must be 16-byte aligned", name);
  GROVE_CHECK(!p->rel || (p->rel_kw > 0 && (p->Lk + p->rel_kw - 1) / p->rel_kw <= p->rel_kh && p->rel_kh + p->rel_kw <= p->rel_ld &&
                          p->rel_ld <= 64 && p->rel_ld % 16 == 0 && ((uintptr_t)p->rel & 15) == 0),
              GROVE_E_SHAPE, "%s: rel dims mismatch (rows of Lk/rel_kw h-bins at 0.., w-bins at rel_kh.., row length rel_ld <= 64)", name);
  return GROVE_OK;
}

template <int HS>
size_t lds_fwd(const grove_flash_attn_params* p) { return 2 * Cfg<HS>::TILEB + (p->rel ? (size_t)ETILEB : 0); }

#define DISPATCH_HS(p, FN)                 \
  switch ((p)->hs) {                       \
    case 32: FN(32); break;                \
    case 64: FN(64); break;                \
    case 96: FN(96); break;                \
    default: FN(128); break;               \
  }

}  // namespace

// win_attn.hip: the LDS-resident kernels for SAM's 14 x 14 windows
bool grove_win_attn_applicable(const grove_flash_attn_params* p);
int grove_win_attn_fwd_launch(const grove_flash_attn_params* p, hipStream_t s);
int grove_win_attn_bwd_launch(const grove_flash_attn_params* p, hipStream_t s);
// flash_attn2.hip: the round-5 eight-wave kernels
bool grove_flash2_fwd_applicable(const grove_flash_attn_params* p);
int grove_flash2_fwd_launch(const grove_flash_attn_params* p, hipStream_t s);
bool grove_flash2_bwd_dq_applicable(const grove_flash_attn_params* p);
int grove_flash2_bwd_dq_launch(const grove_flash_attn_params* p, int make_delta, hipStream_t s);
bool grove_flash2_bwd_dkv_applicable(const grove_flash_attn_params* p);
int grove_flash2_bwd_dkv_launch(const grove_flash_attn_params* p, hipStream_t s);
static int g_win_attn = 1;  // 0 = always the general kernels (A/B arm: grove_flash_attn_set_window_kernels)
static int g_reg_e = 1;     // 0 = the LDS indicator tile also where the register form applies (A/B arm: grove_flash_attn_set_register_e)
extern "C" int grove_flash_attn_set_register_e(int32_t on) {
  g_reg_e = on != 0;
  return GROVE_OK;
}
extern "C" int grove_flash_attn_set_window_kernels(int32_t on) {
  g_win_attn = on != 0;
  return GROVE_OK;
}
extern "C" int grove_flash_attn_window_kernels_on(void) { return g_win_attn ? 1 : 0; }

extern "C" int grove_flash_attn_fwd(const grove_flash_attn_params* p, void* stream) {
  int rc = check(p, "flash_attn_fwd");
  if (rc) return rc;
  GROVE_CHECK(p->o, GROVE_E_SHAPE, "flash_attn_fwd: o required");
  hipStream_t s = (hipStream_t)stream;
  GROVE_CHECK(!p->o_map || (g_win_attn && grove_win_attn_applicable(p) && p->q_valid), GROVE_E_SHAPE,
              "flash_attn_fwd: o_map (token-order output) is a window-kernel feature and needs q_valid");
  // q_valid / pad_k / pad_v exist only in the window kernels: the general ones below would read the (uninitialised) pad rows instead
  GROVE_CHECK(!(p->q_valid || p->pad_k || p->pad_v) || (g_win_attn && grove_win_attn_applicable(p)), GROVE_E_SHAPE,
              "flash_attn_fwd: q_valid / pad_k / pad_v are window-kernel features, but this problem does not take the window kernels");
  GROVE_CHECK(!p->rel_table || (g_win_attn && grove_win_attn_applicable(p)), GROVE_E_SHAPE,
              "flash_attn_fwd: rel_table (rel-pos terms made inside the kernel) is a window-kernel feature: 14 x 14-like windows, hs_valid 80, rel_ld 32, "
              "at most 64 table rows, 16-byte aligned table / rel");
  if (g_win_attn && grove_win_attn_applicable(p)) {
    grove_win_attn_fwd_launch(p, s);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  if (grove_flash2_fwd_applicable(p)) {
    grove_flash2_fwd_launch(p, s);
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  dim3 grid((p->Lq + 127) / 128, p->H, p->B);
#define FWD_L(HS, NRK)                                                                                     \
  {                                                                                                        \
    hipFuncSetAttribute((const void*)flash_fwd_kernel<HS, NRK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((flash_fwd_kernel<HS, NRK>), grid, dim3(NTHR), lds, s, *p);                          \
  }
#define FWD(HS)                                                                                            \
  {                                                                                                        \
    const size_t lds = lds_fwd<HS>(p);                                                                     \
    if (!p->rel) FWD_L(HS, 0) else if (p->rel_ld <= 32) FWD_L(HS, 1)                                        \
    else if (HS == 96 && g_reg_e && p->rel_kw == 32 && p->rel_kh == 32 && p->rel_ld == 64) FWD_L(HS == 96 ? 96 : 32, 3) \
    else FWD_L(HS, 2)                                                                                       \
  }
  DISPATCH_HS(p, FWD)
#undef FWD
#undef FWD_L
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

extern "C" int grove_flash_attn_bwd(const grove_flash_attn_params* p, void* stream) {
  int rc = check(p, "flash_attn_bwd");
  if (rc) return rc;
  GROVE_CHECK(p->o && p->d_o && p->lse && p->delta && p->dq && p->dk && p->dv, GROVE_E_SHAPE, "flash_attn_bwd: o, d_o, lse, delta, dq, dk, dv required");
  GROVE_CHECK(!p->drel || p->rel, GROVE_E_SHAPE, "flash_attn_bwd: drel needs rel");
  GROVE_CHECK(!p->rope || (!p->q_valid && !p->o_map && ((uintptr_t)p->rope & 15) == 0), GROVE_E_SHAPE,
              "flash_attn_bwd: rope (fused inverse RoPE) is a general-kernel feature with a 16-byte aligned table");
  hipStream_t s = (hipStream_t)stream;
  GROVE_CHECK(!p->o_map || (g_win_attn && grove_win_attn_applicable(p) && p->q_valid && p->ld_do % 8 == 0), GROVE_E_SHAPE,
              "flash_attn_bwd: o_map (token-order o / d_o) is a window-kernel feature and needs q_valid");
  GROVE_CHECK(!p->g_tok || (p->o_map && p->pad_k && p->pad_v && p->o_hs > 0 && p->o_hs % 4 == 0), GROVE_E_SHAPE,
              "flash_attn_bwd: g_tok (token-order dq / dk / dv) needs o_map, o_hs (compact head stride, a multiple of 4) and pad_k / pad_v (gradients of "
              "padded positions have no row)");
  const bool win = g_win_attn && !p->rope && grove_win_attn_applicable(p) && p->ld_do % 8 == 0 && p->ld_dq % 4 == 0 && p->ld_dk % 4 == 0 && p->ld_dv % 4 == 0 &&
                   ((uintptr_t)p->d_o & 15) == 0 && ((uintptr_t)p->o & 15) == 0;
  GROVE_CHECK(!(p->q_valid || p->pad_k || p->pad_v) || win, GROVE_E_SHAPE,
              "flash_attn_bwd: q_valid / pad_k / pad_v are window-kernel features, but this problem does not take the window kernels "
              "(shape, alignment of o / d_o / dq / dk / dv, or grove_flash_attn_set_window_kernels(0))");
  GROVE_CHECK(!p->rel_table || (win && p->rel && !p->drel), GROVE_E_SHAPE,
              "flash_attn_bwd: rel_table is a window-kernel feature; it needs rel (the operand the forward left) and no drel (dq leaves complete)");
  if (win) {
    grove_win_attn_bwd_launch(p, s);  // one kernel: delta, dK / dV, then dQ / d rel (win_attn.hip)
    GROVE_LAUNCH_CHECK();
    return GROVE_OK;
  }
  const int64_t nrows = (int64_t)p->B * p->H * p->Lq;
  const bool vec16 = p->ld_o % 8 == 0 && p->ld_do % 8 == 0 && p->so % 8 == 0 && p->sdo % 8 == 0 && (((uintptr_t)p->o | (uintptr_t)p->d_o) & 15) == 0;
  // delta = rowsum(dO * O): made by the dQ kernel (launched first) when the rows allow 16-byte loads, else by its own pass
  const int make_delta = vec16 ? 1 : 0;
  if (!vec16) hipLaunchKernelGGL(flash_delta_kernel<false>, dim3((unsigned)((nrows + 15) / 16)), dim3(NTHR), 0, s, *p);
  const int nrel = p->rel ? p->rel_ld : 0;
  dim3 gk((p->Lk + 127) / 128, p->H, p->B), gq((p->Lq + 127) / 128, p->H, p->B);
  // round 5: the dQ (+ delta, + d rel') kernel in the eight-wave form where it applies; dK / dV stays the four-wave kernel
  const bool dq2 = make_delta && grove_flash2_bwd_dq_applicable(p);
  if (dq2) grove_flash2_bwd_dq_launch(p, make_delta, s);
  const bool dkv2 = grove_flash2_bwd_dkv_applicable(p);  // (delta is in p.delta by now: made by the dQ kernel or by flash_delta_kernel, same stream)
#define BWD_L(HS, NRK)                                                                                     \
  {                                                                                                        \
    hipFuncSetAttribute((const void*)flash_bwd_dkv_kernel<HS, NRK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1); \
    hipFuncSetAttribute((const void*)flash_bwd_dq_kernel<HS, NRK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);  \
    if (!dq2) hipLaunchKernelGGL((flash_bwd_dq_kernel<HS, NRK>), gq, dim3(NTHR), l2, s, *p, make_delta);    \
    if (!dkv2) hipLaunchKernelGGL((flash_bwd_dkv_kernel<HS, NRK>), gk, dim3(NTHR), l1, s, *p);              \
  }
#define BWD(HS)                                                                                            \
  {                                                                                                        \
    const size_t l1 = 2 * Cfg<HS>::TILEB + 2 * BKV * 4 + (nrel ? (size_t)BKV * (64 * 2 + 32) : 0);          \
    const size_t l2 = lds_fwd<HS>(p);                                                                       \
    if (!p->rel) BWD_L(HS, 0) else if (p->rel_ld <= 32) BWD_L(HS, 1) else BWD_L(HS, 2)                       \
  }
  DISPATCH_HS(p, BWD)
#undef BWD
#undef BWD_L
  if (dkv2) grove_flash2_bwd_dkv_launch(p, s);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
