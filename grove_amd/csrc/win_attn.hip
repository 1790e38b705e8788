// SAM window attention for gfx950: the 14 x 14 windows of ImageEncoderViT's non-global blocks
// (image_encoder.py:301-326 Attention with add_decomposed_rel_pos :420-458; window_partition :329-353).
//
// One (window, head) problem is tiny and fixed: L = 196 tokens, head dim 80 (stored with a row stride of 96), bias from
// 14 + 14 relative-position bins. The general flash kernels (flash_attn.hip) ran it at 6-8 % of the MFMA peak: 128-query
// blocks (196 = 128 + 68), 64-key tiles staged global -> registers -> LDS behind two __syncthreads each, the head dim padded
// to 96 in every product. Here the WHOLE problem lives in LDS:
//   * K and V of the (window, head) are copied ONCE by LDS-DMA (global_load_lds_dwordx4, no register staging) into dense
//     [208][80] bf16 images, 160-byte rows. That stride needs no padding and no swizzle: ds_read_b128 row fragments (rows fr,
//     16-byte chunk g) and ds_read_b64_tr_b16 transposed fragments (8 consecutive rows x 32 bytes per 32-lane half) both hit
//     64 distinct banks (row * 160 mod 256 walks the 8 multiples of 32; the b128 lane groups pair chunk c of 8 rows with
//     chunk c + 1 of the other 8);
//   * one barrier per block; after it every wave owns PAIRS of 16-query tiles (a K / V / indicator fragment read from LDS feeds
//     two MFMAs) and computes the full score row block S^T[208 keys][32 queries] in registers — no online softmax, no rescale;
//   * the head dim is 80 = 32 + 32 + 16: two v_mfma_f32_16x16x32_bf16 and one v_mfma_f32_16x16x16_bf16 per score tile, five
//     16-wide output tiles in P V — nothing is multiplied by the padding;
//   * the rel-pos bias is one more MFMA per score tile against a 0 / 1 indicator image E[key][bin] (flash_attn.hip's scheme;
//     here E is built once per block, 64-byte rows with a chunk XOR that makes its b128 reads conflict-free);
//   * the softmax denominator comes off the matrix cores too: P^T times an all-ones operand (one MFMA per 32 keys) instead of
//     52 VALU adds per lane — the VALU, not the MFMA pipe, is what this kernel saturates first.
// 76 KB of LDS per block -> two blocks per CU: one block's DMA prologue hides under the other's MFMAs.
#include "common.h"

namespace {

constexpr int WNT = 13;                 // 16-row tiles of queries / keys: 192 < L <= 208
constexpr int WROWB = 160;              // LDS row stride of the K / V images (80 bf16, dense)
constexpr int WNCH = WNT * 16 * 10;     // 16-byte chunks per image (2080)
constexpr int WINSTR = (WNCH + 63) / 64;  // wave-level LDS-DMA instructions per image (33; the last one runs with 32 lanes)
constexpr int WIMGB = WNCH * 16;        // bytes per image (33,280)
constexpr int WEB = WNT * 16 * 64;      // indicator image E[208][32 bins] bf16
constexpr int WTHR = 256;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ float wexp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ f32x4_t mfma32(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// The 16-deep steps (head dim 64..79; the 13th key / query tile) run as a 16x16x32 MFMA whose operands carry four zeros per lane:
// lane group g then contributes k-slots 8g .. 8g+3 = its four real products, the same sum as v_mfma_f32_16x16x16_bf16. The native
// 16-deep instruction inside an accumulation chain of 32-deep ones gave wrong tiles here (hipcc 7.2 schedules a dependent MFMA
// of a DIFFERENT shape too close behind its producer: only same-shape back-to-back accumulation is forwarded by the hardware);
// one shape throughout costs 8 cycles per such step and needs no hazard padding.
__device__ __forceinline__ f32x4_t mfma16(s16x4_t a, s16x4_t b, f32x4_t c) {
  const s16x8_t a8 = s16x8_t{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = s16x8_t{b[0], b[1], b[2], b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a8), __builtin_bit_cast(bf16x8_t, b8), c, 0, 0, 0);
}
#define KEEP_ALIVE4(a, b, c, d)
#define KEEP_ALIVE3(a, b, c)

// THE QUERIES THAT COUNT. window_partition pads the token grid to whole windows (image_encoder.py:329-353): at 32 x 32 -> 3 x 3
// windows of 14 x 14, 42 % of the windowed rows are padding. As KEYS they are real (k, v = the bias row) — as QUERIES their outputs
// are dropped by window_unpartition and carry no gradient, so the kernels work through the valid queries only: window b keeps
// its top-left vy x vx positions (p.q_valid[b] = {vy, vx}; NULL = all), compact query i is position (i / vx) * kw + i % vx, and
// the query-indexed loops, LDS images and stores run over nq = vy * vx rows (196, 56 or 16 instead of 196: 59 % of the
// query tiles per frame). Rows of o / lse / dq / d rel' at padded positions are not written, d_o there is not read.
struct QList {
  int nq, vx, kw, mvx;  // mvx = ceil(2^16 / vx): i / vx = (i * mvx) >> 16 for i < 208 (vx <= 14: exact below 334)
  int vy, mkw;          // mkw = ceil(2^16 / kw)
  __device__ __forceinline__ int pos(int i) const {
    const int r = (i * mvx) >> 16;
    return r * kw + (i - r * vx);
  }
  __device__ __forceinline__ bool real(int position) const {  // is this position of the window a real token?
    const int r = (position * mkw) >> 16;
    return r < vy && position - r * kw < vx;
  }
};
__device__ __forceinline__ QList make_qlist(const grove_flash_attn_params& p, int b) {
  QList ql;
  ql.kw = p.rel_kw;
  int vy = p.Lq / p.rel_kw;
  ql.vx = p.rel_kw;
  if (p.q_valid) {
    vy = p.q_valid[2 * b];
    ql.vx = p.q_valid[2 * b + 1];
  }
  ql.nq = p.q_valid ? vy * ql.vx : p.Lq;
  ql.mvx = (65536 + ql.vx - 1) / ql.vx;
  ql.vy = vy;
  ql.mkw = (65536 + ql.kw - 1) / ql.kw;
  return ql;
}

// rows 0 .. 207 (clamped to L - 1) x 10 chunks of one head's [*, 80] slice -> dense LDS image; every lane copies 16 bytes per
// instruction, the wave's 64 chunks land contiguously (LDS-DMA destinations are lane-linear)
__device__ __forceinline__ void dma_image(char* img, const bf16_raw* __restrict__ src, int ld, int L, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < (WINSTR + 3) / 4; ++j) {
    const int i = wave + 4 * j;
    if (i < WINSTR) {
      const int c = i * 64 + lane;
      const int row = (c * 6554) >> 16;  // c / 10 (exact for c < 16384)
      const int col = c - row * 10;
      const int gr = min(row, L - 1);
      if (c < WNCH)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)gr * ld + col * 8),
                                         (__attribute__((address_space(3))) void*)(img + i * 1024), 16, 0, 0);
    }
  }
}
// the same for a KEY-indexed operand whose rows at padded positions are all one row (pad != NULL: k / v of a zero token = the
// bias): those chunks come from `pad` (L2-resident), and the caller need not have filled the padded rows of src
__device__ __forceinline__ void dma_image_k(char* img, const bf16_raw* __restrict__ src, int ld, int L, const bf16_raw* __restrict__ pad,
                                            const QList& ql, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < (WINSTR + 3) / 4; ++j) {
    const int i = wave + 4 * j;
    if (i < WINSTR) {
      const int c = i * 64 + lane;
      const int row = (c * 6554) >> 16;
      const int col = c - row * 10;
      const int gr = min(row, L - 1);
      const bf16_raw* g = (pad && !ql.real(gr)) ? pad + col * 8 : src + (int64_t)gr * ld + col * 8;
      if (c < WNCH)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(img + i * 1024), 16,
                                         0, 0);
    }
  }
}
// the same for a QUERY-indexed operand: image row i = compact query min(i, nq - 1)
// rowmap (or NULL): position -> row of src (an operand kept in token order: grove_flash_attn_params.o_map)
// Split in two so that the row indices of a mapped operand can be fetched well before its DMA is issued (image_q_rows first,
// other work, then dma_image_rows): a DMA whose address hangs on an index load issued right before it pays that load's latency.
struct ImageRows { int r[(WINSTR + 3) / 4]; };
__device__ __forceinline__ void image_q_rows(ImageRows& rows, const QList& ql, int wave, int lane, const int32_t* __restrict__ rowmap) {
#pragma unroll
  for (int j = 0; j < (WINSTR + 3) / 4; ++j) {
    const int c = min((wave + 4 * j) * 64 + lane, WNCH - 1);
    const int row = (c * 6554) >> 16;
    const int gr = ql.pos(min(row, ql.nq - 1));
    rows.r[j] = rowmap ? rowmap[gr] : gr;
  }
}
__device__ __forceinline__ void dma_image_rows(char* img, const bf16_raw* __restrict__ src, int ld, const ImageRows& rows, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < (WINSTR + 3) / 4; ++j) {
    const int i = wave + 4 * j;
    if (i < WINSTR) {
      const int c = i * 64 + lane;
      const int row = (c * 6554) >> 16;
      const int col = c - row * 10;
      if (c < WNCH)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)rows.r[j] * ld + col * 8),
                                         (__attribute__((address_space(3))) void*)(img + i * 1024), 16, 0, 0);
    }
  }
}
__device__ __forceinline__ void dma_image_q(char* img, const bf16_raw* __restrict__ src, int ld, const QList& ql, int wave, int lane) {
  ImageRows rows;
  image_q_rows(rows, ql, wave, lane, nullptr);
  dma_image_rows(img, src, ld, rows, wave, lane);
}

// chunk position of E[key][8-bin chunk ch]: ch ^ m[(key >> 2) & 3], m = {0, 2, 3, 1}
__device__ __forceinline__ int e_swz(int key) { return (0x78 >> (((key >> 2) & 3) * 2)) & 3; }

__device__ __forceinline__ bf16x8_t e_chunk(unsigned kb, int bin0, int KH) {
  const int kh = kb & 0xff, kwb = KH + (kb >> 8);
  s16x8_t e;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) e[jj] = ((bin0 + jj) == kh || (bin0 + jj) == kwb) ? (short)0x3F80 : (short)0;
  return __builtin_bit_cast(bf16x8_t, e);
}

__device__ __forceinline__ void build_e(char* Es, int L, int kw, int KH, int tid) {
  for (int c = tid; c < WNT * 16 * 4; c += WTHR) {
    const int key = c >> 2, ch = c & 3;
    const unsigned kb = key < L ? (unsigned)((key / kw) | ((key % kw) << 8)) : 0xFFFFu;
    *(bf16x8_t*)(Es + key * 64 + ((ch ^ e_swz(key)) << 4)) = e_chunk(kb, ch * 8, KH);
  }
}

__device__ __forceinline__ bf16x8_t wscale(bf16x8_t f, float sc) {
  const u32x4_t u = __builtin_bit_cast(u32x4_t, f);
  const u32x4_t o = u32x4_t{pack2bf(bf_lo(u.x) * sc, bf_hi(u.x) * sc), pack2bf(bf_lo(u.y) * sc, bf_hi(u.y) * sc),
                            pack2bf(bf_lo(u.z) * sc, bf_hi(u.z) * sc), pack2bf(bf_lo(u.w) * sc, bf_hi(u.w) * sc)};
  return __builtin_bit_cast(bf16x8_t, o);
}
__device__ __forceinline__ s16x4_t wscale4(u32x2_t u, float sc) {
  const u32x2_t o = u32x2_t{pack2bf(bf_lo(u.x) * sc, bf_hi(u.x) * sc), pack2bf(bf_lo(u.y) * sc, bf_hi(u.y) * sc)};
  return __builtin_bit_cast(s16x4_t, o);
}

__device__ __forceinline__ bf16x8_t wpack(const f32x4_t a, const f32x4_t b) {
  const u32x4_t u = u32x4_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ s16x4_t wpack4(const f32x4_t a) {
  const u32x2_t u = u32x2_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
  return __builtin_bit_cast(s16x4_t, u);
}

__device__ __forceinline__ float gmax4(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// transposed fragments of a row-major [rows][80] image: {rows r0+4g..+3 | rows r0+16+4g..+3} of column c0 + fr (32-deep k-step),
// or the first half alone (16-deep k-step)
__device__ __forceinline__ s16x4_t tr4(const char* img, int r0, int c0, int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const char* a = img + (r0 + 4 * g + (fr >> 2)) * WROWB + (c0 + 4 * (fr & 3)) * 2;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a));
}
__device__ __forceinline__ bf16x8_t tr8(const char* img, int r0, int c0, int lane) {
  const s16x4_t lo = tr4(img, r0, c0, lane), hi = tr4(img, r0 + 16, c0, lane);
  const s16x8_t v = s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// Workgroups are dealt to the 8 XCDs round-robin (block i -> XCD i % 8). The heads of one window share cache lines (a head's row
// is 160 of 192 bytes, so every other 128-byte line holds two heads): problem = (window, head) with the head fastest, and
// consecutive problems go to the SAME XCD so that its L2 sees each line once (bijective for any grid size).
__device__ __forceinline__ int problem_of_block(int bid, int n) {
  const int per = n >> 3, rem = n & 7;   // XCD x owns per (+1 if x < rem) consecutive problems
  const int x = bid & 7, s = bid >> 3;
  return x * per + min(x, rem) + s;
}

struct QPair {          // B operands of one pair of 16-query tiles, pre-scaled into the exp2 domain
  bf16x8_t f[2][2];     // head-dim k-steps 0..31, 32..63
  s16x4_t t[2];         // head dim 64..79 (16-deep step)
  bf16x8_t rel[2];      // rel'[q][32 bins]
};

// REL_MODE 0: rel' = q . Rcat / alpha from HBM, scaled here; 1: the score-domain operand itself from HBM (written by the forward of the
// rel_table form); 2: not loaded (the caller makes it: rel_from_table)
template <int REL_MODE = 0>
__device__ __forceinline__ void load_qpair(QPair& q, const bf16_raw* __restrict__ Q, int ld_q, const bf16_raw* __restrict__ REL, int q0,
                                           const QList& ql, float sc, int fr, int g) {
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int qi = ql.pos(min(q0 + mi * 16 + fr, ql.nq - 1));
    const bf16_raw* row = Q + (int64_t)qi * ld_q;
    q.f[mi][0] = wscale(*(const bf16x8_t*)(row + g * 8), sc);
    q.f[mi][1] = wscale(*(const bf16x8_t*)(row + 32 + g * 8), sc);
    q.t[mi] = wscale4(*(const u32x2_t*)(row + 64 + g * 4), sc);
    if (REL_MODE == 0) q.rel[mi] = wscale(*(const bf16x8_t*)(REL + (int64_t)qi * 32 + g * 8), sc);
    if (REL_MODE == 1) q.rel[mi] = *(const bf16x8_t*)(REL + (int64_t)qi * 32 + g * 8);
  }
}

// ================================================================================ rel-pos inside the kernels
// grove_flash_attn_params.rel_table: the bias operand rel'[q][32 bins] of a query is 14 + 14 of the 27 + 27 products q . T[row] — which
// 14 depends on the query's own (row, column) in the window: bin b of the h part is T[(n - 1) - q_row + b] (the table is stored in
// key-minus-query order, so the bins of one query are CONSECUTIVE rows). The products of 16 queries with all 64 table rows are
// 12 MFMAs on the Q fragments the score chain holds anyway (G^T = T Q^T: lane (fr = query, g) gets table rows 16t + 4g + r); the
// per-query shift is a trip through a wave-private LDS scratch [16 queries][72] bf16 (4 x 8-byte writes, 8 two-byte reads per
// lane and tile). MEASURED (tools/bench_flash.py, 32 frames): forward 143 -> 171 us (182 with the operand kept for the backward) against
// the 27 us rel_bias_fwd stream it replaces; backward 402 -> 455 us against the 46 us rel_bias_bwd stream; config 2 end to end
// 1.4129 -> 1.4102 s. The extra work sits in the prologue of a latency-bound kernel (dependent MFMA chains and LDS round trips before
// the first score tile, one more barrier) — a wash, so sam.py keeps the two streams by default (GROVE_SAM_REL_IN_KERNEL=1 turns this on).
constexpr int WSCRB = 144;  // scratch row stride in bytes (72 bf16: rows 16-byte aligned, 36 banks apart)
struct RelTable {           // A operands: four tiles of 16 table rows
  bf16x8_t f[4][2];
  s16x4_t t[4];
};
__device__ __forceinline__ void load_rel_table(RelTable& rt, const bf16_raw* __restrict__ T, int fr, int g) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bf16_raw* row = T + (16 * t + fr) * 80;
    rt.f[t][0] = *(const bf16x8_t*)(row + g * 8);
    rt.f[t][1] = *(const bf16x8_t*)(row + 32 + g * 8);
    rt.t[t] = __builtin_bit_cast(s16x4_t, *(const u32x2_t*)(row + 64 + g * 4));
  }
}
// where bin b of the query at window position qi lives among the 64 table rows (-1: a pad bin)
struct RelShift {
  int hb, wb, nkh, KH, kw;
  __device__ __forceinline__ int col(int b) const {
    const bool h = b < nkh, w = b >= KH && b < KH + kw;
    return h ? hb + b : w ? wb + b : -1;
  }
};
__device__ __forceinline__ RelShift make_shift(int qi, int nkh, int KH, const QList& ql) {
  const int qy = (qi * ql.mkw) >> 16, qx = qi - qy * ql.kw;
  RelShift s;
  s.nkh = nkh, s.KH = KH, s.kw = ql.kw;
  s.hb = nkh - 1 - qy;
  s.wb = (2 * nkh - 1) + (ql.kw - 1) - qx - KH;
  return s;
}
// qf / qt: the pre-scaled Q fragments of 16 queries (lane: query fr, dims 8g.. of each k-step); returns their bias operand
__device__ __forceinline__ bf16x8_t rel_from_table(const RelTable& rt, const bf16x8_t (&qf)[2], s16x4_t qt, char* scr, const RelShift& sh, int fr, int g) {
  const f32x4_t z4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4_t a = mfma32(rt.f[t][0], qf[0], z4);
    a = mfma32(rt.f[t][1], qf[1], a);
    a = mfma16(rt.t[t], qt, a);
    *(u32x2_t*)(scr + fr * WSCRB + 32 * t + 8 * g) = u32x2_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the wave's own LDS traffic is in order; this is the compiler's fence)
  const unsigned short* row = (const unsigned short*)(scr + fr * WSCRB);
  unsigned v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = sh.col(8 * g + j);
    v[j] = c >= 0 ? (unsigned)row[max(c, 0)] : 0u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const u32x4_t u = u32x4_t{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
  return __builtin_bit_cast(bf16x8_t, u);
}

// ================================================================================ forward
__device__ __forceinline__ void fwd_pair(const QPair& q, const char* Ks, const char* Vs, const char* Es, int q0, int L, const QList& ql, int lane,
                                         bf16_raw* __restrict__ O, int ld_o, float* __restrict__ LSE, const int32_t* __restrict__ omap, bool o_pad) {
  const int fr = lane & 15, g = lane >> 4;
  const int esw = e_swz(fr);
  f32x4_t S[2][WNT];
#pragma unroll
  for (int nt = 0; nt < WNT; ++nt) {
    const char* krow = Ks + (nt * 16 + fr) * WROWB;
    const bf16x8_t k0 = *(const bf16x8_t*)(krow + g * 16);
    const bf16x8_t k1 = *(const bf16x8_t*)(krow + 64 + g * 16);
    const s16x4_t kt = *(const s16x4_t*)(krow + 128 + g * 8);
    const bf16x8_t ef = *(const bf16x8_t*)(Es + (nt * 16 + fr) * 64 + ((g ^ esw) << 4));
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
      a = mfma32(k0, q.f[mi][0], a);
      a = mfma32(k1, q.f[mi][1], a);
      a = mfma16(kt, q.t[mi], a);
      S[mi][nt] = mfma32(ef, q.rel[mi], a);
    }
    KEEP_ALIVE4(k0, k1, kt, ef);
  }
  __builtin_amdgcn_sched_barrier(0);
  // keys >= L exist only in the last tile: rows 4g + r of it
  const int kvalid = L - (WNT - 1) * 16 - 4 * g;
  float mrow[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r >= kvalid) S[mi][WNT - 1][r] = -INFINITY;
    float mx = fmaxf(fmaxf(S[mi][0][0], S[mi][0][1]), fmaxf(S[mi][0][2], S[mi][0][3]));
#pragma unroll
    for (int nt = 1; nt < WNT; ++nt) mx = fmaxf(mx, fmaxf(fmaxf(S[mi][nt][0], S[mi][nt][1]), fmaxf(S[mi][nt][2], S[mi][nt][3])));
    mrow[mi] = gmax4(mx);
#pragma unroll
    for (int nt = 0; nt < WNT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[mi][nt][r] = wexp2(S[mi][nt][r] - mrow[mi]);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4_t Oa[2][5], Ls[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    Ls[mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) Oa[mi][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const s16x8_t ones8s = s16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones8 = __builtin_bit_cast(bf16x8_t, ones8s);
  const s16x4_t ones4 = s16x4_t{0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
  for (int s2 = 0; s2 < WNT / 2; ++s2) {
    const bf16x8_t p0 = wpack(S[0][2 * s2], S[0][2 * s2 + 1]), p1 = wpack(S[1][2 * s2], S[1][2 * s2 + 1]);
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const bf16x8_t vf = tr8(Vs, s2 * 32, dt * 16, lane);
      Oa[0][dt] = mfma32(vf, p0, Oa[0][dt]);
      Oa[1][dt] = mfma32(vf, p1, Oa[1][dt]);
    }
    Ls[0] = mfma32(ones8, p0, Ls[0]);
    Ls[1] = mfma32(ones8, p1, Ls[1]);
  }
  {
    const s16x4_t p0 = wpack4(S[0][WNT - 1]), p1 = wpack4(S[1][WNT - 1]);
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const s16x4_t vt = tr4(Vs, (WNT - 1) * 16, dt * 16, lane);
      Oa[0][dt] = mfma16(vt, p0, Oa[0][dt]);
      Oa[1][dt] = mfma16(vt, p1, Oa[1][dt]);
    }
    Ls[0] = mfma16(ones4, p0, Ls[0]);
    Ls[1] = mfma16(ones4, p1, Ls[1]);
  }
  // lane holds O^T[d = dt*16 + 4g + r][q = fr]; every row of Ls is the softmax denominator of query fr
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    if (q0 + mi * 16 + fr >= ql.nq) continue;
    const int qi = ql.pos(q0 + mi * 16 + fr);
    const float l = Ls[mi][0];
    const float inv = __builtin_amdgcn_rcpf(l);
    bf16_raw* orow = O + (int64_t)(omap ? omap[qi] : qi) * ld_o;
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const f32x4_t o = Oa[mi][dt] * inv;
      *(u32x2_t*)(orow + dt * 16 + g * 4) = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    }
    if (o_pad) *(u32x2_t*)(orow + 80 + g * 4) = u32x2_t{0u, 0u};  // the 16 pad columns of a 96-wide head slot
    if (LSE && g == 0) LSE[qi] = (mrow[mi] + log2f(l)) * 0.6931471805599453f;
  }
}

template <bool TABLE>
__global__ __launch_bounds__(WTHR, 2) void win_attn_fwd_kernel(const grove_flash_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + WIMGB;
  char* Es = smem + 2 * WIMGB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int pidx = problem_of_block(blockIdx.x, gridDim.x);
  const int b = pidx / p.H, h = pidx - b * p.H;
  const int L = p.Lq;
  const bf16_raw* Q = (const bf16_raw*)p.q + (int64_t)b * p.sq + h * p.hs;
  const bf16_raw* K = (const bf16_raw*)p.k + (int64_t)b * p.sk + h * p.hs;
  const bf16_raw* V = (const bf16_raw*)p.v + (int64_t)b * p.sv + h * p.hs;
  // o in token order (o_map: row of (b, position), head h at column h * o_hs) or in the layout of q
  const int ohs = p.o_map && p.o_hs ? p.o_hs : p.hs;
  const int32_t* omap = p.o_map ? p.o_map + (int64_t)b * L : nullptr;
  bf16_raw* O = (bf16_raw*)p.o + (p.o_map ? (int64_t)0 : (int64_t)b * p.so) + h * ohs;
  float* LSE = p.lse ? p.lse + (int64_t)(b * p.H + h) * L : nullptr;
  const float sc = p.alpha * 1.4426950408889634f;
  const QList ql = make_qlist(p, b);
  const bf16_raw* PK = p.pad_k ? (const bf16_raw*)p.pad_k + h * p.hs : nullptr;
  const bf16_raw* PV = p.pad_v ? (const bf16_raw*)p.pad_v + h * p.hs : nullptr;
  if constexpr (TABLE) {
    // a wave owns at most two pairs of query tiles (nq <= 208): both pairs' Q fragments now, their bias operands from the table
    // through the wave's scratch — which lives where the indicator image goes afterwards. The table and Q loads are issued BEFORE
    // the image DMAs: vmcnt retires in order, so the products below start when these have landed, under the DMAs' flight
    QPair qs[2];
    RelTable rt;
    load_rel_table(rt, (const bf16_raw*)p.rel_table, fr, g);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
      if (wave * 32 + 128 * pi < ql.nq) load_qpair<2>(qs[pi], Q, p.ld_q, nullptr, wave * 32 + 128 * pi, ql, sc, fr, g);
    asm volatile("" ::: "memory");
    dma_image_k(Ks, K, p.ld_k, L, PK, ql, wave, lane);
    dma_image_k(Vs, V, p.ld_v, L, PV, ql, wave, lane);
    bf16_raw* RO = p.rel ? (bf16_raw*)p.rel + (int64_t)(b * p.H + h) * L * 32 : nullptr;
    char* scr = Es + wave * (16 * WSCRB);
    const int nkh = L / p.rel_kw;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int q0 = wave * 32 + 128 * pi;
      if (q0 < ql.nq) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int qc = q0 + mi * 16 + fr;
          const int qi = ql.pos(min(qc, ql.nq - 1));
          qs[pi].rel[mi] = rel_from_table(rt, qs[pi].f[mi], qs[pi].t[mi], scr, make_shift(qi, nkh, p.rel_kh, ql), fr, g);
          if (RO && qc < ql.nq) *(bf16x8_t*)(RO + (int64_t)qi * 32 + g * 8) = qs[pi].rel[mi];  // kept for the backward
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has read its scratch back: the indicator image may overwrite it
    asm volatile("" ::: "memory");
    build_e(Es, L, p.rel_kw, p.rel_kh, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave * 32 < ql.nq) fwd_pair(qs[0], Ks, Vs, Es, wave * 32, L, ql, lane, O, p.ld_o, LSE, omap, ohs >= 96);
    if (wave * 32 + 128 < ql.nq) fwd_pair(qs[1], Ks, Vs, Es, wave * 32 + 128, L, ql, lane, O, p.ld_o, LSE, omap, ohs >= 96);
  } else {
    dma_image_k(Ks, K, p.ld_k, L, PK, ql, wave, lane);
    dma_image_k(Vs, V, p.ld_v, L, PV, ql, wave, lane);
    const bf16_raw* REL = (const bf16_raw*)p.rel + (int64_t)(b * p.H + h) * L * 32;
    build_e(Es, L, p.rel_kw, p.rel_kh, tid);
    QPair q;
    load_qpair(q, Q, p.ld_q, REL, wave * 32, ql, sc, fr, g);  // in flight together with the DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma nounroll
    for (int q0 = wave * 32; q0 < ql.nq; q0 += 128) {
      QPair qn;
      const bool more = q0 + 128 < ql.nq;
      if (more) load_qpair(qn, Q, p.ld_q, REL, q0 + 128, ql, sc, fr, g);  // lands under this pair's MFMAs
      fwd_pair(q, Ks, Vs, Es, q0, L, ql, lane, O, p.ld_o, LSE, omap, ohs >= 96);
      if (more) q = qn;
    }
  }
}


// ================================================================================ backward
// One kernel, two phases over the same 80 KB of LDS (two blocks per CU):
//   A (waves own KEY tiles, lane = key): Q, dO and rel' of the (window, head) are in LDS, the wave's K / V fragments in
//     registers; S[q][key] and dP[q][key] per 32 queries, P = exp2(S - lse), dS = P (dP - delta) alpha, and the two products that
//     sum over queries, dV^T += dO^T P and dK^T += Q^T dS, take P / dS straight from the accumulators (flash_attn.hip's
//     accumulator-as-operand chaining) with dO^T / Q^T as transposed LDS reads;
//   B (waves own QUERY tile pairs, lane = query): K, V and the indicator image E replace Q, dO, rel' in LDS (second LDS-DMA; the
//     bytes come back from L2), Q / dO / rel' fragments in registers; S^T, dP^T per 32 keys, dQ^T += K^T dS^T and
//     d rel'^T += E^T dS^T.
// delta = rowsum(dO o O) is computed in the prologue (O straight from global, dO from its LDS image): no separate launch.
// HBM sees every operand once (q, k, v, dO, O, rel', lse in; dq, dk, dv, d rel' out).
__device__ __forceinline__ bf16x8_t ld_rows(const char* img, int row, int byte_off) { return *(const bf16x8_t*)(img + row * WROWB + byte_off); }

__device__ __forceinline__ s16x4_t tr4e(const char* Es, int r0, int c0, int lane) {  // transposed read of the swizzled 64-byte-row image
  const int fr = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * g + (fr >> 2);
  const int cb = (c0 + 4 * (fr & 3)) * 2;
  const char* a = Es + row * 64 + ((((cb >> 4) ^ e_swz(row)) << 4) | (cb & 15));
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a));
}
__device__ __forceinline__ bf16x8_t tr8e(const char* Es, int r0, int c0, int lane) {
  const s16x4_t lo = tr4e(Es, r0, c0, lane), hi = tr4e(Es, r0 + 16, c0, lane);
  const s16x8_t v = s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

struct KVFrag {      // B operands of one 16-key tile (phase A): K pre-scaled into the exp2 domain, V, the indicator columns
  bf16x8_t k[2];
  s16x4_t kt;
  bf16x8_t v[2];
  s16x4_t vt;
  bf16x8_t e;
};

__device__ __forceinline__ void store_t(bf16_raw* __restrict__ dst, int ld, int row, const f32x4_t (&acc)[5], int g) {
  bf16_raw* r = dst + (int64_t)row * ld;
#pragma unroll
  for (int dt = 0; dt < 5; ++dt) *(u32x2_t*)(r + dt * 16 + g * 4) = u32x2_t{pack2bf(acc[dt][0], acc[dt][1]), pack2bf(acc[dt][2], acc[dt][3])};
  *(u32x2_t*)(r + 80 + g * 4) = u32x2_t{0u, 0u};
}

// token-order gradient rows (grove_flash_attn_params.g_tok): compact heads, no pad columns
__device__ __forceinline__ void store_tok(bf16_raw* __restrict__ dst, int ld, int64_t row, const f32x4_t (&acc)[5], int g) {
  bf16_raw* r = dst + row * ld;
#pragma unroll
  for (int dt = 0; dt < 5; ++dt) *(u32x2_t*)(r + dt * 16 + g * 4) = u32x2_t{pack2bf(acc[dt][0], acc[dt][1]), pack2bf(acc[dt][2], acc[dt][3])};
}

// phase A for NTILE (1 or 2) key tiles kt0, kt0 + 4 of this wave
template <int NTILE>
__device__ __forceinline__ void load_kv(KVFrag (&kv)[NTILE], int kt0, const bf16_raw* __restrict__ K, int ld_k, const bf16_raw* __restrict__ V,
                                        int ld_v, int L, float sc, int kw, int KH, int lane, const bf16_raw* __restrict__ PK,
                                        const bf16_raw* __restrict__ PV, const QList& ql) {
  const int fr = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nj = 0; nj < NTILE; ++nj) {
    const int key = (kt0 + 4 * nj) * 16 + fr;
    const int kc = min(key, L - 1);
    const bool padded = PK && !ql.real(kc);  // (k / v of a padded position: the bias row)
    const bf16_raw* kr = padded ? PK : K + (int64_t)kc * ld_k;
    const bf16_raw* vr = padded ? PV : V + (int64_t)kc * ld_v;
    kv[nj].k[0] = wscale(*(const bf16x8_t*)(kr + g * 8), sc);
    kv[nj].k[1] = wscale(*(const bf16x8_t*)(kr + 32 + g * 8), sc);
    kv[nj].kt = wscale4(*(const u32x2_t*)(kr + 64 + g * 4), sc);
    kv[nj].v[0] = *(const bf16x8_t*)(vr + g * 8);
    kv[nj].v[1] = *(const bf16x8_t*)(vr + 32 + g * 8);
    kv[nj].vt = __builtin_bit_cast(s16x4_t, *(const u32x2_t*)(vr + 64 + g * 4));
    const unsigned kb = key < L ? (unsigned)((key / kw) | ((key % kw) << 8)) : 0xFFFFu;
    kv[nj].e = e_chunk(kb, g * 8, KH);
  }
}

template <int NTILE>
__device__ __forceinline__ void bwd_keys(const KVFrag (&kv)[NTILE], int kt0,
                                         const char* Qs, const char* dOs, const char* Rs, const float* lse_s, const float* del_s, int L, int nq,
                                         float alpha, int lane, bf16_raw* __restrict__ DK, int ld_dk, bf16_raw* __restrict__ DV, int ld_dv,
                                         const QList& ql, bool drop_padded, const int32_t* __restrict__ gmap) {
  const int fr = lane & 15, g = lane >> 4;
  const int esw = e_swz(fr);
  constexpr int ntile = NTILE;
  f32x4_t dk[NTILE][5], dv[NTILE][5];
#pragma unroll
  for (int nj = 0; nj < NTILE; ++nj)
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) { dk[nj][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[nj][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const f32x4_t z4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // the images hold the nq valid queries in rows 0 .. nq - 1 (rows beyond: copies of the last one with lse = +inf, i.e. P = 0):
  // nqt 16-row tiles = nqt / 2 steps of 32 and, if nqt is odd, one 16-deep step
  const int nqt = (nq + 15) >> 4;
#pragma nounroll
  for (int qs = 0; qs < (nqt >> 1); ++qs) {  // 32 queries per step
    f32x4_t sacc[2][NTILE], pacc[2][NTILE];
#pragma unroll
    for (int qi_ = 0; qi_ < 2; ++qi_) {
      const int row = (2 * qs + qi_) * 16 + fr;
      const bf16x8_t qa0 = ld_rows(Qs, row, g * 16), qa1 = ld_rows(Qs, row, 64 + g * 16);
      const s16x4_t qat = *(const s16x4_t*)(Qs + row * WROWB + 128 + g * 8);
      const bf16x8_t da0 = ld_rows(dOs, row, g * 16), da1 = ld_rows(dOs, row, 64 + g * 16);
      const s16x4_t dat = *(const s16x4_t*)(dOs + row * WROWB + 128 + g * 8);
      const bf16x8_t ra = *(const bf16x8_t*)(Rs + row * 64 + ((g ^ esw) << 4));
#pragma unroll
      for (int nj = 0; nj < NTILE; ++nj) {
        if (nj < ntile) {
          f32x4_t a = mfma32(qa0, kv[nj].k[0], z4);
          a = mfma32(qa1, kv[nj].k[1], a);
          a = mfma16(qat, kv[nj].kt, a);
          sacc[qi_][nj] = mfma32(ra, kv[nj].e, a);
          f32x4_t c = mfma32(da0, kv[nj].v[0], z4);
          c = mfma32(da1, kv[nj].v[1], c);
          pacc[qi_][nj] = mfma16(dat, kv[nj].vt, c);
        }
      }
      KEEP_ALIVE4(qa0, qa1, qat, ra);
      KEEP_ALIVE3(da0, da1, dat);
    }
    f32x4_t lse4[2], del4[2];
#pragma unroll
    for (int qi_ = 0; qi_ < 2; ++qi_) {
      lse4[qi_] = *(const f32x4_t*)(lse_s + (2 * qs + qi_) * 16 + g * 4);
      del4[qi_] = *(const f32x4_t*)(del_s + (2 * qs + qi_) * 16 + g * 4);
    }
    bf16x8_t pfr[NTILE], dsfr[NTILE];
#pragma unroll
    for (int nj = 0; nj < NTILE; ++nj) {
      if (nj < ntile) {
        f32x4_t pp[2], dd[2];
#pragma unroll
        for (int qi_ = 0; qi_ < 2; ++qi_)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = wexp2(sacc[qi_][nj][r] - lse4[qi_][r]);
            pp[qi_][r] = pr;
            dd[qi_][r] = pr * (pacc[qi_][nj][r] - del4[qi_][r]) * alpha;
          }
        pfr[nj] = wpack(pp[0], pp[1]);
        dsfr[nj] = wpack(dd[0], dd[1]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const bf16x8_t dob = tr8(dOs, qs * 32, dt * 16, lane);
      const bf16x8_t qb = tr8(Qs, qs * 32, dt * 16, lane);
#pragma unroll
      for (int nj = 0; nj < NTILE; ++nj) {
        if (nj < ntile) {
          dv[nj][dt] = mfma32(dob, pfr[nj], dv[nj][dt]);
          dk[nj][dt] = mfma32(qb, dsfr[nj], dk[nj][dt]);
        }
      }
    }
  }
  if (nqt & 1) {  // the odd last query tile (rows >= nq carry no probability)
    const int lt = nqt - 1;
    const int row = lt * 16 + fr;
    const bf16x8_t qa0 = ld_rows(Qs, row, g * 16), qa1 = ld_rows(Qs, row, 64 + g * 16);
    const s16x4_t qat = *(const s16x4_t*)(Qs + row * WROWB + 128 + g * 8);
    const bf16x8_t da0 = ld_rows(dOs, row, g * 16), da1 = ld_rows(dOs, row, 64 + g * 16);
    const s16x4_t dat = *(const s16x4_t*)(dOs + row * WROWB + 128 + g * 8);
    const bf16x8_t ra = *(const bf16x8_t*)(Rs + row * 64 + ((g ^ esw) << 4));
    const f32x4_t lse4 = *(const f32x4_t*)(lse_s + lt * 16 + g * 4), del4 = *(const f32x4_t*)(del_s + lt * 16 + g * 4);
    const int qvalid = nq - lt * 16 - 4 * g;
    s16x4_t p4[NTILE], d4[NTILE];
#pragma unroll
    for (int nj = 0; nj < NTILE; ++nj) {
      if (nj < ntile) {
        f32x4_t a = mfma32(qa0, kv[nj].k[0], z4);
        a = mfma32(qa1, kv[nj].k[1], a);
        a = mfma16(qat, kv[nj].kt, a);
        a = mfma32(ra, kv[nj].e, a);
        f32x4_t c = mfma32(da0, kv[nj].v[0], z4);
        c = mfma32(da1, kv[nj].v[1], c);
        c = mfma16(dat, kv[nj].vt, c);
        f32x4_t pp, dd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = r < qvalid ? wexp2(a[r] - lse4[r]) : 0.f;
          pp[r] = pr;
          dd[r] = pr * (c[r] - del4[r]) * alpha;
        }
        p4[nj] = wpack4(pp);
        d4[nj] = wpack4(dd);
      }
    }
    KEEP_ALIVE4(qa0, qa1, qat, ra);
    KEEP_ALIVE3(da0, da1, dat);
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const s16x4_t dob = tr4(dOs, lt * 16, dt * 16, lane);
      const s16x4_t qb = tr4(Qs, lt * 16, dt * 16, lane);
#pragma unroll
      for (int nj = 0; nj < NTILE; ++nj) {
        if (nj < ntile) {
          dv[nj][dt] = mfma16(dob, p4[nj], dv[nj][dt]);
          dk[nj][dt] = mfma16(qb, d4[nj], dk[nj][dt]);
        }
      }
    }
  }
  // lane holds dK^T / dV^T[d = dt*16 + 4g + r][key = fr]
#pragma unroll
  for (int nj = 0; nj < NTILE; ++nj) {
    const int key = (kt0 + 4 * nj) * 16 + fr;
    if (nj < ntile && key < L && !(drop_padded && !ql.real(key))) {  // (nobody reads dk / dv of a padded position)
      if (gmap) {  // token order: the row of this (real) position
        const int64_t row = gmap[key];
        store_tok(DK, ld_dk, row, dk[nj], g);
        store_tok(DV, ld_dv, row, dv[nj], g);
      } else {
        store_t(DK, ld_dk, key, dk[nj], g);
        store_t(DV, ld_dv, key, dv[nj], g);
      }
    }
  }
}

struct QDPair {         // B operands of one pair of 16-query tiles (phase B)
  QPair q;              // Q and rel', pre-scaled into the exp2 domain
  bf16x8_t d[2][2];     // dO
  s16x4_t dt[2];
  float lse2[2], del[2];
};

// drp (or NULL): the pair's d rel' rows stay in registers as bf16 — [mi][bt] = bins 16 bt + 4g .. + 3 of query fr — for rel_tail
__device__ __forceinline__ void bwd_queries(const QDPair& x, const char* Ks, const char* Vs, const char* Es, int q0, int L, const QList& ql, float alpha,
                                            int lane, bf16_raw* __restrict__ DQ, int ld_dq, bf16_raw* __restrict__ DR, const int32_t* __restrict__ gmap,
                                            u32x2_t (*drp)[2] = nullptr) {
  const int fr = lane & 15, g = lane >> 4;
  const int esw = e_swz(fr);
  const f32x4_t z4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t dq[2][5], drl[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    drl[mi][0] = z4;
    drl[mi][1] = z4;
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) dq[mi][dt] = z4;
  }
#pragma nounroll
  for (int s2 = 0; s2 < WNT / 2; ++s2) {  // 32 keys per step
    f32x4_t s[2][2], dp[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int row = (2 * s2 + ni) * 16 + fr;
      const bf16x8_t k0 = ld_rows(Ks, row, g * 16), k1 = ld_rows(Ks, row, 64 + g * 16);
      const s16x4_t kt = *(const s16x4_t*)(Ks + row * WROWB + 128 + g * 8);
      const bf16x8_t v0 = ld_rows(Vs, row, g * 16), v1 = ld_rows(Vs, row, 64 + g * 16);
      const s16x4_t vt = *(const s16x4_t*)(Vs + row * WROWB + 128 + g * 8);
      const bf16x8_t ef = *(const bf16x8_t*)(Es + row * 64 + ((g ^ esw) << 4));
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f32x4_t a = mfma32(k0, x.q.f[mi][0], z4);
        a = mfma32(k1, x.q.f[mi][1], a);
        a = mfma16(kt, x.q.t[mi], a);
        s[mi][ni] = mfma32(ef, x.q.rel[mi], a);
        f32x4_t c = mfma32(v0, x.d[mi][0], z4);
        c = mfma32(v1, x.d[mi][1], c);
        dp[mi][ni] = mfma16(vt, x.dt[mi], c);
      }
      KEEP_ALIVE4(k0, k1, kt, ef);
      KEEP_ALIVE3(v0, v1, vt);
    }
    bf16x8_t dsf[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = wexp2(s[mi][ni][r] - x.lse2[mi]);
          s[mi][ni][r] = pr * (dp[mi][ni][r] - x.del[mi]) * alpha;
        }
      dsf[mi] = wpack(s[mi][0], s[mi][1]);
    }
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const bf16x8_t ktr = tr8(Ks, s2 * 32, dt * 16, lane);
      dq[0][dt] = mfma32(ktr, dsf[0], dq[0][dt]);
      dq[1][dt] = mfma32(ktr, dsf[1], dq[1][dt]);
    }
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const bf16x8_t et = tr8e(Es, s2 * 32, bt * 16, lane);
      drl[0][bt] = mfma32(et, dsf[0], drl[0][bt]);
      drl[1][bt] = mfma32(et, dsf[1], drl[1][bt]);
    }
  }
  {  // the 13th key tile (keys >= L carry no probability)
    const int row = (WNT - 1) * 16 + fr;
    const bf16x8_t k0 = ld_rows(Ks, row, g * 16), k1 = ld_rows(Ks, row, 64 + g * 16);
    const s16x4_t kt = *(const s16x4_t*)(Ks + row * WROWB + 128 + g * 8);
    const bf16x8_t v0 = ld_rows(Vs, row, g * 16), v1 = ld_rows(Vs, row, 64 + g * 16);
    const s16x4_t vt = *(const s16x4_t*)(Vs + row * WROWB + 128 + g * 8);
    const bf16x8_t ef = *(const bf16x8_t*)(Es + row * 64 + ((g ^ esw) << 4));
    const int kvalid = L - (WNT - 1) * 16 - 4 * g;
    s16x4_t d4[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f32x4_t a = mfma32(k0, x.q.f[mi][0], z4);
      a = mfma32(k1, x.q.f[mi][1], a);
      a = mfma16(kt, x.q.t[mi], a);
      a = mfma32(ef, x.q.rel[mi], a);
      f32x4_t c = mfma32(v0, x.d[mi][0], z4);
      c = mfma32(v1, x.d[mi][1], c);
      c = mfma16(vt, x.dt[mi], c);
      f32x4_t dd;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = r < kvalid ? wexp2(a[r] - x.lse2[mi]) : 0.f;
        dd[r] = pr * (c[r] - x.del[mi]) * alpha;
      }
      d4[mi] = wpack4(dd);
    }
    KEEP_ALIVE4(k0, k1, kt, ef);
    KEEP_ALIVE3(v0, v1, vt);
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const s16x4_t ktr = tr4(Ks, (WNT - 1) * 16, dt * 16, lane);
      dq[0][dt] = mfma16(ktr, d4[0], dq[0][dt]);
      dq[1][dt] = mfma16(ktr, d4[1], dq[1][dt]);
    }
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const s16x4_t et = tr4e(Es, (WNT - 1) * 16, bt * 16, lane);
      drl[0][bt] = mfma16(et, d4[0], drl[0][bt]);
      drl[1][bt] = mfma16(et, d4[1], drl[1][bt]);
    }
  }
  // lane holds dQ^T[d][q = fr] and d rel'^T[bin = bt*16 + 4g + r][q = fr]
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    if (q0 + mi * 16 + fr >= ql.nq) continue;
    const int qi = ql.pos(q0 + mi * 16 + fr);
    if (gmap) store_tok(DQ, ld_dq, gmap[qi], dq[mi], g);
    else store_t(DQ, ld_dq, qi, dq[mi], g);
    if (drp) {
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) drp[mi][bt] = u32x2_t{pack2bf(drl[mi][bt][0], drl[mi][bt][1]), pack2bf(drl[mi][bt][2], drl[mi][bt][3])};
    }
    if (DR) {
      bf16_raw* r = DR + (int64_t)qi * 32;
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
        *(u32x2_t*)(r + bt * 16 + g * 4) = u32x2_t{pack2bf(drl[mi][bt][0], drl[mi][bt][1]), pack2bf(drl[mi][bt][2], drl[mi][bt][3])};
    }
  }
}

// The rel-pos term of dq with grove_flash_attn_params.rel_table: dq[q][d] += sum_bin d rel'[q][bin] T[row of (q, bin)][d]. The wave
// scatters the d rel' rows of 16 queries to their table rows in a private scratch dG[16 queries][64 rows] (the inverse of
// rel_from_table's shift), and dq^T += T^T dG^T is ten MFMAs whose accumulators land on the lanes that stored dq[q][d] (a lane adds
// to its own stores: bf16 + bf16, the two roundings the grove_rel_bias_bwd stream had). Runs after the last pair of every wave:
// the scratch takes the place of the K image.
struct RelTableT { bf16x8_t f[5][2]; };  // A operands: T^T rows d = 16 dt + fr, table rows 32 ks + 8g ..
__device__ __forceinline__ void load_rel_table_t(RelTableT& tt, const bf16_raw* __restrict__ TT, int fr, int g) {
#pragma unroll
  for (int dt = 0; dt < 5; ++dt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) tt.f[dt][ks] = *(const bf16x8_t*)(TT + (16 * dt + fr) * 64 + 32 * ks + 8 * g);
}
__device__ __forceinline__ void rel_tail(const RelTableT& tt, const u32x2_t (&drp)[2], char* scr, const RelShift& sh, bf16_raw* __restrict__ dq_row, bool live,
                                         int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
  *(u32x4_t*)(scr + lane * 16) = z;  // 16 rows x 144 bytes = 144 chunks of 16
  *(u32x4_t*)(scr + 1024 + lane * 16) = z;
  if (lane < 16) *(u32x4_t*)(scr + 2048 + lane * 16) = z;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned short* row = (unsigned short*)(scr + fr * WSCRB);
#pragma unroll
  for (int bt = 0; bt < 2; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = sh.col(16 * bt + 4 * g + r);
      const unsigned w = r < 2 ? drp[bt].x : drp[bt].y;
      if (c >= 0) row[c] = (unsigned short)((r & 1) ? (w >> 16) : (w & 0xffffu));
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const bf16x8_t d0 = *(const bf16x8_t*)(scr + fr * WSCRB + g * 16), d1 = *(const bf16x8_t*)(scr + fr * WSCRB + 64 + g * 16);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const f32x4_t z4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dt = 0; dt < 5; ++dt) {
    f32x4_t a = mfma32(tt.f[dt][0], d0, z4);
    a = mfma32(tt.f[dt][1], d1, a);
    if (live) {
      u32x2_t* at = (u32x2_t*)(dq_row + dt * 16 + g * 4);
      const u32x2_t o = *at;
      *at = u32x2_t{pack2bf(a[0] + bf_lo(o.x), a[1] + bf_hi(o.x)), pack2bf(a[2] + bf_lo(o.y), a[3] + bf_hi(o.y))};
    }
  }
}

template <bool TABLE>
__global__ __launch_bounds__(WTHR, 2) void win_attn_bwd_kernel(const grove_flash_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;                  // Q, then K
  char* Ys = smem + WIMGB;          // dO, then V
  char* Rs = smem + 2 * WIMGB;      // rel' (pre-scaled), then the indicator image E
  float* lse_s = (float*)(Rs + WEB);
  float* del_s = lse_s + WNT * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int pidx = problem_of_block(blockIdx.x, gridDim.x);
  const int b = pidx / p.H, h = pidx - b * p.H;
  const int L = p.Lq;
  const int64_t bh = (int64_t)(b * p.H + h) * L;
  const bf16_raw* Q = (const bf16_raw*)p.q + (int64_t)b * p.sq + h * p.hs;
  const bf16_raw* K = (const bf16_raw*)p.k + (int64_t)b * p.sk + h * p.hs;
  const bf16_raw* V = (const bf16_raw*)p.v + (int64_t)b * p.sv + h * p.hs;
  const int ohs = p.o_map && p.o_hs ? p.o_hs : p.hs;
  const int32_t* omap = p.o_map ? p.o_map + (int64_t)b * L : nullptr;
  const bf16_raw* dO = (const bf16_raw*)p.d_o + (p.o_map ? (int64_t)0 : (int64_t)b * p.sdo) + h * ohs;
  const bf16_raw* Og = (const bf16_raw*)p.o + (p.o_map ? (int64_t)0 : (int64_t)b * p.so) + h * ohs;
  const bf16_raw* REL = (const bf16_raw*)p.rel + bh * 32;
  const float sc = p.alpha * 1.4426950408889634f;
  // ---- prologue: Q, dO -> LDS (DMA); rel' (scaled) -> LDS; lse; delta
  const QList ql = make_qlist(p, b);
  const bf16_raw* PK = p.pad_k ? (const bf16_raw*)p.pad_k + h * p.hs : nullptr;
  const bf16_raw* PV = p.pad_v ? (const bf16_raw*)p.pad_v + h * p.hs : nullptr;
  // token-order d_o / o (o_map): the indices first — the d_o image's rows and this thread's o row — so that they arrive under the
  // Q image's DMA and the rel' staging instead of in front of the d_o DMA
  ImageRows do_rows;
  image_q_rows(do_rows, ql, wave, lane, omap);
  const int myrow = ql.pos(min(tid, ql.nq - 1));
  const int64_t orow_i = omap ? omap[myrow] : myrow;
  dma_image_q(Xs, Q, p.ld_q, ql, wave, lane);
  for (int c = tid; c < WNT * 16 * 4; c += WTHR) {
    const int row = c >> 2, ch = c & 3;
    u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
    if (row < ql.nq) v = *(const u32x4_t*)(REL + (int64_t)ql.pos(row) * 32 + ch * 8);
    // (rel_table: the forward left the score-domain operand itself)
    *(bf16x8_t*)(Rs + row * 64 + ((ch ^ e_swz(row)) << 4)) = TABLE ? __builtin_bit_cast(bf16x8_t, v) : wscale(__builtin_bit_cast(bf16x8_t, v), sc);
  }
  dma_image_rows(Ys, dO, p.ld_do, do_rows, wave, lane);
  u32x4_t orow[10];
  if (tid < WNT * 16) {
#pragma unroll
    for (int c = 0; c < 10; ++c) orow[c] = *(const u32x4_t*)(Og + orow_i * p.ld_o + c * 8);
    lse_s[tid] = tid < ql.nq ? p.lse[bh + myrow] * 1.4426950408889634f : INFINITY;  // (+inf: a row past the last query has P = exp2(s - inf) = 0)
  }
  KVFrag kv0[2];
  load_kv<2>(kv0, wave, K, p.ld_k, V, p.ld_v, L, sc, p.rel_kw, p.rel_kh, lane, PK, PV, ql);  // pass 0's fragments: in flight with the DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < WNT * 16) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const u32x4_t d = *(const u32x4_t*)(Ys + tid * WROWB + c * 16);
      const u32x4_t o = orow[c];
      acc += bf_lo(o.x) * bf_lo(d.x) + bf_hi(o.x) * bf_hi(d.x) + bf_lo(o.y) * bf_lo(d.y) + bf_hi(o.y) * bf_hi(d.y) +
             bf_lo(o.z) * bf_lo(d.z) + bf_hi(o.z) * bf_hi(d.z) + bf_lo(o.w) * bf_lo(d.w) + bf_hi(o.w) * bf_hi(d.w);
    }
    del_s[tid] = acc;
  }
  __syncthreads();
  // ---- phase A: dK, dV of this wave's key tiles {w, w+4} and {w+8, w+12}
  // g_tok: dq / dk / dv in token order through o_map (compact heads of o_hs columns), like o and d_o
  const int32_t* gmap = (p.g_tok && omap) ? omap : nullptr;
  bf16_raw* DK = (bf16_raw*)p.dk + (gmap ? (int64_t)h * ohs : (int64_t)b * p.sdk + h * p.hs);
  bf16_raw* DV = (bf16_raw*)p.dv + (gmap ? (int64_t)h * ohs : (int64_t)b * p.sdv + h * p.hs);
  bwd_keys<2>(kv0, wave, Xs, Ys, Rs, lse_s, del_s, L, ql.nq, p.alpha, lane, DK, p.ld_dk, DV, p.ld_dv, ql, PK != nullptr, gmap);
  if (wave == 0) {  // key tiles {8, 12}
    KVFrag kv1[2];
    load_kv<2>(kv1, 8, K, p.ld_k, V, p.ld_v, L, sc, p.rel_kw, p.rel_kh, lane, PK, PV, ql);
    bwd_keys<2>(kv1, 8, Xs, Ys, Rs, lse_s, del_s, L, ql.nq, p.alpha, lane, DK, p.ld_dk, DV, p.ld_dv, ql, PK != nullptr, gmap);
  } else {          // key tile 8 + wave
    KVFrag kv1[1];
    load_kv<1>(kv1, 8 + wave, K, p.ld_k, V, p.ld_v, L, sc, p.rel_kw, p.rel_kh, lane, PK, PV, ql);
    bwd_keys<1>(kv1, 8 + wave, Xs, Ys, Rs, lse_s, del_s, L, ql.nq, p.alpha, lane, DK, p.ld_dk, DV, p.ld_dv, ql, PK != nullptr, gmap);
  }
  __syncthreads();
  // ---- phase B: K, V, E replace Q, dO, rel' in LDS; dQ and d rel' of this wave's query-tile pairs
  int tok[2];  // d_o rows of this lane's queries in the first pass of the loop below, fetched ahead of the DMA
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int qp = ql.pos(min(wave * 32 + mi * 16 + fr, ql.nq - 1));
    tok[mi] = omap ? omap[qp] : qp;
  }
  dma_image_k(Xs, K, p.ld_k, L, PK, ql, wave, lane);
  dma_image_k(Ys, V, p.ld_v, L, PV, ql, wave, lane);
  build_e(Rs, L, p.rel_kw, p.rel_kh, tid);
  bf16_raw* DQ = (bf16_raw*)p.dq + (gmap ? (int64_t)h * ohs : (int64_t)b * p.sdq + h * p.hs);
  bf16_raw* DR = p.drel ? (bf16_raw*)p.drel + bh * 32 : nullptr;
  auto load_x = [&](QDPair& x, int q0, const int* tk) {
    load_qpair<TABLE ? 1 : 0>(x.q, Q, p.ld_q, REL, q0, ql, sc, fr, g);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int qc = min(q0 + mi * 16 + fr, ql.nq - 1);  // compact index: lse_s / del_s; position: the global rows
      int trow = ql.pos(qc);
      if (tk) trow = tk[mi]; else if (omap) trow = omap[trow];
      const bf16_raw* row = dO + (int64_t)trow * p.ld_do;
      x.d[mi][0] = *(const bf16x8_t*)(row + g * 8);
      x.d[mi][1] = *(const bf16x8_t*)(row + 32 + g * 8);
      x.dt[mi] = __builtin_bit_cast(s16x4_t, *(const u32x2_t*)(row + 64 + g * 4));
      x.lse2[mi] = lse_s[qc];   // (lse_s / del_s are not overwritten by phase B's images)
      x.del[mi] = del_s[qc];
    }
  };
  QDPair x;
  load_x(x, wave * 32, tok);  // in flight with the DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (TABLE) {
    u32x2_t drp[2][2][2];  // [pair][mi][bt]
    if (wave * 32 < ql.nq) bwd_queries(x, Xs, Ys, Rs, wave * 32, L, ql, p.alpha, lane, DQ, p.ld_dq, nullptr, gmap, drp[0]);
    if (wave * 32 + 128 < ql.nq) {
      load_x(x, wave * 32 + 128, nullptr);
      bwd_queries(x, Xs, Ys, Rs, wave * 32 + 128, L, ql, p.alpha, lane, DQ, p.ld_dq, nullptr, gmap, drp[1]);
    }
    RelTableT tt;
    load_rel_table_t(tt, (const bf16_raw*)p.rel_table + 64 * 80, fr, g);
    __syncthreads();  // every wave is through with the K image (and its dq stores are ordered before the adds below)
    char* scr = Xs + wave * (16 * WSCRB);
    const int nkh = L / p.rel_kw;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int q0 = wave * 32 + 128 * pi;
      if (q0 < ql.nq) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int qc = q0 + mi * 16 + fr;
          const int qi = ql.pos(min(qc, ql.nq - 1));
          bf16_raw* dq_row = DQ + (gmap ? (int64_t)gmap[qi] : (int64_t)qi) * p.ld_dq;
          rel_tail(tt, drp[pi][mi], scr, make_shift(qi, nkh, p.rel_kh, ql), dq_row, qc < ql.nq, lane);
        }
      }
    }
  } else {
#pragma nounroll
    for (int q0 = wave * 32; q0 < ql.nq; q0 += 128) {
      bwd_queries(x, Xs, Ys, Rs, q0, L, ql, p.alpha, lane, DQ, p.ld_dq, DR, gmap);
      if (q0 + 128 < ql.nq) load_x(x, q0 + 128, nullptr);
    }
  }
}

}  // namespace

// Does this problem fit the window kernels? (192 < L <= 208 tokens, head dim 80 in a 96-wide slot, 32 rel bins, no masks)
bool grove_win_attn_applicable(const grove_flash_attn_params* p) {
  if (p->rel_table && !(p->rel_kw > 0 && p->Lq % p->rel_kw == 0 && 2 * (p->Lq / p->rel_kw) + 2 * p->rel_kw - 2 <= 64 && p->Lq / p->rel_kw <= p->rel_kh &&
                        p->rel_kh + p->rel_kw <= 32 && ((uintptr_t)p->rel_table & 15) == 0 && ((uintptr_t)p->rel & 15) == 0))
    return false;
  return p->hs == 96 && p->hs_valid == 80 && p->Lq == p->Lk && p->Lk > (WNT - 1) * 16 && p->Lk <= WNT * 16 && (p->rel || p->rel_table) && p->rel_ld == 32 &&
         !p->causal && !p->kv_len && p->ld_q % 8 == 0 && p->ld_k % 8 == 0 && p->ld_v % 8 == 0 && p->ld_o % 8 == 0;
}

int grove_win_attn_bwd_launch(const grove_flash_attn_params* p, hipStream_t s) {
  const size_t lds = 2 * WIMGB + WEB + 2 * WNT * 16 * sizeof(float);
  if (p->rel_table) {
    hipFuncSetAttribute((const void*)win_attn_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(win_attn_bwd_kernel<true>, dim3((unsigned)(p->B * p->H)), dim3(WTHR), lds, s, *p);
  } else {
    hipFuncSetAttribute((const void*)win_attn_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(win_attn_bwd_kernel<false>, dim3((unsigned)(p->B * p->H)), dim3(WTHR), lds, s, *p);
  }
  return 0;
}

int grove_win_attn_fwd_launch(const grove_flash_attn_params* p, hipStream_t s) {
  const size_t lds = 2 * WIMGB + WEB;
  if (p->rel_table) {
    hipFuncSetAttribute((const void*)win_attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(win_attn_fwd_kernel<true>, dim3((unsigned)(p->B * p->H)), dim3(WTHR), lds, s, *p);
  } else {
    hipFuncSetAttribute((const void*)win_attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(win_attn_fwd_kernel<false>, dim3((unsigned)(p->B * p->H)), dim3(WTHR), lds, s, *p);
  }
  return 0;
}
