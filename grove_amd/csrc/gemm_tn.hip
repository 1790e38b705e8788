// "TN" bf16 GEMM for weight gradients on gfx950:  C[m, n] += scale * sum_k A[k, m] * B[row_b(k, n), n']
//
// Both operands are K-major (row k = one token): A = dY [tokens, M], B = X [tokens, N] — exactly what the
// backward pass has in HBM, so dW = dY^T X needs no transposed copies. Tiles are staged as plain [64 k][128]
// bf16 images with a 288-byte row stride and consumed with ds_read_b64_tr_b16 (the transposed read validated
// in flash_attn.hip: conflict-free at that stride); both MFMA operands use the same permuted k-order.
// B rows may be gathered per tap (b_idx[tap*K + k], -1 = zero row, tap = n / (N / b_taps)): the weight gradient
// of the implicit-GEMM Conv3d/Conv2d (SAM / CLIP adapters, neck) in ONE launch, no im2col in HBM.
// C is fp32 and always accumulated; K is split over blockIdx.z (fp32 atomics) when the tile grid cannot fill
// the chip. 128 x 128 x 64 tile, 4 waves (2 x 2), register-staged prefetch of the next K tile.
#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_tn_zero_page[64];  // 256 B of zeros (gathered zero rows)

constexpr int TBM = 128, TBN = 128, TBK = 64, TNT = 256;
constexpr int TROWB = 128 * 2 + 32;           // LDS row stride (bytes)
constexpr int TTILEB = TBK * TROWB;           // one operand tile
constexpr int TLPT = (TBK * 16) / TNT;        // 16-byte chunks per thread per operand tile (= 4)

__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int row0, int c0, int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const int qq = fr >> 2, pp = fr & 3;
  const char* a0 = tile + (row0 + 4 * g + qq) * TROWB + (c0 + 4 * pp) * 2;
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 16 * TROWB));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(TNT) void gemm_tn_kernel(const grove_gemm_tn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  const int tiles_m = (p.M + TBM - 1) / TBM;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;  // consecutive blocks share the B (gathered) panel
  const int m0 = tm * TBM, n0 = tn * TBN;
  const bf16_raw* __restrict__ A = (const bf16_raw*)p.A;
  const bf16_raw* __restrict__ B = (const bf16_raw*)p.B;
  const int n_per_tap = p.N / p.b_taps;
  const int tap = n0 / n_per_tap;
  const int nb0 = n0 - tap * n_per_tap;  // column inside the gathered row
  const int32_t* bidx = p.b_idx ? p.b_idx + (int64_t)tap * p.K : nullptr;

  const int nkt = (p.K + TBK - 1) / TBK;
  const int nsplit = gridDim.z;
  const int per = (nkt + nsplit - 1) / nsplit;
  const int kt0 = blockIdx.z * per;
  const int kt1 = min(nkt, kt0 + per);
  if (kt0 >= kt1) return;

  const int st_c = tid & 15;   // chunk (8 columns) inside the 128-wide tile row
  const int st_r = tid >> 4;   // row, + 16 per pass
  const bool a_col_ok = m0 + st_c * 8 < p.M;
  const bool b_col_ok = n0 + st_c * 8 < p.N && nb0 + st_c * 8 < n_per_tap;
  u32x4_t ra[TLPT], rb[TLPT];
  // gathered-row indices are fetched ONE tile ahead of the loads that use them, so the dependent
  // index -> row load chain never sits on the critical path of a K tile
  int rowidx[TLPT];
  auto fetch_idx = [&](int kt) {
#pragma unroll
    for (int i = 0; i < TLPT; ++i) {
      const int k = kt * TBK + st_r + 16 * i;
      rowidx[i] = k < p.K ? (bidx ? bidx[k] : k) : -1;
    }
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int i = 0; i < TLPT; ++i) {
      const int k = kt * TBK + st_r + 16 * i;
      ra[i] = u32x4_t{0u, 0u, 0u, 0u};
      rb[i] = u32x4_t{0u, 0u, 0u, 0u};
      if (k < p.K && a_col_ok) ra[i] = *(const u32x4_t*)(A + (int64_t)k * p.lda + m0 + st_c * 8);
      const int row = rowidx[i];
      if (row >= 0 && b_col_ok) rb[i] = *(const u32x4_t*)(B + (int64_t)row * p.ldb + nb0 + st_c * 8);
    }
    fetch_idx(kt + 1);
  };
  auto commit = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < TLPT; ++i) {
      const int off = (st_r + 16 * i) * TROWB + st_c * 16;
      *(u32x4_t*)(buf + off) = ra[i];
      *(u32x4_t*)(buf + TTILEB + off) = rb[i];
    }
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const char* buf) {
    const char* As = buf;
    const char* Bs = buf + TTILEB;
#pragma unroll
    for (int ks = 0; ks < TBK / 32; ++ks) {
      bf16x8_t af[4], bfg[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(As, ks * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfg[j] = tr_frag(Bs, ks * 32, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  char* buf0 = smem;
  char* buf1 = smem + 2 * TTILEB;
  fetch_idx(kt0);
  issue(kt0);
  commit(buf0);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    char* cur = ((kt - kt0) & 1) ? buf1 : buf0;
    char* nxt = ((kt - kt0) & 1) ? buf0 : buf1;
    const bool more = kt + 1 < kt1;
    if (more) issue(kt + 1);
    compute(cur);
    if (more) commit(nxt);
    __syncthreads();
  }
  float scale = p.alpha;
  if (p.scale_ptr) scale *= p.scale_tanh ? tanhf(*p.scale_ptr) : *p.scale_ptr;
  // acc[i][j] holds D[n = 4g + r][m = fr] of m-tile i, n-tile j
  float* C = p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      float* c = C + (int64_t)m * p.ldc + n;
      if (nsplit > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) atomicAdd(c + e, acc[i][j][e] * scale);
      } else if (n + 3 < p.N && (p.ldc & 3) == 0) {
        const f32x4_t old = *(const f32x4_t*)c;
        *(f32x4_t*)c = old + acc[i][j] * scale;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) c[e] += acc[i][j][e] * scale;
      }
    }
  }
}


// =====================================================================================================
// Persistent pipelined form (the structure of gemm_nt_pp_kernel in gemm.hip, which documents the phase plan and the
// hazard argument): 256 (M) x 256 (N) x 64 (K) tiles, 8 waves = 2 (M) x 4 (N), one block per CU, two 64 KB stages of four
// K-major half-tiles [64 k][128 columns] (A_lo, B_lo, B_hi, A_hi) filled by LDS-DMA and read with ds_read_b64_tr_b16.
//
// LDS image of a half-tile: plain 256-byte rows (LDS-DMA cannot pad), 16-byte chunk c of row k stored at chunk
// c ^ ((k & 7) << 1): a transposed fragment read touches 8 rows x 32 bytes per 32-lane group, and the XOR spreads those
// eight 32-byte pieces over the eight 32-byte bank groups (conflict-free, checked against the §LDS table of
// MI355X_MICROARCH.md: ds_read_b64_tr_b16 = 2 x 32 lanes, bank = (addr / 4) mod 64). The swizzle is applied on the
// SOURCE address of the DMA (lane -> physical chunk is fixed).
// Gathered B rows (Conv3d weight gradient): the 64 row indices of a K tile are fetched with scalar loads (4 consecutive
// k per wave instruction) when the tile's A_lo is issued — no vector load inside the loop (see sload8 in gemm.hip).
// =====================================================================================================
constexpr int Q_NT = 512, Q_BK = 64;
int g_tn_split_tail = 1;  // cut the tiles of a partial last round into K ranges: 0 never, 1 gathered launches, 2 every launch (grove_gemm_tn_set_split_tail)
int g_tn_last_parts = 1;  // K ranges per cut tile in the last pipelined launch
int g_tn_last_skip = 0;   // the last pipelined launch skipped the temporal-padding K tiles
constexpr int Q_ROWB = 256;                 // bytes per LDS row (128 bf16 columns)
constexpr int Q_HALF = Q_BK * Q_ROWB;       // 16 KB
constexpr int Q_STAGE = 4 * Q_HALF;

typedef int i32x4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4s_t sload4(const int* p) {
  i32x4s_t v;
  asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wait_vm_even12(int n) {  // exact even counts up to 12 (the rebalanced staging schedule's tail)
  if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wait_vm_even(int n) {
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Transposed fragment read as inline asm: through the builtin, hipcc guards every LDS read that may alias an LDS-DMA
// destination with s_waitcnt vmcnt(0) (it cannot see that the counted waits already ordered them), which drains the
// staging stream every phase. The asm is invisible to that pass; the lgkmcnt wait before the MFMAs is written by hand,
// followed by sched_barrier(0) (cdna_hip_programming.md rule 18), and the two halves of an operand are only combined
// AFTER that wait so that no register move can read them early.
typedef __attribute__((ext_vector_type(4))) short q_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short q_s16x8_t;
template <int OFF>
__device__ __forceinline__ q_s16x4_t ds_tr16(unsigned addr) {
  q_s16x4_t v;
#if defined(TN_ABL) && TN_ABL == 1
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));  // ablation: the same bytes without the transpose
#else
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
#endif
  return v;
}
struct TrFrag {
  q_s16x4_t lo, hi;
};
template <int OFF>
__device__ __forceinline__ TrFrag tr_pair(unsigned addr) {  // rows r .. r+3 and r+16 .. r+19 of a 16-column block
  TrFrag f;
  f.lo = ds_tr16<OFF>(addr);
  f.hi = ds_tr16<OFF + 16 * Q_ROWB>(addr);
  return f;
}
__device__ __forceinline__ bf16x8_t tr_join(const TrFrag& f) {
  return __builtin_bit_cast(bf16x8_t, (q_s16x8_t{f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]}));
}

// TSKIP (round 4; gathered launches with the temporal-padding promise b_frame_rows / b_frames, grove_hip.h): the N columns are three
// equal tap groups (temporal offset -1, 0, +1) and the K rows are frames of fk K tiles, T frames per group. For the first tap group every
// gathered row of a group's FIRST frame is the zero row, for the last tap group every row of its LAST frame: those K tiles only ever add
// +0.0 and are left out. A tile of tap group 0 / 2 walks `run` = (T - 1) fk valid K tiles per T fk (LOGICAL K tiles; the physical one
// adds the skipped frames back), and the tiles are dealt full ones (tap group 1) first, so that rounds and the cut tail hold tiles of
// one length. Whole tiles stay bit-identical to the un-skipped launch.
#ifndef TN_SCHED
// 1 = two pieces per phase (rounds 2-5, shipped); 2 = the rebalanced 0 / 2 / 2 / 4 schedule (round 5 experiment, -DTN_SCHED=2: correct —
// the TN tests and the guard screen pass — and 3 % faster on the plain (1280, 34560, 32768) launch, nothing on the gathered Conv3d
// launches the step runs (2696 / 2720 against 2717 / 2717 us): where in the K tile the pieces are issued is not what they cost);
// 3 = the NT kernel's schedule 2 (every piece one phase later, 0 / 2 / 2 / 4): plain launch -3 % (2736 -> 2655 us), the gathered
// Conv3d launches 10 % SLOWER (2762 -> 3063: the row-index hand-over lands in the phase that now issues four pieces)
#define TN_SCHED 1
#endif
struct tn_skip {
  int fk, T;  // K tiles per frame, frames per group
};
// KBATCH (round 6; grove_gemm_tn_params.k_batches — the Winograd weight gradient, 64 transform points in one launch): the operands are
// k_batches stacked K ranges and the tile index carries the batch: L = batch * (tiles_m * tiles_n) + tile; batch b reads rows
// [b K, (b + 1) K) of A and B and accumulates into C + b * sC_batch. Everything else (units, rounds, the cut tail) is unchanged.
template <bool GATHER, bool TSKIP = false, bool KBATCH = false>
__global__ __launch_bounds__(Q_NT) void gemm_tn_pp_kernel(const grove_gemm_tn_params p, const int tiles_m, const int tiles_n, const int tiles_whole,
                                                          const int parts, const tn_skip sk = tn_skip{0, 0}) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, g = lane >> 4;
  const bf16_raw* __restrict__ A = (const bf16_raw*)p.A;
  const bf16_raw* __restrict__ B = (const bf16_raw*)p.B;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  // Work units, dealt round-robin (unit u = wgid + G r): the first tiles_whole tiles whole, then the tiles of the partial last
  // round cut into `parts` equal K ranges, part-major (all first parts, then all second parts: the blocks of a round walk the same
  // K range, as they do in a round of whole tiles). A partial range is added to C with fp32 atomics (launch_tn_pp picks parts).
  const int tiles1 = tiles_m * tiles_n;
  const int tiles = KBATCH ? tiles1 * p.k_batches : tiles1;
  const int tail_tiles = tiles - tiles_whole;
  const int units = tiles_whole + tail_tiles * parts;
  const int nk = p.K / Q_BK;
  // TSKIP: tile order = tap group 1 (full) first, then groups 0 and 2; LOGICAL K tiles per tile
  const int tiles_g = TSKIP ? tiles / 3 : 0;                                 // tiles of one tap group (tn-major: a contiguous third)
  const int sk_run = TSKIP ? (sk.T - 1) * sk.fk : 0, sk_per = TSKIP ? sk.T * sk.fk : 1;
  const int nk_short = TSKIP ? (nk / sk_per) * sk_run : nk;
  auto tile_of = [&](int Lo) { return !TSKIP ? Lo : (Lo < tiles_g ? Lo + tiles_g : (Lo < 2 * tiles_g ? Lo - tiles_g : Lo)); };
  auto grp_of = [&](int L) { return !TSKIP ? 1 : (L >= 2 * tiles_g) + (L >= tiles_g); };
  auto unit_of = [&](int u, int& L, int& ka, int& kb) {
    if (u < tiles_whole) {
      L = tile_of(u);
      ka = 0;
      kb = (TSKIP && grp_of(L) != 1) ? nk_short : nk;
    } else {
      const int v = u - tiles_whole;
      const int part = (v >= tail_tiles) + (v >= 2 * tail_tiles) + (v >= 3 * tail_tiles);
      L = tile_of(tiles_whole + v - part * tail_tiles);
      const int nkl = (TSKIP && grp_of(L) != 1) ? nk_short : nk;
      ka = part * nkl / parts;
      kb = (part + 1) * nkl / parts;
    }
  };
  // physical K tile of logical K tile k of a tile in tap group grp, and the valid tiles left in its run (TSKIP)
  auto phys_of = [&](int k, int grp, int& left) {
    if (!TSKIP || grp == 1) {
      left = 0x7fffffff;
      return k;
    }
    const int q = k / sk_run, r = k - q * sk_run;
    left = sk_run - r;
    return q * sk_per + r + (grp == 0 ? sk.fk : 0);
  };
  int NT = 0;
  for (int u = wgid; u < units; u += G) {
    int L, ka, kb;
    unit_of(u, L, ka, kb);
    NT += kb - ka;
  }
  const int NH = 4 * NT;
  const int n_per_tap = p.N / p.b_taps;

  // staging: thread -> (row k = 32 i + tid / 16, physical chunk tid % 16) of a half-tile, two instructions per half
  const int st_k = tid >> 4;                      // + 32 i
  const int lc = (tid & 15) ^ ((st_k & 7) << 1);  // logical chunk this lane must fetch (st_k + 32 i has the same low 3 bits)
  const bf16_raw* pa[2];     // A_lo / A_hi column bases (row 0)
  int colb[2];               // B_lo / B_hi column offsets inside a (gathered) row
  const bf16_raw* pbrow[2];  // GATHER: my two B rows of the K tile being issued (nullptr = zero row)
  const bf16_raw* Bt = B;    // KBATCH: row 0 of the tile's batch
  const int32_t* bidx = nullptr;
  int is_u = wgid, is_k = 0, is_kb = 0;
  int is_kp = 0, is_left = 0x7fffffff, is_grp = 1;  // physical K tile being issued, valid tiles left in its run, tap group (TSKIP)
  auto set_tile = [&](int u) {  // consecutive tiles share the B panel
    int L;
    unit_of(u, L, is_k, is_kb);
    is_grp = grp_of(L);
    is_kp = phys_of(is_k, is_grp, is_left);
    const int bt = KBATCH ? L / tiles1 : 0, Lt = KBATCH ? L - bt * tiles1 : L;
    const int tm = Lt % tiles_m, tn = Lt / tiles_m;
    const int m0 = tm * 256, n0 = tn * 256;
    const int tap = n0 / n_per_tap, nb0 = n0 - tap * n_per_tap;
    const bf16_raw* At = A;
    if constexpr (KBATCH) At = A + (int64_t)bt * p.K * p.lda, Bt = B + (int64_t)bt * p.K * p.ldb;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pa[h] = At + min(m0 + 128 * h + lc * 8, p.M - 8);
      colb[h] = min(nb0 + 128 * h + lc * 8, n_per_tap - 8);
    }
    if (GATHER) bidx = p.b_idx + (int64_t)tap * p.K;
  };
  // GATHER: the 64 row indices of a K tile — 4 consecutive k per wave instruction, so two 4-dword SCALAR loads per wave —
  // are prefetched one K tile ahead and taken after a wait that is already satisfied (every MFMA segment starts with
  // lgkmcnt(0)); the "+s" ties make the wait the definition point of the values, so no use can be scheduled before it.
  i32x4s_t nx0 = {0, 0, 0, 0}, nx1 = {0, 0, 0, 0};
  auto prefetch_rows = [&](const int32_t* base) {
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x0" : "=&s"(nx0), "=&s"(nx1) : "s"(base + 4 * wave), "s"(base + 32 + 4 * wave));
  };
  auto take_rows = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nx0), "+s"(nx1)::"memory");
    int r0 = nx0[0], r1 = nx1[0];
    r0 = g == 1 ? nx0[1] : r0; r0 = g == 2 ? nx0[2] : r0; r0 = g == 3 ? nx0[3] : r0;
    r1 = g == 1 ? nx1[1] : r1; r1 = g == 2 ? nx1[2] : r1; r1 = g == 3 ? nx1[3] : r1;
    pbrow[0] = r0 >= 0 ? B + (int64_t)r0 * p.ldb : nullptr;
    pbrow[1] = r1 >= 0 ? B + (int64_t)r1 * p.ldb : nullptr;
  };
  auto next_rows_base = [&]() -> const int32_t* {  // index window of the K tile AFTER (is_u, is_k)
    int k2 = is_k + 1, kp2 = is_kp + 1;
    if (TSKIP && is_left == 1) kp2 += sk.fk;  // the run ends here: the next valid K tile is one frame on
    const int32_t* bb = bidx;
    if (k2 == is_kb) {
      int L2 = 0, kb2;
      k2 = 0, kp2 = 0;
      if (is_u + G < units) {
        int left2;
        unit_of(is_u + G, L2, k2, kb2);
        kp2 = phys_of(k2, grp_of(L2), left2);
        bb = p.b_idx + (int64_t)(((L2 / tiles_m) * 256) / n_per_tap) * p.K;
      }
    }
    return bb + kp2 * Q_BK;
  };
  auto issue = [&](int x, int stream_t) {
    char* dst = smem + (stream_t & 1) * Q_STAGE + x * Q_HALF + wave * (64 * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_raw* src;
      if (x == 0 || x == 3) {
        src = pa[x == 3] + (int64_t)(is_kp * Q_BK + 32 * i + st_k) * p.lda;
      } else if (GATHER) {
        src = pbrow[i] ? pbrow[i] + colb[x == 2] : (const bf16_raw*)g_tn_zero_page;
      } else {
        src = Bt + (int64_t)(is_kp * Q_BK + 32 * i + st_k) * p.ldb + colb[x == 2];
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + i * (Q_NT * 16)), 16, 0, 0);
    }
  };
  auto advance_issue = [&]() {
    if (++is_k == is_kb) {
      is_u += G;
      set_tile(is_u);
    } else {
      ++is_kp;
      if (TSKIP && --is_left == 0) is_kp += sk.fk, is_left = sk_run;
    }
    if (GATHER) {
      take_rows();
      prefetch_rows(next_rows_base());
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transposed fragment reads: lane (fr, g) reads 8 bytes of row 4 g + (fr >> 2) (+16) at column 4 (fr & 3) of the fragment
  const int qq = fr >> 2, pp = fr & 3;
  const int R2 = ((4 * (g & 1) + qq) & 7) << 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (4 * g + qq) * Q_ROWB + pp * 8;
  unsigned a_adr[4], b_adr[2];  // LDS byte address of fragment i / j in stage 0, region 0, k-step 0 (swizzled column)
#pragma unroll
  for (int i = 0; i < 4; ++i) a_adr[i] = lds0 + ((wr * 8 + 2 * i) ^ R2) * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) b_adr[j] = lds0 + ((wc * 4 + 2 * j) ^ R2) * 16;
  TrFrag af[4][2], b0[2][2], b1[2][2];
#if defined(TN_ABL) && (TN_ABL == 3 || TN_ABL == 4)
#define TN_RD (T == 0)   /* ablation: fragments read for the first K tile only */
#else
#define TN_RD true
#endif
#if defined(TN_ABL) && (TN_ABL == 2 || TN_ABL == 4)
#define TN_DMA(Q) ((Q) < 4)  /* ablation: no LDS-DMA after the first K tile */
#else
#define TN_DMA(Q) true
#endif
#define QQ_READ_A(X)                                                              \
  if (TN_RD) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
    af[i][0] = tr_pair<(X) * Q_HALF>(a_adr[i] + st_off);                         \
    af[i][1] = tr_pair<(X) * Q_HALF + 32 * Q_ROWB>(a_adr[i] + st_off);           \
  }
#define QQ_READ_B(X, BB)                                                          \
  if (TN_RD) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                \
    BB[j][0] = tr_pair<(X) * Q_HALF>(b_adr[j] + st_off);                         \
    BB[j][1] = tr_pair<(X) * Q_HALF + 32 * Q_ROWB>(b_adr[j] + st_off);           \
  }
#define QQ_MMA(IO, JO, BB)                                                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                                      \
  __builtin_amdgcn_s_setprio(1);                                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[IO + i][JO + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_join(BB[j][ks]), tr_join(af[i][ks]), acc[IO + i][JO + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                                          \
  __builtin_amdgcn_sched_barrier(0);
#define QQ_MEM_END(Q, X, TOFF, WAIT)                                                                                      \
  if ((Q) + 6 < NH) {                                                                                                     \
    if (TN_DMA(Q)) issue(X, T + TOFF);                                                                                                   \
    if (WAIT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                            \
  } else if (WAIT) {                                                                                                      \
    wait_vm_even(2 * max(NH - 3 - (Q), 0));                                                                               \
  }                                                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                      \
  __builtin_amdgcn_s_barrier();

  float scale = p.alpha;
  if (p.scale_ptr) scale *= p.scale_tanh ? tanhf(*p.scale_ptr) : *p.scale_ptr;
  asm volatile("" ::"v"(scale));  // fetched and used before the loop (see gemm_nt_pp_kernel)
  set_tile(is_u);
  if (GATHER) {
    prefetch_rows(bidx + is_kp * Q_BK);
    take_rows();
    prefetch_rows(next_rows_base());
  }
#if TN_SCHED == 2
  // Rebalanced staging schedule (round 5). A K tile's fragment reads fall 24 / 8 / 16 / 0 on its four phases; with two LDS-DMA pieces
  // in every phase the memory segments of phases 1 and 3 outlast the partner group's MFMA segment. Here the pieces of K tile T + 2 are
  // issued 0 / 2 / 2 / 4 (regions A_lo, B_lo, then B_hi + A_hi) into the slots tile T frees as it goes — regions 0 and 1 are read in
  // phase 1, region 2 in phase 2, region 3 in phase 3 — so a piece is in flight for 6-7 phases instead of 5-6 and up to 16 pieces
  // per wave instead of 10. Waits (vmcnt retires in issue order; n1 / n2 = tile T + 1 / T + 2 exists):
  //   phase 1 retires region 2 of T (younger: region 3 of T, all of T + 1):                 2 + 8 n1
  //   phase 2 retires region 3 of T (younger: T + 1, region 0 of T + 2 just issued):        8 n1 + 2 n2
  //   phase 4 retires regions 0, 1 of T + 1 (younger: regions 2, 3 of T + 1, T + 2):        4 + 8 n2
  // WAR: region 0 of T + 2 is issued in phase 2 of T by a wave whose partner group read region 0 of T before the barrier both have
  // passed; the reads were queued at the LDS before that barrier, the piece lands hundreds of cycles after its issue.
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  if (NT > 1) {
    advance_issue();
    issue(0, 1); issue(1, 1); issue(2, 1); issue(3, 1);
  }
  wait_vm_even12(4 + (NT > 1 ? 8 : 0));
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
#else
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  if (NT > 1) {
    advance_issue();
    issue(0, 1); issue(1, 1);
  }
  wait_vm_even(2 * (min(5, NH - 1) - 1));
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
#endif

  int c_u = wgid, c_L, c_k, c_kb;
  unit_of(c_u, c_L, c_k, c_kb);
  for (int T = 0; T < NT; ++T) {
    const unsigned st_off = (T & 1) * Q_STAGE;
    const int q = 4 * T;
#if TN_SCHED == 2
    const bool n1 = T + 1 < NT, n2 = T + 2 < NT;
    (void)q;
    // phase 1: 24 reads, no piece
    QQ_READ_B(1, b0)
    QQ_READ_A(0)
    if (n1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(0, 0, b0)
    __builtin_amdgcn_s_barrier();
    // phase 2: 8 reads, region 0 (A_lo) of T + 2
    if (n2) advance_issue();  // before this phase's reads: its (already satisfied) lgkmcnt wait must not cover them
    QQ_READ_B(2, b1)
    if (n2) {
      issue(0, T);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
      wait_vm_even12(n1 ? 8 : 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(0, 2, b1)
    __builtin_amdgcn_s_barrier();
    // phase 3: 16 reads, region 1 (B_lo) of T + 2; nothing to retire (phase 4 reads nothing)
    QQ_READ_A(3)
    if (n2) issue(1, T);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(4, 2, b1)
    __builtin_amdgcn_s_barrier();
    // phase 4: no reads, regions 2 and 3 (B_hi, A_hi) of T + 2
    if (n2) {
      issue(2, T);
      issue(3, T);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (n1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(4, 0, b0)
    __builtin_amdgcn_s_barrier();
#elif TN_SCHED == 3
    // the NT kernel's schedule 2: every piece one phase later, both half-tiles of the K tile after next in phase 4 (pieces 0 / 2 / 2 / 4);
    // half-tile h = 4 T + region in issue order; phase 1 retires h = q + 2 (younger: q + 3 .. q + 5), phase 2 q + 3 (.. q + 6), phase 4
    // q + 5 (.. q + 9); same prologue as schedule 1
    auto tail_wait = [&](int needed, int last) { wait_vm_even(2 * max(min(last, NH - 1) - needed, 0)); };
    const bool steady = q + 9 < NH;
    QQ_READ_B(1, b0)
    QQ_READ_A(0)
    if (steady) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else tail_wait(q + 2, q + 5);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(0, 0, b0)
    __builtin_amdgcn_s_barrier();
    QQ_READ_B(2, b1)
    if (q + 6 < NH) issue(2, T + 1);
    if (steady) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else tail_wait(q + 3, q + 6);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(0, 2, b1)
    __builtin_amdgcn_s_barrier();
    QQ_READ_A(3)
    if (q + 7 < NH) issue(3, T + 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(4, 2, b1)
    __builtin_amdgcn_s_barrier();
    if (q + 8 < NH) {
      advance_issue();
      issue(0, T + 2);
    }
    if (q + 9 < NH) issue(1, T + 2);
    if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else tail_wait(q + 5, q + 9);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    QQ_MMA(4, 0, b0)
    __builtin_amdgcn_s_barrier();
#else
    QQ_READ_B(1, b0)
    QQ_READ_A(0)
    QQ_MEM_END(q, 2, 1, true)
    QQ_MMA(0, 0, b0)
    __builtin_amdgcn_s_barrier();
    QQ_READ_B(2, b1)
    QQ_MEM_END(q + 1, 3, 1, true)
    QQ_MMA(0, 2, b1)
    __builtin_amdgcn_s_barrier();
    if (q + 8 < NH) advance_issue();  // before this phase's reads: its (already satisfied) lgkmcnt wait must not cover them
    QQ_READ_A(3)
    QQ_MEM_END(q + 2, 0, 2, false)
    QQ_MMA(4, 2, b1)
    __builtin_amdgcn_s_barrier();
    QQ_MEM_END(q + 3, 1, 2, true)
    QQ_MMA(4, 0, b0)
    __builtin_amdgcn_s_barrier();
#endif
    if (++c_k == c_kb) {
      // acc[i][j] holds D[n = 4 g + e][m = fr] of A fragment i (m) and B fragment j (n): C += scale * D, 16 bytes per lane
      const int cbt = KBATCH ? c_L / tiles1 : 0, cLt = KBATCH ? c_L - cbt * tiles1 : c_L;
      const int tm = cLt % tiles_m, tn = cLt / tiles_m;
      const int m0 = tm * 256 + wr * 64 + fr, n0 = tn * 256 + wc * 32 + 4 * g;
      float* const Cb = KBATCH ? p.C + (int64_t)cbt * p.sC_batch : p.C;
      if (__builtin_expect(c_u >= tiles_whole && parts > 1, 0)) {  // one K range of a cut tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + (j >> 1) * 128 + (j & 1) * 16;
          if (n >= p.N) continue;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int m = m0 + (i >> 2) * 128 + (i & 3) * 16;
            if (m < p.M) {
              float* c = Cb + (int64_t)m * p.ldc + n;
#pragma unroll
              for (int e = 0; e < 4; ++e) unsafeAtomicAdd(c + e, acc[i][j][e] * scale);
            }
          }
        }
      } else
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + (j >> 1) * 128 + (j & 1) * 16;
        if (n >= p.N) continue;
        f32x4_t old[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = min(m0 + (i >> 2) * 128 + (i & 3) * 16, p.M - 1);
          old[i] = (KBATCH && p.overwrite) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : *(const f32x4_t*)(Cb + (int64_t)m * p.ldc + n);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = m0 + (i >> 2) * 128 + (i & 3) * 16;
          if (m < p.M) *(f32x4_t*)(Cb + (int64_t)m * p.ldc + n) = old[i] + acc[i][j] * scale;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      c_u += G;
      if (c_u < units) unit_of(c_u, c_L, c_k, c_kb);
#ifndef TN_NO_EPI_DRAIN
      // The epilogue's stores (and atomics) are still outstanding here, and hipcc guards their DATA registers against the next K tile's
      // fragment reads: with the loop's LDS-DMA pieces in the same counter it can only do that with `s_waitcnt vmcnt(0)` — which it
      // placed at the LOOP HEADER, i.e. in every K tile, for a path taken once per 512: the staging queue drained once per K tile, a
      // quarter of the kernel's time (round 5: tools/dev/tn_ablate.py, 2855 us with the pieces, 2126 without). A wait the compiler
      // can see, here, once per output tile, clears its scoreboard before the back edge.
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15)
#endif
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef QQ_MMA
#undef QQ_MEM_END
#undef QQ_READ_A
#undef QQ_READ_B
}

int g_tn_tap_skip = 1;  // honour b_frame_rows / b_frames (grove_gemm_tn_set_tap_skip; A/B knob)
template <bool GATHER>
int launch_tn_pp(const grove_gemm_tn_params& p, hipStream_t s) {
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const size_t lds = 2 * (size_t)Q_STAGE;
  static bool attr_set = false;
  static int num_cus = 0;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_tn_pp_kernel<GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (GATHER) hipFuncSetAttribute((const void*)gemm_tn_pp_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (num_cus <= 0) num_cus = 256;
    attr_set = true;
  }
  const int kb = (!GATHER && p.k_batches > 1) ? p.k_batches : 1;
  const int tiles = tiles_m * tiles_n * kb;
  const int cap = grove_gemm_persistent_blocks();  // the resident-block cap of the persistent kernels (N > 1: CUs left to RCCL)
  const int cus = cap > 0 && cap < num_cus ? cap : num_cus;
  const int G = tiles < cus ? tiles : cus;
  // The partial last round (675 tiles on 256 CUs: 2 rounds + 163 tiles) is cut into 2-4 K ranges when that shortens it:
  // ceil(tail * parts / G) rounds of 1 / parts each, e.g. 163 x 3 = 489 units = 2 rounds of a third (0.67 instead of 1).
  // Measured (tools/bench_gemm_tn.py, (1280, 34560, 32768)): gathered taps 2.90-2.98 -> 2.85 ms; the plain form does not move
  // (2.73 ms either way: at the board's power limit a launch costs its energy, not its rounds), so by default only gathered
  // launches are cut and plain ones keep their fixed sum order.
  // temporal tap skipping: three equal tap groups of whole column tiles, frames of whole K tiles, K = whole groups of frames
  tn_skip sk{0, 0};
  if (GATHER && g_tn_tap_skip && p.b_frame_rows > 0 && p.b_frames >= 2 && p.b_taps % 3 == 0 && tiles_n % 3 == 0 && p.b_frame_rows % Q_BK == 0 &&
      p.K % ((long)p.b_frame_rows * p.b_frames) == 0 && p.M % 256 == 0 && p.N % 256 == 0)
    sk = tn_skip{p.b_frame_rows / Q_BK, p.b_frames};
  const int nk_eff = sk.fk ? (p.K / Q_BK) / sk.T * (sk.T - 1) : p.K / Q_BK;  // K tiles of the tiles a cut tail holds (the short ones come last)
  int parts = 1, tail = tiles % G;
  if (tail && !grove_det_on() && !(kb > 1 && p.overwrite) && (g_tn_split_tail == 2 || (g_tn_split_tail == 1 && GATHER))) {
    double best = 1.0;
    for (int s2 = 2; s2 <= 4; ++s2) {
      if (nk_eff / s2 < 32) break;
      const double c = (double)((tail * s2 + G - 1) / G) / s2 + 0.02 * s2;  // + the atomics of a part
      if (c < best) { best = c; parts = s2; }
    }
  }
  g_tn_last_parts = parts;
  g_tn_last_skip = sk.fk != 0;
  if constexpr (GATHER) {
    if (sk.fk) {
      hipLaunchKernelGGL((gemm_tn_pp_kernel<true, true>), dim3(G), dim3(Q_NT), lds, s, p, tiles_m, tiles_n, parts > 1 ? tiles - tail : tiles, parts, sk);
      GROVE_LAUNCH_CHECK();
      return GROVE_OK;
    }
  }
  if constexpr (!GATHER) {
    if (kb > 1) {
      static bool attr_kb = false;
      if (!attr_kb) {
        hipFuncSetAttribute((const void*)gemm_tn_pp_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_kb = true;
      }
      hipLaunchKernelGGL((gemm_tn_pp_kernel<false, false, true>), dim3(G), dim3(Q_NT), lds, s, p, tiles_m, tiles_n, parts > 1 ? tiles - tail : tiles, parts, sk);
      GROVE_LAUNCH_CHECK();
      return GROVE_OK;
    }
  }
  hipLaunchKernelGGL((gemm_tn_pp_kernel<GATHER, false>), dim3(G), dim3(Q_NT), lds, s, p, tiles_m, tiles_n, parts > 1 ? tiles - tail : tiles, parts, sk);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}

}  // namespace

static int g_tn_pipelined = -1;  // -1 auto, 0 never, 1 whenever the problem is eligible (A/B runs and tests)
extern "C" int grove_gemm_tn_set_pipelined(int mode) {
  g_tn_pipelined = mode;
  return GROVE_OK;
}

extern "C" int grove_gemm_tn_set_split_tail(int on) {
  g_tn_split_tail = on;
  return GROVE_OK;
}
extern "C" int grove_gemm_tn_last_parts(void) { return g_tn_last_parts; }
extern "C" int grove_gemm_tn_set_tap_skip(int on) {
  g_tn_tap_skip = on ? 1 : 0;
  return GROVE_OK;
}
extern "C" int grove_gemm_tn_last_skip(void) { return g_tn_last_skip; }

extern "C" int grove_gemm_tn_bf16(const grove_gemm_tn_params* pp, void* stream) {
  GROVE_CHECK(pp && pp->M > 0 && pp->N > 0 && pp->K > 0, GROVE_E_SHAPE, "gemm_tn: bad shape");
  grove_gemm_tn_params p = *pp;
  if (p.b_taps <= 0) p.b_taps = 1;
  GROVE_CHECK(p.M % 8 == 0 && p.N % 8 == 0, GROVE_E_SHAPE, "gemm_tn: M=%d and N=%d must be multiples of 8", p.M, p.N);
  GROVE_CHECK(p.N % p.b_taps == 0 && (p.N / p.b_taps) % TBN == 0 || p.b_taps == 1, GROVE_E_SHAPE, "gemm_tn: N/b_taps must be a multiple of %d", TBN);
  GROVE_CHECK(p.lda % 8 == 0 && p.ldb % 8 == 0, GROVE_E_ALIGN, "gemm_tn: lda/ldb must be multiples of 8");
  GROVE_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && ((uintptr_t)p.C & 15) == 0, GROVE_E_ALIGN, "gemm_tn: pointers must be 16-byte aligned");
  const int tiles = ((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
  const int nkt = (p.K + TBK - 1) / TBK;
  int split = p.split_k;
  if (split <= 0) {
    split = 1;
    if (tiles < 256 && nkt >= 8) {
      split = (768 + tiles - 1) / tiles;
      if (split > nkt / 2) split = nkt / 2;
      if (split > 256) split = 256;
      // a few output tiles under a very long K (the box decoder's key-side weight gradients: 128 x 256 from K = 98,304): every K range adds
      // its whole tile with fp32 atomics onto the SAME addresses, and that serialises — measured time ~ 1.05 us x (k tiles per range) +
      // 0.28 us x tiles x ranges (tools/dev/tn_split_sweep.py: 256 ranges 145-215 us, 32-64 ranges 63-87 us), minimal at the root below
      if (tiles <= 8) {
        int best = 1;
        while ((long)(best + 1) * (best + 1) * tiles * 4 <= 15L * nkt) ++best;  // best = floor(sqrt(3.75 nkt / tiles))
        if (split > best) split = best;
      }
      if (split < 1) split = 1;
    }
  }
  if (p.k_batches > 1) {  // the K-batched form exists in the pipelined kernel only
    GROVE_CHECK(!p.b_idx && p.b_taps == 1 && p.K % 64 == 0 && p.split_k <= 1 && p.ldc % 4 == 0 && p.M >= 8 && p.N >= 8 && p.sC_batch % 4 == 0 &&
                    (long)p.k_batches * p.K < (1L << 31),
                GROVE_E_SHAPE, "gemm_tn: k_batches needs plain operands (no b_idx / taps / split_k), K %% 64 == 0 and 16-byte aligned batch strides");
    return launch_tn_pp<false>(p, (hipStream_t)stream);
  }
  if (grove_det_on()) split = 1;  // deterministic mode: whole-K tiles only (no fp32 atomics into C), here and in the pipelined kernel's tail
  // the persistent pipelined kernel: un-split problems with enough 256 x 256 tiles whose column tiles stay inside one tap
  const int g_force = g_tn_pipelined;
  const int n_per_tap = p.N / p.b_taps;
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const bool pp_ok = p.K % 64 == 0 && p.split_k <= 1 && (p.b_taps == 1 || n_per_tap % 256 == 0) && p.ldc % 4 == 0 && p.M >= 8 &&
                     n_per_tap >= 8 && (!p.b_idx || p.K >= 64);
  if (pp_ok && g_force != 0 && (g_force == 1 || (split == 1 && t256 >= 192 && nkt >= 16)))
    return p.b_idx ? launch_tn_pp<true>(p, (hipStream_t)stream) : launch_tn_pp<false>(p, (hipStream_t)stream);
  static bool attr = false;
  const size_t lds = 4 * TTILEB;
  if (!attr) {
    hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, 1, split), dim3(TNT), lds, (hipStream_t)stream, p);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}
