// bf16 "NT" GEMM with fused epilogue and gather/scatter row maps for gfx950 (MI355X).
//
//   C[m, n] = epi( alpha * sum_k A[row_a(m, k), k] * B[n, k] )
//
// Tiling (wave64, v_mfma_f32_16x16x32_bf16):
//   workgroup = 256 threads = 4 waves laid out 2 (M) x 2 (N); macro tile 128 x 128 x BK
//   (BK = 64, or 32 when K % 64 != 0); each wave owns 64 x 64 = 4 x 4 MFMA tiles.
//   A and B tiles are staged global -> LDS either with LDS-DMA (global_load_lds_dwordx4,
//   lane-linear destination, swizzle applied to the per-lane SOURCE address) or through
//   registers (global_load_dwordx4 + ds_write_b128); both produce the same XOR-swizzled image
//   that ds_read_b128 fragment reads hit conflict-free (cdna_hip_programming.md T2, rule 21).
//   LDS is double buffered: the next K tile is in flight while the current one feeds the MFMAs.
//   The MFMA is issued with swapped operands (D = Btile . Atile^T) so that each lane ends up
//   with 4 consecutive n for one m -> 8-byte (bf16) / 16-byte (f32) epilogue accesses.
//   blockIdx is remapped XCD-aware (bijective, cdna_hip_programming.md §5) and grouped so that
//   tiles sharing an A row panel / B column panel run on the same XCD L2.
#include <type_traits>

#include "common.h"
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

constexpr int BN = 128;
constexpr int NT = 256;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];  // 256 B of zeros

// physical 16-byte chunk of logical chunk c in LDS row r
template <int BK>
__device__ __forceinline__ int swz(int r, int c) {
  if constexpr (BK == 64) {
    return c ^ ((r >> 1) & 7);
  } else {
    // 64-byte rows: 4 rows share one 256-byte bank row; band permutation {0,2,3,1}
    const int band = (r >> 2) & 3;
    return c ^ ((0x1320 >> (4 * band)) & 3);
  }
}

struct TileCoord {
  int tm, tn;
};

__device__ __forceinline__ TileCoord map_block(int tiles_m, int tiles_n) {
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  constexpr int GM = 8;
  const int per_band = GM * tiles_n;
  const int band = wgid / per_band;
  const int first_m = band * GM;
  const int gm = min(tiles_m - first_m, GM);
  const int in_band = wgid - band * per_band;
  TileCoord t;
  t.tm = first_m + in_band % gm;
  t.tn = in_band / gm;
  return t;
}

// Fused epilogue shared by all NT kernels. acc[i][j] holds D[n = nw0 + j*16 + 4*fq + e][m = mw0 + i*16 + fr] (contiguous form).
// MG / NG: fragments per contiguous group and MGS / NGS: the stride between groups (the pipelined 256 x 256 kernel gives
// each wave two row groups and two column groups, one per staged half-tile).
template <int MI, int NJ, int MG = MI, int MGS = 0, int NG = NJ, int NGS = 0>
__device__ __forceinline__ void gemm_epilogue(const grove_gemm_params& p, const int vec_ok, f32x4_t (&acc)[MI][NJ], const int mw0,
                                              const int nw0, const int fr, const int fq, const int b1, const int b2, const int nsplit) {
  float scale = 1.f;
  if (p.scale_ptr) {
    scale = *p.scale_ptr;
    if (p.scale_tanh) scale = tanhf(scale);
  }
  const bf16_raw* __restrict__ bias = (const bf16_raw*)p.bias;
  const int64_t c_boff = (int64_t)b1 * p.sC1 + (int64_t)b2 * p.sC2;
  const int64_t r_boff = (int64_t)b1 * p.sR1 + (int64_t)b2 * p.sR2;
  const bool vec4 = vec_ok && ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0);

#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mw0 + (i / MG) * MGS + (i % MG) * 16 + fr;
    if (m >= p.M) continue;
    int crow = m;
    if (p.c_idx) {
      crow = p.c_idx[m];
      if (crow < 0) continue;
    }
    int rrow = crow;
    if (p.r_idx) rrow = p.r_idx[m];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = nw0 + (j / NG) * NGS + (j % NG) * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha;
      const bool full = vec4 && (n + 3 < p.N);
      if (bias) {
        if (full) {
          const u32x2_t bb = *(const u32x2_t*)(bias + n);
          v[0] += bf_lo(bb.x); v[1] += bf_hi(bb.x); v[2] += bf_lo(bb.y); v[3] += bf_hi(bb.y);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) v[e] += bf2f(bias[n + e]);
        }
      }
      if (p.aux) {
        bf16_raw* aux = (bf16_raw*)p.aux + c_boff + (int64_t)crow * p.ldc + n;
        float av[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = (MI <= 4 && p.aux_grad) ? act_grad(p.act, v[e]) : v[e];  // (no sharing with act(v) here; not in the 192-row variant: registers)
        if (full) {
          *(u32x2_t*)aux = u32x2_t{pack2bf(av[0], av[1]), pack2bf(av[2], av[3])};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) aux[e] = f2bf(av[e]);
        }
      }
      if (p.act != GROVE_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(p.act, v[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= scale;
      if (p.residual && rrow >= 0) {
        const bf16_raw* res = (const bf16_raw*)p.residual + r_boff + (int64_t)rrow * p.ldr + n;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) {
          const u32x2_t rr = *(const u32x2_t*)res;
          rv[0] = bf_lo(rr.x); rv[1] = bf_hi(rr.x); rv[2] = bf_lo(rr.y); rv[3] = bf_hi(rr.y);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) rv[e] = bf2f(res[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (MI <= 4 && p.residual_mul) ? v[e] * rv[e] : v[e] + rv[e];
      }
      if (p.c_dtype == GROVE_BF16) {
        bf16_raw* c = (bf16_raw*)p.C + c_boff + (int64_t)crow * p.ldc + n;
        if (full) {
          *(u32x2_t*)c = u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) c[e] = f2bf(v[e]);
        }
      } else {
        float* c = (float*)p.C + c_boff + (int64_t)crow * p.ldc + n;
        if (nsplit > 1) {
          // split-K partials meet in C through fp32 atomics (C pre-initialised by the caller: accumulate semantics)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) atomicAdd(c + e, v[e]);
        } else if (full) {
          f32x4_t o = f32x4_t{v[0], v[1], v[2], v[3]};
          if (p.accumulate) {
            const f32x4_t old = *(const f32x4_t*)c;
            o += old;
          }
          *(f32x4_t*)c = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) c[e] = p.accumulate ? c[e] + v[e] : v[e];
        }
      }
    }
  }
}

template <int BK, bool GLDS, int NJ, int MI>
__global__ __launch_bounds__(NT) void gemm_nt_kernel(const grove_gemm_params p, const int vec_ok) {
  constexpr int BM_ = 32 * MI;                  // macro tile M: 128 (MI = 4) or 192 (MI = 6, fewer / fuller rounds of blocks)
  constexpr int WM = 16 * MI;                   // per-wave M extent
  constexpr int BN_ = 32 * NJ;                  // macro tile N: 128 (NJ = 4) or 64 (NJ = 2, for wave-quantisation-bound shapes)
  constexpr int WN = 16 * NJ;                   // per-wave N extent
  constexpr int CPR = BK / 8;                   // 16-byte chunks per LDS row
  constexpr int ROW_BYTES = BK * 2;
  constexpr int TILE_BYTES = BM_ * ROW_BYTES;   // A tile
  constexpr int B_BYTES = BN_ * ROW_BYTES;      // B tile
  constexpr int LPT = (BM_ * CPR) / NT;         // 16-byte loads per thread per A tile
  constexpr int LPTM = LPT > ((32 * NJ) * CPR) / NT ? LPT : ((32 * NJ) * CPR) / NT;
  constexpr int LPTB = (BN_ * CPR) / NT;        // ... per B tile
  constexpr int ROWS_PER_PASS = NT / CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [buf0: A | B][buf1: A | B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_m = (p.M + BM_ - 1) / BM_;
  const int tiles_n = (p.N + BN_ - 1) / BN_;
  const TileCoord tc = map_block(tiles_m, tiles_n);
  const int m0 = tc.tm * BM_, n0 = tc.tn * BN_;

  const int bz = blockIdx.y;
  const int b1 = bz / p.batch2, b2 = bz - b1 * p.batch2;
  const bf16_raw* __restrict__ Ab = (const bf16_raw*)p.A + (int64_t)b1 * p.sA1 + (int64_t)b2 * p.sA2;
  const bf16_raw* __restrict__ Bb = (const bf16_raw*)p.B + (int64_t)b1 * p.sB1 + (int64_t)b2 * p.sB2;

  // ---- staging geometry (fixed per thread) ----
  const int st_c = tid % CPR;          // logical chunk within the row
  const int st_r0 = tid / CPR;         // first row, + ROWS_PER_PASS per pass
  int a_m[LPT];                        // clamped global m of each staged A row
  const bf16_raw* b_ptr[LPTB];
  int lds_off[LPTM];                   // byte offset inside a tile for the reg-staged write
  // For LDS-DMA the destination is lane-linear: lane -> (row, physical chunk); the lane must
  // therefore FETCH the logical chunk that belongs at that physical slot (swz is an involution).
  int src_c[LPTM];
#pragma unroll
  for (int i = 0; i < LPTM; ++i) {
    const int r = st_r0 + i * ROWS_PER_PASS;
    if (i < LPT) a_m[i] = min(m0 + r, p.M - 1);
    src_c[i] = GLDS ? swz<BK>(r, st_c) : st_c;
    lds_off[i] = r * ROW_BYTES + swz<BK>(r, st_c) * 16;
    if (i < LPTB) {
      const int n = min(n0 + r, p.N - 1);
      b_ptr[i] = Bb + (int64_t)n * p.ldb + src_c[i] * 8;
    }
  }

  const int taps = p.a_taps;
  const int k_per_tap = p.K / taps;
  const bf16_raw* a_ptr[LPT];
  int cur_tap = -1;
  auto resolve_a_rows = [&](int tap) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      int row = a_m[i];
      if (p.a_idx) row = p.a_idx[(int64_t)tap * p.M + row];
      a_ptr[i] = row < 0 ? nullptr : Ab + (int64_t)row * p.lda + src_c[i] * 8;
    }
  };

  const int nk_total = p.K / BK;
  const int nsplit = gridDim.z;
  const int nk_per = (nk_total + nsplit - 1) / nsplit;
  const int kt0 = blockIdx.z * nk_per;
  const int nk = min(nk_total, kt0 + nk_per);  // exclusive end tile of this split
  if (kt0 >= nk) return;                         // empty split (uniform for the whole block)
  u32x4_t ra[LPT], rb[LPTB];

  auto issue_loads = [&](int kt, char* buf) {
    const int k0 = kt * BK;
    const int tap = k0 / k_per_tap;
    if (tap != cur_tap) {
      resolve_a_rows(tap);
      cur_tap = tap;
    }
    const int ka = k0 - tap * k_per_tap;  // column inside the gathered A row
    if constexpr (GLDS) {
      char* la = buf + wave * (64 * 16);
      char* lb = buf + TILE_BYTES + wave * (64 * 16);
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        const bf16_raw* src = a_ptr[i] ? a_ptr[i] + ka : (const bf16_raw*)g_zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(la + i * (NT * 16)), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < LPTB; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[i] + k0),
                                         (__attribute__((address_space(3))) void*)(lb + i * (NT * 16)), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        if (a_ptr[i]) ra[i] = *(const u32x4_t*)(a_ptr[i] + ka);
        else ra[i] = u32x4_t{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int i = 0; i < LPTB; ++i) rb[i] = *(const u32x4_t*)(b_ptr[i] + k0);
    }
  };
  auto commit_loads = [&](char* buf) {
    if constexpr (!GLDS) {
#pragma unroll
      for (int i = 0; i < LPT; ++i) *(u32x4_t*)(buf + lds_off[i]) = ra[i];
#pragma unroll
      for (int i = 0; i < LPTB; ++i) *(u32x4_t*)(buf + TILE_BYTES + lds_off[i]) = rb[i];
    }
  };

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes within a tile), per k-step
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](const char* buf) {
    const char* As = buf;
    const char* Bs = buf + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t af[MI], bfg[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * WM + i * 16 + fr;
        af[i] = *(const bf16x8_t*)(As + r * ROW_BYTES + swz<BK>(r, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wn * WN + j * 16 + fr;
        bfg[j] = *(const bf16x8_t*)(Bs + r * ROW_BYTES + swz<BK>(r, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  char* buf0 = smem;
  char* buf1 = smem + TILE_BYTES + B_BYTES;

  // ---- prologue ----
  issue_loads(kt0, buf0);
  if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  commit_loads(buf0);
  __syncthreads();

  for (int kt = kt0; kt < nk; ++kt) {
    char* cur = ((kt - kt0) & 1) ? buf1 : buf0;
    char* nxt = ((kt - kt0) & 1) ? buf0 : buf1;
    const bool more = kt + 1 < nk;
    if (more) issue_loads(kt + 1, nxt);
    compute(cur);
    if (more) commit_loads(nxt);
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue ----
  gemm_epilogue<MI, NJ>(p, vec_ok, acc, m0 + wm * WM, n0 + wn * WN, fr, fq, b1, b2, nsplit);
}

// What the current grove_gemm_* call was given / asked for (the entry points set it, launch_pp_act reads it: the dispatcher's
// fifteen launch sites stay as they are). Thread-local: the library holds no mutable state shared between threads.
struct gemm_call_ctx {
  const void* image;
  size_t image_bytes;
  void* scratch;
  size_t scratch_bytes;
  int mode;                 // 0 = launch, 1 = plan only (fill *plan), 2 = write the image into host_image
  grove_gemm_plan* plan;
  void* host_image;
  size_t host_image_bytes;
};
static thread_local gemm_call_ctx t_call = {nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0};

template <int BK, bool GLDS, int NJ, int MI>
int launch(const grove_gemm_params& p, int vec_ok, hipStream_t s) {
  if (t_call.mode) return GROVE_OK;  // plan / image requests: the non-persistent kernels need no workspace (sizes stay 0)
  constexpr int BN_ = 32 * NJ;
  constexpr int BM_ = 32 * MI;
  const int tiles_m = (p.M + BM_ - 1) / BM_, tiles_n = (p.N + BN_ - 1) / BN_;
  int split = p.split_k;
  const bool can_split = p.accumulate && p.c_dtype == GROVE_F32 && !p.bias && !p.residual && !p.aux && p.act == GROVE_ACT_NONE;
  if (split == 0) {
    // auto: pure accumulating GEMMs (weight gradients) that cannot fill 256 CUs get their K range split
    split = 1;
    const long tiles = (long)tiles_m * tiles_n * p.batch1 * p.batch2;
    const int nk = p.K / BK;
    if (can_split && tiles < 256 && nk >= 16) {
      split = (int)((768 + tiles - 1) / tiles);
      if (split > nk / 4) split = nk / 4;
      if (split > 128) split = 128;
      if (split < 1) split = 1;
    }
  }
  if (split > 1 && !can_split) {
    grove_set_error("gemm: split_k needs an accumulating f32 C without bias/act/residual/aux");
    return GROVE_E_SHAPE;
  }
  if (grove_det_on()) split = 1;  // deterministic mode: no K ranges meeting in C through atomics (the tile's own read-add-store instead)
  dim3 grid(tiles_m * tiles_n, p.batch1 * p.batch2, split);
  const size_t lds = 2 * (size_t)(BM_ + BN_) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<BK, GLDS, NJ, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BK, GLDS, NJ, MI>), grid, dim3(NT), lds, s, p, vec_ok);
  GROVE_LAUNCH_CHECK();
  return GROVE_OK;
}


// =====================================================================================================
// 256 x 256 x 64 pipelined kernel for the large plain GEMMs (8 waves = 2 (M) x 4 (N), one block per CU, 128 KB of LDS:
// two stages of four 16 KB half-tiles). LDS-DMA loads stay in flight across raw s_barriers under COUNTED vmcnt waits
// (a __syncthreads() would drain the queue) — cdna_hip_programming.md §5 "Pipelining across barriers", T3 + T4 + T5; the
// phase plan and the hazard argument are spelled out at the kernel below.
// =====================================================================================================
constexpr int P_BN = 256, P_BK = 64, P_NT = 512;
constexpr int P_ROWB = P_BK * 2;                 // 128-byte rows, XOR-swizzled like the BK = 64 path above
constexpr int P_HALF = 128 * P_ROWB;             // region of one half-tile (up to 128 rows) = 16 KB
constexpr int P_STAGE = 4 * P_HALF;              // A_lo | B_lo | B_hi | A_hi (staging order)

// Epilogue of the pipelined kernel: lane (fr, fq) holds, for row fragment i and column half g, the 8 consecutive
// columns n = nw0 + 128 g + 8 fq + (0..7) in acc[i][2g][0..3], acc[i][2g+1][0..3] (see the permuted B staging), so every
// global access of the epilogue is 16 bytes per lane. The host only selects this kernel when every epilogue operand
// is 16-byte aligned with row strides that keep it so and N % 8 == 0 (no partial column groups).
// ACT is a compile-time activation (GROVE_ACT_*; -1 = "plain": alpha == 1, no activation, no scale): the persistent
// kernel has no co-resident block to hide epilogue VALU work behind, so the activation must be straight-line code.
// The epilogue of an INTERIOR tile with plain addressing (bf16 C, no c_idx / r_idx / n_group): no masks, no per-row index
// arithmetic. With one wave per SIMD running (the other group waits at a barrier) the epilogue is bound by its own VALU
// instruction count — s_memtime stamps: ~9000 clocks per group with per-row 64-bit multiplies and validity tests, of which
// the 16 stores take under 2000 — so: one 64-bit row pointer per A half, uniform (scalar) row steps, and every residual
// load issued before a store it would otherwise queue behind (vmcnt retires in order):
//   loads(h0) compute(h0) loads(h1) stores(h0) compute(h1) stores(h1)
template <int MIH, int BMH, int ACT>
__device__ __forceinline__ void gemm_epilogue_fast(const grove_gemm_params& p, f32x4_t (&acc)[2 * MIH][4], const int mw0, const int nw0,
                                                   const int fr, const int fq, const float scale) {
  constexpr bool PLAIN = ACT < 0;
  constexpr bool PAIR = ACT == GROVE_ACT_SWIGLU_PAIR;
  const int n0 = nw0 + fq * 8;  // column group 0; group 1 is 128 columns on
  const int64_t row0 = mw0 + fr;
  float bv[2][8];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[g][e] = 0.f;
  if (p.bias) {
    const u32x4_t b0 = *(const u32x4_t*)((const bf16_raw*)p.bias + n0), b1 = *(const u32x4_t*)((const bf16_raw*)p.bias + n0 + 128);
    const float t0[8] = {bf_lo(b0.x), bf_hi(b0.x), bf_lo(b0.y), bf_hi(b0.y), bf_lo(b0.z), bf_hi(b0.z), bf_lo(b0.w), bf_hi(b0.w)};
    const float t1[8] = {bf_lo(b1.x), bf_hi(b1.x), bf_lo(b1.y), bf_hi(b1.y), bf_lo(b1.z), bf_hi(b1.z), bf_lo(b1.w), bf_hi(b1.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[0][e] = t0[e];
      bv[1][e] = t1[e];
    }
  }
  const int64_t ld_aux = p.ld_aux ? p.ld_aux : p.ldc;
  u32x4_t rr[MIH][2], outp[MIH][2];
  auto load_res = [&](const int h) {
    const bf16_raw* rp = (const bf16_raw*)p.residual + (row0 + h * BMH) * p.ldr + n0;
#pragma unroll
    for (int i = 0; i < MIH; ++i) {
      rr[i][0] = *(const u32x4_t*)(rp + (int64_t)(i * 16) * p.ldr);
      rr[i][1] = *(const u32x4_t*)(rp + (int64_t)(i * 16) * p.ldr + 128);
    }
  };
  auto compute = [&](const int h) {
    bf16_raw* ap = p.aux ? (bf16_raw*)p.aux + (row0 + h * BMH) * ld_aux + (PAIR ? (n0 >> 3) * 4 : n0) : nullptr;
#pragma unroll
    for (int i = 0; i < MIH; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4_t a0 = acc[h * MIH + i][2 * g], a1 = acc[h * MIH + i][2 * g + 1];
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (PLAIN) {
            v[e] = a0[e] + bv[g][e];
            v[4 + e] = a1[e] + bv[g][4 + e];
          } else {
            v[e] = a0[e] * p.alpha + bv[g][e];
            v[4 + e] = a1[e] * p.alpha + bv[g][4 + e];
          }
        }
        if constexpr (PAIR) {  // see gemm_epilogue_wide
          const unsigned g01 = pack2bf(v[0], v[1]), g23 = pack2bf(v[2], v[3]), u01 = pack2bf(v[4], v[5]), u23 = pack2bf(v[6], v[7]);
          if (ap) {
            bf16_raw* ax = ap + (int64_t)(i * 16) * ld_aux + g * 64;
            *(u32x2_t*)ax = u32x2_t{g01, g23};
            *(u32x2_t*)(ax + (p.N >> 1)) = u32x2_t{u01, u23};
          }
          const float gg[4] = {bf_lo(g01), bf_hi(g01), bf_lo(g23), bf_hi(g23)};
          const float uu[4] = {bf_lo(u01), bf_hi(u01), bf_lo(u23), bf_hi(u23)};
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = gg[e] * fast_sigmoid(gg[e]) * uu[e];
          outp[i][g] = u32x4_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), 0u, 0u};
        } else {
          if (ap && p.aux_grad) {  // aux = act'(v): computed together with act(v)
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) act_apply_grad(ACT < 0 ? GROVE_ACT_NONE : ACT, v[e], v[e], av[e]);
            *(u32x4_t*)(ap + (int64_t)(i * 16) * ld_aux + g * 128) =
                u32x4_t{pack2bf(av[0], av[1]), pack2bf(av[2], av[3]), pack2bf(av[4], av[5]), pack2bf(av[6], av[7])};
          } else {
            if (ap)
              *(u32x4_t*)(ap + (int64_t)(i * 16) * ld_aux + g * 128) =
                  u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            if (!PLAIN && ACT != GROVE_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = act_apply(ACT, v[e]);
            }
          }
          if (!PLAIN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= scale;
          }
          if (p.residual) {
            const u32x4_t q = rr[i][g];
            const float rv[8] = {bf_lo(q.x), bf_hi(q.x), bf_lo(q.y), bf_hi(q.y), bf_lo(q.z), bf_hi(q.z), bf_lo(q.w), bf_hi(q.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p.residual_mul ? v[e] * rv[e] : v[e] + rv[e];
          }
          outp[i][g] = u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        }
      }
  };
  auto store = [&](const int h) {
    bf16_raw* cp = (bf16_raw*)p.C + (row0 + h * BMH) * p.ldc + (PAIR ? (n0 >> 3) * 4 : n0);
#pragma unroll
    for (int i = 0; i < MIH; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if constexpr (PAIR) *(u32x2_t*)(cp + (int64_t)(i * 16) * p.ldc + g * 64) = u32x2_t{outp[i][g].x, outp[i][g].y};
        else *(u32x4_t*)(cp + (int64_t)(i * 16) * p.ldc + g * 128) = outp[i][g];
      }
  };
  if (p.residual) load_res(0);
  compute(0);
  if (p.residual) load_res(1);
  store(0);
  compute(1);
  store(1);
}

// Three passes, so that no vector load is ever pending while stores are issued (hipcc guards a load's first use with
// s_waitcnt vmcnt(0) whenever control flow makes the count ambiguous; with stores in the queue that wait is a store
// round trip per row — measured 9100 clocks per epilogue, against ~2000 for the stores themselves):
//   rows:   destination / residual rows of my 2 * MIH row fragments (c_idx / r_idx loads back to back);
//   values: per column group, bias + all residual rows in flight together, then the arithmetic IN PLACE in the accumulators
//           (aux, which keeps the pre-activation value, is stored here);
//   stores: conversions and stores only.
template <int MIH, int BMH, int ACT>
__device__ __forceinline__ void gemm_epilogue_wide(const grove_gemm_params& p, f32x4_t (&acc)[2 * MIH][4], const int mw0, const int nw0,
                                                   const int fr, const int fq, const float scale) {
  constexpr bool PLAIN = ACT < 0;
  constexpr bool PAIR = ACT == GROVE_ACT_SWIGLU_PAIR;
  const bf16_raw* __restrict__ bias = (const bf16_raw*)p.bias;
  const bool f32_out = p.c_dtype != GROVE_BF16;
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // the two A halves in turn: MIH row fragments each (register budget)
    int crow[MIH], rrow[MIH];
    {
      int mc[MIH];
#pragma unroll
      for (int i = 0; i < MIH; ++i) {
        crow[i] = mw0 + h * BMH + i * 16 + fr;
        mc[i] = min(crow[i], p.M - 1);
      }
      if (p.c_idx) {
#pragma unroll
        for (int i = 0; i < MIH; ++i) crow[i] = p.c_idx[mc[i]];
      }
#pragma unroll
      for (int i = 0; i < MIH; ++i) {
        if (mw0 + h * BMH + i * 16 + fr >= p.M) crow[i] = -1;
        rrow[i] = crow[i];
      }
      if (p.r_idx) {
        int ri[MIH];
#pragma unroll
        for (int i = 0; i < MIH; ++i) ri[i] = p.r_idx[mc[i]];
#pragma unroll
        for (int i = 0; i < MIH; ++i)
          if (crow[i] >= 0) rrow[i] = ri[i];
      }
    }
    u32x4_t outp[MIH][2];  // bf16 C: the finished rows, packed (SWIGLU_PAIR: .x .y only)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = nw0 + g * 128 + fq * 8;
      const bool n_ok = n < p.N;
      const int nc8 = min(n, p.N - 8);
      const int nd = n + (p.n_group ? n / p.n_group : 0) * p.n_pad;  // padded-head output (n_group / n_pad): destination column
      u32x4_t bb = u32x4_t{0u, 0u, 0u, 0u};
      if (bias) bb = *(const u32x4_t*)(bias + nc8);
      u32x4_t rr[MIH];
      if (p.residual) {
#pragma unroll
        for (int i = 0; i < MIH; ++i) rr[i] = *(const u32x4_t*)((const bf16_raw*)p.residual + (int64_t)max(rrow[i], 0) * p.ldr + nc8);
      }
      const float bv[8] = {bf_lo(bb.x), bf_hi(bb.x), bf_lo(bb.y), bf_hi(bb.y), bf_lo(bb.z), bf_hi(bb.z), bf_lo(bb.w), bf_hi(bb.w)};
#pragma unroll
      for (int i = 0; i < MIH; ++i) {
        const f32x4_t a0 = acc[h * MIH + i][2 * g], a1 = acc[h * MIH + i][2 * g + 1];
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (PLAIN) {
            v[e] = a0[e] + bv[e];
            v[4 + e] = a1[e] + bv[4 + e];
          } else {
            v[e] = a0[e] * p.alpha + bv[e];
            v[4 + e] = a1[e] * p.alpha + bv[4 + e];
          }
        }
        const bool ok = n_ok && crow[i] >= 0;
        if constexpr (PAIR) {
          // v[0..3] = gate, v[4..7] = up of columns (n / 8) * 4 .. + 3 (B rows interleaved by the caller). Rounded to bf16 first:
          // the same values the unfused path stores and reads back, so the product is bit-identical to grove_swiglu_fwd's.
          const unsigned g01 = pack2bf(v[0], v[1]), g23 = pack2bf(v[2], v[3]), u01 = pack2bf(v[4], v[5]), u23 = pack2bf(v[6], v[7]);
          if (p.aux && ok) {
            bf16_raw* ax = (bf16_raw*)p.aux + (int64_t)crow[i] * (p.ld_aux ? p.ld_aux : p.ldc) + (n >> 3) * 4;
            *(u32x2_t*)ax = u32x2_t{g01, g23};
            *(u32x2_t*)(ax + (p.N >> 1)) = u32x2_t{u01, u23};
          }
          const float gg[4] = {bf_lo(g01), bf_hi(g01), bf_lo(g23), bf_hi(g23)};
          const float uu[4] = {bf_lo(u01), bf_hi(u01), bf_lo(u23), bf_hi(u23)};
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = gg[e] * fast_sigmoid(gg[e]) * uu[e];
          outp[i][g] = u32x4_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), 0u, 0u};
          continue;
        }
        if (p.aux && p.aux_grad) {  // aux = act'(v): computed together with act(v)
          float av[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) act_apply_grad(ACT < 0 ? GROVE_ACT_NONE : ACT, v[e], v[e], av[e]);
          if (ok)
            *(u32x4_t*)((bf16_raw*)p.aux + (int64_t)crow[i] * (p.ld_aux ? p.ld_aux : p.ldc) + n) =
                u32x4_t{pack2bf(av[0], av[1]), pack2bf(av[2], av[3]), pack2bf(av[4], av[5]), pack2bf(av[6], av[7])};
        } else {
          if (p.aux && ok)
            *(u32x4_t*)((bf16_raw*)p.aux + (int64_t)crow[i] * (p.ld_aux ? p.ld_aux : p.ldc) + n) =
                u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          if (!PLAIN && ACT != GROVE_ACT_NONE && !PAIR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_apply(ACT, v[e]);
          }
        }
        if (!PLAIN) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= scale;
        }
        if (p.residual) {
          const bool has_r = rrow[i] >= 0;  // a row without a residual row (r_idx < 0) adds nothing
          const u32x4_t q = rr[i];
          const float rv[8] = {bf_lo(q.x), bf_hi(q.x), bf_lo(q.y), bf_hi(q.y), bf_lo(q.z), bf_hi(q.z), bf_lo(q.w), bf_hi(q.w)};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = has_r ? (p.residual_mul ? v[e] * rv[e] : v[e] + rv[e]) : v[e];
        }
        if (f32_out) {  // fp32 C (rare on this path): stored row by row
          float* c = (float*)p.C + (int64_t)crow[i] * p.ldc + nd;
          if (ok) {
            *(f32x4_t*)c = f32x4_t{v[0], v[1], v[2], v[3]};
            *(f32x4_t*)(c + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
          }
        } else {
          outp[i][g] = u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        }
      }
    }
    if (f32_out) continue;
    // stores only
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = nw0 + g * 128 + fq * 8;
      const bool n_ok = n < p.N;
      const int nd = PAIR ? (n >> 3) * 4 : n + (p.n_group ? n / p.n_group : 0) * p.n_pad;
#pragma unroll
      for (int i = 0; i < MIH; ++i) {
        bf16_raw* c = (bf16_raw*)p.C + (int64_t)crow[i] * p.ldc + nd;
        if (n_ok && crow[i] >= 0) {
          if constexpr (PAIR) *(u32x2_t*)c = u32x2_t{outp[i][g].x, outp[i][g].y};
          else *(u32x4_t*)c = outp[i][g];
        }
      }
    }
    if (p.n_group) {  // chunks that close a head group zero-fill the pad columns behind it (bf16 C only)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int n = nw0 + g * 128 + fq * 8;
        const int ngrp = n / p.n_group;
        if (n >= p.N || n - ngrp * p.n_group + 8 != p.n_group) continue;
#pragma unroll
        for (int i = 0; i < MIH; ++i) {
          if (crow[i] < 0) continue;
          bf16_raw* c = (bf16_raw*)p.C + (int64_t)crow[i] * p.ldc + n + ngrp * p.n_pad + 8;
          for (int z = 0; z < p.n_pad; z += 8) *(u32x4_t*)(c + z) = u32x4_t{0u, 0u, 0u, 0u};
        }
      }
    }
  }
}

// GROVE_ACT_SWIGLU_BWD (round 4): the backward of LlamaMLP's silu(gate) * up in the epilogue of the down-projection's dgrad GEMM. The
// accumulators are d a (rounded to bf16 first: the value the unfused path stores and reads back, so the outputs are grove_swiglu_bwd's
// bit for bit); gate | up = the saved pre-activations (p.residual: [M, >= 2N], up at column N + n); C [M, >= 2N] gets d gate at column n
// and d up at N + n. One function for interior and edge tiles (rows clamped for the loads, stores predicated; N % 8 == 0). Per A half: all
// four 16-byte loads of a row fragment pair are issued first (they queue behind the previous half's stores once, not once per store),
// then every (row fragment, column group) is computed and stored straight away — 2 loads + 2 stores per 8 outputs.
template <int MIH, int BMH>
__device__ __forceinline__ void gemm_epilogue_swiglu_bwd(const grove_gemm_params& p, f32x4_t (&acc)[2 * MIH][4], const int mw0, const int nw0,
                                                         const int fr, const int fq) {
  const bf16_raw* __restrict__ gu = (const bf16_raw*)p.residual;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u32x4_t gq[MIH][2], uq[MIH][2];
    const int row0 = mw0 + h * BMH + fr;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int nc8 = min(nw0 + g * 128 + fq * 8, p.N - 8);
#pragma unroll
      for (int i = 0; i < MIH; ++i) {
        const bf16_raw* rp = gu + (int64_t)min(row0 + i * 16, p.M - 1) * p.ldr + nc8;
        gq[i][g] = *(const u32x4_t*)rp;
        uq[i][g] = *(const u32x4_t*)(rp + p.N);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = nw0 + g * 128 + fq * 8;
#pragma unroll
      for (int i = 0; i < MIH; ++i) {
        const f32x4_t a0 = acc[h * MIH + i][2 * g], a1 = acc[h * MIH + i][2 * g + 1];
        const unsigned d01 = pack2bf(a0[0] * p.alpha, a0[1] * p.alpha), d23 = pack2bf(a0[2] * p.alpha, a0[3] * p.alpha);
        const unsigned d45 = pack2bf(a1[0] * p.alpha, a1[1] * p.alpha), d67 = pack2bf(a1[2] * p.alpha, a1[3] * p.alpha);
        const float dv[8] = {bf_lo(d01), bf_hi(d01), bf_lo(d23), bf_hi(d23), bf_lo(d45), bf_hi(d45), bf_lo(d67), bf_hi(d67)};
        const u32x4_t gp = gq[i][g], up = uq[i][g];
        const float gv[8] = {bf_lo(gp.x), bf_hi(gp.x), bf_lo(gp.y), bf_hi(gp.y), bf_lo(gp.z), bf_hi(gp.z), bf_lo(gp.w), bf_hi(gp.w)};
        const float uv[8] = {bf_lo(up.x), bf_hi(up.x), bf_lo(up.y), bf_hi(up.y), bf_lo(up.z), bf_hi(up.z), bf_lo(up.w), bf_hi(up.w)};
        float dg[8], du[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) swiglu_bwd_elem(dv[e], gv[e], uv[e], dg[e], du[e]);
        if (n < p.N && row0 + i * 16 < p.M) {
          bf16_raw* c = (bf16_raw*)p.C + (int64_t)(row0 + i * 16) * p.ldc + n;
          *(u32x4_t*)c = u32x4_t{pack2bf(dg[0], dg[1]), pack2bf(dg[2], dg[3]), pack2bf(dg[4], dg[5]), pack2bf(dg[6], dg[7])};
          *(u32x4_t*)(c + p.N) = u32x4_t{pack2bf(du[0], du[1]), pack2bf(du[2], du[3]), pack2bf(du[4], du[5]), pack2bf(du[6], du[7])};
        }
      }
    }
  }
}

#ifndef PP_SCHED
#define PP_SCHED 1
#endif
#ifndef PP_SCHED2_EXTRA
#define PP_SCHED2_EXTRA false   // (A/B builds: an expression over BM / GATHER / ACT that puts more instances on schedule 2)
#endif
__device__ __forceinline__ void wait_vm_loads(int n) {  // n (even) = vector-memory operations allowed to stay in flight
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the kernel -------------------------------------------------------------------------------------------------------
// PERSISTENT: one block per CU walks its output tiles (tile L = wgid + k * gridDim, XCD-aware wgid, band-major tile
// order) and the staging stream runs straight through the tile seams, so a tile's first operands land while the
// previous tile's epilogue runs.
//
// The four half-tiles of a K tile (A_lo, B_lo, B_hi, A_hi: BM/2 or 128 rows x 64 k) are staged, waited for and freed one
// at a time. A wave owns BM/4 rows of each A half and 32 columns of each B half, so the quadrant order
//   ph1 A_lo x B_lo | ph2 A_lo x B_hi | ph3 A_hi x B_hi | ph4 A_hi x B_lo      (4 MIH MFMAs each; 12 / 4 / 8 / 0 LDS reads)
// needs the half-tiles exactly in staging order and releases each one phase after its last read. A phase is
//   { ds_reads of this phase's operands; LDS-DMA of half-tile q + 6; counted vmcnt } barrier { MFMAs } barrier
// and the wr = 1 waves run one barrier behind the wr = 0 waves, so each SIMD always has one wave in its MFMA segment
// and the other in its memory segment. Hazards (q = phase number, h = half-tile number of the block's stream, h issued at q = h - 6):
//   WAR  the region of h held half-tile h - 8, whose last ds_read is at least two phases (four barriers) before the re-issue,
//        retired by the lgkmcnt the compiler places before the MFMAs that consume it;
//   RAW  phase q + 1 reads half-tiles <= q + 2; every wave retires them with a counted vmcnt (the loads of the half-tiles
//        issued after q + 2 may stay in flight) before the first barrier of phase q, the lagging group included, and the
//        reader passes one more barrier before its ds_reads. vmcnt retires in issue order, stores included: the epilogue's
//        stores sit between loads in the queue, so a counted wait after an epilogue is merely conservative (it also waits
//        for the stores), never early.
// 8 consecutive int32 through the scalar cache (wave-uniform address). The gather indices of the GATHER instance are read
// this way on purpose: a VECTOR load inside the persistent loop would sit in the vmcnt queue among the LDS-DMA loads and
// make hipcc's own waits drain the staging stream.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8_t sload8(const int* p) {
  i32x8_t v;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t sload4(const void* p) {  // 4 consecutive int32, same reasoning (p is wave-uniform by construction)
  i32x4_t v;
  const uint64_t a = (uint64_t)p;
  const uint64_t u = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                     ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
  asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"((const void*)u) : "memory");
  return v;
}

// GATHER: A rows are looked up per (tap, m) in p.a_idx (implicit-GEMM Conv3d: K = taps x C_in; window (un)partition: one tap),
// -1 = a zero row.
// THE WORK LIST. The host writes, once per shape, what every block does (pp_table below); the kernel only walks its column:
//   row 0        header  {NT = K tiles of the block's stream, number of segments, 0, 0}
//   row 1 + i    segment {m0, n0, k0 | k1 << 16, part}: K tiles [k0, k1) of the output tile at (m0, n0)
// one 16-byte scalar load per segment, no index arithmetic (and none of its scalar registers) in the kernel.
//   part 0       a whole tile (k0 = 0, k1 = nk): accumulate, epilogue. The data-parallel rounds: tile L = wgid + i * G, band-major.
//   part 1 + s   STREAM-K: a K range of a tile of the last, partial round; the raw fp32 accumulators go to workspace slot s
//                and gemm_pp_fixup_kernel (next launch on the stream) sums a tile's slots in K order and runs the epilogue.
// Stream-K tail (plan_stream_k): tiles = rounds * G + tail with tail <= G / 2 — the tiles of the partial round are cut into 2..4
// equal K ranges (up to 8 when the partial round is the only one), one block each, so that (nearly) every CU works through the last round instead of `tail` CUs for a whole
// tile. The order of the sum is fixed by the list, so results are deterministic.
struct pp_work {
  const i32x4_t* table;
  float* ws;  // slots of 8 waves x 32 accumulators x 64 lanes x 16 B (lane-linear: the fix-up's same wave reads them with the same map)
  const float* row_scale;  // FP8 instances: de-quantisation scales of the activation rows [M] and of the weight rows [N]
  const float* col_scale;
};
constexpr size_t P_SLOT = (size_t)8 * 32 * 64 * 16;

// FP8 instances: the accumulators hold sums of products of e4m3 CODES; times row_scale[m] * col_scale[n] they are the product
// the epilogue expects (same map as the epilogues: row fragment i of A half h, column group g, 8 consecutive columns).
template <int MIH, int BMH>
__device__ __forceinline__ void fp8_scale_acc(const grove_gemm_params& p, const pp_work& work, f32x4_t (&acc)[2 * MIH][4], const int mw0,
                                              const int nw0, const int fr, const int fq) {
  float cs[2][8];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int n = min(nw0 + g * 128 + fq * 8, p.N - 8);  // (N % 8 == 0: a column group is whole or absent)
    const f32x4_t c0 = *(const f32x4_t*)(work.col_scale + n), c1 = *(const f32x4_t*)(work.col_scale + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[g][e] = c0[e], cs[g][4 + e] = c1[e];
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < MIH; ++i) {
      const float rs = work.row_scale[min(mw0 + h * BMH + i * 16 + fr, p.M - 1)];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[h * MIH + i][2 * g][e] *= rs * cs[g][e];
          acc[h * MIH + i][2 * g + 1][e] *= rs * cs[g][4 + e];
        }
    }
}

// ACT: the epilogue compiled in (-1 = plain: act NONE, alpha 1, no scale) — one per kernel: with all of them in one kernel the
// register allocator spills inside the K loop.
// FP8: A and B hold e4m3 codes; the kernel moves the same bytes (the host passes K, lda, ldb in 2-byte units: a 128-byte LDS row is
// 128 codes instead of 64 bf16) and a phase's MFMAs are v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales — one per
// (row fragment, column fragment) on the SAME two 16-byte reads per operand that feed the two bf16 k-steps: a lane's 32 bytes are
// chunks fq and fq + 4 of the row for A and B alike, and which k a byte is does not matter as long as both operands agree.
// GROUPED (round 6; grove_gemm_params.b_group_rows — the Winograd form of the Conv3d adapters, 64 transform points in one launch):
// the rows of A are groups of b_group_rows rows (whole tiles) and group g multiplies its own B matrix, the g-th [N, ldb] block of B.
template <int BM, bool GATHER, int ACT, bool FP8 = false, bool GROUPED = false>
__global__ __launch_bounds__(P_NT) void gemm_nt_pp_kernel(const grove_gemm_params p, const pp_work work) {
  constexpr int BMH = BM / 2;    // rows of an A half-tile: 128 or 96
  constexpr int WRH = BMH / 2;   // ... of which one wave group owns 64 or 48
  constexpr int MIH = WRH / 16;  // row fragments per half per wave: 4 or 3
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const bf16_raw* __restrict__ A = (const bf16_raw*)p.A;
  const bf16_raw* __restrict__ B = (const bf16_raw*)p.B;

  // my column of the work list: blocks of one XCD (blockIdx % 8) take neighbouring tiles of every round
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const i32x4_t* my_work = work.table + wgid;
  const i32x4_t head = sload4(my_work);
  const int NT = head[0];    // K tiles of my stream
  const int nseg = head[1];  // segments of my stream
  const int NH = 4 * NT;     // half-tiles of my stream
  if (NT == 0) return;       // (a block past the end of a stream-K tail)

  // staging: regions of a stage in staging order 0 = A_lo, 1 = B_lo, 2 = B_hi, 3 = A_hi. A half-tile region is 128 rows x 8
  // chunks = 2 LDS-DMA instructions per thread. (BM = 192 uses 96 rows of an A region; every wave still issues both
  // instructions — rows 96..127 are fetched and never read — so that the vmcnt arithmetic is the same for all waves.)
  const int st_c = tid & 7, st_r = tid >> 3;
  const bf16_raw* src[4][2];
  bool a_zero[2][2] = {{false, false}, {false, false}};
  const int k_per_tap = GATHER ? p.K / p.a_taps : p.K;
  const int kt_per_tap = k_per_tap / P_BK;
  int is_m0 = 0, is_tap = 0;  // GATHER: origin row and tap of the A pointers currently loaded
  // K map (padded-head A): this lane's logical chunk offset inside a K tile is the same for both of its rows (r and r + 64 share
  // r & 7 ... the swizzle term (r >> 1) & 7 is the same for r and r + 64), so one correction serves both loads
  const unsigned a_lc8 = swz<64>(st_r, st_c) * 8;
  const unsigned k_magic = p.k_group ? ((1u << 24) + p.k_group - 1) / p.k_group : 0;
  // GATHER: my two rows of each A half for tap `tap` of the tile at row m0. The wave's 8 rows per LDS-DMA instruction are 8
  // consecutive m: one 8-dword scalar load, then a per-lane pick. Positions past the end of the index array belong to
  // rows >= M (never stored): the load window is shifted back inside the array and those lanes take any entry.
  auto set_src_a = [&](int m0, int tap) {
    const int total = p.a_taps * p.M;
    const int j = lane >> 3;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int P = tap * p.M + m0 + h * BMH + 64 * i + wave * 8;
        const int Pc = max(min(P, total - 8), 0);
        const i32x8_t v = sload8(p.a_idx + Pc);
        const int jj = min(j + (P - Pc), 7);
        int idx = v[0];
        idx = jj == 1 ? v[1] : idx; idx = jj == 2 ? v[2] : idx; idx = jj == 3 ? v[3] : idx; idx = jj == 4 ? v[4] : idx;
        idx = jj == 5 ? v[5] : idx; idx = jj == 6 ? v[6] : idx; idx = jj == 7 ? v[7] : idx;
        const int r = st_r + 64 * i;
        src[h ? 3 : 0][i] = idx >= 0 ? A + (int64_t)idx * p.lda + swz<64>(r, st_c) * 8 : (const bf16_raw*)g_zero_page;
        a_zero[h][i] = idx < 0;
      }
    is_m0 = m0;
    is_tap = tap;
  };
  auto set_src = [&](int m0, int n0, int k0) {
    if constexpr (GATHER) set_src_a(m0, k0 ? __builtin_amdgcn_readfirstlane(k0 / kt_per_tap) : 0);  // (a stream-K part starts inside the K range)
    // (the un-grouped instances keep their round-5 source text below, token for token: routing their B through a variable that merely
    // equals B moved hipcc's register allocation in the gathered plain instance and put a vmcnt(0) into its K loop, +7 % on the 21
    // window-block launches — tests/test_isa_host.py screens for exactly that)
    const bf16_raw* Bg = nullptr;
    if constexpr (GROUPED) Bg = B + (int64_t)(m0 / p.b_group_rows) * p.N * p.ldb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = st_r + 64 * i;
      const int c = swz<64>(r, st_c) * 8;
      if constexpr (!GATHER) {
        src[0][i] = A + (int64_t)min(m0 + r, p.M - 1) * p.lda + c;
        src[3][i] = A + (int64_t)min(m0 + BMH + r, p.M - 1) * p.lda + c;
      }
      // LDS row R of a B half holds column pn(R): fragment j, operand row r of a wave's 32-column group lands on column
      // 8 * (r >> 2) + 4 * j + (r & 3), so a lane's accumulators for (j = 0, 1) are 8 consecutive columns -> 16-byte stores
      const int pn = (r & ~31) | (((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3));
      if constexpr (GROUPED) {
        src[1][i] = Bg + (int64_t)min(n0 + pn, p.N - 1) * p.ldb + c;
        src[2][i] = Bg + (int64_t)min(n0 + 128 + pn, p.N - 1) * p.ldb + c;
      } else {
        src[1][i] = B + (int64_t)min(n0 + pn, p.N - 1) * p.ldb + c;
        src[2][i] = B + (int64_t)min(n0 + 128 + pn, p.N - 1) * p.ldb + c;
      }
    }
  };
  int is_seg = 0, is_k, is_kend;  // segment and K tile of the half-tile being issued
  {
    const i32x4_t e = sload4(my_work + G);
    is_k = e[2] & 0xffff, is_kend = (unsigned)e[2] >> 16;
    set_src(e[0], e[1], is_k);
  }
  auto issue = [&](int x, int stream_t) {
    char* dst = smem + (stream_t & 1) * P_STAGE + x * P_HALF + wave * (64 * 16);
    int64_t koff0 = (int64_t)is_k * P_BK, koff1 = koff0;
    if (GATHER && (x == 0 || x == 3)) {  // A: the K offset is local to the tap; a zero row stays on the zero page
      int64_t ka = (int64_t)(is_k - is_tap * kt_per_tap) * P_BK;
      if (p.k_group) {  // padded-head A: my chunk's logical k -> its column (k_magic = ceil(2^24 / k_group): exact for k < 2^16, k % 8 == 0)
        const unsigned kl = (unsigned)ka + a_lc8;
        ka = (int64_t)(kl + (unsigned)(((uint64_t)kl * k_magic) >> 24) * p.k_pad) - a_lc8;
      }
      koff0 = a_zero[x == 3][0] ? 0 : ka;
      koff1 = a_zero[x == 3][1] ? 0 : ka;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[x][0] + koff0),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[x][1] + koff1),
                                     (__attribute__((address_space(3))) void*)(dst + P_NT * 16), 16, 0, 0);
  };
  auto advance_issue = [&]() {  // before the A_lo of every K tile but the first
    if (++is_k == is_kend) {
      ++is_seg;
      const i32x4_t e = sload4(my_work + (is_seg + 1) * G);
      is_k = e[2] & 0xffff, is_kend = (unsigned)e[2] >> 16;
      set_src(e[0], e[1], is_k);
    } else if (GATHER && is_k == (is_tap + 1) * kt_per_tap) {
      set_src_a(is_m0, is_tap + 1);  // next tap of the same tile: new A rows
    }
  };

  f32x4_t acc[2 * MIH][4];
#pragma unroll
  for (int i = 0; i < 2 * MIH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a region (the swizzle term depends on fr only: the row bases are multiples of 16)
  const int sw = (fr >> 1) & 7;
  const int a_off = (wr * WRH + fr) * P_ROWB;
  const int b_off = (wc * 32 + fr) * P_ROWB;
  const int kc0 = ((0 + fq) ^ sw) * 16, kc1 = ((4 + fq) ^ sw) * 16;

  bf16x8_t af[MIH][2], b0[2][2], b1[2][2];
  auto read_a = [&](const char* st, int x) {
#pragma unroll
    for (int i = 0; i < MIH; ++i) {
      af[i][0] = *(const bf16x8_t*)(st + x * P_HALF + a_off + i * 16 * P_ROWB + kc0);
      af[i][1] = *(const bf16x8_t*)(st + x * P_HALF + a_off + i * 16 * P_ROWB + kc1);
    }
  };
  auto read_b = [&](const char* st, int x, bf16x8_t (&bb)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bb[j][0] = *(const bf16x8_t*)(st + x * P_HALF + b_off + j * 16 * P_ROWB + kc0);
      bb[j][1] = *(const bf16x8_t*)(st + x * P_HALF + b_off + j * 16 * P_ROWB + kc1);
    }
  };
  auto cat8 = [](const bf16x8_t lo, const bf16x8_t hi) __attribute__((always_inline)) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, lo), b = __builtin_bit_cast(u32x4_t, hi);
    return i32x8_t{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
  };
#define PP_MMA(IO, JO, BB)                                                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                                                      \
  __builtin_amdgcn_s_setprio(1);                                                                                          \
  if constexpr (FP8) {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < MIH; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                         \
        acc[IO + i][JO + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(BB[j][0], BB[j][1]), cat8(af[i][0], af[i][1]), \
                                                                               FIRST ? zero4 : acc[IO + i][JO + j], 0, 0, 0, 127, 0, 127); \
  } else {                                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < MIH; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        acc[IO + i][JO + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BB[j][ks], af[i][ks], (FIRST && ks == 0) ? zero4 : acc[IO + i][JO + j], 0, 0, 0); \
  }                                                                                                                       \
  __builtin_amdgcn_s_setprio(0);                                                                                          \
  __builtin_amdgcn_sched_barrier(0);
  // end of a phase's memory segment: stage half-tile q + 6, retire what phase q + 1 reads, meet the other group.
  // STEADY (every K tile but the stream's last two): the half-tile exists, no tests — the memory segment has to fit
  // beside the other wave group's MFMA segment (192-256 cycles), scalar branches included.
#define PP_MEM_END(STEADY, Q, X, TOFF, WAIT)                                                                              \
  if (STEADY || (Q) + 6 < NH) {                                                                                           \
    if ((X) == 0) advance_issue();                                                                                        \
    issue(X, T + TOFF);                                                                                                   \
    if (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + RELAX) : "memory"); /* half-tiles q+3 .. q+6 (and RELAX stores) stay in flight */ \
  } else if (WAIT) {                                                                                                      \
    wait_vm_loads(2 * max(NH - 3 - (Q), 0)); /* the stream's tail: half-tiles q+3 .. NH-1 */                              \
  }                                                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                      \
  __builtin_amdgcn_s_barrier();

  // The epilogue scale is fetched and USED here, once: a global load left pending on any path into the K loop makes hipcc
  // guard the loop's first ds_read with s_waitcnt vmcnt(0), which drains the staging queue every K tile.
  float scale = 1.f;
  if (p.scale_ptr) {
    scale = *p.scale_ptr;
    if (p.scale_tanh) scale = tanhf(scale);
  }
  asm volatile("" ::"v"(scale));
  // prologue: half-tiles 0..5 of my stream, the first two landed before anyone reads
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  if (NT > 1) {
    advance_issue();
    issue(0, 1); issue(1, 1);
  }
  wait_vm_loads(2 * (min(5, NH - 1) - 1));
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the stagger

  // RELAX: the first K tile after an epilogue that left exactly RELAX stores per wave in the queue (the interior-tile epilogue
  // without aux: 4 * MIH, every lane, no branches). vmcnt retires in issue order, so for these four phases the stores sit
  // between the half-tiles already issued and the ones issued now: allowing 8 + RELAX operations in flight retires the same
  // loads as vmcnt(8) does elsewhere and lets the stores drain behind the MFMAs instead of stalling the first phase (~1500
  // clocks per output tile). From the next K tile on, vmcnt(8) covers only loads issued after the stores.
  // FIRST: a segment's first K tile starts its accumulators from the zero operand of the MFMA (no 128-register clear per tile)
  // schedule 2 ships in the two instances that only ever run the Conv3d adapters' 27-tap launches (K = 34560: -2 ... -3 %); -DPP_SCHED=2
  // puts every instance on it (the A/B build)
  // and in the plain 192-row instance (LLaMA's 2812-row launches: down 193 -> 189 us, o_proj 80 -> 78, qkv dgrad 235 -> 230, the others
  // and CLIP's within +-1 %: tools/dev/pp_shapes_ab.py)
  constexpr bool SCHED2 = PP_SCHED == 2 || (!FP8 && GATHER && BM == 256 && (ACT == GROVE_ACT_NONE || ACT == GROVE_ACT_RELU)) ||
                          (!FP8 && !GATHER && BM == 192 && ACT < 0) ||
                          (!FP8 && !GATHER && (ACT == GROVE_ACT_SWIGLU_PAIR || ACT == GROVE_ACT_SWIGLU_BWD)) ||  // LLaMA's MLP: 431-439 -> 422-425 us, 236-242 -> 234
                          PP_SCHED2_EXTRA;
  auto k_tile = [&](auto steady, auto relax_stores, auto first, int T) {
    constexpr bool STEADY = decltype(steady)::value;
    constexpr int RELAX = decltype(relax_stores)::value;
    constexpr bool FIRST = decltype(first)::value && !FP8;  // (the FP8 instances keep the explicit clear: with the zero operand hipcc spills 4x more there)
    const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* st = smem + (T & 1) * P_STAGE;
    const int q = 4 * T;
    if constexpr (SCHED2) {
    // Staging schedule 2 (round 5 experiment, -DPP_SCHED=2; correct: the GEMM tests, the guard screen and the ABI tests pass on it;
    // measured same-box against schedule 1, tools/dev/pp_drain_ab.py product _s2: long-K launches 2-3 % faster — the gathered Conv3d
    // forward / dgrad 1993 -> 1940 / 1973 -> 1917 us, LLaMA o_proj 85.5 -> 83.1 — the K <= 5120 SAM launches 1-2.5 % SLOWER — fc2 +
    // residual 341.5 -> 349.6, fc2 dgrad 336.7 -> 346.3, fc1 + GELU + aux 466 -> 471-484: a wash over the step, so schedule 1 ships everywhere but in the two adapter-only instances):
    // every piece one phase later, phase 4 issues two half-tiles — the pieces
    // fall 0 / 2 / 2 / 4 on the phases whose fragment reads fall 12 / 4 / 8 / 0 (2 / 2 / 2 / 2 in schedule 1). Half-tile q + 6 is
    // issued in phase 2, q + 7 in phase 3, q + 8 and q + 9 in phase 4; phase 1 retires q + 2 (younger: q + 3 .. q + 5), phase 2
    // retires q + 3 (younger: q + 4 .. q + 6), phase 4 retires q + 5 (younger: q + 6 .. q + 9). Slots are refilled three phases or
    // more after their last read. The prologue (half-tiles 0 .. 5) is the same.
    auto tail_wait = [&](int needed, int last) { wait_vm_loads(2 * max(min(last, NH - 1) - needed, 0)); };
    // ph1: 12 reads, no piece
    read_b(st, 1, b0);
    read_a(st, 0);
    if (STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + RELAX) : "memory");
    else tail_wait(q + 2, q + 5);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    PP_MMA(0, 0, b0)
    __builtin_amdgcn_s_barrier();
    // ph2: 4 reads, B_hi of the next K tile
    read_b(st, 2, b1);
    if (STEADY || q + 6 < NH) issue(2, T + 1);
    if (STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + RELAX) : "memory");
    else tail_wait(q + 3, q + 6);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    PP_MMA(0, 2, b1)
    __builtin_amdgcn_s_barrier();
    // ph3: 8 reads, A_hi of the next K tile; nothing to retire (phase 4 reads nothing)
    read_a(st, 3);
    if (STEADY || q + 7 < NH) issue(3, T + 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    PP_MMA(MIH, 2, b1)
    __builtin_amdgcn_s_barrier();
    // ph4: no reads, A_lo and B_lo of the K tile after the next
    if (STEADY || q + 8 < NH) {
      advance_issue();
      issue(0, T + 2);
    }
    if (STEADY || q + 9 < NH) issue(1, T + 2);
    if (STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + RELAX) : "memory");
    else tail_wait(q + 5, q + 9);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    PP_MMA(MIH, 0, b0)
    __builtin_amdgcn_s_barrier();
    } else {
    // ph1
    read_b(st, 1, b0);
    read_a(st, 0);
    PP_MEM_END(STEADY, q, 2, 1, true)
    PP_MMA(0, 0, b0)
    __builtin_amdgcn_s_barrier();
    // ph2
    read_b(st, 2, b1);
    PP_MEM_END(STEADY, q + 1, 3, 1, true)
    PP_MMA(0, 2, b1)
    __builtin_amdgcn_s_barrier();
    // ph3
    read_a(st, 3);
    PP_MEM_END(STEADY, q + 2, 0, 2, false)
    PP_MMA(MIH, 2, b1)
    __builtin_amdgcn_s_barrier();
    // ph4
    PP_MEM_END(STEADY, q + 3, 1, 2, true)
    PP_MMA(MIH, 0, b0)
    __builtin_amdgcn_s_barrier();
    }
  };
  const bool fast_addr = p.c_dtype == GROVE_BF16 && !p.c_idx && !p.r_idx && !p.n_group;  // gemm_epilogue_fast's case
  // my output tiles in turn: their K tiles (STEADY while the stream still has a half-tile to issue six phases ahead, i.e. all
  // but the stream's last two), then the epilogue (no barriers inside) — the next tile's operands are already landing
  int T = 0;
  bool relax = false;
  for (int sg = 0; sg < nseg; ++sg) {
    const i32x4_t e = sload4(my_work + (sg + 1) * G);
    const int m0 = e[0], n0 = e[1], nks = (int)((unsigned)e[2] >> 16) - (e[2] & 0xffff), part = e[3];
    const int ns = min(max(NT - 2 - T, 0), nks);
    int k = 0;
    using no_relax = std::integral_constant<int, 0>;
    if constexpr (!FP8) {  // (the FP8 instances clear explicitly below and never relax: two K-tile bodies instead of five)
      if (relax && ns > 0) k_tile(std::true_type{}, std::integral_constant<int, 4 * MIH>{}, std::true_type{}, T);
      else if (ns > 0) k_tile(std::true_type{}, no_relax{}, std::true_type{}, T);
      else k_tile(std::false_type{}, no_relax{}, std::true_type{}, T);
      ++k, ++T;
    }
    for (; k < ns; ++k, ++T) k_tile(std::true_type{}, no_relax{}, std::false_type{}, T);
    for (; k < nks; ++k, ++T) k_tile(std::false_type{}, no_relax{}, std::false_type{}, T);
    // Both groups run their epilogues in the same barrier interval: the leading group waits one barrier here (the lagging
    // group is in its last MFMA segment), the lagging group waits one after its epilogue, which restores the one-barrier lag.
    // An epilogue is bound by the issue latency of its own VALU / store stream, so two waves per SIMD take little longer
    // than one, where the groups one after the other took twice as long. No LDS access and no staging in the interval.
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if (__builtin_expect(part != 0, 0)) {
      // stream-K part: raw accumulators to my slot — wave-uniform base + lane offset + immediate, one address register for all
      // the stores; the empty asm pins that arithmetic here (hoisted out of the segment loop it would hold registers across
      // the K loops, and so would this branch without the "unlikely"). The stores are not counted for RELAX: the next K tile's
      // vmcnt(8) simply waits for them too.
      char* wb = (char*)work.ws + ((size_t)(part - 1) * 8 + wave) * 32768;
      int lo = lane * 16;
      asm volatile("" : "+s"(wb), "+v"(lo));
#pragma unroll
      for (int i = 0; i < 2 * MIH; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4_t*)(wb + (i * 4 + j) * 1024 + lo) = acc[i][j];
        __builtin_amdgcn_sched_barrier(0);
      }
      relax = false;
    } else {
      const int mw0 = m0 + wr * WRH, nw0 = n0 + wc * 32;
      const bool interior = fast_addr && m0 + BM <= p.M && n0 + P_BN <= p.N;
      if constexpr (FP8) fp8_scale_acc<MIH, BMH>(p, work, acc, mw0, nw0, fr, fq);
      if constexpr (ACT == GROVE_ACT_SWIGLU_BWD) gemm_epilogue_swiglu_bwd<MIH, BMH>(p, acc, mw0, nw0, fr, fq);
      else if (interior) gemm_epilogue_fast<MIH, BMH, ACT>(p, acc, mw0, nw0, fr, fq, ACT < 0 ? 1.f : scale);
      else gemm_epilogue_wide<MIH, BMH, ACT>(p, acc, mw0, nw0, fr, fq, ACT < 0 ? 1.f : scale);
      relax = interior && !p.aux && !FP8 && ACT != GROVE_ACT_SWIGLU_BWD;  // exactly 4 * MIH stores per wave were issued (FP8: the scale loads sit in the queue too)
      // Round 5: in the 256-row GELU instance (SAM's fc1: 32 launches, 15 ms per step) hipcc's register allocation leaves epilogue
      // vector-memory operations pending on the back edge into the K loop and guards the loop's first fragment read with
      // `s_waitcnt vmcnt(0)` — at the LOOP HEADER, so the staging queue drained once per K tile, 20 times per output tile
      // (tools/isa_loop_waits.py finds such waits; tools/isa_kloop.py counted 8 instead of the 4 explicit ones). A wait the compiler
      // can SEE, once per output tile, clears its scoreboard: one store round trip per tile (what RELAX otherwise hides) against
      // twenty drains — 507 -> 474 us for (32768, 5120, 1280) + GELU + aux, same box (tools/dev/pp_drain_ab.py). The other instances
      // are clean or (QuickGELU at K = 1024: +2.6 %) lose more to the round trip than they gain: they keep RELAX.
#ifndef PP_NO_EPI_DRAIN  // (-DPP_NO_EPI_DRAIN: the A/B arm of tools/dev/step_ab.sh)
      if constexpr (!FP8 && BM == 256 && !GATHER && ACT == GROVE_ACT_GELU) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15)
        relax = false;
      }
#endif
    }
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 2 * MIH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef PP_MMA
#undef PP_MEM_END
}

// Stream-K fix-up: split tile {m0, n0, first slot, parts}; wave w sums what wave w of the parts left and runs the same epilogue
// the main kernel would have run (same accumulator map, same functions). The waves are independent: FIX_WAVES per block,
// 8 / FIX_WAVES blocks per tile, so that the few split tiles still spread over the whole chip.
constexpr int FIX_WAVES = 2;
template <int BM, int ACT, bool FP8 = false>
__global__ __launch_bounds__(64 * FIX_WAVES) void gemm_pp_fixup_kernel(const grove_gemm_params p, const i32x4_t* __restrict__ list, const float* __restrict__ ws,
                                                                       const pp_work work) {
  constexpr int BMH = BM / 2, WRH = BMH / 2, MIH = WRH / 16;
  constexpr int BPT = 8 / FIX_WAVES;  // blocks per tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % BPT) * FIX_WAVES + (tid >> 6));
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const i32x4_t e = list[blockIdx.x / BPT];
  const int m0 = e[0], n0 = e[1], s0 = e[2], np = e[3];
  float scale = 1.f;
  if (p.scale_ptr) {
    scale = *p.scale_ptr;
    if (p.scale_tanh) scale = tanhf(scale);
  }
  f32x4_t acc[2 * MIH][4];
  {
    const char* wb = (const char*)ws + ((size_t)s0 * 8 + wave) * 32768 + lane * 16;
#pragma unroll
    for (int i = 0; i < 2 * MIH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = *(const f32x4_t*)(wb + (i * 4 + j) * 1024);
  }
  for (int q = 1; q < np; ++q) {
    const char* wb = (const char*)ws + ((size_t)(s0 + q) * 8 + wave) * 32768 + lane * 16;
#pragma unroll
    for (int i = 0; i < 2 * MIH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += *(const f32x4_t*)(wb + (i * 4 + j) * 1024);
  }
  const bool fast_addr = p.c_dtype == GROVE_BF16 && !p.c_idx && !p.r_idx && !p.n_group;
  const int mw0 = m0 + wr * WRH, nw0 = n0 + wc * 32;
  const bool interior = fast_addr && m0 + BM <= p.M && n0 + P_BN <= p.N;
  if constexpr (FP8) fp8_scale_acc<MIH, BMH>(p, work, acc, mw0, nw0, fr, fq);
  if constexpr (ACT == GROVE_ACT_SWIGLU_BWD) gemm_epilogue_swiglu_bwd<MIH, BMH>(p, acc, mw0, nw0, fr, fq);
  else if (interior) gemm_epilogue_fast<MIH, BMH, ACT>(p, acc, mw0, nw0, fr, fq, ACT < 0 ? 1.f : scale);
  else gemm_epilogue_wide<MIH, BMH, ACT>(p, acc, mw0, nw0, fr, fq, ACT < 0 ? 1.f : scale);
}

#ifdef GROVE_EXPERIMENT_W4
// =====================================================================================================
// EXPERIMENT (round 3), compiled only with -DGROVE_EXPERIMENT_W4 (`make W4=1`; tools/bench_gemm.py waves): measured 10-25 % SLOWER than
// the eight-wave kernel above, bit-identical results. Kept as the record of why: DESIGN.md section 7a "four-wave form".
// FOUR-WAVE form of the 256 x 256 kernel: ONE wave per SIMD, wave tile 128 x 128 (two of the eight-wave kernel's wave tiles side by
// side), 512 registers per wave. Same work list, same epilogues and same stream-K slots as the eight-wave kernel (wave (wr, wc2) here
// = waves (wr, 2 wc2) and (wr, 2 wc2 + 1) there) and the same sum order, so the two are interchangeable per launch, bit for bit.
// What changes is the K loop. There is no second wave on the SIMD to run MFMAs while this one waits, so nothing may ever wait:
//   * the unit is a k-step of 32 (64 MFMAs per wave), staged into a ring of W4_NS = 5 LDS stages of 32 KB (four regions A_lo, B_lo, B_hi,
//     A_hi of 128 rows x 64 B): the loads of k-step t + 5 are issued during k-step t, four k-steps (~4000 MFMA cycles) before their sync
//     (a first cut with two 64-deep stages, i.e. one K tile of lookahead, stalled ~1900 cycles per K tile on vmcnt: 52 % MFMA-busy);
//   * the loop is software-pipelined across its one barrier per k-step:
//       sync(t):   my loads of k-step t + 1 landed (counted vmcnt: k-steps t + 2 .. t + 4 stay in flight), my fragment reads of stage
//                  t retired (lgkmcnt(0)); barrier
//       body(t):   64 MFMAs on the fragments of k-step t (registers) || 8 LDS-DMA loads of k-step t + 5 into stage t (free: everyone's
//                  reads of it retired before the barrier) || 16 ds_reads of the fragments of k-step t + 1 into the other register buffer
//     RAW: a wave retires its own loads of k-step t + 1 before barrier(t) and everyone reads them after it. WAR: see body(t).
// hipcc cannot allocate this loop (256 accumulators + 128 fragment registers: it parks fragments in AGPRs and spills accumulators,
// with builtins and with inline asm under register-class constraints alike), so a k-step is ONE hand-scheduled asm statement with
// FIXED registers — accumulators a[0:255], fragments v[128:255], read bases v[126:127] — generated by tools/gen/gen_gemm_w4_body.py
// (gemm_w4_body.inc). The fixed registers carry values from one statement to the next, which the compiler does not know: everything
// between two bodies is scalar (loop control, the staging stream's scalar bases), no fragment is kept across an epilogue (re-read
// after it), the accumulators leave through w4_read_acc, and tools/gen/check_w4_isa.py greps the ISA for compiler-made uses.
// RESULT (tools/bench_gemm.py waves, ablations of the body by W4X_* switches of the generator, (8192)^3, eight-wave kernel 1.49-1.54 PFLOP/s):
//   MFMAs only 1.87 | + reads + barrier 1.59 | + loads, no reads 1.39 | everything 1.34 (loads in the first 8 gaps: 1.19) | no barrier 1.40
// The 8 LDS-DMA instructions of a k-step cost the issuing wave 60+ cycles EACH (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") and
// with one wave per SIMD nobody issues MFMAs meanwhile: ~480 of a k-step's 1024 MFMA cycles. The eight-wave form pays the same issue
// cost inside the memory segment of the wave that is NOT on the matrix pipe — that alternation is what hides it, and why it stays.
// Staging addresses: a scalar base per operand (tile origin + K offset) plus a 32-bit per-lane row offset that is the SAME for every
// tile — which is why this form takes M % 256 == 0 and N % 256 == 0 only (no clamped edge rows). Past the end of its stream a block
// keeps re-issuing the last k-step's loads into free stages, so that the counted vmcnt means the same thing in every body.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const uint64_t a = (uint64_t)p;
  return (const char*)((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32));
}
struct w4_operands {
  unsigned ra, rb;        // LDS byte offsets of my A / B fragment reads inside a stage (region and fragment offsets are immediates)
  unsigned vo[8];         // [region * 2 + i]: byte offset of my chunk from the operand's scalar base
  unsigned st;            // LDS byte offset of the stage the current k-step's fragments came from (scalar)
  unsigned ldst;          // LDS byte address my LDS-DMA instructions start from inside a stage (lds base + wave * 1024; scalar)
  const char *sba, *sbb;  // A + (m0 * lda + k * 32) * 2 and B + (n0 * ldb + k * 32) * 2 of the k-step being issued
};
#include "gemm_w4_body.inc"

constexpr int W4_NT = 256, W4_BK = 32, W4_REGION = 8192, W4_STAGE = 4 * W4_REGION;
template <int ACT>
__global__ __launch_bounds__(W4_NT) void gemm_nt_w4_kernel(const grove_gemm_params p, const pp_work work) {
  constexpr int BM = 256, BMH = 128, WRH = 64, MIH = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc2 = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const char* __restrict__ A = (const char*)p.A;
  const char* __restrict__ B = (const char*)p.B;

  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const i32x4_t* my_work = work.table + wgid;
  const i32x4_t head = sload4(my_work);
  const int NT = head[0], nseg = head[1];
  if (NT == 0) return;

  w4_operands o;
  o.sba = o.sbb = nullptr;
  o.st = 0;
  // The per-lane operands of the bodies. Called again after every epilogue (on a laundered lane id, so that they ARE recomputed): kept
  // live across an epilogue, which needs most of the VGPRs, these registers get spilled — and hipcc spills into AGPRs, i.e. into the
  // accumulators it knows nothing about.
  auto init_o = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int ln = t & 63;
    const int fr_ = ln & 15, fq_ = ln >> 4;
    // staging: a region = 128 rows x 4 chunks = 2 LDS-DMA instructions per thread (rows st_r and st_r + 64: same swizzle band)
    const int st_c = t & 3, st_r = t >> 2;
    const unsigned lc = (unsigned)swz<32>(st_r, st_c) * 16;
    const unsigned lda2 = (unsigned)p.lda * 2, ldb2 = (unsigned)p.ldb * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = st_r + 64 * i;
      const int pn = (r & ~31) | (((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3));  // (see gemm_nt_pp_kernel: permuted B rows)
      o.vo[0 * 2 + i] = (unsigned)r * lda2 + lc;
      o.vo[3 * 2 + i] = (unsigned)(BMH + r) * lda2 + lc;
      o.vo[1 * 2 + i] = (unsigned)pn * ldb2 + lc;
      o.vo[2 * 2 + i] = (unsigned)(128 + pn) * ldb2 + lc;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned kc = (unsigned)swz<32>(fr_, fq_) * 16;  // (row bases are multiples of 16: the band depends on fr only)
    o.ra = lds0 + (wr * WRH + fr_) * 64 + kc;
    o.rb = lds0 + (wc2 * 64 + fr_) * 64 + kc;
    o.ldst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  };
  init_o();
  // the staging stream, in k-steps: is_k in [2 k0, 2 k1) of segment is_seg; at the end of the stream it stays on the last k-step
  int is_seg = 0, is_k, is_kend;
  auto seg_src = [&](const i32x4_t e) {
    is_k = 2 * (e[2] & 0xffff), is_kend = 2 * (int)((unsigned)e[2] >> 16);
    o.sba = uniform_ptr(A + ((int64_t)e[0] * p.lda + (int64_t)is_k * W4_BK) * 2);
    o.sbb = uniform_ptr(B + ((int64_t)e[1] * p.ldb + (int64_t)is_k * W4_BK) * 2);
  };
  seg_src(sload4(my_work + G));
  auto advance_issue = [&]() {  // scalar only
    if (is_k + 1 < is_kend) {
      ++is_k;
      o.sba += W4_BK * 2;
      o.sbb += W4_BK * 2;
    } else if (is_seg + 1 < nseg) {
      ++is_seg;
      seg_src(sload4(my_work + (is_seg + 1) * G));
    }
  };
  auto read_p0 = [&]() {  // the fragments of the k-step in stage o.st into register buffer 0
    unsigned st = __builtin_amdgcn_readfirstlane(o.st);
    asm volatile(W4_READ_P0 : : [st] "s"(st), [ra] "v"(o.ra), [rb] "v"(o.rb) : W4_CLOBBERS);
  };

  float scale = 1.f;
  if (p.scale_ptr) {
    scale = *p.scale_ptr;
    if (p.scale_tanh) scale = tanhf(scale);
  }
  asm volatile("" ::"v"(scale));
  // prologue: k-steps 0 .. NS - 1 of my stream in flight (nothing lives in the fixed registers yet), the first landed and in registers
#pragma unroll
  for (int sgi = 0; sgi < W4_NS; ++sgi) {
    if (sgi) advance_issue();
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      char* dst = smem + sgi * W4_STAGE + (n >> 1) * W4_REGION + wave * (64 * 16) + (n & 1) * (W4_NT * 16);
      const char* sb = ((n >> 1) == 0 || (n >> 1) == 3) ? o.sba : o.sbb;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + o.vo[n]), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (W4_NS - 1)) : "memory");
  __builtin_amdgcn_s_barrier();
  read_p0();

  const bool fast_addr = p.c_dtype == GROVE_BF16 && !p.c_idx && !p.r_idx && !p.n_group;
  bool relax = false;  // the last epilogue left at least 32 vector-memory operations behind the staged loads
  for (int sg = 0; sg < nseg; ++sg) {
    const i32x4_t e = sload4(my_work + (sg + 1) * G);
    const int m0 = e[0], n0 = e[1], nks = (int)((unsigned)e[2] >> 16) - (e[2] & 0xffff), part = e[3];
    for (int k = 0; k < nks; ++k) {  // two k-steps per K tile of the list; every body issues the stream's next k-step first
      advance_issue();
      if (k == 0) {  // a segment's first k-step starts the accumulators from the MFMA's zero operand
        if (relax) w4_body<0, 1, 1, W4_VMC_RELAX>(o);
        else w4_body<0, 1, 1, W4_VMC>(o);
        relax = false;
      } else {
        w4_body<0, 1, 0, W4_VMC>(o);
      }
      advance_issue();
      if (k == nks - 1) w4_body<1, 0, 0, W4_VMC>(o);  // (the next k-step's fragments would not survive the epilogue)
      else w4_body<1, 1, 0, W4_VMC>(o);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (the compiler's hazard recogniser does not see inside the bodies)
    f32x4_t acc[2 * MIH][4];
    if (__builtin_expect(part != 0, 0)) {
      // stream-K part: the raw accumulators to the slots of the two eight-wave-form waves this wave stands for
      auto put = [&](const int c) {
        char* wb = (char*)work.ws + ((size_t)(part - 1) * 8 + (wr * 4 + wc2 * 2 + c)) * 32768;
        int lo = lane * 16;
#pragma unroll
        for (int i = 0; i < 2 * MIH; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) *(f32x4_t*)(wb + (i * 4 + j) * 1024 + lo) = acc[i][j];
      };
      w4_read_acc<0>(acc);
      put(0);
      w4_read_acc<1>(acc);
      put(1);
      relax = false;
    } else {
      const int mw0 = m0 + wr * WRH;
      const bool interior = fast_addr && m0 + BM <= p.M && n0 + P_BN <= p.N;
      auto epi = [&](const int c) {
        const int nw0 = n0 + (wc2 * 2 + c) * 32;
        if (interior) gemm_epilogue_fast<MIH, BMH, ACT>(p, acc, mw0, nw0, fr, fq, ACT < 0 ? 1.f : scale);
        else gemm_epilogue_wide<MIH, BMH, ACT>(p, acc, mw0, nw0, fr, fq, ACT < 0 ? 1.f : scale);
      };
      w4_read_acc<0>(acc);
      epi(0);
      w4_read_acc<1>(acc);
      epi(1);
      relax = interior;  // at least 32 vector-memory operations were issued after the staged loads: the counted wait may allow 32 more
    }
    init_o();
    if (sg + 1 < nseg) read_p0();  // the next segment's first k-step (landed and visible since the last body's sync)
  }
}
#endif  // GROVE_EXPERIMENT_W4

static int g_num_cus = 0;
static int g_persistent_blocks = 0;   // 0 = one resident block per CU; else the grid of the persistent kernels (grove_gemm_set_persistent_blocks)
static int g_gemm_last_epilogue = 0;  // ACT template argument of the last pipelined launch (see grove_gemm_last_epilogue)
static int g_gemm_stream_k = 1;       // plan_stream_k mode (grove_gemm_set_stream_k)
static int g_gemm_tap_skip = 1;       // temporal tap skipping of the Conv3d implicit GEMMs (grove_gemm_set_tap_skip; A/B knob)
static int g_gemm_last_stream_k = 0;  // S of the last pipelined launch
#ifdef GROVE_EXPERIMENT_W4
static int g_gemm_waves = 8;          // 8 = the eight-wave pipelined kernel, 4 = gemm_nt_w4_kernel where it applies (grove_gemm_set_waves)
#endif

inline int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    g_num_cus = n > 0 ? n : 256;
  }
  return g_persistent_blocks > 0 && g_persistent_blocks < g_num_cus ? g_persistent_blocks : g_num_cus;
}

// Stream-K plan for `tiles` output tiles of nk K tiles on G blocks: after the whole rounds, a partial round of `tail` tiles
// is cut into s = G / tail (2..4) K ranges of S = ceil(nk / s) K tiles, one block each. S = 0: whole tiles only. kt_units: the
// length of the launch in K tiles either way; a split costs the raw 256 KB stores of the parts, the fix-up launch that reads
// them back and some of the L2 sharing between neighbouring tiles: SK_FIXED K tiles' worth (~30 us, measured).
// (Equal parts, not one evenly dealt stream: blocks that share an operand panel stay in the same K phase, so the panel is
// fetched into L2 once — dealt out unevenly, 240 tiles on 256 CUs ran 1.5x SLOWER than whole tiles.)
constexpr int SK_FIXED = 20;  // ... at 256 parts (128 tiles in halves); the stores and the fix-up scale with the parts, a third of it is launch gaps
struct sk_plan { int S, parts, rounds, tail; double kt_units; };
inline sk_plan plan_stream_k(long tiles, int nk, int G, int mode) {  // mode: 0 = never, 1 = where it pays, 2 = wherever it applies (tests)
  sk_plan pl;
  pl.rounds = (int)(tiles / G), pl.tail = (int)(tiles % G), pl.S = 0, pl.parts = 1;
  pl.kt_units = (double)((tiles + G - 1) / G) * nk;
  if (!mode || pl.tail == 0) return pl;
  // Round 4: FEWER tiles than CUs (rounds == 0: the few-row GEMMs of LLaMA's last-layer tail, prefill, the decoder's K = 2048 layers) —
  // the only round is a partial one; with a long K it is cut the same way, into up to 8 ranges (32 tiles x 8 = the whole chip).
  if (pl.rounds == 0 && nk < 32) return pl;
  const int s = std::min(G / pl.tail, pl.rounds == 0 ? 8 : 4);
  if (s < 2) return pl;
  const int S = (nk + s - 1) / s;
  if (pl.rounds == 0 && S < 4) return pl;
  const double pf = (double)(pl.tail * s) / G;  // part blocks / CUs: 5 K tiles for the extra launch + the stores and read-backs (measured: 20 at pf = 1, ~8 at 0.56)
  const double fixed = 5.0 + (SK_FIXED - 5.0) * pf * pf;
  const double t = (double)pl.rounds * nk + S + fixed;
  if (S >= 2 && (mode == 2 || t < 0.97 * pl.kt_units)) pl.S = S, pl.parts = (nk + S - 1) / S, pl.kt_units = t;
  return pl;
}

// The work list of a shape (see pp_work) and the fix-up's list of split tiles. Round 3: both live in CALLER-OWNED memory — the
// library allocates nothing (grove_hip.h "Workspaces of the persistent GEMMs"): the caller asks grove_gemm_plan for the sizes, has
// grove_gemm_plan_image write the lists into a host buffer, uploads that image once per plan key, and passes image + scratch
// (the stream-K partial tiles) with every launch.
struct pp_table_key {
  int dev, bm, tiles_m, tiles_n, nk, G, S;
  int ft = 0, T = 0, parts = 0;  // temporal tap skipping (see tap_skip): tile rows per frame, frames per group; parts of a split tail
};
struct pp_table_dev { const i32x4_t* table; const i32x4_t* fixups; int n_fixups, n_slots; };

inline uint64_t plan_key(const pp_table_key& k) {  // the image depends on exactly these (not on the epilogue or the operand type)
  uint64_t h = 1469598103934665603ull;
  const int v[9] = {k.bm, k.tiles_m, k.tiles_n, k.nk, k.G, k.S, k.ft, k.T, k.parts};
  for (int i = 0; i < 9; ++i) h = (h ^ (uint64_t)(uint32_t)v[i]) * 1099511628211ull;
  return h;
}

// Temporal tap skipping (grove_gemm_params.a_frame_rows / a_frames; the Conv3d 3 x 3 x 3 adapters as implicit GEMMs): the output rows
// are frames of `ft` tiles, T frames per group, and the K range is three equal tap groups (temporal offset -1, 0, +1). For the rows of a
// group's FIRST frame every row index of the first tap group is -1 (the temporal zero padding), for its LAST frame the last group's: a
// third of those tiles' K range multiplies zeros. A tile's K range [ka, kb) leaves those K tiles out — the accumulators then miss
// additions of +0.0 only, so whole tiles are bit-identical to the un-skipped GEMM.
inline void tile_k_range(const pp_table_key& key, int m0, int& ka, int& kb) {
  ka = 0, kb = key.nk;
  if (!key.ft) return;
  const int t = (m0 / key.bm / key.ft) % key.T;
  if (t == 0) ka = key.nk / 3;
  if (t == key.T - 1) kb = key.nk - key.nk / 3;
}

// tile L -> origin: bands of 8 tile rows, column-major inside a band (the last band may be shorter)
inline void tile_origin(const pp_table_key& key, int L, int& m0, int& n0) {
  const int per_band = 8 * key.tiles_n, band = L / per_band, in_band = L - band * per_band;
  const int rows = std::min(8, key.tiles_m - band * 8);
  m0 = (band * 8 + in_band % rows) * key.bm;
  n0 = (in_band / rows) * P_BN;
}

// processing order of the tiles. Plain shapes: L itself. Tap-skipping shapes: the FULL tiles first, then the short ones, each class in
// band order — a round of the persistent grid lasts as long as its longest tile, so short tiles only pay when they share rounds (and
// the stream-K tail) with each other.
inline void tile_order(const pp_table_key& key, std::vector<int>& order) {
  const int tiles = key.tiles_m * key.tiles_n;
  order.resize(tiles);
  if (!key.ft) {
    for (int L = 0; L < tiles; ++L) order[L] = L;
    return;
  }
  int n = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int L = 0; L < tiles; ++L) {
      int m0, n0, ka, kb;
      tile_origin(key, L, m0, n0);
      tile_k_range(key, m0, ka, kb);
      if ((kb - ka == key.nk) == (pass == 0)) order[n++] = L;
    }
}

// Plan of a tap-skipping shape: rounds of whole tiles in tile_order, then the partial round either whole or cut into `parts` K ranges
// per tile (each tile's OWN range [ka, kb) in `parts` equal pieces). Same cost terms as plan_stream_k, with every round priced at its
// longest tile.
inline sk_plan plan_tap_skip(const pp_table_key& key0, int G, int mode, int* parts_out) {
  pp_table_key key = key0;
  const int tiles = key.tiles_m * key.tiles_n, nk = key.nk;
  std::vector<int> order;
  tile_order(key, order);
  auto len = [&](int i) {
    int m0, n0, ka, kb;
    tile_origin(key, order[i], m0, n0);
    tile_k_range(key, m0, ka, kb);
    return kb - ka;
  };
  sk_plan pl;
  pl.rounds = tiles / G, pl.tail = tiles % G, pl.S = 0, pl.parts = 1;
  double whole = 0;
  for (int r = 0; r < pl.rounds; ++r) {
    int mx = 0;
    for (int i = r * G; i < (r + 1) * G; ++i) mx = std::max(mx, len(i));
    whole += mx;
  }
  int tail_max = 0, tail_min = nk;
  for (int i = pl.rounds * G; i < tiles; ++i) tail_max = std::max(tail_max, len(i)), tail_min = std::min(tail_min, len(i));
  pl.kt_units = whole + tail_max;
  *parts_out = 1;
  if (!mode || pl.tail == 0 || pl.rounds == 0) return pl;
  const int s = std::min(G / pl.tail, 4);
  if (s < 2 || tail_min < 2 * s) return pl;
  for (int i = pl.rounds * G; i < tiles; ++i)  // every piece of every tail tile must hold at least one K tile
    if ((s - 1) * ((len(i) + s - 1) / s) >= len(i)) return pl;
  const int S = (tail_max + s - 1) / s;
  const double pf = (double)(pl.tail * s) / G;
  const double fixed = 5.0 + (SK_FIXED - 5.0) * pf * pf;
  const double t = whole + S + fixed;
  if (mode == 2 || t < 0.97 * pl.kt_units) pl.S = S, pl.parts = s, pl.kt_units = t, *parts_out = s;
  return pl;
}

// the lists of a shape on the host: `t` = the work list (rows of G entries, see pp_work), `fix` = the split tiles {m0, n0, first slot, parts}
inline void build_work_list(const pp_table_key& key, const sk_plan& pl, std::vector<i32x4_t>& t, std::vector<i32x4_t>& fix) {
  const int G = key.G, nk = key.nk, S = key.S;
  const int tiles = key.tiles_m * key.tiles_n;
  std::vector<int> order;
  tile_order(key, order);
  const int n_dp_max = S ? pl.rounds : (tiles + G - 1) / G;
  const int rows = 1 + n_dp_max + 1;
  t.assign((size_t)rows * G, i32x4_t{0, 0, 0, 0});
  fix.clear();
  for (int w = 0; w < G; ++w) {
    int n = 0, NT = 0;
    auto push = [&](int L, int j, int part) {  // j < 0: the tile's whole K range; else piece j of pl.parts
      int m0, n0, ka, kb;
      tile_origin(key, L, m0, n0);
      tile_k_range(key, m0, ka, kb);
      int k0 = ka, k1 = kb;
      if (j >= 0) {
        const int Sa = key.ft ? (kb - ka + pl.parts - 1) / pl.parts : S;  // (plain shapes: pieces of S K tiles, the last one shorter)
        k0 = ka + j * Sa, k1 = std::min(kb, ka + (j + 1) * Sa);
      }
      t[(size_t)(1 + n) * G + w] = i32x4_t{m0, n0, k0 | (k1 << 16), part};
      ++n, NT += k1 - k0;
    };
    const int n_dp = S ? pl.rounds : (tiles - w + G - 1) / G;
    for (int i = 0; i < n_dp; ++i) push(order[w + i * G], -1, 0);
    if (S && w < pl.tail * pl.parts) {  // part-major: neighbouring blocks hold the same K range of neighbouring tiles
      const int j = w / pl.tail, a = w % pl.tail;
      push(order[pl.rounds * G + a], j, 1 + a * pl.parts + j);  // slot: a tile's parts in K order
    }
    t[w] = i32x4_t{NT, n, 0, 0};
  }
  if (S)
    for (int a = 0; a < pl.tail; ++a) {
      int m0, n0;
      tile_origin(key, order[pl.rounds * G + a], m0, n0);
      fix.push_back(i32x4_t{m0, n0, a * pl.parts, pl.parts});
    }
  (void)nk;
}

template <int BM, bool GATHER, int ACT, bool FP8 = false, bool GROUPED = false>
int launch_pp_act(const grove_gemm_params& p, hipStream_t s, const float* row_scale = nullptr, const float* col_scale = nullptr) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + P_BN - 1) / P_BN;
  const size_t lds = 2 * (size_t)P_STAGE;
  const int G = num_cus(), nk = p.K / P_BK;
  const long tiles = (long)tiles_m * tiles_n;
  GROVE_CHECK(tiles < (1L << 24) && nk < 65536, GROVE_E_SHAPE, "gemm: %d x %d tiles x %d K tiles overflow the pipelined kernel's work list", tiles_m, tiles_n, nk);
  // temporal tap skipping (Conv3d adapters): whole tiles inside one frame, three equal tap groups on K-tile boundaries
  pp_table_key skey{0, BM, tiles_m, tiles_n, nk, G, 0};
  if (GATHER && g_gemm_tap_skip && p.a_frame_rows > 0 && p.a_frames >= 1 && p.a_frame_rows % BM == 0 && p.a_taps % 3 == 0 && nk % 3 == 0 &&
      (nk / 3) % (p.a_taps / 3) == 0 && p.M % ((long)p.a_frame_rows * p.a_frames) == 0 && !p.k_group)
    skey.ft = p.a_frame_rows / BM, skey.T = p.a_frames;
  int sparts = 0;
  const sk_plan pl = skey.ft ? plan_tap_skip(skey, G, g_gemm_stream_k, &sparts) : plan_stream_k(tiles, nk, G, g_gemm_stream_k);
  const int grid = pl.S ? G : (int)(tiles < G ? tiles : G);  // (the list is laid out for this grid: its row stride and wgid map)
  pp_table_key key{0, BM, tiles_m, tiles_n, nk, grid, pl.S};
  key.ft = skey.ft, key.T = skey.T, key.parts = skey.ft ? sparts : 0;
  const int n_dp_max = pl.S ? pl.rounds : (int)((tiles + G - 1) / G);
  const size_t list_bytes = (size_t)(1 + n_dp_max + 1) * grid * sizeof(i32x4_t);
  const int n_fix = pl.S ? pl.tail : 0, n_slots = pl.S ? pl.tail * pl.parts : 0;
  const size_t image_bytes = list_bytes + (size_t)n_fix * sizeof(i32x4_t);
  const size_t scratch_bytes = (size_t)n_slots * P_SLOT;
  if (t_call.mode == 1) {  // grove_gemm_make_plan: sizes and key only
    grove_gemm_plan* o = t_call.plan;
    o->bm = BM, o->tiles_m = tiles_m, o->tiles_n = tiles_n, o->k_tiles = nk, o->grid = grid, o->stream_k = pl.S;
    o->image_bytes = (int64_t)image_bytes, o->scratch_bytes = (int64_t)scratch_bytes;
    o->key = plan_key(key);
    return GROVE_OK;
  }
  if (t_call.mode == 2) {  // grove_gemm_plan_image: the lists, as the device will read them
    GROVE_CHECK(t_call.host_image && t_call.host_image_bytes >= image_bytes, GROVE_E_WORKSPACE, "gemm_plan_image: buffer of %zu bytes, the image needs %zu",
                t_call.host_image_bytes, image_bytes);
    std::vector<i32x4_t> t, fix;
    build_work_list(key, pl, t, fix);
    GROVE_CHECK(t.size() * sizeof(i32x4_t) == list_bytes && (int)fix.size() == n_fix, GROVE_E_WORKSPACE, "gemm_plan_image: list size mismatch");
    memcpy(t_call.host_image, t.data(), list_bytes);
    if (n_fix) memcpy((char*)t_call.host_image + list_bytes, fix.data(), (size_t)n_fix * sizeof(i32x4_t));
    return GROVE_OK;
  }
  GROVE_CHECK(t_call.image && t_call.image_bytes >= image_bytes && ((uintptr_t)t_call.image & 15) == 0, GROVE_E_WORKSPACE,
              "gemm: the persistent kernel needs its work-list image (%zu bytes, 16-byte aligned; got %zu): grove_gemm_make_plan + grove_gemm_plan_image", image_bytes,
              t_call.image_bytes);
  GROVE_CHECK(!scratch_bytes || (t_call.scratch && t_call.scratch_bytes >= scratch_bytes && ((uintptr_t)t_call.scratch & 15) == 0), GROVE_E_WORKSPACE,
              "gemm: this launch cuts %d tiles into stream-K parts and needs %zu bytes of scratch (got %zu)", n_fix, scratch_bytes, t_call.scratch_bytes);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<BM, GATHER, ACT, FP8, GROUPED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  g_gemm_last_epilogue = ACT;
  const pp_table_dev td{(const i32x4_t*)t_call.image, (const i32x4_t*)((const char*)t_call.image + list_bytes), n_fix, n_slots};
  pp_work work{td.table, (float*)t_call.scratch, row_scale, col_scale};
  g_gemm_last_stream_k = pl.S;
  bool w4 = false;
#ifdef GROVE_EXPERIMENT_W4
  if constexpr (BM == 256 && !GATHER && !FP8) {
    w4 = g_gemm_waves == 4 && !p.k_group && p.M % 256 == 0 && p.N % 256 == 0 && 255ll * 2 * std::max(p.lda, p.ldb) < (1ll << 32);
    if (w4) {
      static bool attr4_set = false;
      if (!attr4_set) {
        hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NS * W4_STAGE);
        attr4_set = true;
      }
      hipLaunchKernelGGL((gemm_nt_w4_kernel<ACT>), dim3(grid, 1, 1), dim3(W4_NT), (size_t)W4_NS * W4_STAGE, s, p, work);
    }
  }
#endif
  if (!w4) hipLaunchKernelGGL((gemm_nt_pp_kernel<BM, GATHER, ACT, FP8, GROUPED>), dim3(grid, 1, 1), dim3(P_NT), lds, s, p, work);
  GROVE_LAUNCH_CHECK();
  if (td.n_fixups) {
    hipLaunchKernelGGL((gemm_pp_fixup_kernel<BM, ACT, FP8>), dim3(td.n_fixups * (8 / FIX_WAVES), 1, 1), dim3(64 * FIX_WAVES), 0, s, p, td.fixups,
                       (const float*)work.ws, work);
    GROVE_LAUNCH_CHECK();
  }
  return GROVE_OK;
}
// the gathered instances carry the epilogues their callers use (plain, scaled, ReLU: window (un)partition, Conv3d adapters)
inline bool pp_act_ok(const grove_gemm_params& p) {
  if (p.a_idx) return p.act == GROVE_ACT_NONE || p.act == GROVE_ACT_RELU;
  return p.act == GROVE_ACT_NONE || p.act == GROVE_ACT_GELU || p.act == GROVE_ACT_QUICKGELU || p.act == GROVE_ACT_RELU || p.act == GROVE_ACT_SWIGLU_PAIR ||
         p.act == GROVE_ACT_SWIGLU_BWD;
}
template <int BM, bool GATHER>
int launch_pp(const grove_gemm_params& p, hipStream_t s) {
  if (p.act == GROVE_ACT_NONE) {
    if (p.alpha == 1.f && !p.scale_ptr) return launch_pp_act<BM, GATHER, -1>(p, s);
    return launch_pp_act<BM, GATHER, GROVE_ACT_NONE>(p, s);
  }
  if (p.act == GROVE_ACT_RELU) return launch_pp_act<BM, GATHER, GROVE_ACT_RELU>(p, s);
  if constexpr (!GATHER) {
    if (p.act == GROVE_ACT_GELU) return launch_pp_act<BM, false, GROVE_ACT_GELU>(p, s);
    if (p.act == GROVE_ACT_QUICKGELU) return launch_pp_act<BM, false, GROVE_ACT_QUICKGELU>(p, s);
    if (p.act == GROVE_ACT_SWIGLU_PAIR) return launch_pp_act<BM, false, GROVE_ACT_SWIGLU_PAIR>(p, s);
    if (p.act == GROVE_ACT_SWIGLU_BWD) return launch_pp_act<BM, false, GROVE_ACT_SWIGLU_BWD>(p, s);
  }
  GROVE_CHECK(false, GROVE_E_SHAPE, "gemm: no pipelined instance for act %d%s", p.act, GATHER ? " with gathered A" : "");
}

// fraction of the 256 CUs' block slots that do useful work when `tiles` equal blocks are spread over them
inline double wave_util(long tiles, int per_cu) {
  const double slots = 256.0 * per_cu;
  if (tiles >= slots) {
    const double rounds = tiles / slots;
    return rounds / (double)(long)(rounds + 0.999999);
  }
  return tiles >= 256 ? 1.0 : tiles / 256.0;
}

}  // namespace

// staging variant: 1 = LDS-DMA (default), 0 = register staged (kept for A/B and as the
// conservative path); switchable at run time for in-process A/B (cdna guide §5.4 rule 24).
static int g_gemm_glds = 1;
static int g_gemm_last_variant = 0;  // kernel chosen by the last grove_gemm_bf16 call (see grove_gemm_last_variant)
static int g_gemm_bk = 0;      // 0 = auto (64 when K allows), 32 = forced
extern "C" int grove_gemm_set_bk(int bk) {
  g_gemm_bk = bk;
  return GROVE_OK;
}
static int g_gemm_tile_m = 0;  // 0 = auto, 128 / 192 = forced
extern "C" int grove_gemm_set_tile_m(int tile_m) {
  g_gemm_tile_m = tile_m;
  return GROVE_OK;
}
static int g_gemm_tile_n = 0;  // 0 = auto, 64 / 128 = forced (A/B runs)
extern "C" int grove_gemm_set_tile_n(int tile_n) {
  g_gemm_tile_n = tile_n;
  return GROVE_OK;
}
extern "C" int grove_gemm_set_staging(int use_lds_dma) {
  g_gemm_glds = use_lds_dma ? 1 : 0;
  return GROVE_OK;
}

// The e4m3 GEMM (grove_gemm_fp8, gemm_fp8.hip) on the FP8 instances of the pipelined kernel: same staging stream and phase
// structure, twice the math per byte. Returns 1 when the problem does not fit them (the caller falls back to its own kernel).
// (gemm_fp8.hip's entry points set the call context through these: mode 0 launch / 1 plan / 2 image, as above)
void grove_gemm_ctx_set(const grove_gemm_workspace* w, int mode, grove_gemm_plan* plan, void* host_image, size_t host_bytes) {
  t_call = gemm_call_ctx{w ? w->image : nullptr, w ? w->image_bytes : 0, w ? w->scratch : nullptr, w ? w->scratch_bytes : 0, mode, plan, host_image, host_bytes};
}
void grove_gemm_ctx_clear() { t_call = gemm_call_ctx{nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0}; }
bool grove_gemm_ctx_has_image() { return t_call.mode != 0 || t_call.image != nullptr; }

int grove_gemm_fp8_pipelined(const grove_gemm_fp8_params* q, hipStream_t s) {
  if (!grove_gemm_ctx_has_image()) return 1;  // no work-list image: the caller's own (two-barrier) kernel runs
  const bool al = ((((uintptr_t)q->C | (uintptr_t)q->bias | (uintptr_t)q->residual | (uintptr_t)q->scale_b) & 15) == 0) && q->ldc % 8 == 0 &&
                  (!q->residual || q->ldr % 8 == 0);
  if (!al || q->N % 8 != 0 || q->K % 128 != 0 || q->lda % 16 != 0 || q->ldb % 16 != 0) return 1;
  if (q->act != GROVE_ACT_NONE && q->act != GROVE_ACT_QUICKGELU && q->act != GROVE_ACT_GELU) return 1;
  grove_gemm_params p;
  memset(&p, 0, sizeof(p));
  p.A = q->A, p.B = q->B, p.C = q->C, p.bias = q->bias, p.residual = q->residual;
  p.M = q->M, p.N = q->N, p.K = q->K / 2, p.lda = q->lda / 2, p.ldb = q->ldb / 2, p.ldc = q->ldc, p.ldr = q->residual ? q->ldr : 0;
  p.act = q->act, p.alpha = 1.f, p.c_dtype = GROVE_BF16, p.batch1 = p.batch2 = 1, p.a_taps = 1;
  // 256- or 192-row tiles by the cost model of the bf16 instances (same bytes per K tile, same phase lengths)
  const int G = num_cus(), nk = q->K / 128;
  const long tn = (q->N + 255) / 256;
  auto cost = [&](long tiles, double kt, double fixed) {
    const sk_plan pl = plan_stream_k(tiles, nk, G, g_gemm_stream_k);
    const double r = (double)((tiles + G - 1) / G);
    if (pl.S) return pl.kt_units * kt + r * fixed;
    return r * (nk * kt * (0.6 + 0.4 * (double)tiles / (r * G)) + fixed);
  };
  const bool big = cost((long)((q->M + 255) / 256) * tn, 1.5, 6.0) <= cost((long)((q->M + 191) / 192) * tn, 1.17, 4.8);
  if (q->act == GROVE_ACT_NONE) return big ? launch_pp_act<256, false, -1, true>(p, s, q->scale_a, q->scale_b) : launch_pp_act<192, false, -1, true>(p, s, q->scale_a, q->scale_b);
  if (q->act == GROVE_ACT_GELU)  // round 6: SAM's mlp.lin1 (fp8_policy "sam_mlp"; image_encoder.py:243-259, common.py:21-26)
    return big ? launch_pp_act<256, false, GROVE_ACT_GELU, true>(p, s, q->scale_a, q->scale_b) : launch_pp_act<192, false, GROVE_ACT_GELU, true>(p, s, q->scale_a, q->scale_b);
  return big ? launch_pp_act<256, false, GROVE_ACT_QUICKGELU, true>(p, s, q->scale_a, q->scale_b)
             : launch_pp_act<192, false, GROVE_ACT_QUICKGELU, true>(p, s, q->scale_a, q->scale_b);
}

// Host-only view of the pipelined kernels' work list (no device needed): what `grid` blocks do for tiles_m x tiles_n output tiles of
// BM x 256 with nk K tiles. list: int32 [rows][grid][4] (row 0 = {K tiles of the block's stream, segments, 0, 0}; row 1 + i = segment
// {m0, n0, k0 | k1 << 16, part}), fixups: int32 [n][4] = {m0, n0, first slot, parts}. Returns the number of rows (negative: an error or a
// buffer too small); *n_fixups and *k_tiles_per_part (0 = whole tiles only) describe the stream-K tail. mode as grove_gemm_set_stream_k.
extern "C" int grove_gemm_work_list(int bm, int tiles_m, int tiles_n, int nk, int num_cus, int mode, int32_t* list, int64_t list_cap,
                                    int32_t* fixups, int64_t fixups_cap, int* n_fixups, int* k_tiles_per_part) {
  GROVE_CHECK((bm == 192 || bm == 256) && tiles_m > 0 && tiles_n > 0 && nk > 0 && nk < 65536 && num_cus > 0, GROVE_E_SHAPE, "gemm_work_list: bad arguments");
  const long tiles = (long)tiles_m * tiles_n;
  const sk_plan pl = plan_stream_k(tiles, nk, num_cus, mode);
  const int grid = pl.S ? num_cus : (int)(tiles < num_cus ? tiles : num_cus);
  std::vector<i32x4_t> t, fix;
  build_work_list(pp_table_key{0, bm, tiles_m, tiles_n, nk, grid, pl.S}, pl, t, fix);
  GROVE_CHECK((int64_t)t.size() * 4 <= list_cap && (int64_t)fix.size() * 4 <= fixups_cap, GROVE_E_WORKSPACE, "gemm_work_list: %zu + %zu entries do not fit the buffers", t.size(), fix.size());
  memcpy(list, t.data(), t.size() * sizeof(i32x4_t));
  if (!fix.empty()) memcpy(fixups, fix.data(), fix.size() * sizeof(i32x4_t));
  *n_fixups = (int)fix.size();
  *k_tiles_per_part = pl.S;
  return (int)(t.size() / grid);
}

// The persistent kernels launch one block per CU and a block owns its CU for the whole GEMM: when another long-lived kernel (an RCCL
// collective overlapped with the backward at N > 1) holds some CUs, the blocks dealt to them wait for a free CU — i.e. for another
// block's whole share — and the GEMM takes up to twice as long. `n` < CU count makes plans, images and launches use n resident
// blocks, leaving the other CUs to the collective (0 = all CUs). A/B knob for the first multi-GPU run: unmeasured (no N > 1 hardware).
extern "C" int grove_gemm_set_persistent_blocks(int n) {
  g_persistent_blocks = n < 0 ? 0 : n;
  return GROVE_OK;
}
extern "C" int grove_gemm_persistent_blocks(void) { return g_persistent_blocks; }
extern "C" int grove_gemm_set_tap_skip(int on) {
  g_gemm_tap_skip = on ? 1 : 0;
  return GROVE_OK;
}
#ifdef GROVE_EXPERIMENT_W4
extern "C" int grove_gemm_set_waves(int waves) {  // 8 (default) or 4: the four-wave form of the 256-row instances (A/B knob)
  g_gemm_waves = waves == 4 ? 4 : 8;
  return GROVE_OK;
}
#endif
extern "C" int grove_gemm_set_stream_k(int mode) {
  g_gemm_stream_k = mode < 0 ? 0 : mode > 2 ? 2 : mode;
  return GROVE_OK;
}
extern "C" int grove_gemm_last_stream_k(void) { return g_gemm_last_stream_k; }
extern "C" int grove_gemm_last_variant(void) { return g_gemm_last_variant; }
extern "C" int grove_gemm_last_epilogue(void) { return g_gemm_last_epilogue; }

static int gemm_bf16_dispatch(const grove_gemm_params* pp, void* stream);

namespace {
struct call_guard {  // the call context never outlives its call
  ~call_guard() { t_call = gemm_call_ctx{nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0}; }
};
inline void set_launch_ctx(const grove_gemm_workspace* w) {
  t_call = gemm_call_ctx{w ? w->image : nullptr, w ? w->image_bytes : 0, w ? w->scratch : nullptr, w ? w->scratch_bytes : 0, 0, nullptr, nullptr, 0};
}
}  // namespace

extern "C" int grove_gemm_bf16(const grove_gemm_params* pp, const grove_gemm_workspace* w, void* stream) {
  call_guard g;
  set_launch_ctx(w);
  return gemm_bf16_dispatch(pp, stream);
}

extern "C" int grove_gemm_make_plan(const grove_gemm_params* pp, grove_gemm_plan* out) {
  GROVE_CHECK(out != nullptr, GROVE_E_SHAPE, "gemm_plan: null output");
  memset(out, 0, sizeof(*out));
  call_guard g;
  t_call = gemm_call_ctx{nullptr, 0, nullptr, 0, 1, out, nullptr, 0};
  const int rc = gemm_bf16_dispatch(pp, nullptr);
  out->variant = g_gemm_last_variant;
  return rc;
}

extern "C" int grove_gemm_plan_image(const grove_gemm_params* pp, void* host_image, size_t bytes) {
  call_guard g;
  t_call = gemm_call_ctx{nullptr, 0, nullptr, 0, 2, nullptr, host_image, bytes};
  return gemm_bf16_dispatch(pp, nullptr);
}

size_t grove_gemm_workspace_bytes_impl(const grove_gemm_params* pp) {
  grove_gemm_plan pl;
  if (grove_gemm_make_plan(pp, &pl) != GROVE_OK) return 0;
  return (size_t)pl.image_bytes + (size_t)pl.scratch_bytes;
}
extern "C" size_t grove_gemm_workspace_bytes(const grove_gemm_params* pp) { return grove_gemm_workspace_bytes_impl(pp); }

static int gemm_bf16_dispatch(const grove_gemm_params* pp, void* stream) {
  GROVE_CHECK(pp != nullptr, GROVE_E_SHAPE, "gemm: null params");
  grove_gemm_params p = *pp;
  if (p.batch1 <= 0) p.batch1 = 1;
  if (p.batch2 <= 0) p.batch2 = 1;
  if (p.a_taps <= 0) p.a_taps = 1;
  GROVE_CHECK(p.M > 0 && p.N > 0 && p.K > 0, GROVE_E_SHAPE, "gemm: M,N,K must be > 0 (got %d,%d,%d)", p.M, p.N, p.K);
  GROVE_CHECK(p.K % 32 == 0, GROVE_E_SHAPE, "gemm: K=%d must be a multiple of 32 (pad the operands)", p.K);
  GROVE_CHECK(p.K % p.a_taps == 0 && (p.K / p.a_taps) % 32 == 0, GROVE_E_SHAPE,
              "gemm: K/a_taps must be a multiple of 32 (K=%d taps=%d)", p.K, p.a_taps);
  GROVE_CHECK(p.lda % 8 == 0 && p.ldb % 8 == 0, GROVE_E_ALIGN, "gemm: lda=%d ldb=%d must be multiples of 8", p.lda, p.ldb);
  GROVE_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, GROVE_E_ALIGN, "gemm: A/B must be 16-byte aligned");
  GROVE_CHECK((p.sA1 % 8) == 0 && (p.sA2 % 8) == 0 && (p.sB1 % 8) == 0 && (p.sB2 % 8) == 0, GROVE_E_ALIGN,
              "gemm: A/B batch strides must be multiples of 8 elements");
  GROVE_CHECK(p.c_dtype == GROVE_BF16 || p.c_dtype == GROVE_F32, GROVE_E_DTYPE, "gemm: bad c_dtype %d", p.c_dtype);
  GROVE_CHECK(!(p.accumulate && p.c_dtype != GROVE_F32), GROVE_E_DTYPE, "gemm: accumulate needs f32 C");
  GROVE_CHECK(!(p.aux && p.c_dtype != GROVE_BF16), GROVE_E_DTYPE, "gemm: aux needs bf16 C");
  if (!p.residual) p.ldr = 0;
  // vector epilogue needs 8-byte aligned rows for every epilogue operand; otherwise force the
  // scalar path by making ldc look unaligned to the kernel's vec4 test (handled via flag below).
  const bool c_al = p.c_dtype == GROVE_BF16 ? (((uintptr_t)p.C & 7) == 0) : (((uintptr_t)p.C & 15) == 0);
  const int vec_ok = c_al && (p.sC1 % 4 == 0) && (p.sC2 % 4 == 0) &&
                      (!p.aux || ((uintptr_t)p.aux & 7) == 0) &&
                      (!p.residual || ((((uintptr_t)p.residual & 7) == 0) && p.sR1 % 4 == 0 && p.sR2 % 4 == 0)) &&
                      (!p.bias || ((uintptr_t)p.bias & 7) == 0);
  hipStream_t s = (hipStream_t)stream;
  const bool bk64 = (p.K % 64 == 0) && ((p.K / p.a_taps) % 64 == 0) && g_gemm_bk != 32;
  // Tile choice by block-round quantisation. All variants keep 2 blocks per CU resident (512 slots); a block of a
  // partial last round runs alone on its CU at about half the CU's rate, so time ~ ceil(tiles / 512) * tile_rows.
  //   128 x 128: the default.   192 x 128: e.g. M = 2812, N = 4096 -> 480 tiles = ONE round instead of 1.375.
  //   128 x 64 : only when even 128 x 128 tiles cannot fill the chip.
  const long bt = (long)p.batch1 * p.batch2;
  const long tn128 = (p.N + 127) / 128;
  const long t128 = (long)((p.M + 127) / 128) * tn128 * bt;
  const long t192 = (long)((p.M + 191) / 192) * tn128 * bt;
  // the pipelined 256 x 256 kernel: plain (un-gathered, un-batched, un-split) GEMMs whose epilogue operands allow 16-byte accesses
  const bool wide_ok = (((uintptr_t)p.C & 15) == 0) && (p.ldc % (p.c_dtype == GROVE_BF16 ? 8 : 4) == 0) &&
                       (!p.aux || ((uintptr_t)p.aux & 15) == 0) && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                       (!p.residual || ((((uintptr_t)p.residual & 15) == 0) && p.ldr % 8 == 0));
  const bool pp_act = pp_act_ok(p);  // compiled-in epilogues
  const bool maps = p.n_group || p.k_group;
  const bool have_ws = t_call.mode != 0 || t_call.image != nullptr;  // no work-list image given: only the non-persistent kernels can run
  const bool p256_ok = have_ws && g_gemm_glds && bk64 && (!p.a_idx || (long)p.a_taps * p.M >= 8) && bt == 1 && p.split_k <= 1 && !p.accumulate && wide_ok && p.N % 8 == 0 && pp_act;
  // Tile choice by a measured cost model (tools/bench_gemm.py tiles, microseconds): time = rounds of resident blocks x
  // (K tiles x per-K-tile time + fixed per-tile time). The 128- and 192-row kernels keep 2 blocks per CU (512 slots; a
  // lone block of a partial round still takes a full round); the pipelined kernels are persistent, one block per CU,
  // and a partly filled chip runs each block faster (L2, clocks).
  const double nk64 = p.K / 64.0;
  const double out_scale = (p.c_dtype == GROVE_F32 ? 2.0 : 1.0) + (p.aux ? 1.0 : 0.0);
  auto rounds = [](long tiles, long slots) { return (double)((tiles + slots - 1) / slots); };
  const double c128 = rounds(t128, 512) * (nk64 * 0.88 + 7.0 + 4.7 * out_scale);
  const double c192 = rounds(t192, 512) * (nk64 * 1.38 + 1.5 * (7.0 + 4.7 * out_scale));
  const long tn256 = (p.N + 255) / 256;
  const long tp256 = (long)((p.M + 255) / 256) * tn256, tp192 = (long)((p.M + 191) / 192) * tn256;
  auto pp_cost = [&](long tiles, double kt, double fixed) {
    const double r = rounds(tiles, 256);
    const sk_plan pl = plan_stream_k(tiles, p.K / 64, 256, bk64 ? g_gemm_stream_k : 0);
    if (pl.S) return pl.kt_units * kt + r * fixed;  // the tail is dealt out by K tiles: every CU busy for the same time
    const double fill = (double)tiles / (r * 256.0);
    return r * (nk64 * kt * (0.7 + 0.3 * fill) + fixed);
  };
  const double cp256 = pp_cost(tp256, 1.5, 3.0 + 3.0 * out_scale);
  const double cp192 = pp_cost(tp192, 1.17, 0.8 * (3.0 + 3.0 * out_scale));
  const double c_old = c128 < c192 ? c128 : c192;
  if (p.b_group_rows) {  // grouped B: the plain 256-row pipelined instance only
    GROVE_CHECK(p256_ok && !p.a_idx && !maps && p.act == GROVE_ACT_NONE && p.alpha == 1.f && !p.scale_ptr && p.c_dtype == GROVE_BF16 && !p.aux && !p.c_idx && !p.r_idx &&
                    !p.residual && p.b_group_rows > 0 && p.b_group_rows % 256 == 0 && p.M % p.b_group_rows == 0,
                GROVE_E_SHAPE, "gemm: b_group_rows needs the plain pipelined kernel (bf16 C, no epilogue operands beyond bias, no row maps), groups of whole 256-row "
                               "tiles (b_group_rows = %d, M = %d)", p.b_group_rows, p.M);
    g_gemm_last_variant = GROVE_GEMM_PP256;
    return launch_pp_act<256, false, -1, false, true>(p, s);
  }
  if (p256_ok && (g_gemm_tile_m == 193 || g_gemm_tile_m == 256)) {
    g_gemm_last_variant = g_gemm_tile_m == 256 ? (p.a_idx ? GROVE_GEMM_PP256_GATHER : GROVE_GEMM_PP256) : (p.a_idx ? GROVE_GEMM_PP192_GATHER : GROVE_GEMM_PP192);
    if (p.a_idx) return g_gemm_tile_m == 256 ? launch_pp<256, true>(p, s) : launch_pp<192, true>(p, s);
    return g_gemm_tile_m == 256 ? launch_pp<256, false>(p, s) : launch_pp<192, false>(p, s);
  }
  if (maps) {  // only the pipelined kernel implements the padded-head maps
    GROVE_CHECK(p256_ok && p.c_dtype == GROVE_BF16 && !p.aux && p.n_group % 8 == 0 && p.n_pad % 8 == 0 && p.k_group % 8 == 0 && p.k_pad % 8 == 0 &&
                    (!p.k_group || (p.a_idx && p.a_taps == 1 && p.K < 65536)) && (!p.n_group || p.N % p.n_group == 0),
                GROVE_E_SHAPE, "gemm: n_group/k_group maps need the pipelined kernel (bf16 C, no aux; k map: gathered A with one tap)");
    g_gemm_last_variant = cp256 <= cp192 ? (p.a_idx ? GROVE_GEMM_PP256_GATHER : GROVE_GEMM_PP256) : (p.a_idx ? GROVE_GEMM_PP192_GATHER : GROVE_GEMM_PP192);
    if (p.a_idx) return cp256 <= cp192 ? launch_pp<256, true>(p, s) : launch_pp<192, true>(p, s);
    return cp256 <= cp192 ? launch_pp<256, false>(p, s) : launch_pp<192, false>(p, s);
  }
  if (p.act == GROVE_ACT_SWIGLU_BWD) {  // only the pipelined kernel's epilogue implements it
    GROVE_CHECK(p256_ok && !p.a_idx && p.residual && !p.bias && !p.aux && !p.c_idx && !p.r_idx && !maps && p.c_dtype == GROVE_BF16 && !p.scale_ptr &&
                    !p.residual_mul && p.ldc >= 2 * p.N && p.ldr >= 2 * p.N && p.ldc % 8 == 0,
                GROVE_E_SHAPE, "gemm: act SWIGLU_BWD needs the pipelined kernel (plain un-batched bf16 GEMM, K %% 64 == 0, N %% 8 == 0) with residual = "
                               "gate | up [M, >= 2N], C [M, >= 2N] and no bias / aux / row maps / scale");
    g_gemm_last_variant = cp256 <= cp192 ? GROVE_GEMM_PP256 : GROVE_GEMM_PP192;
    return cp256 <= cp192 ? launch_pp<256, false>(p, s) : launch_pp<192, false>(p, s);
  }
  if (p.act == GROVE_ACT_SWIGLU_PAIR) {  // only the pipelined kernel's epilogue implements it
    GROVE_CHECK(p256_ok && p.N % 16 == 0 && p.c_dtype == GROVE_BF16 && !p.residual && p.ldc % 4 == 0 && (p.ld_aux % 4 == 0), GROVE_E_SHAPE,
                "gemm: act SWIGLU_PAIR needs the pipelined kernel (plain un-batched bf16 GEMM, K %% 64 == 0, N %% 16 == 0, aligned operands)");
    g_gemm_last_variant = cp256 <= cp192 ? GROVE_GEMM_PP256 : GROVE_GEMM_PP192;
    return cp256 <= cp192 ? launch_pp<256, false>(p, s) : launch_pp<192, false>(p, s);
  }
  // (under 48 tiles the persistent kernel only pays when the stream-K plan spreads the K range over the idle CUs)
  const bool few_tiles_cut = tp192 < 48 && bk64 && plan_stream_k(tp192, p.K / 64, num_cus(), g_gemm_stream_k).S > 0;
  if (p256_ok && g_gemm_tile_m == 0 && g_gemm_tile_n == 0 && (tp192 >= 48 || few_tiles_cut) && (cp256 < c_old || cp192 < c_old))
    {
      g_gemm_last_variant = cp256 <= cp192 ? (p.a_idx ? GROVE_GEMM_PP256_GATHER : GROVE_GEMM_PP256) : (p.a_idx ? GROVE_GEMM_PP192_GATHER : GROVE_GEMM_PP192);
      if (p.a_idx) return cp256 <= cp192 ? launch_pp<256, true>(p, s) : launch_pp<192, true>(p, s);
      return cp256 <= cp192 ? launch_pp<256, false>(p, s) : launch_pp<192, false>(p, s);
    }
  int variant = 128;
  if (g_gemm_tile_m == 192 || (g_gemm_tile_m == 0 && t128 > 300 && c192 <= c128)) variant = 192;
  if (p.aux_grad || p.residual_mul) variant = 128;  // the 192-row variant's epilogue does not carry them
  const bool narrow = g_gemm_tile_n == 64 || (g_gemm_tile_n == 0 && ((t128 < 160 && p.N > 64) || p.N <= 64));  // N <= 64: half of a 128-wide tile would be padding
  g_gemm_last_variant = narrow ? GROVE_GEMM_T128X64 : variant == 192 && g_gemm_glds ? GROVE_GEMM_T192X128 : GROVE_GEMM_T128X128;
  if (narrow) {
    if (g_gemm_glds) return bk64 ? launch<64, true, 2, 4>(p, vec_ok, s) : launch<32, true, 2, 4>(p, vec_ok, s);
    return bk64 ? launch<64, false, 2, 4>(p, vec_ok, s) : launch<32, false, 2, 4>(p, vec_ok, s);
  }
  if (variant == 192 && g_gemm_glds) return bk64 ? launch<64, true, 4, 6>(p, vec_ok, s) : launch<32, true, 4, 6>(p, vec_ok, s);
  if (g_gemm_glds) return bk64 ? launch<64, true, 4, 4>(p, vec_ok, s) : launch<32, true, 4, 4>(p, vec_ok, s);
  return bk64 ? launch<64, false, 4, 4>(p, vec_ok, s) : launch<32, false, 4, 4>(p, vec_ok, s);
}
