/*
 * grove_hip.h — C-ABI of libgrove_hip.so, the MI355X (gfx950) kernel library behind
 * grove_amd's GROVEForCausalLM hot path.
 *
 * The reference (ekazakos/grove) has NO native layer: every device kernel it runs comes from
 * PyTorch/cuBLAS/cuDNN, flash-attn and DeepSpeed (SURVEY.md §2 "Third-party native/fused ops").
 * Each entry point below therefore cites the reference op sequence (file:line under
 * /root/reference) that it replaces, not a reference FFI symbol.
 *
 * Conventions
 *  - plain pointers + sizes only; every buffer — inputs, outputs AND workspaces — is caller-owned device memory (in
 *    practice a torch tensor). The library never allocates or frees device memory (there is no hipMalloc / hipFree /
 *    hipMemcpy in it) and retains no pointer across calls; the only kernels that need memory beyond their operands are the
 *    persistent GEMMs, whose work-list image and stream-K scratch the caller supplies (grove_gemm_workspace, below).
 *  - all calls are asynchronous on `stream` (a hipStream_t passed as void*) and graph-capturable on any stream, first use
 *    of a shape included (nothing is allocated or copied behind the call).
 *  - no mutable state shared between threads drives results: the grove_*_set_* switches are process-wide TEST / A-B knobs
 *    (kernel-selection overrides; every setting computes the same function), never needed by a caller; the
 *    grove_*_last_* getters are measurement aids of the calling process.
 *  - return 0 on success, negative GROVE_E_* otherwise; grove_last_error() has the text.
 *  - bf16 tensors are raw uint16 storage ("bf16"); accumulation is always fp32.
 */
#ifndef GROVE_HIP_H
#define GROVE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GROVE_OK 0
#define GROVE_E_SHAPE (-1)
#define GROVE_E_DTYPE (-2)
#define GROVE_E_ALIGN (-3)
#define GROVE_E_WORKSPACE (-4)
#define GROVE_E_HIP (-5)

enum grove_act {
  GROVE_ACT_NONE = 0,
  GROVE_ACT_RELU = 1,      /* SAM adapters / decoder MLP / text_hidden_fcs / bbox head */
  GROVE_ACT_GELU = 2,      /* erf GELU: mm_projector (llava_with_region_arch.py:17), SAM MLPBlock */
  GROVE_ACT_QUICKGELU = 3, /* CLIP MLP (modeling_clip.py:336-348) */
  GROVE_ACT_SILU = 4,      /* LLaMA SwiGLU gate */
  GROVE_ACT_SIGMOID = 5,
  GROVE_ACT_SWIGLU_PAIR = 6, /* LlamaMLP's silu(gate) * up folded into the gate|up GEMM: B rows interleaved [4 gate, 4 up] per 8, C gets
                               N / 2 columns (the product), aux (optional, row stride ld_aux) the un-interleaved gate | up pre-activations;
                               pipelined kernel only */
  GROVE_ACT_SWIGLU_BWD = 7   /* round 4: the BACKWARD of that product folded into the down-projection's dgrad GEMM. With d = the GEMM's own
                               output (d a = dy . W_down, N = I columns, rounded to bf16 as the unfused path stores it) and the saved
                               pre-activations gate | up = residual[m, n], residual[m, N + n] (row stride ldr >= 2N):
                                 C[m, n] = d u (s + g s (1 - s)),  C[m, N + n] = d g s,  s = sigmoid(g)      (C: bf16, ldc >= 2N)
                               = grove_swiglu_bwd's d(gate | up), bit for bit, without the [M, I] round trip. No bias / aux / c_idx /
                               n_group; pipelined kernel only */
};

enum grove_dtype { GROVE_BF16 = 0, GROVE_F32 = 1 };

int grove_version(void);
/* copies the last error message of the calling thread into buf (NUL-terminated) */
int grove_last_error(char* buf, size_t n);
/* Deterministic mode (round 4; the Python host turns it on when GROVE_DETERMINISTIC=1): `tickets` = n >= 64 ZEROED 32-bit words of
 * device memory owned by the caller (the library allocates nothing) that stay valid until the mode is turned off with (NULL, 0);
 * one ring per process, on the device the caller launches on. Every sum that otherwise meets in global
 * memory through fp32 atomics gets a fixed order, so two runs of the same program on the same inputs agree bit for bit — the mode the
 * stream-overlap race tests assert equality in. How: split-K / cut-tail GEMM launches run whole-K tiles (grove_gemm_bf16,
 * grove_gemm_tn_bf16); the block-level reductions (cross entropy and box-loss sums, norm dweight / dbias, dot, sum of squares, the
 * few-key attention dK / dV, the box heads' weight gradients) make their blocks add in blockIdx order through a ticket
 * (csrc/common.h: det_wait / det_pass); scatter_add walks its source rows in order, colsum uses one block per column group.
 * Slower (the add phases are serialised) and meant for race hunting, not for training runs. Results differ from the default mode's
 * only in summation order. */
int grove_set_deterministic(void* tickets, int32_t n);
int grove_deterministic(void);
/* sizeof(struct <name>) as compiled into the library (binding self-check), -1 if unknown */
int grove_sizeof(const char* name);

/* ------------------------------------------------------------------------------------------
 * GEMM:  C[m, n] = epilogue( alpha * sum_k A[row_a(m,k), k] * B[n, k] )
 *   A: bf16 [*, lda] row-major (activations), B: bf16 [N, ldb] row-major (nn.Linear weight
 *   layout, i.e. "NT" GEMM), fp32 accumulate on v_mfma_f32_16x16x32_bf16.
 *   epilogue(v):  v += bias[n];  aux[m,n] = v (optional pre-activation copy for backward; act'(v) if aux_grad);
 *                 v = act(v);  v *= scale (scale = *scale_ptr, tanh'd if scale_tanh);
 *                 v += residual[row_r(m), n] (v *= ... if residual_mul);  (C += v if accumulate)  store as bf16 / f32.
 *   Gather/scatter row maps (int32, -1 = zero row on A / skipped row on C):
 *     a_idx[tap*M + m] with tap = k / (K / a_taps)  -> implicit-GEMM convolution and window
 *     partition without materialising im2col; c_idx[m], r_idx[m] for scatter / broadcast.
 *   Batched over batch1*batch2 with independent element strides.
 * Replaces: every nn.Linear / torch.bmm / `@` / Conv2d(k=stride) / Conv3d 3x3x3 on the path:
 *   modeling_clip.py:174-180,269-271,279,319,331,344-348,594; image_encoder.py:43,152-168,
 *   305-326,484-486; common.py:21-26; transformer.py:205-242; llava_with_region_arch.py:16-19;
 *   HF LlamaAttention/LlamaMLP (llava_llama.py:100-112); GROVE.py:75-79; mask_decoder.py:80-84.
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_gemm_params {
  const void* A;
  const void* B;
  void* C;
  const void* bias;       /* bf16 [N] or NULL */
  const void* residual;   /* bf16 [*, ldr] or NULL */
  void* aux;              /* bf16 [*, ldc] pre-activation copy or NULL (same row map as C) */
  const float* scale_ptr; /* device scalar or NULL */
  const int32_t* a_idx;   /* [a_taps, M] or NULL */
  const int32_t* c_idx;   /* [M] or NULL */
  const int32_t* r_idx;   /* [M] or NULL (NULL -> residual uses the C row) */
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2; /* batch strides in elements */
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldr;
  int32_t batch1, batch2;
  int32_t a_taps;     /* >=1 */
  int32_t act;        /* enum grove_act */
  int32_t c_dtype;    /* enum grove_dtype */
  int32_t accumulate; /* C += result (c_dtype f32 only) */
  int32_t scale_tanh; /* scale = tanh(*scale_ptr) */
  float alpha;
  int32_t split_k;    /* 0 = auto (accumulating f32 GEMMs only), 1 = off, >1 = K range split over blockIdx.z,
                         partials combined with fp32 atomics into the pre-initialised C */
  int32_t ld_aux;     /* row stride of aux; 0 = ldc */
  /* Padded-head layouts (SAM: head dim 80 stored as 96) without multiplying by the padding — pipelined kernel only:
   *   n_group / n_pad: logical output column n is stored at n + (n / n_group) * n_pad and the n_pad columns after every
   *                    group are written as zeros (N, bias and B are the compact sizes);
   *   k_group / k_pad: logical reduction index k of A is read from column k + (k / k_group) * k_pad (K and B compact;
   *                    needs a_idx: the gathered-A instance). Groups and pads are multiples of 8; 0 = off. */
  int32_t n_group, n_pad, k_group, k_pad;
  /* Activation backward folded into the two GEMMs either side of it (replaces autograd's GeluBackward / the elementwise
   * dY * act'(x) pass between an MLP's two dgrad GEMMs — image_encoder.py:43 MLPBlock):
   *   aux_grad:     aux receives act'(v), the activation's derivative at the pre-activation, instead of v itself;
   *   residual_mul: the residual operand multiplies the result (v *= residual[row_r(m), n]) instead of being added. */
  int32_t aux_grad, residual_mul;
  /* Temporal tap skipping (round 4; pipelined gathered-A kernel with 256-row tiles, a_taps % 3 == 0 — the Conv3d 3 x 3 x 3 adapters,
   * image_encoder.py:43 / modeling_clip.py:594): a PROMISE about a_idx that lets the planner leave out K ranges that multiply zeros.
   * The M output rows are frames of a_frame_rows rows (a multiple of 256), a_frames frames per group, and the taps are three equal
   * groups in K order (temporal offset -1, 0, +1): a_idx[tap][m] == -1 for EVERY tap of the first group when m lies in the first
   * frame of its group, and for every tap of the last group when m lies in the last frame (conv3d_gather_index builds exactly that:
   * the temporal zero padding). Those tiles then run 2/3 of the K range; whole tiles stay bit-identical (the skipped products only
   * ever added +0.0), a stream-K tail is cut at other K tiles (deterministic, different fp32 sum order). 0 / 0 = no promise. */
  int32_t a_frame_rows, a_frames;
  /* Grouped B (round 6; the plain pipelined kernel with 256-row tiles only — the Winograd form of the Conv3d adapters, see
   * grove_wino3d_*): the M rows are groups of b_group_rows rows (a multiple of 256) and group g multiplies its own weight matrix,
   * the g-th of M / b_group_rows matrices [N, ldb] stacked in B (group g at B + g * N * ldb elements). One launch for the 64
   * independent [tiles, C] x [C, C] products of the 64 transform points. 0 = off. Needs a_idx NULL, no batch, no split, bf16 C.
   * (One int32 in what was the struct's tail padding: sizeof(grove_gemm_params) is what it was in round 5.) */
  int32_t b_group_rows;
} grove_gemm_params;

/* Workspaces of the persistent GEMMs (`grove_<op>_workspace_bytes` + `workspace, ws_bytes` arguments).
 * The persistent pipelined kernels (GROVE_GEMM_PP*) walk a host-made WORK LIST (which output tiles / K ranges each of the
 * resident blocks computes) and, when the last round of tiles is cut into K ranges (stream-K), park the raw fp32 partial
 * tiles in SCRATCH until the fix-up launch sums them. Both belong to the caller:
 *   1. grove_gemm_make_plan(p, &plan)                  host only: which kernel the call will run, image_bytes, scratch_bytes, key;
 *   2. grove_gemm_plan_image(p, host_buf, bytes)  host only: writes the image (work list + fix-up list) into a host buffer;
 *      the caller uploads it to device memory ONCE per plan.key (read-only afterwards, shareable by all streams, by bf16 and
 *      fp8 launches, and by every epilogue: the key depends on tile geometry only);
 *   3. grove_gemm_bf16(p, &ws, stream)            ws.image = that device image, ws.scratch = scratch_bytes of device memory
 *      that no other launch IN FLIGHT uses (one scratch buffer per stream is enough: launches of a stream are ordered).
 * A call without an image (ws NULL / image NULL) runs the non-persistent kernels; a problem only the persistent kernel
 * implements (padded-head maps, SWIGLU_PAIR, fp8 pipelined) then fails with GROVE_E_WORKSPACE. Too small an image / scratch:
 * GROVE_E_WORKSPACE. grove_gemm_workspace_bytes(p) = image_bytes + scratch_bytes of the plan (0: none needed). */
typedef struct grove_gemm_workspace {
  const void* image;   /* device copy of the plan image (16-byte aligned) or NULL */
  size_t image_bytes;
  void* scratch;       /* stream-K partial tiles (16-byte aligned) or NULL when the plan's scratch_bytes is 0 */
  size_t scratch_bytes;
} grove_gemm_workspace;
typedef struct grove_gemm_plan {
  int32_t variant;                 /* enum grove_gemm_variant the call will launch (bf16 entry; 0 for the fp8 entry) */
  int32_t bm, tiles_m, tiles_n;    /* persistent kernels: tile rows (192 / 256), output tiles */
  int32_t k_tiles, grid, stream_k; /* K tiles of 64 (128 fp8 codes), resident blocks, K tiles per stream-K part (0 = whole tiles) */
  int32_t reserved;
  int64_t image_bytes, scratch_bytes;
  uint64_t key;                    /* equal keys (on devices with equal CU counts) <=> identical images */
} grove_gemm_plan;
int grove_gemm_make_plan(const grove_gemm_params* p, grove_gemm_plan* out);
int grove_gemm_plan_image(const grove_gemm_params* p, void* host_image, size_t bytes);
size_t grove_gemm_workspace_bytes(const grove_gemm_params* p);
int grove_gemm_bf16(const grove_gemm_params* p, const grove_gemm_workspace* ws, void* stream);
/* Which kernel the last grove_gemm_bf16 call launched (measurement aid: bench.py prices each kernel on its own launches). */
enum grove_gemm_variant {
  GROVE_GEMM_T128X128 = 1, /* gemm_nt_kernel, 128 x 128 tile */
  GROVE_GEMM_T192X128 = 2, /* gemm_nt_kernel, 192 x 128 tile */
  GROVE_GEMM_T128X64 = 3,  /* gemm_nt_kernel, 128 x 64 tile */
  GROVE_GEMM_PP256 = 4,        /* gemm_nt_pp_kernel<256, false>: persistent pipelined 256 x 256 x 64 */
  GROVE_GEMM_PP192 = 5,        /* gemm_nt_pp_kernel<192, false>: persistent pipelined 192 x 256 x 64 */
  GROVE_GEMM_PP256_GATHER = 6, /* gemm_nt_pp_kernel<256, true>: the same with gathered A rows (a_idx) */
  GROVE_GEMM_PP192_GATHER = 7  /* gemm_nt_pp_kernel<192, true> */
};
int grove_gemm_last_variant(void);
/* The epilogue compiled into the pipelined kernel of the last call — its third template argument, as profilers print it:
 * -1 = plain (act NONE, alpha 1, no scale), else the enum grove_act value. Meaningless after a non-pipelined launch. */
int grove_gemm_last_epilogue(void);
/* Stream-K tail of the pipelined kernels. When the output tiles do not fill a last round of the persistent grid (at most half
 * of the CUs would work), that round's tiles are cut into 2..4 equal K ranges, one block each; the raw fp32 parts go through the
 * caller's scratch (grove_gemm_workspace) and a fix-up launch sums them in K order and runs the epilogue (deterministic). TEST / A-B
 * knob, like every grove_*_set_*: plan, image and launch must run under the same setting. mode 1 (default) =
 * where the cost model says it pays, 0 = never (whole tiles only), 2 = wherever it applies (tests).
 * grove_gemm_last_stream_k: K tiles per part of the last pipelined launch, 0 = it ran whole tiles only. */
int grove_gemm_set_stream_k(int mode);
/* Grid of the persistent kernels: 0 (default) = one block per CU, n < CU count = n resident blocks — leaves CUs to an overlapped
 * RCCL collective at N > 1, which would otherwise make the blocks dealt to its CUs wait for another block's whole share (TEST / A-B
 * knob like the other setters: plan, image and launch under one setting; `bench.py --gemm_blocks n`). */
int grove_gemm_set_persistent_blocks(int n);
/* The current setting (0 = one per CU); the persistent weight-gradient kernel (grove_gemm_tn_bf16) follows the same cap. Round 4:
 * GroveEngine's gradient exchange lowers the cap to (CUs - 16) from the first bucket it hands to RCCL until the optimizer step has
 * waited for the collectives (train.GradExchange: the RULE, not an A/B value), because a persistent block needs a whole CU
 * (128 KB LDS, 8 x 256 VGPRs) and RCCL's channel blocks can only start on a CU a GEMM block has left. */
int grove_gemm_persistent_blocks(void);
/* A/B knob: 0 = ignore a_frame_rows / a_frames (every tile runs the whole K range), 1 (default) = skip. */
int grove_gemm_set_tap_skip(int on);
/* Host-only view of the persistent kernels' work list (needs no device): what each of the `grid` = min(tiles, num_cus) (num_cus with a
 * stream-K tail) blocks does for tiles_m x tiles_n output tiles of bm x 256 (bm = 192 / 256) with nk K tiles of 64.
 * list: int32 [rows][grid][4] — row 0 = {K tiles of the block's stream, its segments, 0, 0}, row 1 + i = segment {m0, n0, k0 | k1 << 16,
 * part}: part 0 = whole tile with epilogue, part 1 + s = K range [k0, k1) whose raw fp32 sum goes to workspace slot s.
 * fixups: int32 [n][4] = split tile {m0, n0, first slot, parts}. Returns rows (< 0: error / buffers too small). */
int grove_gemm_work_list(int bm, int tiles_m, int tiles_n, int nk, int num_cus, int mode, int32_t* list, int64_t list_cap,
                         int32_t* fixups, int64_t fixups_cap, int* n_fixups, int* k_tiles_per_part);
int grove_gemm_last_stream_k(void);
/* A/B staging variant: 1 = LDS-DMA (global_load_lds, default), 0 = register staged */
int grove_gemm_set_staging(int use_lds_dma);
/* macro-tile N: 0 = auto (by wave quantisation), 64 or 128 = forced (for A/B measurements) */
int grove_gemm_set_tile_n(int tile_n);
/* macro-tile M: 0 = auto, 128 or 192 = forced */
int grove_gemm_set_tile_m(int tile_m);
/* K tile: 0 = auto (64 when K % 64 == 0), 32 = forced (A/B measurements) */
int grove_gemm_set_bk(int bk);

/* ------------------------------------------------------------------------------------------
 * "TN" GEMM for weight gradients:  C[m, n] += f * sum_k A[k, m] * B[row_b(k, n), n mod (N / b_taps)]
 *   A: bf16 [K, lda] (dY: tokens x out-features), B: bf16 [K, ldb] (X: tokens x in-features), both K-major as
 *   the backward pass holds them; C: fp32 [M, ldc], always accumulated (fp32 atomics when K is split).
 *   b_idx[tap*K + k] (-1 = zero row): per-tap row gather of B — Conv3d/Conv2d weight gradient in one launch.
 *   f = alpha * (scale_ptr ? (scale_tanh ? tanh(*scale_ptr) : *scale_ptr) : 1).
 * Replaces autograd's weight-gradient GEMMs/convs of every trainable layer (train.py:279-316 freeze policy).
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_gemm_tn_params {
  const void* A;
  const void* B;
  float* C;
  const float* scale_ptr;
  const int32_t* b_idx; /* [b_taps, K] or NULL */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t b_taps;       /* >= 1; N / b_taps must be a multiple of 128 when > 1 */
  int32_t scale_tanh;
  int32_t split_k;      /* 0 = auto */
  float alpha;
  /* Temporal tap skipping (round 4; pipelined gathered kernel): the PROMISE of grove_gemm_params.a_frame_rows / a_frames for this
   * product — the K rows are frames of b_frame_rows rows (a multiple of 64), b_frames frames per group; the taps are three equal groups
   * in N order; b_idx[tap][k] == -1 for every tap of the first group when k lies in the first frame of its group and for every tap of
   * the last group when k lies in the last frame. Those K tiles are skipped (1 / b_frames of the K range of two thirds of the output
   * tiles) and the full tiles are dealt first. Whole tiles stay bit-identical. 0 / 0 = no promise. */
  int32_t b_frame_rows, b_frames;
  /* K-batched form (round 6; the pipelined kernel only — the Winograd weight gradient, see grove_wino3d_*): A and B are k_batches
   * stacked operands of K rows each ([k_batches * K, ld]) and batch b accumulates its own product into C + b * sC_batch
   * (elements; [M, ldc] each): sum_k A[b K + k, m] * B[b K + k, n]. One launch for the 64 transform points (64 x the output
   * tiles of one product fill the chip where one product's 25 tiles cannot). 0 / 1 = off. Needs b_idx NULL, K % 64 == 0.
   * overwrite (K-batched form only): C = f * product instead of C += (every tile whole: no cut tail, fixed sum order) — the
   * transform-point sums are temporaries, and zero-filling 64 x [M, N] fp32 first would cost a pass of its own. */
  int32_t k_batches, overwrite;
  int64_t sC_batch;
} grove_gemm_tn_params;
int grove_gemm_tn_bf16(const grove_gemm_tn_params* p, void* stream);
/* kernel choice of grove_gemm_tn_bf16: -1 = auto (default), 0 = the 128 x 128 kernel always, 1 = the persistent pipelined
 * 256 x 256 kernel whenever the problem is eligible (K % 64 == 0, no split-K, column tiles inside one tap) */
int grove_gemm_tn_set_pipelined(int mode);
/* The pipelined kernel cuts the tiles of a partial last round into 2-4 equal K ranges when that shortens the round (675 tiles on
 * 256 CUs: 2 rounds + 163 tiles -> 2 + 2/3); a cut tile's ranges are added to C with fp32 atomics, so its sum order is not fixed
 * (as with split_k > 1). on: 0 = every tile whole, 1 = gathered (b_idx) launches only (default: the plain form does not gain),
 * 2 = every eligible launch. grove_gemm_tn_last_parts: K ranges per cut tile in the last pipelined launch. */
int grove_gemm_tn_set_split_tail(int on);
int grove_gemm_tn_last_parts(void);
/* A/B knob: 0 = ignore b_frame_rows / b_frames; grove_gemm_tn_last_skip: whether the last pipelined launch skipped. */
int grove_gemm_tn_set_tap_skip(int on);
int grove_gemm_tn_last_skip(void);

/* out[c, r] = in[r, c] for a batch of 2-D bf16 matrices (used for V^T, dY^T, X^T, NCHW<->NHWC).
 * rows beyond `rows` in the output's padded leading dim (ld_out > rows) are zero filled up to
 * pad_to columns. Replaces .transpose().contiguous() / .permute() copies
 * (modeling_clip.py:255; image_encoder.py:305-323; transformer.py:83-84). */
typedef struct grove_transpose_params {
  const void* in;
  void* out;
  int64_t s_in1, s_in2, s_out1, s_out2;
  int32_t rows, cols;     /* input is [rows, cols] with leading dim ld_in */
  int32_t ld_in, ld_out;  /* output is [cols, ld_out], columns [rows, pad_to) zero-filled */
  int32_t pad_to;
  int32_t batch1, batch2;
} grove_transpose_params;
int grove_transpose_bf16(const grove_transpose_params* p, void* stream);

/* Round 6b: many small transposes in one launch. items_dev: DEVICE array of n_items descriptors; item i transposes src (bf16 [rows, cols], row stride
 * ld_src) into dst (bf16 [cols, rows], row stride ld_dst) and owns the blocks tile0 <= block < next item's tile0, tile0 = the running sum of
 * ceil(rows / 64) * ceil(cols / 64) (total_tiles = the sum over all items). The box decoder's backward transposes its ~40 trainable weights
 * (prompt / mask decoder, transformer.py:151-242) this way at the start instead of one 8-15 us launch per linear on the serial dgrad chain. */
typedef struct grove_transpose_item {
  const void* src;
  void* dst;
  int32_t rows, cols, ld_src, ld_dst, tile0, pad_;
} grove_transpose_item;
int grove_transpose_many(const grove_transpose_item* items_dev, int32_t n_items, int32_t total_tiles, void* stream);

/* ------------------------------------------------------------------------------------------
 * Norms. x: bf16 [rows, ld_x]; y: bf16 or f32 [*, ld_y]; statistics fp32.
 * layernorm: nn.LayerNorm (modeling_clip.py:381-395,915; image_encoder.py:243-259;
 *   transformer.py:151-182) and LayerNorm2d in channels-last (common.py:32-43).
 *   out_idx (optional) scatters row r of the input to row out_idx[r] of y (SAM window partition,
 *   image_encoder.py:329-352; rows never written keep their caller-provided zero padding).
 * rmsnorm: HF LlamaRMSNorm (fp32 normalise, cast, times weight).
 * Residual-stream form (res != NULL): the pre-norm residual connections of the three towers
 *   (`hidden_states = residual + hidden_states` then the next block's norm: HF LlamaDecoderLayer; modeling_clip.py:386-396;
 *   image_encoder.py:256-258) fused into the norm that follows them, with the stream itself held in FP32:
 *   v = res[row] + x[row] (x = the branch output, bf16, or NULL = no update); res[row] = v; res_bf16[row] = bf16(v)
 *   (optional: [rows, C] contiguous, the copy a backward pass / a GEMM that consumes the stream reads); y = norm(v)
 *   (y == NULL: stream update only). The reference rounds the stream to bf16 after every block; here it is not rounded.
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_norm_params {
  const void* x;
  const void* weight; /* bf16 [C] */
  const void* bias;   /* bf16 [C] or NULL (rmsnorm) */
  void* y;
  float* mean;        /* [rows] or NULL (layernorm only; saved for backward) */
  float* rstd;        /* [rows] or NULL */
  const int32_t* out_idx;
  int32_t rows, C, ld_x, ld_y;
  int32_t y_dtype;
  float eps;
  float* res;      /* f32 [rows, ld_res] residual stream, updated in place, or NULL (plain norm of x) */
  void* res_bf16;  /* bf16 [rows, C] rounded copy of the updated stream, or NULL */
  int32_t ld_res;
} grove_norm_params;
int grove_layernorm_fwd(const grove_norm_params* p, void* stream);
int grove_rmsnorm_fwd(const grove_norm_params* p, void* stream);

typedef struct grove_norm_bwd_params {
  const void* x;      /* bf16 [rows, ld_x] forward input */
  const void* weight; /* bf16 [C] */
  const void* dy;     /* bf16 [*, ld_dy]; row r read at in_idx[r] if in_idx */
  void* dx;           /* bf16 [rows, ld_dx]; dx (+)= ... when accumulate */
  const float* mean;  /* layernorm: saved stats; rmsnorm: NULL */
  const float* rstd;
  float* dweight;     /* f32 [C] accumulated with atomics, or NULL */
  float* dbias;       /* f32 [C] or NULL */
  const int32_t* in_idx;
  int32_t rows, C, ld_x, ld_dy, ld_dx;
  int32_t accumulate; /* dx += */
  float eps;
} grove_norm_bwd_params;
int grove_layernorm_bwd(const grove_norm_bwd_params* p, void* stream);
int grove_rmsnorm_bwd(const grove_norm_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row softmax over attention scores, fp32 in -> bf16 probabilities, with the masks / biases the
 * three towers need. scores: f32 [batch, Lq, ld_s]; probs: bf16 [batch, Lq, ld_p], columns
 * [Lk, ld_p) are written as zero so that P can feed the PV GEMM with a padded K.
 *   causal: key j allowed iff j <= i + (Lk - Lq)           (HF Llama eager mask)
 *   kv_len[batch / heads]: keys >= kv_len masked            (right padding, llava arch :392-418)
 *   rel-pos: s[i, j] += relh[b, i, j / kw] + relw[b, i, j % kw]
 *            (image_encoder.py:420-458 add_decomposed_rel_pos)
 * Replaces nn.functional.softmax at modeling_clip.py:305, image_encoder.py:317,
 * transformer.py:235 and the fp32 softmax of HF eager Llama attention.
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_softmax_params {
  const float* scores;
  void* probs;
  const int32_t* kv_len; /* [batch / heads] or NULL */
  const float* rel;      /* f32 [batch, Lq, rel_kh + rel_kw] or NULL */
  int32_t batch, heads, Lq, Lk, ld_s, ld_p;
  int32_t causal;
  int32_t rel_kh, rel_kw;
} grove_softmax_params;
int grove_softmax_fwd(const grove_softmax_params* p, void* stream);

/* dS = P * (dP - rowsum(P * dP)) * scale; optionally reduces dS over kw / kh into drel.
 * dP: f32 [batch, Lq, ld_s]; P: bf16; dS: bf16 [batch, Lq, ld_p] (pad columns zero). */
typedef struct grove_softmax_bwd_params {
  const float* dprobs;
  const void* probs;
  void* dscores;
  float* drel; /* f32 [batch, Lq, rel_kh + rel_kw] or NULL */
  int32_t batch, Lq, Lk, ld_s, ld_p;
  int32_t rel_kh, rel_kw;
  float scale;
} grove_softmax_bwd_params;
int grove_softmax_bwd(const grove_softmax_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused (flash-style) attention, forward and backward; no score matrix in HBM.
 * q/k/v/o (+ gradients): bf16, row r of batch b at base + b*s? + r*ld_?, head h at column h*hs,
 * hs in {32, 64, 96, 128} (zero-padded head dims: pad columns must be exact zeros).
 *   S = alpha * (Q K^T + rel[q][j / rel_kw] + rel[q][rel_kh + j % rel_kw]), causal / kv_len masks as
 *   grove_softmax_fwd; O = softmax(S) V; lse[b*H+h, q] = log sum_j exp(S[q, j]).
 * bwd: needs o, d_o, lse and a delta scratch [B*H, Lq] f32; writes dq, dk, dv (bf16, every valid element)
 * and optionally drel (bf16), the gradient of the pre-scaled rel table (bias and d rel run on the matrix cores).
 * Replaces bmm+softmax+bmm at modeling_clip.py:279-319, image_encoder.py:310-319 (+ add_decomposed_rel_pos
 * :420-458) and flash-attn-2 inside HF LlamaAttention (llava_llama.py:100-109), forward and backward.
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_flash_attn_params {
  const void* q; const void* k; const void* v;
  void* o;
  const void* d_o;
  void* dq; void* dk; void* dv;
  float* lse;
  float* delta;
  const int32_t* kv_len; /* [B] or NULL */
  const void* rel;       /* bf16 [B*H, Lq, rel_ld] or NULL: rel / alpha; h-bins at columns 0.., w-bins at rel_kh.. */
  void* drel;            /* bwd: bf16 [B*H, Lq, rel_ld] gradient w.r.t. rel / alpha, or NULL */
  int64_t sq, sk, sv, so, sdo, sdq, sdk, sdv; /* batch strides in elements */
  int32_t B, H, Lq, Lk, hs;
  int32_t ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
  int32_t causal, rel_kh, rel_kw, rel_ld; /* key j -> bins j / rel_kw and rel_kh + j % rel_kw */
  float alpha;
  int32_t hs_valid; /* real head dim inside the hs-wide slot (0 = hs). SAM: 80 in 96 — with Lq == Lk in (192, 208] and 32 rel bins
                       (the 14 x 14 windows of image_encoder.py:329-353) the LDS-resident window kernels run (win_attn.hip): nothing
                       is multiplied by the padding, and the pad columns of o / dq / dk / dv are written as zeros */
  const int32_t* q_valid; /* window kernels only: int32 [B][2] = {vy, vx} or NULL. Batch b keeps the top-left vy x vx positions of its
                       (Lq / rel_kw) x rel_kw window as QUERIES (window_partition's zero padding, image_encoder.py:329-353: real as keys,
                       dropped as queries): rows of o / lse / dq / drel at the other positions are NOT written and d_o there is NOT read.
                       The general kernels ignore the field and process every row (the caller then supplies zero d_o rows there);
                       grove_flash_attn_window_kernels_on() tells which family runs a fitting problem */
  const int32_t* o_map; /* with q_valid, window kernels only: int32 [B * Lq] or NULL. Given, o and d_o are NOT in the layout of q but in
                       TOKEN order: the row of (batch b, position i) is o_map[b * Lq + i] (valid at every position q_valid keeps), head h at
                       column h * o_hs (o_hs = 0: hs; SAM: 80 — compact heads), row strides ld_o / ld_do, so / sdo unused. window_unpartition
                       (image_encoder.py:356-384) and its backward then need no gather: the projection reads / writes plain matrices */
  int32_t o_hs;
  int32_t g_tok;  /* backward, window kernels with o_map only (round 6; fills the padding after o_hs): 1 = dq / dk / dv are in TOKEN order too —
                     the row of (batch b, position i) is o_map[b * Lq + i], head h at column h * o_hs, row strides ld_dq / ld_dk / ld_dv (sdq /
                     sdk / sdv unused), no pad columns; rows of padded positions do not exist. The qkv dgrad is then a plain GEMM over the
                     tokens instead of a gathered one over the windowed rows (image_encoder.py:329-353 window_partition's backward). */
  const void* pad_k; const void* pad_v; /* with q_valid, window kernels only: bf16 rows [H*hs] = k / v of a padded position (a zero token's
                       projection = the bias), or NULL. Given, the k / v rows at padded positions are NOT read (the caller need not
                       fill them) and dk / dv there are NOT written */
  const float* rope; /* backward, general kernels only (round 4): f32 [positions, hs] = cos[hs / 2] | sin[hs / 2] per position, or NULL.
                       Given, q and k are the ROTATED tensors (rotate-half RoPE, HF apply_rotary_pos_emb) and dq / dk leave the kernels
                       already rotated BACK to the un-rotated projections' gradients: dx1 = dy1 cos + dy2 sin, dx2 = dy2 cos - dy1 sin on
                       the halves of every head, in the epilogues of the two backward kernels (a lane holds both halves of its dims) — the
                       separate inverse-RoPE pass over dq | dk (46 MB read + written per LLaMA layer) is gone. Query i sits at position
                       i + Lk - Lq (the causal mask's alignment), key j at position j. */
  const void* rel_table; /* window kernels only (round 6b): bf16 [2][64 * 80] or NULL — the decomposed rel-pos terms are then computed INSIDE the
                       kernels from the relative-position embeddings themselves (image_encoder.py:420-458 add_decomposed_rel_pos + :387-417
                       get_rel_pos with q_size == k_size), no grove_rel_bias_fwd / _bwd launch and no rel / d rel stream through HBM. First image
                       T [64 rows][80 dims]: with n = Lq / rel_kw window rows, row r < 2n - 1 = rel_pos_h[2n - 2 - r] / alpha (the embedding of
                       key row - query row = r - (n - 1)), row 2n - 1 + r (r < 2 rel_kw - 1) = rel_pos_w[2 rel_kw - 2 - r] / alpha, zero rows
                       after; second image T^T [80][64]. Needs hs_valid = 80 and 2n + 2 rel_kw - 2 <= 64; rel_kh / rel_kw / rel_ld = 32 as with rel.
                       fwd: rel, if given, is an OUTPUT — bf16 [B*H, Lq, 32], the score-domain (x alpha log2 e) bias operand of every valid
                       query, kept for the backward; bwd: rel = that tensor, drel must be NULL, and dq already contains
                       sum_bin d rel[q][bin] * T[bin's row for q] (what grove_rel_bias_bwd added). */
} grove_flash_attn_params;
int grove_flash_attn_fwd(const grove_flash_attn_params* p, void* stream);
int grove_flash_attn_bwd(const grove_flash_attn_params* p, void* stream);
/* 1 (default): problems that fit them run on the window kernels; 0: always the general kernels (A/B arm of tests and tools) */
int grove_flash_attn_set_window_kernels(int32_t on);
/* A/B knob (round 4): 0 = the general kernels build their rel-pos indicator tile in LDS for every key tile also where the register
 * form applies (rel_kw == rel_kh == 32, rel_ld == 64, head dim 96: SAM's global blocks); 1 (default) = indicator fragments in registers.
 * Same MFMAs on the same operand values: bit-identical results. */
int grove_flash_attn_set_register_e(int32_t on);
/* A/B knob (round 5): bit 0 = forward, bit 1 = backward dQ, bit 2 = backward dK / dV (head dim 96), bit 3 = the role-split
 * dK / dV of head dim 128, bit 4 = the dQ kernel at head dim 128 too (off by default: slower there) — the eight-wave ping-pong kernels (flash_attn2.hip: 512-thread workgroups, LDS-DMA rings, 32x32x16
 * MFMAs, rel-pos bias on the score accumulators / the matrix pipe) for head dims 64 / 96 / 128 without rel-pos or with SAM's
 * 32 x 32 global form; 0 = the round-2 four-wave kernels everywhere. Bit 5 (round 6b): the non-causal launches of those kernels deal
 * whole (batch, head) groups to an XCD — the blocks that stream one head's K / V (or Q / dO) share one L2 instead of eight; bit-identical
 * results, a launch-order change only. Default 47. Results of the kernel families agree to fp32 sum order (different tile shapes), not
 * bit for bit. */
int grove_flash_attn_set_v2(int32_t mask);
int grove_flash_attn_window_kernels_on(void);

/* Decomposed relative-position terms of SAM attention (image_encoder.py:420-458):
 * rel[b*heads + h, q, 0:kh]     = sum_c qv[q, h, c] * Rh[qh(q), :, c]
 * rel[b*heads + h, q, kh:kh+kw] = sum_c qv[q, h, c] * Rw[qw(q), :, c]
 * q: bf16 rows [b, q] with leading dim ld_q, head h at column offset h*hd_stride; hd = real head
 * dim; Rh: f32 [qh, kh, hd]; Rw: f32 [qw, kw, hd]. bwd adds  dq += drel . R  into dq (bf16). */
typedef struct grove_relpos_params {
  const void* q;
  const float* Rh;
  const float* Rw;
  float* rel;        /* fwd: out; bwd: d rel in */
  void* dq;          /* bwd only: bf16, accumulated */
  int32_t batch, heads, qh, qw, kh, kw, hd, hd_stride, ld_q;
} grove_relpos_params;
int grove_relpos_fwd(const grove_relpos_params* p, void* stream);
int grove_relpos_bwd(const grove_relpos_params* p, void* stream);

/* The same terms in the form the fused attention kernels consume, as two streams on the matrix cores (rel_bias.hip):
 *   fwd  rel[(b*nh + h), q, bin] = sum_d q[(b*L + q), h*hp + d] * table[q][bin][d]        table = Rcat  bf16 [L, rel_ld, hp]
 *   bwd  dq[(b*L + q), h*hp + d] += sum_bin rel[(b*nh + h), q, bin] * table[q][d][bin]    table = RcatT bf16 [L, hp, rel_ld]
 * bins = [rel_h | rel_w] padded to rel_ld (32 or 64), tables pre-divided by the softmax scale and zero in the pad bins and in
 * the pad dims hd..hp-1 (sam.py:_rcat_tables). q / dq: bf16 rows, head h at column h*hp (hp % 32 == 0, <= 128), nh <= 16.
 * Replaces the two GEMMs batched over the L query positions (4608 x 32 x 96 per position at SAM-H window size). */
typedef struct grove_rel_bias_params {
  const void* q;     /* fwd: bf16 [nb*L, ld_q] */
  const void* table; /* fwd: Rcat; bwd: RcatT */
  void* rel;         /* bf16 [nb*nh, L, rel_ld]: fwd out; bwd: d rel in */
  void* dq;          /* bwd: bf16 [nb*L, ld_dq], accumulated in place */
  int32_t nb, nh, L, hp, hd, rel_ld, ld_q, ld_dq;
  const int32_t* q_valid; /* int32 [nb][2] = {vy, vx} or NULL: grove_flash_attn_params.q_valid's rule — positions outside the top-left
                             vy x vx block of window b get no rel row (fwd) and no dq contribution (bwd; their rel rows are not read) */
  int32_t kw;             /* window width (positions per window row); with q_valid only */
  int32_t dq_hs;          /* bwd with dq_map: column stride between heads of dq (compact heads: hd) */
  const int32_t* dq_map;  /* bwd (round 6): int32 [nb * L] or NULL. Given, dq is in TOKEN order (grove_flash_attn_params.g_tok): the row of
                             (window b, position q) is dq_map[b * L + q] (valid wherever q_valid keeps the position), head h at column h * dq_hs */
} grove_rel_bias_params;
int grove_rel_bias_fwd(const grove_rel_bias_params* p, void* stream);
int grove_rel_bias_bwd(const grove_rel_bias_params* p, void* stream);

/* Rotary embedding, HF rotate_half convention, applied in place to q and k heads inside a fused
 * qkv activation: x: bf16 [rows, ld]; row r has position pos[r]; heads at columns
 * col0 + h*hd for h < nheads. inverse=1 applies the transpose rotation (backward).
 * cos/sin are computed in fp32 from theta (HF LlamaRotaryEmbedding) and applied in fp32. */
typedef struct grove_rope_params {
  void* x;
  const int32_t* pos;
  int32_t rows, ld, col0, nheads, hd;
  int32_t inverse;
  float theta;
  const float* table; /* round 4: f32 [positions, hd] = cos[hd / 2] | sin[hd / 2] per position (the table of grove_flash_attn_params.rope), or
                         NULL. Given (vector kernel: hd % 16 == 0), the angles are read from it instead of evaluating powf + sincosf eight
                         times per thread: the pass becomes a plain HBM stream. 16-byte aligned, at least max(pos) + 1 rows. */
} grove_rope_params;
int grove_rope_inplace(const grove_rope_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight-streaming GEMV for the cached greedy-decode step (HF `generate(use_cache=True)` feeds one
 * token per sequence after step 0: llava_llama.py:144-180, GROVE.py:418-422).
 *   y[b, n] = act( sum_k x[b, k] * W[n, k] + bias[n] ) + residual[b, n],  1 <= M <= 8 rows
 * x: bf16 [M, ldx]; W: bf16 [N, ldw] (nn.Linear layout); y: bf16 or f32 [M, ldy]. HBM-bound: every
 * weight byte is read once. x, residual and y have exactly M rows: for M of 3 (5..7) the 4- (8-)row
 * instance runs on row M - 1 repeated and stores M rows (round 5; earlier builds read and wrote the padding rows).
 * act GROVE_ACT_SWIGLU_PAIR (HF LlamaMLP's silu(gate) * up in the gate|up projection's epilogue): W rows interleaved
 * [4 gate, 4 up] per 8 as for grove_gemm_bf16, N % 16 == 0, no bias / residual; y gets N / 2 columns.
 * ------------------------------------------------------------------------------------------ */
enum grove_gemv_x_mode {
  GROVE_GEMV_X_PLAIN = 0,
  GROVE_GEMV_X_RMSNORM = 1, /* x' = bf16(x * rsqrt(mean(x^2) + eps) * norm_weight): the layer's RMSNorm folded into the load */
  GROVE_GEMV_X_SWIGLU = 2   /* x is a fused [M, 2K] gate|up row; x' = bf16(silu(gate) * up): LlamaMLP's product folded in */
};
typedef struct grove_gemv_params {
  const void* x;
  const void* W;
  void* y;
  const void* bias;        /* bf16 [N] or NULL */
  const void* residual;    /* bf16 [M, ldr] or NULL */
  const void* norm_weight; /* bf16 [K], x_mode RMSNORM */
  int32_t M, N, K;
  int32_t ldx, ldw, ldy, ldr;
  int32_t act;     /* GROVE_ACT_* */
  int32_t y_dtype; /* GROVE_BF16 / GROVE_F32 */
  int32_t x_mode;  /* grove_gemv_x_mode */
  float eps;
  /* the decode step's residual stream in FP32 (inference models): x is f32 [M, ldx] (x_mode PLAIN / RMSNORM: statistics and
   * normalisation on the fp32 values), residual is f32 [M, ldr]; combine with y_dtype F32 for the stream's next value */
  int32_t x_f32, res_f32;
  /* 1 = the matrix-core kernel for EVERY M it can take (default: from 3 sequences up; 1 and 2 run the VALU kernel, whose fp32 sum order
   * is different). A row's result is then the same bits whichever other sequences share its launch: batch-invariant decode (round 6;
   * one int32 in what was the struct's tail padding). */
  int32_t force_mfma;
  /* DEFERRED RMSNorm between two launches of the decode step (round 6; matrix-core kernel, plain bf16 x): HF LlamaDecoderLayer's
   * `hidden = residual + proj(...)` followed by the next `*_layernorm` and projection, without a norm launch in between.
   *   producer (o_proj / down_proj, y = the fp32 residual stream v): xs_out bf16 [M, ld_xs] = bf16(v * xs_weight[n]) (NULL: not wanted),
   *     ssq_out f32 [ceil(N / 16)][8] = this launch's sums of v^2 over each group of 16 output columns, per row (every entry written);
   *   consumer (q|k|v / gate|up / lm_head, x = that xs_out): ssq_in = the producer's ssq_out, ssq_in_blocks = its entry count — row m of
   *     the product is multiplied by rsqrt(sum_b ssq_in[b][m] / K + eps) (fixed summation order) before bias / activation.
   * Same function as x_mode RMSNORM up to where the bf16 rounding of the normalised input falls (before instead of after the rstd). */
  void* xs_out;
  const void* xs_weight;
  float* ssq_out;
  const float* ssq_in;
  int32_t ssq_in_blocks, ld_xs;
} grove_gemv_params;
int grove_gemv_bf16(const grove_gemv_params* p, void* stream);
/* A/B knob (round 5): 1 (default) = 3..8 sequences with K % 128 == 0 run on the matrix-core kernel (one weight stream feeds one
 * v_mfma_f32_16x16x32_bf16 per 32-deep k-step: HBM-bound at 8 sequences, where the VALU kernel is compute-bound); 0 = the VALU
 * kernel for every M. fp32 sum order differs between the two (both accumulate in fp32). Bit 1 (on = 3): the matrix-core kernel
 * with 16-row workgroups for every N (default: 32-row workgroups when N >= 16384 and x is plain — the same results bit for bit);
 * bit 2 (on = 5): the VALU kernel at M = 1 with two rows per wave for N <= 4096 (default: one — the same results);
 * bit 3 (on = 9, round 6): the matrix-core kernel's weight loads in MFMA operand shape (64 contiguous bytes per row per instruction; default:
 * 128 bytes per row per instruction + a lane-pair exchange, the same results bit for bit, 15-20 % more bytes per second). */
int grove_gemv_set_mfma(int32_t on);
/* 1 when grove_gemv_bf16 would run `p` on the matrix-core kernel under the current knob (host only; the dispatcher's own predicate, so a
 * caller that splits rows or un-folds a norm around it cannot disagree with the library). */
int grove_gemv_uses_mfma(const grove_gemv_params* p);

/* One cached decode step of causal self-attention for ONE new token per sequence (HF LlamaAttention with a KV cache):
 * rotates q and k of the new token in place (rotate-half RoPE at position pos[b], fp32), appends k | v to the cache row
 * pos[b] and attends the query to cache rows 0..pos[b]. qkv: bf16 [B, ld_qkv] = q | k | v (H heads of hd each);
 * cache: bf16 [B, 2, H, S_max, hd] = (keys, values) planes, head-major — the positions of one head are contiguous rows, which is what
 * the one-block-per-head stream reads (round 2's [B, S_max, 2*H*hd] put every row of a head in its own DRAM page);
 * out: bf16 [B, H*hd]. hd must be 32, 64 or 128. */
typedef struct grove_decode_attn_params {
  void* qkv;
  void* cache;
  void* out;
  const int32_t* pos; /* [B] device */
  int32_t B, H, hd, S_max, ld_qkv;
  float theta, alpha;
  /* flash-decoding split: the cached rows of a head are dealt to n_split blocks (one CU streams ~23 GB/s: 32 heads alone cannot
   * use the chip), partial {max, sum, output} per block in the caller's `partial` (f32 [B, H, n_split, hd + 2]), merged by a second
   * launch. n_split <= 1 (partial may be NULL): one block per head. */
  void* partial;
  int32_t n_split;
} grove_decode_attn_params;
int grove_decode_attn(const grove_decode_attn_params* p, void* stream);

/* HF greedy search between two decoder steps, on the device (GenerationMixin as GROVE.py:418-422 uses it: num_beams 1, do_sample
 * False; llava_llama.py:144-180 feeds the picked token back): per sequence b — nxt = finished[b] ? pad : argmax(logits[b, :V]) (first
 * maximum); finished[b] |= nxt == eos; tok[b] = nxt; step = pos[b] - pos0; ids_out[b, step] = nxt; hid_out[step, b, :] = hidden[b, :]
 * (bf16, and the fp32 pair when given); pos[b] += 1. Everything is device data, so the whole decode step — gather, 32 layers, this —
 * replays from one HIP graph with no host read-back per token. */
typedef struct grove_greedy_pick_params {
  const float* logits;     /* f32 [B, ld_logits] */
  int64_t ld_logits;
  uint8_t* finished;       /* [B] 0 / 1 (torch.bool storage) */
  int32_t* tok;            /* [B] out */
  int32_t* pos;            /* [B] in / out */
  int64_t* ids_out;        /* [B, ld_ids] */
  int64_t ld_ids;
  const void* hidden;      /* bf16 [B, H] or NULL */
  void* hid_out;           /* bf16 [max_steps, B, H] or NULL */
  const float* hidden_f32; /* f32 [B, H] or NULL */
  float* hid_out_f32;      /* f32 [max_steps, B, H] or NULL */
  int32_t B, V, H, eos, pad, pos0, max_steps;
} grove_greedy_pick_params;
int grove_greedy_pick(const grove_greedy_pick_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frame preprocessing on the device. One separable pass of Pillow's 8-bit resampler
 * (ImagingResample: 22-bit fixed-point coefficients, half-up rounding, uint8 clamp) over uint8 RGB frames
 * [F, H, W, 3]; the reference runs it on the host through torchvision / transformers:
 * transforms.py:27-34 (`resize(to_pil_image(image), target_size)`, bilinear) and CLIPImageProcessor.preprocess
 * (bicubic, HowTo100M.py:309). kk: int32 [out_size, ksize] coefficients, bounds: int32 [out_size, 2] = (first input
 * index, count) per output index (grove_amd/preprocess.py builds both exactly as Pillow does).
 * axis 0 = horizontal (Hout == Hin), axis 1 = vertical (Wout == Win).
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_resample_params {
  const void* src; /* u8 [F, Hin, Win, 3] */
  void* dst;       /* u8 [F, Hout, Wout, 3] */
  const int32_t* kk;
  const int32_t* bounds;
  int32_t F, Hin, Win, Hout, Wout;
  int32_t ksize, axis;
} grove_resample_params;
int grove_resample_u8(const grove_resample_params* p, void* stream);

/* u8 [F, H, W, 3] -> bf16 / f32 [3, F, Ho, Wo] (the `b c t h w` layout the encoders take):
 * out[c, f, y, x] = (src[f, top + y, left + x, c] * rescale - mean[c]) / std[c], 0 outside the source (SAM pads right /
 * bottom AFTER normalising: HowTo100M.py:168-178; CLIP centre-crops: top, left > 0). */
typedef struct grove_normalize_params {
  const void* src;
  void* dst;
  int32_t F, H, W, Ho, Wo, top, left;
  int32_t out_dtype;
  float rescale;
  float mean[3], std[3];
} grove_normalize_params;
int grove_normalize_pack(const grove_normalize_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / data-movement kernels (all bf16, vectorised 16 B per lane).
 * ------------------------------------------------------------------------------------------ */
/* y = silu(gate) * up over a fused [rows, 2*I] gate|up activation (HF LlamaMLP) */
int grove_swiglu_fwd(const void* gu, void* y, int32_t rows, int32_t I, void* stream);
/* dgu = [dy*up*silu'(gate) | dy*silu(gate)] */
int grove_swiglu_bwd(const void* gu, const void* dy, void* dgu, int32_t rows, int32_t I, void* stream);
/* dx = dy * act'(pre) (in place allowed) */
int grove_act_bwd(const void* pre, const void* dy, void* dx, int64_t n, int32_t act, void* stream);
/* y = act(x) (bf16, in place allowed): the GELU between LayerNorm2d and the second ConvTranspose2d of output_upscaling (mask_decoder.py:58-67) */
int grove_act_fwd(const void* x, void* y, int64_t n, int32_t act, void* stream);
/* Bilinear resize with align_corners = False of fp32 planes [planes, h, w] -> [planes, H, W]; only rows < h_use and columns < w_use of
 * every source plane are sampled (the padding crop of Sam.postprocess_masks, sam.py:137-172: interpolate, crop, interpolate). */
int grove_resize_bilinear_f32(const float* src, float* dst, int32_t planes, int32_t h, int32_t w, int32_t h_use, int32_t w_use, int32_t H,
                              int32_t W, void* stream);
/* y = a + b (bf16), n elements; b may be NULL (copy) */
int grove_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);
/* y[r, :] = a[r, :] + b[r % period, :]   (keys + key_pe, transformer.py:168-170) */
int grove_add_bcast_rows(const void* a, const void* b, void* y, int32_t rows, int32_t C, int32_t period, void* stream);
/* dst[idx_dst[r] or r, :] = src[idx_src[r] or r, :]; idx_src -1 -> zero row; bf16 rows of C elems.
 * accumulate=1: dst row += (fp32 add, rows must be unique or use grove_scatter_add_f32) */
typedef struct grove_rows_params {
  const void* src;
  void* dst;
  const int32_t* idx_src;
  const int32_t* idx_dst;
  int32_t rows, C, ld_src, ld_dst;
  int32_t accumulate;
} grove_rows_params;
int grove_copy_rows(const grove_rows_params* p, void* stream);
/* dst_f32[idx[r], :] += src_bf16[r, :] with float atomics (embed_tokens / broadcast-row grads) */
int grove_scatter_add_f32(const void* src, float* dst, const int32_t* idx, int32_t rows, int32_t C,
                          int32_t ld_src, int32_t ld_dst, void* stream);
/* Round 6b: the same sum when the scattered rows come in consecutive groups — member i (rows_per_seg bf16 rows of C) belongs to segment s iff
 * seg_ptr[s] <= i < seg_ptr[s + 1] (int32 [nseg + 1], non-decreasing): dst_f32[s * rows_per_seg + r, :] += sum_i src[i * rows_per_seg + r, :].
 * One owner per output element: no atomics, the same bits every run. The box decoder's gradient w.r.t. the SAM embeddings (every frame's
 * instances gathered its 1024 embedding rows: mask_decoder.py:181-186) — 165 -> 25 us against grove_scatter_add_f32 on 3 instances per frame. */
int grove_segment_sum_rows(const void* src, float* dst, const int32_t* seg_ptr, int32_t nseg, int32_t rows_per_seg, int32_t C, int32_t ld_src,
                           int32_t ld_dst, void* stream);
/* the same with fp32 source rows: dst_f32[idx[r], :] += src_f32[r, :] (idx -1 = skipped). Sums the all-gathered (row id, row) pairs of
 * the sparse embed_tokens gradient exchange into the dense gradient (train.py:466-478's dense reduce-scatter replaced) */
int grove_scatter_add_rows_f32(const float* src, float* dst, const int32_t* idx, int32_t rows, int32_t C,
                               int32_t ld_src, int32_t ld_dst, void* stream);
/* column sums of a bf16 [rows, ld] matrix into f32 out[C] (bias gradients); accumulate adds */
int grove_colsum_f32(const void* x, float* out, int32_t rows, int32_t C, int32_t ld, int32_t accumulate, void* stream);
/* out[0] += f * sum_i a[i]*b[i] (bf16 in, fp32 accumulate). scale_ptr NULL: f=1; mode 0: f=*scale_ptr;
 * mode 1: f = 1 - tanh(*scale_ptr)^2 (d tanh(alpha), adapter gate gradient); mode 2: f = tanh(*scale_ptr) */
int grove_dot_bf16(const void* a, const void* b, float* out, int64_t n, const float* scale_ptr, int32_t mode, void* stream);
/* y += f * x over n fp32 elements, f as for grove_dot_bf16 (modes 0 and 2) */
int grove_axpy_f32(float* y, const float* x, int64_t n, const float* scale_ptr, int32_t mode, void* stream);
/* casts */
int grove_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int grove_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);

/* Patch im2col for the two ViT stems: images bf16 [B, C, T, H, W] (the reference's 'b c t h w'
 * clip layout, clip_encoder.py:70 / image_encoder.py:174) -> col bf16 [B*T*(H/P)*(W/P), ld_col]
 * with k = (c, py, px) matching Conv2d weight.flatten(1), columns [C*P*P, ld_col) zeroed.
 * Replaces the strided Conv2d at modeling_clip.py:174-180,190 and image_encoder.py:484-492. */
int grove_im2col_patch(const void* img, void* col, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                       int32_t P, int32_t ld_col, void* stream);
/* d img not needed (inputs carry no gradient). */

/* CLIP token pooling: AdaptiveAvgPool3d((8,8,9)) over (t=8, 24, 24) patch tokens, skipping CLS
 * (pooling.py:6-25, clip_encoder.py:45-50,74-76). x: bf16 [G*8, 577, C] -> y: bf16 [G, 576, C]. */
int grove_clip_pool(const void* x, void* y, int32_t G, int32_t C, void* stream);

/* Cross entropy over rows of bf16 logits [R, ld] with V valid columns, labels int32 [R]
 * (every label valid; rows are pre-gathered). loss_sum += sum_r (lse_r - logit[r, label_r]);
 * dlogits (bf16, may alias logits) = (softmax - onehot) * (*grad_scale).
 * Replaces CrossEntropyLoss at llava_llama.py:115-125. */
int grove_cross_entropy(const void* logits, const int32_t* labels, float* loss_sum, void* dlogits,
                        const float* grad_scale, int32_t R, int32_t V, int32_t ld, void* stream);

/* Small attention for the SAM two-way decoder (transformer.py:185-242): per (instance, head)
 * softmax(q k^T / sqrt(d)) v with Lq*Lk small on one side (6 tokens x 1024 image tokens, either
 * direction). q: bf16 [inst, Lq, ld_q], k/v: bf16 [inst, Lk, ld_k]/[.., ld_v], heads packed
 * along columns with head dim d (16 or 32). out bf16 [inst, Lq, ld_o]. Wavefront-reduced
 * (no MFMA: the products are <= 6 rows). bwd recomputes the probabilities. */
typedef struct grove_small_attn_params {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  const void* d_o; /* bwd */
  void* dq; /* bwd outputs are f32, dense [inst, L, heads*d] */
  void* dk;
  void* dv;
  int32_t inst, heads, d, Lq, Lk, ld_q, ld_k, ld_v, ld_o;
  int32_t q_f32, kv_f32, o_f32; /* forward only: q / k,v / o are f32 arrays (the decoder's fp32 token path); leading dims in elements */
  int32_t grad_bf16; /* bwd (round 6b): 1 = dq / dk / dv are BF16 arrays of the same dense layout — only where grove_small_attn_bwd_stores_bf16 says
                        the kernel stores every gradient element exactly once (no fp32 atomics): the caller's three cast passes go away */
} grove_small_attn_params;
int grove_small_attn_fwd(const grove_small_attn_params* p, void* stream);
int grove_small_attn_bwd(const grove_small_attn_params* p, void* stream);
/* 1 if grove_small_attn_bwd on this problem (shapes, leading dims, q / k / v / o / d_o pointers set) runs a kernel that can store bf16 gradients */
int grove_small_attn_bwd_stores_bf16(const grove_small_attn_params* p);
/* A/B knob (round 6b): 1 (default) = Lq, Lk <= 8 at head dim 32 (the box decoder's token self attention) runs the lane-per-(pair, row) kernels —
 * no atomics, every gradient element stored once; 0 = the generic few-keys kernels. Results agree to fp32 sum order. */
int grove_small_attn_set_tiny(int32_t on);

/* Exact-fp32 small GEMM on v_mfma_f32_16x16x4_f32: C f32 [M, N] = act(A f32 [M, K] . W bf16 [N, K]^T + bias bf16 [N]) + residual f32.
 * K % 16 == 0. C_bf16 (optional, same leading dim as C) receives the bf16 rounding. The token side of the two-way decoder
 * (transformer.py:151-242: 6 tokens per box instance) and text_hidden_fcs on the [DET] rows (GROVE.py:75-79, 248-268) run through
 * it at inference, so the box path carries no bf16 activation rounding between the LLaMA hidden state and the box head. */
typedef struct grove_gemm_f32_params {
  const float* A;
  const void* W;
  const void* bias;      /* bf16 [N] or NULL */
  const float* residual; /* f32 [M, ldr] or NULL */
  float* C;
  void* C_bf16;          /* bf16 [M, ldc] or NULL */
  int32_t M, N, K, lda, ldw, ldc, ldr, act;
} grove_gemm_f32_params;
int grove_gemm_f32(const grove_gemm_f32_params* p, void* stream);

/* FP8 (OCP e4m3fn) GEMM for the frozen linear layers at inference (BASELINE config 5):
 * C bf16 [M, N] = act((A . B^T) * scale_a[m] * scale_b[n] + bias[n]) + residual[m, n]; A [M, lda], B [N, ldb]: e4m3 codes (bytes),
 * K % 128 == 0, rows 16-byte aligned; scale_a f32 [M] / scale_b f32 [N]: de-quantisation scales (amax / 448 per row, as
 * grove_quant_fp8_rows produces them). Runs on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales. Replaces nn.Linear in
 * CLIPAttention / CLIPMLP (modeling_clip.py:257-348) and HF LlamaAttention / LlamaMLP for an fp8-quantised checkpoint. */
typedef struct grove_gemm_fp8_params {
  const void* A; const void* B;
  void* C;                 /* bf16 [M, ldc] */
  const float* scale_a; const float* scale_b;
  const void* bias;        /* bf16 [N] or NULL */
  const void* residual;    /* bf16 [M, ldr] or NULL */
  int32_t M, N, K, lda, ldb, ldc, ldr, act;
} grove_gemm_fp8_params;
/* workspace protocol as grove_gemm_bf16 (the FP8 instances of the persistent kernel share its images: equal plan keys, equal bytes);
 * without an image the two-barrier kernel of gemm_fp8.hip runs (no workspace) */
int grove_gemm_fp8_make_plan(const grove_gemm_fp8_params* p, grove_gemm_plan* out);
int grove_gemm_fp8_plan_image(const grove_gemm_fp8_params* p, void* host_image, size_t bytes);
int grove_gemm_fp8(const grove_gemm_fp8_params* p, const grove_gemm_workspace* ws, void* stream);
/* 1 (default): problems that fit them (K % 128 == 0, N % 8 == 0, 16-byte aligned operands, act NONE / QUICKGELU / GELU) run on the FP8
 * instances of the persistent pipelined kernel; 0: always the two-barrier kernel (A/B arm of tests and tools) */
int grove_gemm_fp8_set_pipelined(int on);
/* x bf16 [rows, ld_x] -> q e4m3 [rows, ld_q] with one scale per row: scale = amax / 448 (1 for a zero row), q = x / scale */
int grove_quant_fp8_rows(const void* x, void* q, float* scale, int32_t rows, int32_t K, int32_t ld_x, int32_t ld_q, void* stream);
/* the same on act(x) (enum grove_act NONE .. SIGMOID; the activation's result is rounded to bf16 first, as a stored activation would be):
 * MLPBlock's lin1 -> GELU -> lin2 (common.py:21-26) with the GELU in the quantisation pass of lin2's input instead of lin1's epilogue.
 * K <= 8192 (the row stays in registers between the amax sweep and the conversion). */
int grove_quant_fp8_rows_act(const void* x, void* q, float* scale, int32_t rows, int32_t K, int32_t ld_x, int32_t ld_q, int32_t act, void* stream);

/* ------------------------------------------------------------------------------------------
 * Winograd F(2 x 2 x 2, 3 x 3 x 3) form of the Conv3d adapters (round 6): the memory-bound transforms either side of the grouped
 * / K-batched GEMMs (grove_gemm_params.b_group_rows, grove_gemm_tn_params.k_batches). Replaces the 27-tap implicit GEMMs of
 * SpatioTemporalConvAdapter (image_encoder.py:43-59: nn.Conv3d(C, C, 3, padding = 1) on '(b t) h w c -> b c t h w', t = 8) with
 * 64 products per 2 x 2 x 2 output tile instead of 216. fp32 arithmetic; transformed operands are rounded to bf16 once.
 *   tokens: bf16 rows [(g, t, y, x)][ld] = ((g T + t) H + y) W + x, groups of T frames of H x W positions; T, H, W even.
 *   tile (g, tt, ty, tx) = outputs (2 tt + {0, 1}, 2 ty + {0, 1}, 2 tx + {0, 1}); tile id = ((g T/2 + tt) H/2 + ty) W/2 + tx.
 *   transformed tensors: POINT-MAJOR [64][tiles or rows][ld], point = (a 4 + b) 4 + c over (t, y, x).
 * grove_wino3d_transform_tokens  mode 0: dst bf16 [64][tiles][ld_dst] = (B^T (x) B^T (x) B^T) of the zero-padded 4 x 4 x 4 input
 *                                tiles of src (tokens); mode 1: = (A (x) A (x) A) of the 2 x 2 x 2 tiles of src (the output gradient).
 * grove_wino3d_transform_weight  src bf16 [rows = Co][27 taps][C = Ci] (tap = (kt 3 + kh) 3 + kw, ld_src >= 27 C) ->
 *                                dst bf16 [64][rows][ld_dst] = (G (x) G (x) G) per (co, ci).
 * grove_wino3d_output            src bf16 [64][tiles][ld_src] (the products) -> dst tokens: v = (A^T (x) A^T (x) A^T) src + bias;
 *                                aux = bf16(v) (optional pre-activation copy); dst = act(v) * f + residual, act NONE / RELU,
 *                                f = alpha * (scale_ptr ? (scale_tanh ? tanh(*scale_ptr) : *scale_ptr) : 1).
 * grove_wino3d_wgrad_output      src f32 [64][rows = Co][ld_src] (sum over tiles of dM (.) V per point) ->
 *                                dst f32 [rows][27][C] += f * (G^T (x) G^T (x) G^T) src.
 * ------------------------------------------------------------------------------------------ */
typedef struct grove_wino3d_params {
  const void* src;
  void* dst;
  const void* bias;        /* bf16 [C] or NULL (output) */
  const void* residual;    /* bf16 tokens [*, ld_res] or NULL (output) */
  void* aux;               /* bf16 tokens [*, ld_aux] or NULL (output) */
  const float* scale_ptr;  /* device scalar or NULL (output, wgrad_output) */
  int32_t groups, T, H, W; /* token geometry (transform_tokens, output) */
  int32_t C;               /* channels per row (even) */
  int32_t rows;            /* weight rows = output channels (transform_weight, wgrad_output) */
  int32_t ld_src, ld_dst, ld_res, ld_aux; /* row strides in elements (even) */
  int32_t mode, act, scale_tanh;
  float alpha;             /* 0 is NOT special: pass 1 for "no scale" */
  /* Token tensors whose frames carry extra rows (CLIP: [frames, 1 CLS + 576 patches, C], modeling_clip.py:599-611 splits the CLS row off
   * and treats the 24 x 24 grid as 16 x 36): token (g, t, y, x) is row (g T + t) * frame_rows + row_offset + y W + x of the token
   * operands (src of transform_tokens; dst / residual / aux of output). 0 / 0 = dense frames of H W rows. */
  int32_t frame_rows, row_offset;
  /* Rows per transform point of the point-major tensor (dst of transform_tokens, src of output): >= tiles, 0 = tiles. A caller pads it to
   * a multiple of 256 so that every point is whole tiles of the grouped GEMM; the pad rows are never written and never read here. */
  int32_t tiles_ld;
} grove_wino3d_params;
int grove_wino3d_transform_tokens(const grove_wino3d_params* p, void* stream);
int grove_wino3d_transform_weight(const grove_wino3d_params* p, void* stream);
int grove_wino3d_output(const grove_wino3d_params* p, void* stream);
int grove_wino3d_wgrad_output(const grove_wino3d_params* p, void* stream);

/* Box + temporal-objectness heads in fp32 (mask_decoder.py:80-84,198-203): x f32 [N, D];
 * box = sigmoid(W2 relu(W1 x + b1) + b2) [N,4]; obj = Wo x + bo [N]. Weights bf16.
 * hidden (f32 [N, D]) is saved for backward. */
typedef struct grove_box_head_params {
  const float* x;
  const void* W1; const void* b1; const void* W2; const void* b2; const void* Wo; const void* bo;
  float* hidden;
  float* box;
  float* obj;
  int32_t N, D;
} grove_box_head_params;
int grove_box_head_fwd(const grove_box_head_params* p, void* stream);
/* backward of the heads. dbox/dobj: f32 upstream grads; dx f32 [N,D]; weight grads f32,
 * ACCUMULATED (+=) with atomics: dW1 [D,D], db1 [D], dW2 [4,D], db2 [4], dWo [D], dbo [1]. */
typedef struct grove_box_head_bwd_params {
  const float* x;
  const void* W1; const void* W2; const void* Wo;
  const float* hidden;
  const float* box;
  const float* dbox;
  const float* dobj; /* may be NULL */
  float* dx;
  float* dW1; float* db1; float* dW2; float* db2; float* dWo; float* dbo;
  int32_t N, D;
} grove_box_head_bwd_params;
int grove_box_head_bwd(const grove_box_head_bwd_params* p, void* stream);

/* GROVE losses on device in fp32 (GROVE.py:339-381; torchvision generalized_box_iou_loss eps 1e-7):
 * pred_box f32 [N,4] cxcywh, obj_logit f32 [N], gt_box f32 [N,4] (rows with visible==0 ignored),
 * visible f32 [N] in {0,1}. sums[0..2] += (giou_sum, l1_sum, bce_sum); dbox/dobj = gradients of
 * w_box*(giou+l1)/n_gt + w_obj*bce/N  when dbox != NULL. */
int grove_box_losses(const float* pred_box, const float* obj_logit, const float* gt_box, const float* visible,
                     float* sums, float* dbox, float* dobj, int32_t N, float w_box_over_ngt, float w_obj_over_n,
                     void* stream);

/* Fused AdamW step over a flat fp32 master / bf16 model-weight pair (DeepSpeed AdamW config,
 * train.py:466-475). grad is the (already all-reduced) fp32 flat gradient. */
int grove_adamw_step(float* master, void* model_bf16, const float* grad, float* m, float* v, int64_t n,
                     float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                     int32_t step, void* stream);
/* The same step over all trainable tensors in ONE launch (DeepSpeed's multi-tensor FusedAdam, train.py:466-475): the fp32
 * state (master / grad / m / v) is one flat buffer of `total` elements in which tensor s occupies
 * [seg_off[s], seg_off[s] + seg_len[s]) (16-byte aligned starts, zero-filled gaps); model_bf16[s] is that tensor's bf16
 * working copy (NULL = none). seg_off / seg_len / model_bf16 are DEVICE arrays of nseg entries, seg_off ascending.
 * Global-norm clipping ("gradient_clipping": 1.0, train.py:475) without a host round trip: sumsq (device, NULL = off) holds the
 * sum of squares of `grad` (grove_sumsq_f32); the kernel scales by grad_scale * min(1, clip / (sqrt(sumsq) * grad_scale + 1e-6))
 * and writes the pre-clip norm sqrt(sumsq) * grad_scale to norm_out (device, may be NULL). */
int grove_adamw_step_multi(float* master, const float* grad, float* m, float* v, const int64_t* seg_off, const int64_t* seg_len,
                           void* const* model_bf16, int32_t nseg, int64_t total, float lr, float beta1, float beta2, float eps,
                           float weight_decay, float grad_scale, int32_t step, const float* sumsq, float clip, float* norm_out,
                           void* stream);
/* sum of squares of a flat f32 buffer into out[0] (+=), for grad clipping */
int grove_sumsq_f32(const float* x, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GROVE_HIP_H */
