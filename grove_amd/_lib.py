"""ctypes binding of libgrove_hip.so (include/grove_hip.h).

The product path has no fallback: if the shared library is missing this module raises at the
first op, and every non-zero status becomes a RuntimeError carrying grove_last_error().
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GROVE_HIP_LIB") or os.path.join(_HERE, "csrc", "libgrove_hip.so")  # (GROVE_HIP_LIB: another build of the library, for A/B runs)

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

ACT_NONE, ACT_RELU, ACT_GELU, ACT_QUICKGELU, ACT_SILU, ACT_SIGMOID, ACT_SWIGLU_PAIR, ACT_SWIGLU_BWD = range(8)
BF16, F32 = 0, 1


class GemmParams(C.Structure):
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("bias", c_vp), ("residual", c_vp), ("aux", c_vp),
                ("scale_ptr", c_vp), ("a_idx", c_vp), ("c_idx", c_vp), ("r_idx", c_vp),
                ("sA1", c_i64), ("sA2", c_i64), ("sB1", c_i64), ("sB2", c_i64),
                ("sC1", c_i64), ("sC2", c_i64), ("sR1", c_i64), ("sR2", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("lda", c_i32), ("ldb", c_i32), ("ldc", c_i32), ("ldr", c_i32),
                ("batch1", c_i32), ("batch2", c_i32), ("a_taps", c_i32), ("act", c_i32),
                ("c_dtype", c_i32), ("accumulate", c_i32), ("scale_tanh", c_i32), ("alpha", c_f32), ("split_k", c_i32), ("ld_aux", c_i32),
                ("n_group", c_i32), ("n_pad", c_i32), ("k_group", c_i32), ("k_pad", c_i32), ("aux_grad", c_i32), ("residual_mul", c_i32),
                ("a_frame_rows", c_i32), ("a_frames", c_i32), ("b_group_rows", c_i32)]


class TransposeParams(C.Structure):
    _fields_ = [("inp", c_vp), ("out", c_vp), ("s_in1", c_i64), ("s_in2", c_i64), ("s_out1", c_i64), ("s_out2", c_i64),
                ("rows", c_i32), ("cols", c_i32), ("ld_in", c_i32), ("ld_out", c_i32), ("pad_to", c_i32),
                ("batch1", c_i32), ("batch2", c_i32)]


class NormParams(C.Structure):
    _fields_ = [("x", c_vp), ("weight", c_vp), ("bias", c_vp), ("y", c_vp), ("mean", c_vp), ("rstd", c_vp),
                ("out_idx", c_vp), ("rows", c_i32), ("C", c_i32), ("ld_x", c_i32), ("ld_y", c_i32),
                ("y_dtype", c_i32), ("eps", c_f32), ("res", c_vp), ("res_bf16", c_vp), ("ld_res", c_i32)]


class NormBwdParams(C.Structure):
    _fields_ = [("x", c_vp), ("weight", c_vp), ("dy", c_vp), ("dx", c_vp), ("mean", c_vp), ("rstd", c_vp),
                ("dweight", c_vp), ("dbias", c_vp), ("in_idx", c_vp),
                ("rows", c_i32), ("C", c_i32), ("ld_x", c_i32), ("ld_dy", c_i32), ("ld_dx", c_i32),
                ("accumulate", c_i32), ("eps", c_f32)]


class SoftmaxParams(C.Structure):
    _fields_ = [("scores", c_vp), ("probs", c_vp), ("kv_len", c_vp), ("rel", c_vp),
                ("batch", c_i32), ("heads", c_i32), ("Lq", c_i32), ("Lk", c_i32), ("ld_s", c_i32), ("ld_p", c_i32),
                ("causal", c_i32), ("rel_kh", c_i32), ("rel_kw", c_i32)]


class SoftmaxBwdParams(C.Structure):
    _fields_ = [("dprobs", c_vp), ("probs", c_vp), ("dscores", c_vp), ("drel", c_vp),
                ("batch", c_i32), ("Lq", c_i32), ("Lk", c_i32), ("ld_s", c_i32), ("ld_p", c_i32),
                ("rel_kh", c_i32), ("rel_kw", c_i32), ("scale", c_f32)]


class RelposParams(C.Structure):
    _fields_ = [("q", c_vp), ("Rh", c_vp), ("Rw", c_vp), ("rel", c_vp), ("dq", c_vp),
                ("batch", c_i32), ("heads", c_i32), ("qh", c_i32), ("qw", c_i32), ("kh", c_i32), ("kw", c_i32),
                ("hd", c_i32), ("hd_stride", c_i32), ("ld_q", c_i32)]



class RelBiasParams(C.Structure):
    _fields_ = [("q", c_vp), ("table", c_vp), ("rel", c_vp), ("dq", c_vp),
                ("nb", c_i32), ("nh", c_i32), ("L", c_i32), ("hp", c_i32), ("hd", c_i32), ("rel_ld", c_i32), ("ld_q", c_i32), ("ld_dq", c_i32),
                ("q_valid", c_vp), ("kw", c_i32), ("dq_hs", c_i32), ("dq_map", c_vp)]

class RopeParams(C.Structure):
    _fields_ = [("x", c_vp), ("pos", c_vp), ("rows", c_i32), ("ld", c_i32), ("col0", c_i32), ("nheads", c_i32),
                ("hd", c_i32), ("inverse", c_i32), ("theta", c_f32), ("table", c_vp)]


class RowsParams(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("idx_src", c_vp), ("idx_dst", c_vp),
                ("rows", c_i32), ("C", c_i32), ("ld_src", c_i32), ("ld_dst", c_i32), ("accumulate", c_i32)]


class SmallAttnParams(C.Structure):
    _fields_ = [("q", c_vp), ("k", c_vp), ("v", c_vp), ("o", c_vp), ("d_o", c_vp), ("dq", c_vp), ("dk", c_vp), ("dv", c_vp),
                ("inst", c_i32), ("heads", c_i32), ("d", c_i32), ("Lq", c_i32), ("Lk", c_i32),
                ("ld_q", c_i32), ("ld_k", c_i32), ("ld_v", c_i32), ("ld_o", c_i32), ("q_f32", c_i32), ("kv_f32", c_i32), ("o_f32", c_i32), ("grad_bf16", c_i32)]


class TransposeItem(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("rows", c_i32), ("cols", c_i32), ("ld_src", c_i32), ("ld_dst", c_i32), ("tile0", c_i32), ("pad_", c_i32)]


class BoxHeadParams(C.Structure):
    _fields_ = [("x", c_vp), ("W1", c_vp), ("b1", c_vp), ("W2", c_vp), ("b2", c_vp), ("Wo", c_vp), ("bo", c_vp),
                ("hidden", c_vp), ("box", c_vp), ("obj", c_vp), ("N", c_i32), ("D", c_i32)]


class BoxHeadBwdParams(C.Structure):
    _fields_ = [("x", c_vp), ("W1", c_vp), ("W2", c_vp), ("Wo", c_vp), ("hidden", c_vp), ("box", c_vp),
                ("dbox", c_vp), ("dobj", c_vp), ("dx", c_vp),
                ("dW1", c_vp), ("db1", c_vp), ("dW2", c_vp), ("db2", c_vp), ("dWo", c_vp), ("dbo", c_vp),
                ("N", c_i32), ("D", c_i32)]


class GemmTnParams(C.Structure):
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("scale_ptr", c_vp), ("b_idx", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("lda", c_i32), ("ldb", c_i32), ("ldc", c_i32),
                ("b_taps", c_i32), ("scale_tanh", c_i32), ("split_k", c_i32), ("alpha", c_f32),
                ("b_frame_rows", c_i32), ("b_frames", c_i32), ("k_batches", c_i32), ("overwrite", c_i32), ("sC_batch", c_i64)]


class FlashAttnParams(C.Structure):
    _fields_ = [("q", c_vp), ("k", c_vp), ("v", c_vp), ("o", c_vp), ("d_o", c_vp), ("dq", c_vp), ("dk", c_vp), ("dv", c_vp),
                ("lse", c_vp), ("delta", c_vp), ("kv_len", c_vp), ("rel", c_vp), ("drel", c_vp),
                ("sq", c_i64), ("sk", c_i64), ("sv", c_i64), ("so", c_i64), ("sdo", c_i64), ("sdq", c_i64), ("sdk", c_i64), ("sdv", c_i64),
                ("B", c_i32), ("H", c_i32), ("Lq", c_i32), ("Lk", c_i32), ("hs", c_i32),
                ("ld_q", c_i32), ("ld_k", c_i32), ("ld_v", c_i32), ("ld_o", c_i32), ("ld_do", c_i32), ("ld_dq", c_i32),
                ("ld_dk", c_i32), ("ld_dv", c_i32), ("causal", c_i32), ("rel_kh", c_i32), ("rel_kw", c_i32), ("rel_ld", c_i32), ("alpha", c_f32),
                ("hs_valid", c_i32), ("q_valid", c_vp), ("o_map", c_vp), ("o_hs", c_i32), ("g_tok", c_i32), ("pad_k", c_vp), ("pad_v", c_vp), ("rope", c_vp), ("rel_table", c_vp)]


class GemvParams(C.Structure):
    _fields_ = [("x", c_vp), ("W", c_vp), ("y", c_vp), ("bias", c_vp), ("residual", c_vp), ("norm_weight", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("ldx", c_i32), ("ldw", c_i32), ("ldy", c_i32), ("ldr", c_i32),
                ("act", c_i32), ("y_dtype", c_i32), ("x_mode", c_i32), ("eps", c_f32), ("x_f32", c_i32), ("res_f32", c_i32), ("force_mfma", c_i32),
                ("xs_out", c_vp), ("xs_weight", c_vp), ("ssq_out", c_vp), ("ssq_in", c_vp), ("ssq_in_blocks", c_i32), ("ld_xs", c_i32)]


class DecodeAttnParams(C.Structure):
    _fields_ = [("qkv", c_vp), ("cache", c_vp), ("out", c_vp), ("pos", c_vp), ("B", c_i32), ("H", c_i32), ("hd", c_i32),
                ("S_max", c_i32), ("ld_qkv", c_i32), ("theta", c_f32), ("alpha", c_f32), ("partial", c_vp), ("n_split", c_i32)]


class ResampleParams(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("kk", c_vp), ("bounds", c_vp), ("F", c_i32), ("Hin", c_i32), ("Win", c_i32),
                ("Hout", c_i32), ("Wout", c_i32), ("ksize", c_i32), ("axis", c_i32)]


class NormalizeParams(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("F", c_i32), ("H", c_i32), ("W", c_i32), ("Ho", c_i32), ("Wo", c_i32),
                ("top", c_i32), ("left", c_i32), ("out_dtype", c_i32), ("rescale", c_f32), ("mean", c_f32 * 3), ("std", c_f32 * 3)]


class GemmF32Params(C.Structure):
    _fields_ = [("A", c_vp), ("W", c_vp), ("bias", c_vp), ("residual", c_vp), ("C", c_vp), ("C_bf16", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("lda", c_i32), ("ldw", c_i32), ("ldc", c_i32), ("ldr", c_i32), ("act", c_i32)]


class GemmFp8Params(C.Structure):
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("scale_a", c_vp), ("scale_b", c_vp), ("bias", c_vp), ("residual", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("lda", c_i32), ("ldb", c_i32), ("ldc", c_i32), ("ldr", c_i32), ("act", c_i32)]


class GreedyPickParams(C.Structure):
    _fields_ = [("logits", c_vp), ("ld_logits", c_i64), ("finished", c_vp), ("tok", c_vp), ("pos", c_vp), ("ids_out", c_vp), ("ld_ids", c_i64),
                ("hidden", c_vp), ("hid_out", c_vp), ("hidden_f32", c_vp), ("hid_out_f32", c_vp),
                ("B", c_i32), ("V", c_i32), ("H", c_i32), ("eos", c_i32), ("pad", c_i32), ("pos0", c_i32), ("max_steps", c_i32)]


class Wino3dParams(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("bias", c_vp), ("residual", c_vp), ("aux", c_vp), ("scale_ptr", c_vp),
                ("groups", c_i32), ("T", c_i32), ("H", c_i32), ("W", c_i32), ("C", c_i32), ("rows", c_i32),
                ("ld_src", c_i32), ("ld_dst", c_i32), ("ld_res", c_i32), ("ld_aux", c_i32), ("mode", c_i32), ("act", c_i32), ("scale_tanh", c_i32),
                ("alpha", c_f32), ("frame_rows", c_i32), ("row_offset", c_i32), ("tiles_ld", c_i32)]


class GemmWorkspace(C.Structure):
    _fields_ = [("image", c_vp), ("image_bytes", C.c_size_t), ("scratch", c_vp), ("scratch_bytes", C.c_size_t)]


class GemmPlan(C.Structure):
    _fields_ = [("variant", c_i32), ("bm", c_i32), ("tiles_m", c_i32), ("tiles_n", c_i32), ("k_tiles", c_i32), ("grid", c_i32),
                ("stream_k", c_i32), ("reserved", c_i32), ("image_bytes", c_i64), ("scratch_bytes", c_i64), ("key", C.c_uint64)]


STRUCTS = {
    "grove_gemm_params": GemmParams, "grove_transpose_params": TransposeParams, "grove_norm_params": NormParams,
    "grove_norm_bwd_params": NormBwdParams, "grove_softmax_params": SoftmaxParams,
    "grove_softmax_bwd_params": SoftmaxBwdParams, "grove_relpos_params": RelposParams, "grove_rel_bias_params": RelBiasParams, "grove_rope_params": RopeParams,
    "grove_rows_params": RowsParams, "grove_small_attn_params": SmallAttnParams, "grove_box_head_params": BoxHeadParams,
    "grove_box_head_bwd_params": BoxHeadBwdParams, "grove_flash_attn_params": FlashAttnParams,
    "grove_gemm_tn_params": GemmTnParams, "grove_gemv_params": GemvParams, "grove_decode_attn_params": DecodeAttnParams, "grove_resample_params": ResampleParams,
    "grove_normalize_params": NormalizeParams, "grove_gemm_f32_params": GemmF32Params, "grove_gemm_fp8_params": GemmFp8Params,
    "grove_gemm_workspace": GemmWorkspace, "grove_gemm_plan": GemmPlan, "grove_greedy_pick_params": GreedyPickParams,
    "grove_wino3d_params": Wino3dParams,
}

# every symbol include/grove_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "grove_version", "grove_last_error", "grove_set_deterministic", "grove_deterministic", "grove_sizeof", "grove_gemm_bf16", "grove_gemm_make_plan", "grove_gemm_plan_image", "grove_gemm_workspace_bytes", "grove_gemm_fp8_make_plan", "grove_gemm_fp8_plan_image", "grove_gemm_last_variant", "grove_gemm_last_epilogue", "grove_gemm_set_staging", "grove_gemm_set_stream_k", "grove_gemm_set_persistent_blocks", "grove_gemm_persistent_blocks", "grove_gemm_set_tap_skip", "grove_gemm_work_list", "grove_gemm_last_stream_k", "grove_gemm_set_tile_n", "grove_gemm_set_tile_m", "grove_gemm_set_bk", "grove_gemm_tn_bf16", "grove_gemm_tn_set_pipelined", "grove_gemm_tn_set_split_tail", "grove_gemm_tn_last_parts", "grove_gemm_tn_set_tap_skip", "grove_gemm_tn_last_skip", "grove_gemv_bf16", "grove_gemv_set_mfma", "grove_gemv_uses_mfma", "grove_decode_attn", "grove_greedy_pick", "grove_resample_u8", "grove_normalize_pack",
    "grove_transpose_bf16", "grove_layernorm_fwd", "grove_rmsnorm_fwd", "grove_layernorm_bwd", "grove_rmsnorm_bwd",
    "grove_flash_attn_fwd", "grove_flash_attn_bwd", "grove_flash_attn_set_window_kernels", "grove_flash_attn_set_register_e", "grove_flash_attn_set_v2", "grove_flash_attn_window_kernels_on", "grove_softmax_fwd", "grove_softmax_bwd", "grove_relpos_fwd", "grove_relpos_bwd", "grove_rel_bias_fwd", "grove_rel_bias_bwd", "grove_rope_inplace",
    "grove_swiglu_fwd", "grove_swiglu_bwd", "grove_act_bwd", "grove_act_fwd", "grove_resize_bilinear_f32", "grove_add_bf16", "grove_add_bcast_rows",
    "grove_copy_rows", "grove_dot_bf16", "grove_axpy_f32", "grove_scatter_add_f32", "grove_segment_sum_rows", "grove_transpose_many", "grove_scatter_add_rows_f32", "grove_colsum_f32", "grove_cast_f32_to_bf16", "grove_cast_bf16_to_f32",
    "grove_im2col_patch", "grove_clip_pool", "grove_cross_entropy", "grove_small_attn_fwd", "grove_small_attn_bwd", "grove_small_attn_set_tiny", "grove_small_attn_bwd_stores_bf16", "grove_gemm_f32", "grove_gemm_fp8", "grove_gemm_fp8_set_pipelined", "grove_quant_fp8_rows", "grove_quant_fp8_rows_act",
    "grove_box_head_fwd", "grove_box_head_bwd", "grove_box_losses", "grove_adamw_step", "grove_adamw_step_multi", "grove_sumsq_f32",
    "grove_wino3d_transform_tokens", "grove_wino3d_transform_weight", "grove_wino3d_output", "grove_wino3d_wgrad_output",
]

_lib = None


def lib():
    """Load libgrove_hip.so once. Raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C grove_amd/csrc`). grove_amd has no non-HIP fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name in SYMBOLS:
            getattr(_lib, name).restype = C.c_int
        _lib.grove_gemm_workspace_bytes.restype = C.c_size_t
        if os.environ.get("GROVE_GEMM_TAP_SKIP") is not None:  # A/B arm: 0 = the Conv3d GEMMs run every tap group on every tile
            _lib.grove_gemm_set_tap_skip(int(os.environ["GROVE_GEMM_TAP_SKIP"]))
            _lib.grove_gemm_tn_set_tap_skip(int(os.environ["GROVE_GEMM_TAP_SKIP"]))
        if os.environ.get("GROVE_GEMM_BLOCKS") is not None:    # A/B runs at N > 1: resident blocks of the persistent GEMMs (grove_hip.h)
            _lib.grove_gemm_set_persistent_blocks(int(os.environ["GROVE_GEMM_BLOCKS"]))
        if os.environ.get("GROVE_GEMM_STREAM_K") is not None:  # A/B runs of whole programs: 0 = whole tiles only (grove_hip.h)
            _lib.grove_gemm_set_stream_k(int(os.environ["GROVE_GEMM_STREAM_K"]))
        if os.environ.get("GROVE_FLASH_V2") is not None:       # A/B runs of whole programs: which attention launches take the eight-wave kernels
            _lib.grove_flash_attn_set_v2(int(os.environ["GROVE_FLASH_V2"]))
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().grove_last_error(buf, C.c_size_t(512))
    return buf.value.decode("utf-8", "replace")


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}: {last_error()}")
