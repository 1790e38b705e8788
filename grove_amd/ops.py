"""Torch-tensor front end of the C-ABI in include/grove_hip.h.

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic is done by the
hand-written gfx950 kernels in grove_amd/csrc. Nothing in this module falls back to torch ops.
Tensors are bf16 unless stated; every function launches on torch's current stream.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_SIGMOID, ACT_SILU, ACT_SWIGLU_BWD, ACT_SWIGLU_PAIR, BF16, F32)  # noqa: F401

bf16 = torch.bfloat16


_last_dev = [-1]  # device index of the last tensor whose pointer was taken (every launch takes its pointers before its stream)


def _stream():
    """The HIP stream of the launch = torch's current stream on the CURRENT device. Kernels are launched on the calling thread's
    current HIP device, so the tensors must live there: one process per GPU calls torch.cuda.set_device(local_rank) once
    (GROVEForCausalLM / GroveEngine do it); a launch whose tensors sit on another device is refused instead of faulting."""
    s = torch.cuda.current_stream()
    if _det_env:
        set_deterministic(True)
    if _last_dev[0] == -2:
        raise RuntimeError("grove_amd ops need device tensors (no CPU fallback exists)")
    if _last_dev[0] != s.device_index and _last_dev[0] >= 0:
        raise RuntimeError(f"grove_amd: tensors on cuda:{_last_dev[0]} but the current device is cuda:{s.device_index}; "
                           "call torch.cuda.set_device(local_rank) in every rank before using the model")
    return C.c_void_p(s.cuda_stream)


def _p(t):
    if t is None:
        return C.c_void_p(0)
    _last_dev[0] = t.device.index if t.is_cuda else -2
    return C.c_void_p(t.data_ptr())


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("grove_amd ops need device tensors (no CPU fallback exists)")


def pad_to(n, m):
    return (n + m - 1) // m * m


# ---- caller-owned workspaces of the persistent GEMMs (grove_hip.h "Workspaces of the persistent GEMMs")
# The library allocates nothing: this front end keeps, per device, ONE device image of every plan key it has launched (the work list:
# read-only, shared by all streams / epilogues / bf16 and fp8) and, per (device, stream), ONE scratch tensor for the stream-K partial
# tiles (launches of a stream are ordered, so they can share it; two streams may overlap, so they cannot). All of it is torch memory.
_gemm_images = {}    # (device index, plan key) -> uint8 device tensor
_gemm_scratch = {}   # (device index, stream handle) -> uint8 device tensor (grown on demand)


def gemm_workspace(plan, image_fn, device):
    """GemmWorkspace for `plan` (a filled _lib.GemmPlan) on the current stream, or None when the kernel needs none.
    image_fn(host_buffer, nbytes): the library call that writes the plan image into a host buffer."""
    if plan.image_bytes == 0:
        return None
    dev = device.index if device.index is not None else torch.cuda.current_device()
    # (the 64-bit FNV key alone could collide between two images of equal size; the geometry it hashes is part of the key)
    key = (dev, int(plan.key), int(plan.bm), int(plan.tiles_m), int(plan.tiles_n), int(plan.k_tiles), int(plan.grid), int(plan.stream_k),
           int(plan.image_bytes))
    img = _gemm_images.get(key)
    if img is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("grove_amd: first use of a GEMM shape inside a stream capture — its work-list image has to be uploaded "
                               "first (run the shape once before capturing; the C-ABI itself is capture-safe with a caller workspace)")
        n = int(plan.image_bytes)
        host = torch.empty(n, dtype=torch.uint8)
        image_fn(C.c_void_p(host.data_ptr()), C.c_size_t(n))
        img = host.to(device)  # (one synchronous upload per plan key for the life of the process)
        _gemm_images[key] = img
    w = _lib.GemmWorkspace()
    w.image, w.image_bytes = C.c_void_p(img.data_ptr()), img.numel()
    if plan.scratch_bytes:
        skey = (dev, torch.cuda.current_stream().cuda_stream)
        sc = _gemm_scratch.get(skey)
        if sc is None or sc.numel() < plan.scratch_bytes:
            if torch.cuda.is_current_stream_capturing():
                # a capture stream the cache has not seen: a temporary from the graph's private pool. The scratch is live only between
                # the GEMM launch and its fix-up launch (both inside this call), and the pool re-issues a freed block only to LATER
                # nodes of the same capture, which run after them — so it is not kept
                sc = torch.empty(int(plan.scratch_bytes), dtype=torch.uint8, device=device)
            else:
                sc = torch.empty(max(int(plan.scratch_bytes), 64 << 20), dtype=torch.uint8, device=device)
                _gemm_scratch[skey] = sc
        w.scratch, w.scratch_bytes = C.c_void_p(sc.data_ptr()), sc.numel()
        w._keep = sc
    return w


# Set by train.GradExchange while gradient buckets are in flight on RCCL (N > 1): called before every persistent-GEMM launch so the
# exchange can lift its CU reservation as soon as the host sees the collectives complete (None otherwise: no per-launch cost).
_pre_gemm_hook = None


_det_ring = None
_det_env = os.environ.get("GROVE_DETERMINISTIC", "0") not in ("", "0")


def set_deterministic(on: bool, device=None):
    """Deterministic mode (include/grove_hip.h: grove_set_deterministic; GROVE_DETERMINISTIC=1 turns it on at the first op):
    fixed-order sums instead of fp32 atomics, for the race tests. The ticket ring is this module's tensor. Returns the previous setting."""
    global _det_ring, _det_env
    _det_env = False
    lib = _lib.lib()
    prev = bool(lib.grove_deterministic())
    if on:
        if _det_ring is None:
            _det_ring = torch.zeros(4096, dtype=torch.int32, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
            torch.cuda.synchronize()
        _lib.check(lib.grove_set_deterministic(_p(_det_ring), _det_ring.numel()), "grove_set_deterministic")
    else:
        _lib.check(lib.grove_set_deterministic(None, 0), "grove_set_deterministic")
    return prev


_stream_k_mode = [int(os.environ.get("GROVE_GEMM_STREAM_K", "1"))]  # what grove_gemm_set_stream_k was last given (the library has no getter)


def gemm_set_stream_k(mode: int):
    """Stream-K tail of the persistent GEMMs: 0 never / 1 where it pays / 2 wherever it applies; returns the previous setting."""
    prev = _stream_k_mode[0]
    _lib.check(_lib.lib().grove_gemm_set_stream_k(int(mode)), "grove_gemm_set_stream_k")
    _stream_k_mode[0] = int(mode)
    return prev


def gemm_set_persistent_blocks(n: int):
    """Resident blocks of the persistent GEMMs (0 = one per CU); returns the previous setting."""
    lib = _lib.lib()
    prev = int(lib.grove_gemm_persistent_blocks())
    _lib.check(lib.grove_gemm_set_persistent_blocks(int(n)), "grove_gemm_set_persistent_blocks")
    return prev


def gemm_raw(A, B, Cout, M, N, K, lda, ldb, ldc, *, bias=None, residual=None, ldr=0, aux=None, scale_ptr=None,
             scale_tanh=False, a_idx=None, a_taps=1, c_idx=None, r_idx=None, batch=(1, 1), sA=(0, 0), sB=(0, 0),
             sC=(0, 0), sR=(0, 0), act=ACT_NONE, accumulate=False, alpha=1.0, split_k=0, ld_aux=0, n_map=(0, 0), k_map=(0, 0),
             aux_grad=False, residual_mul=False, a_frames=(0, 0), b_group=0):
    """Direct call of grove_gemm_bf16; A/B/Cout are tensors whose storage the pointers refer to.
    a_frames = (rows per frame, frames per group): the temporal-padding promise about a_idx (grove_gemm_params.a_frame_rows).
    b_group = rows per group: grouped B, the groups' [N, ldb] matrices stacked in B (grove_gemm_params.b_group_rows)."""
    _chk_dev(A, B, Cout)
    if _pre_gemm_hook is not None:
        _pre_gemm_hook()
    p = _lib.GemmParams()
    p.A, p.B, p.C = _p(A), _p(B), _p(Cout)
    p.bias, p.residual, p.aux = _p(bias), _p(residual), _p(aux)
    p.scale_ptr, p.a_idx, p.c_idx, p.r_idx = _p(scale_ptr), _p(a_idx), _p(c_idx), _p(r_idx)
    p.sA1, p.sA2, p.sB1, p.sB2 = sA[0], sA[1], sB[0], sB[1]
    p.sC1, p.sC2, p.sR1, p.sR2 = sC[0], sC[1], sR[0], sR[1]
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc, p.ldr = lda, ldb, ldc, ldr
    p.batch1, p.batch2 = batch
    p.a_taps, p.act = a_taps, act
    p.c_dtype = F32 if Cout.dtype == torch.float32 else BF16
    p.accumulate, p.scale_tanh, p.alpha = int(accumulate), int(scale_tanh), float(alpha)
    p.split_k = split_k
    p.ld_aux = ld_aux
    p.aux_grad = int(aux_grad)
    p.residual_mul = int(residual_mul)
    p.n_group, p.n_pad = n_map
    p.k_group, p.k_pad = k_map
    p.a_frame_rows, p.a_frames = a_frames
    p.b_group_rows = int(b_group)
    lib = _lib.lib()
    st = _stream()
    plan = _lib.GemmPlan()
    _lib.check(lib.grove_gemm_make_plan(C.byref(p), C.byref(plan)), "grove_gemm_make_plan")
    w = gemm_workspace(plan, lambda buf, n: _lib.check(lib.grove_gemm_plan_image(C.byref(p), buf, n), "grove_gemm_plan_image"), Cout.device)
    _lib.check(lib.grove_gemm_bf16(C.byref(p), C.byref(w) if w is not None else None, st), "grove_gemm_bf16")
    return Cout


def linear(x, w, bias=None, *, act=ACT_NONE, residual=None, out=None, out_dtype=bf16, aux=None, alpha=1.0,
           scale_ptr=None, scale_tanh=False, a_idx=None, a_taps=1, c_idx=None, r_idx=None, M=None, out_rows=None,
           accumulate=False, ld_aux=0, n_map=(0, 0), k_map=(0, 0), out_cols=None, aux_grad=False, residual_mul=False, a_frames=(0, 0)):
    """y = epilogue(x @ w.T): x [*, K] (row stride lda), w [N, K] (nn.Linear layout)."""
    K = w.shape[1]
    N = w.shape[0]
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
    if M is None:
        M = x2.shape[0]
    assert x2.stride(1) == 1 and w.stride(1) == 1
    if a_idx is None and k_map == (0, 0):
        assert x2.shape[1] == K, f"linear: x has K={x2.shape[1]}, weight has K={K}"
    if out is None:
        rows = out_rows if out_rows is not None else M
        cols = out_cols if out_cols is not None else (N // 2 if act == ACT_SWIGLU_PAIR else 2 * N if act == ACT_SWIGLU_BWD else N)
        out = torch.empty((rows, cols), dtype=out_dtype, device=x.device)
    ldr = residual.stride(0) if residual is not None else 0
    gemm_raw(x2, w, out, M, N, K, x2.stride(0), w.stride(0), out.stride(0), bias=bias, residual=residual, ldr=ldr,
             aux=aux, scale_ptr=scale_ptr, scale_tanh=scale_tanh, a_idx=a_idx, a_taps=a_taps, c_idx=c_idx, r_idx=r_idx,
             act=act, accumulate=accumulate, alpha=alpha, ld_aux=ld_aux, n_map=n_map, k_map=k_map,
             aux_grad=aux_grad, residual_mul=residual_mul, a_frames=a_frames)
    return out


def wgrad(dy, x, grad, *, b_idx=None, b_taps=1, scale_ptr=None, scale_tanh=False, alpha=1.0, split_k=0, K=None, b_frames=(0, 0),
          k_batches=0, sC_batch=0, overwrite=False, M=None, N=None):
    """grad[M, N] (fp32) += alpha * dy[K, M]^T @ x[K, N]  (x rows optionally gathered per tap).
    b_frames = (rows per frame, frames per group): the temporal-padding promise about b_idx (grove_gemm_tn_params.b_frame_rows).
    k_batches > 1: dy / x are k_batches stacked operands of K rows, batch b lands in grad + b * sC_batch elements (= instead of += with
    `overwrite`); M / N then name one product's shape (grove_gemm_tn_params.k_batches)."""
    _chk_dev(dy, x, grad)
    if _pre_gemm_hook is not None:
        _pre_gemm_hook()
    p = _lib.GemmTnParams()
    p.A, p.B, p.C, p.scale_ptr, p.b_idx = _p(dy), _p(x), _p(grad), _p(scale_ptr), _p(b_idx)
    p.M, p.N = (grad.shape[0] if M is None else M), (grad.shape[1] if N is None else N)
    p.K = K if K is not None else dy.shape[0]
    p.k_batches, p.sC_batch, p.overwrite = int(k_batches), int(sC_batch), int(overwrite)
    p.lda, p.ldb, p.ldc = dy.stride(-2), x.stride(-2), grad.stride(-2)
    p.b_taps, p.scale_tanh, p.split_k, p.alpha = b_taps, int(scale_tanh), split_k, float(alpha)
    p.b_frame_rows, p.b_frames = b_frames
    _lib.check(_lib.lib().grove_gemm_tn_bf16(C.byref(p), _stream()), "grove_gemm_tn_bf16")
    return grad


# ---- Winograd F(2x2x2, 3x3x3) form of the Conv3d adapters (grove_hip.h "grove_wino3d_*"; image_encoder.py:43-59)
def _wino_params(src, dst, geom, Cc, frames=(0, 0), tiles_ld=0):
    p = _lib.Wino3dParams()
    p.src, p.dst = _p(src), _p(dst)
    p.groups, p.T, p.H, p.W = geom
    p.C = Cc
    p.alpha = 1.0
    p.frame_rows, p.row_offset = frames
    p.tiles_ld = tiles_ld
    return p


def _wino_rows(geom, frames):
    return geom[0] * geom[1] * (frames[0] if frames[0] else geom[2] * geom[3])


def wino3d_tiles(geom):
    g, T, H, W = geom
    return g * (T // 2) * (H // 2) * (W // 2)


def wino3d_transform_tokens(x, geom, mode, out=None, frames=(0, 0), tiles_ld=0):
    """x bf16 tokens [groups*T*H*W, C] -> bf16 [64, tiles, C]: mode 0 = input tiles (B^T), mode 1 = output-gradient tiles (A).
    frames = (rows per frame, row of a frame's first token): token tensors with extra rows per frame (CLIP's CLS); tiles_ld: rows per
    transform point of the output (>= tiles: padded so that a point is whole GEMM tiles; the pad rows are left as they are)."""
    _chk_dev(x)
    Cc = x.shape[1]
    tiles = wino3d_tiles(geom)
    tl = tiles_ld or tiles
    assert x.shape[0] == _wino_rows(geom, frames) and x.dtype == bf16 and x.stride(1) == 1
    if out is None:
        out = torch.empty((64, tl, Cc), dtype=bf16, device=x.device)
    assert out.is_contiguous() and out.shape == (64, tl, Cc)
    p = _wino_params(x, out, geom, Cc, frames, tiles_ld)
    p.ld_src, p.ld_dst, p.mode = x.stride(0), Cc, mode
    _lib.check(_lib.lib().grove_wino3d_transform_tokens(C.byref(p), _stream()), "grove_wino3d_transform_tokens")
    return out


def wino3d_transform_weight(w, out=None):
    """w bf16 [Co, 27 * Ci] (tap-major packed Conv3d weight) -> bf16 [64, Co, Ci]."""
    _chk_dev(w)
    Co, Ci = w.shape[0], w.shape[1] // 27
    assert w.shape[1] == 27 * Ci and w.dtype == bf16 and w.stride(1) == 1
    if out is None:
        out = torch.empty((64, Co, Ci), dtype=bf16, device=w.device)
    p = _wino_params(w, out, (0, 0, 0, 0), Ci)
    p.rows, p.ld_src, p.ld_dst = Co, w.stride(0), Ci
    _lib.check(_lib.lib().grove_wino3d_transform_weight(C.byref(p), _stream()), "grove_wino3d_transform_weight")
    return out


def wino3d_output(Mh, geom, out, *, bias=None, act=ACT_NONE, scale_ptr=None, scale_tanh=False, residual=None, aux=None, frames=(0, 0)):
    """Mh bf16 [64, tiles_ld, C] -> out tokens: act((A^T..) Mh + bias) * scale + residual; aux = the pre-activation. Rows of `out` that
    are not tokens (frames = (rows per frame, first token row): CLIP's CLS rows) are not touched."""
    _chk_dev(Mh, out)
    Cc = Mh.shape[2]
    assert Mh.is_contiguous() and Mh.shape[0] == 64 and Mh.shape[1] >= wino3d_tiles(geom) and out.shape == (_wino_rows(geom, frames), Cc)
    p = _wino_params(Mh, out, geom, Cc, frames, Mh.shape[1] if Mh.shape[1] != wino3d_tiles(geom) else 0)
    p.bias, p.residual, p.aux, p.scale_ptr = _p(bias), _p(residual), _p(aux), _p(scale_ptr)
    p.ld_src, p.ld_dst = Cc, out.stride(0)
    p.ld_res = residual.stride(0) if residual is not None else 0
    p.ld_aux = aux.stride(0) if aux is not None else 0
    p.act, p.scale_tanh = act, int(scale_tanh)
    _lib.check(_lib.lib().grove_wino3d_output(C.byref(p), _stream()), "grove_wino3d_output")
    return out


def wino3d_wgrad_output(dU, gw, *, scale_ptr=None, scale_tanh=False):
    """gw f32 [Co, 27 * Ci] += scale * (G^T..) dU, dU f32 [64, Co, Ci]."""
    _chk_dev(dU, gw)
    _, Co, Ci = dU.shape
    assert dU.is_contiguous() and dU.dtype == torch.float32 and gw.dtype == torch.float32 and gw.shape == (Co, 27 * Ci) and gw.stride(1) == 1
    p = _wino_params(dU, gw, (0, 0, 0, 0), Ci)
    p.scale_ptr, p.scale_tanh = _p(scale_ptr), int(scale_tanh)
    p.rows, p.ld_src, p.ld_dst = Co, Ci, gw.stride(0)
    _lib.check(_lib.lib().grove_wino3d_wgrad_output(C.byref(p), _stream()), "grove_wino3d_wgrad_output")
    return gw


def _ws_tensor(ws, key, shape, dtype, device):
    """A [shape] view of the caller's persistent workspace buffer `key` (grown when too small): big temporaries that come back every call
    are allocated once — a fresh multi-GB block from the allocator is mapped page by page under the first kernel that writes it (measured:
    +0.35 s on a step whose two 6.7 GB Winograd temporaries were fresh, against 1.40 s when the allocator happened to hand back mapped ones)."""
    n = 1
    for d_ in shape:
        n *= int(d_)
    buf = ws.get(key)
    if buf is None or buf.numel() < n or buf.dtype != dtype or buf.device != device:
        buf = ws[key] = torch.empty(n, dtype=dtype, device=device)
    return buf[:n].view(*shape)


def wino3d_conv(x, U, geom, out, *, bias=None, act=ACT_NONE, scale_ptr=None, scale_tanh=False, residual=None, aux=None, V=None, keep_V=False,
                frames=(0, 0), ws=None):
    """out = epilogue(Conv3d 3x3x3 'same' of the token tensor x with the TRANSFORMED weights U [64, Co, Ci]) — transform, ONE grouped GEMM
    over the 64 transform points, output transform. Returns (out, V) with V the transformed input when keep_V (the weight gradient reads it).
    ws (dict or None): the caller's persistent workspace for the two [64, tiles, C] temporaries (inference: the same sizes every call).
    A tile count that is not a multiple of 256 is padded per point (the grouped GEMM's groups are whole 256-row tiles): the pad rows of V
    are whatever the allocator left — rows of a GEMM do not mix, and the output transform never reads theirs."""
    tiles = wino3d_tiles(geom)
    tl = pad_to(tiles, 256)
    Co, Ci = U.shape[1], U.shape[2]
    if V is None:
        Vb = _ws_tensor(ws, "V", (64, tl, Ci), bf16, x.device) if (ws is not None and not keep_V) else None
        V = wino3d_transform_tokens(x, geom, 0, out=Vb, frames=frames, tiles_ld=tl if tl != tiles else 0)
    Mh = _ws_tensor(ws, "M", (64, tl, Co), bf16, x.device) if ws is not None else torch.empty((64, tl, Co), dtype=bf16, device=x.device)
    gemm_raw(V, U, Mh, 64 * tl, Co, Ci, Ci, Ci, Co, b_group=tl)
    wino3d_output(Mh, geom, out, bias=bias, act=act, scale_ptr=scale_ptr, scale_tanh=scale_tanh, residual=residual, aux=aux, frames=frames)
    return out, (V if keep_V else None)


def wino3d_wgrad(dz, V, geom, gw, *, scale_ptr=None, scale_tanh=False):
    """gw f32 [Co, 27 Ci] += scale * dW of the Conv3d, from the output gradient dz (tokens [rows, Co]) and the transformed input V [64, tiles, Ci]:
    per transform point dU = dM^T V as ONE K-batched TN GEMM, then the G^T transform."""
    tiles = wino3d_tiles(geom)
    Co, Ci = dz.shape[1], V.shape[2]
    dM = wino3d_transform_tokens(dz, geom, 1)
    dU = torch.empty((64, Co, Ci), dtype=torch.float32, device=dz.device)
    wgrad(dM, V, dU, K=tiles, k_batches=64, sC_batch=Co * Ci, overwrite=True, M=Co, N=Ci)
    return wino3d_wgrad_output(dU, gw, scale_ptr=scale_ptr, scale_tanh=scale_tanh)


def gemm_set_tile_n(n: int):
    _lib.check(_lib.lib().grove_gemm_set_tile_n(int(n)), "grove_gemm_set_tile_n")


def gemm_set_staging(use_lds_dma: bool):
    _lib.check(_lib.lib().grove_gemm_set_staging(int(use_lds_dma)), "grove_gemm_set_staging")


def transpose(x, rows, cols, ld_in, out, ld_out, *, pad_to_cols=None, batch=(1, 1), s_in=(0, 0), s_out=(0, 0)):
    _chk_dev(x, out)
    p = _lib.TransposeParams()
    p.inp, p.out = _p(x), _p(out)
    p.s_in1, p.s_in2, p.s_out1, p.s_out2 = s_in[0], s_in[1], s_out[0], s_out[1]
    p.rows, p.cols, p.ld_in, p.ld_out = rows, cols, ld_in, ld_out
    p.pad_to = pad_to_cols if pad_to_cols is not None else rows
    p.batch1, p.batch2 = batch
    _lib.check(_lib.lib().grove_transpose_bf16(C.byref(p), _stream()), "grove_transpose_bf16")
    return out


def transpose2d(x, pad_cols_to=None):
    """[R, C] -> [C, R(pad)] contiguous"""
    R, Cc = x.shape
    ld_out = pad_cols_to if pad_cols_to is not None else R
    out = torch.empty((Cc, ld_out), dtype=bf16, device=x.device)
    return transpose(x, R, Cc, x.stride(0), out, ld_out, pad_to_cols=ld_out)


_TRANSPOSE_MANY = {}  # (data pointers, shapes) of a weight set -> (device item array, persistent outputs, total tiles)


def transpose2d_many(ws):
    """[w.t().contiguous() for w in ws] for a list of bf16 matrices in ONE launch (grove_transpose_many). The outputs are persistent buffers owned
    by this cache — keyed by the set's data pointers — and are REWRITTEN by every call: call it where the values are needed (a backward pass),
    never keep them across an optimizer step."""
    key = tuple((w.data_ptr(), w.shape[0], w.shape[1], w.stride(0)) for w in ws)
    ent = _TRANSPOSE_MANY.get(key)
    if ent is None:
        _chk_dev(*ws)
        outs, items, tile0 = [], (_lib.TransposeItem * len(ws))(), 0
        for i, w in enumerate(ws):
            assert w.dtype == bf16 and w.dim() == 2 and w.stride(1) == 1
            R, Cc = w.shape
            o = torch.empty((Cc, R), dtype=bf16, device=w.device)
            outs.append(o)
            items[i].src, items[i].dst, items[i].rows, items[i].cols = _p(w), _p(o), R, Cc
            items[i].ld_src, items[i].ld_dst, items[i].tile0 = w.stride(0), R, tile0
            tile0 += ((R + 63) // 64) * ((Cc + 63) // 64)
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(ws[0].device)
        ent = (raw, outs, tile0, list(ws))  # (the weights are kept alive with the entry: the key is made of their addresses)
        _TRANSPOSE_MANY[key] = ent
    raw, outs, total, _ = ent
    _lib.check(_lib.lib().grove_transpose_many(_p(raw), len(outs), total, _stream()), "grove_transpose_many")
    return outs


def _norm_fwd(fn, name, x, weight, bias, eps, out, out_idx, save_stats, out_dtype, out_rows, res=None, res_bf16=None, want_y=True):
    """res (fp32 [rows, C], updated in place): the residual-stream form — normalises res + x (x may be None), see grove_hip.h."""
    if res is not None:
        rows, Cc = res.shape
        assert res.dtype == torch.float32 and res.stride(1) == 1
        assert x is None or (x.shape == res.shape and x.dtype == bf16)
        assert res_bf16 is None or (res_bf16.is_contiguous() and res_bf16.shape == res.shape and res_bf16.dtype == bf16)
        dev_t = res
    else:
        rows, Cc = x.shape
        dev_t = x
    _chk_dev(dev_t, weight)
    x_dev = dev_t.device
    if not want_y:
        p = _lib.NormParams()
        p.x, p.res, p.res_bf16 = _p(x), _p(res), _p(res_bf16)
        p.rows, p.C, p.ld_x, p.ld_res, p.eps = rows, Cc, (x.stride(0) if x is not None else 0), res.stride(0), eps
        _lib.check(fn(C.byref(p), _stream()), name)
        return None, None, None
    if out is None:
        orows = out_rows if out_rows is not None else rows
        if out_idx is not None:
            out = torch.zeros((orows, Cc), dtype=out_dtype, device=x_dev)
        else:
            out = torch.empty((orows, Cc), dtype=out_dtype, device=x_dev)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x_dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=x_dev)
    p = _lib.NormParams()
    p.x, p.weight, p.bias, p.y = _p(x), _p(weight), _p(bias), _p(out)
    p.mean, p.rstd, p.out_idx = _p(mean), _p(rstd), _p(out_idx)
    p.res, p.res_bf16 = _p(res), _p(res_bf16)
    p.rows, p.C, p.ld_x, p.ld_y = rows, Cc, (x.stride(0) if x is not None else 0), out.stride(0)
    p.ld_res = res.stride(0) if res is not None else 0
    p.y_dtype = F32 if out.dtype == torch.float32 else BF16
    p.eps = eps
    _lib.check(fn(C.byref(p), _stream()), name)
    return out, mean, rstd


def layernorm(x, weight, bias, eps, *, out=None, out_idx=None, save_stats=False, out_dtype=bf16, out_rows=None, res=None,
              res_bf16=None):
    return _norm_fwd(_lib.lib().grove_layernorm_fwd, "grove_layernorm_fwd", x, weight, bias, eps, out, out_idx,
                     save_stats, out_dtype, out_rows, res=res, res_bf16=res_bf16)


def rmsnorm(x, weight, eps, *, out=None, res=None, res_bf16=None, out_dtype=bf16):
    return _norm_fwd(_lib.lib().grove_rmsnorm_fwd, "grove_rmsnorm_fwd", x, weight, None, eps, out, None, False, out_dtype,
                     None, res=res, res_bf16=res_bf16)[0]


def stream_add(res, x, res_bf16=None):
    """res (fp32 residual stream) += x (bf16 branch output); optionally the bf16 rounding of the updated stream. No norm."""
    _norm_fwd(_lib.lib().grove_layernorm_fwd, "grove_layernorm_fwd", x, None, None, 0.0, None, None, False, bf16, None,
              res=res, res_bf16=res_bf16, want_y=False)
    return res_bf16


def _norm_bwd(fn, name, x, weight, dy, mean, rstd, eps, dx, dweight, dbias, in_idx, accumulate):
    rows, Cc = x.shape
    if dx is None:
        dx = torch.empty((rows, Cc), dtype=bf16, device=x.device)
    p = _lib.NormBwdParams()
    p.x, p.weight, p.dy, p.dx = _p(x), _p(weight), _p(dy), _p(dx)
    p.mean, p.rstd, p.dweight, p.dbias, p.in_idx = _p(mean), _p(rstd), _p(dweight), _p(dbias), _p(in_idx)
    p.rows, p.C, p.ld_x, p.ld_dy, p.ld_dx = rows, Cc, x.stride(0), dy.stride(0), dx.stride(0)
    p.accumulate, p.eps = int(accumulate), eps
    _lib.check(fn(C.byref(p), _stream()), name)
    return dx


def layernorm_bwd(x, weight, dy, mean, rstd, *, dx=None, dweight=None, dbias=None, in_idx=None, accumulate=False):
    return _norm_bwd(_lib.lib().grove_layernorm_bwd, "grove_layernorm_bwd", x, weight, dy, mean, rstd, 0.0, dx,
                     dweight, dbias, in_idx, accumulate)


def rmsnorm_bwd(x, weight, dy, eps, *, dx=None, dweight=None, accumulate=False):
    return _norm_bwd(_lib.lib().grove_rmsnorm_bwd, "grove_rmsnorm_bwd", x, weight, dy, None, None, eps, dx, dweight,
                     None, None, accumulate)


def softmax(scores, Lk, *, heads=1, causal=False, kv_len=None, rel=None, rel_hw=(0, 0), ld_p=None, out=None):
    """scores f32 [batch, Lq, ld_s] -> probs bf16 [batch, Lq, ld_p] (pad columns zeroed)."""
    batch, Lq, ld_s = scores.shape
    if ld_p is None:
        ld_p = pad_to(Lk, 32)
    if out is None:
        out = torch.empty((batch, Lq, ld_p), dtype=bf16, device=scores.device)
    p = _lib.SoftmaxParams()
    p.scores, p.probs, p.kv_len, p.rel = _p(scores), _p(out), _p(kv_len), _p(rel)
    p.batch, p.heads, p.Lq, p.Lk, p.ld_s, p.ld_p = batch, heads, Lq, Lk, ld_s, ld_p
    p.causal, p.rel_kh, p.rel_kw = int(causal), rel_hw[0], rel_hw[1]
    _lib.check(_lib.lib().grove_softmax_fwd(C.byref(p), _stream()), "grove_softmax_fwd")
    return out


def softmax_bwd(dprobs, probs, Lk, scale, *, drel=None, rel_hw=(0, 0), out=None):
    batch, Lq, ld_s = dprobs.shape
    ld_p = probs.shape[2]
    if out is None:
        out = torch.empty_like(probs)
    p = _lib.SoftmaxBwdParams()
    p.dprobs, p.probs, p.dscores, p.drel = _p(dprobs), _p(probs), _p(out), _p(drel)
    p.batch, p.Lq, p.Lk, p.ld_s, p.ld_p = batch, Lq, Lk, ld_s, ld_p
    p.rel_kh, p.rel_kw, p.scale = rel_hw[0], rel_hw[1], scale
    _lib.check(_lib.lib().grove_softmax_bwd(C.byref(p), _stream()), "grove_softmax_bwd")
    return out


def window_kernels_take(L, hs, hs_valid, rel_ld):
    """Will the LDS-resident window kernels (win_attn.hip) run this attention problem? (They honour q_valid; the general ones do not.)"""
    return bool(_lib.lib().grove_flash_attn_window_kernels_on()) and hs == 96 and hs_valid == 80 and 192 < L <= 208 and rel_ld == 32


def flash_attn(qkv, B, L, H, hs, q_off, k_off, v_off, alpha, *, causal=False, kv_len=None, rel=None, rel_hw=(0, 0), out=None,
               want_lse=False, hs_valid=0, q_valid=None, pad_row=None, o_map=None, o_rows=0, rel_table=None, rel_out=None):
    """Fused attention over a fused qkv activation [B*L, ld]; returns (out [B*L, H*hs], lse or None).
    o_map (window kernels, with q_valid): out is [o_rows, H*hs_valid] in TOKEN order, row o_map[b*L + i] for position i of batch b.
    rel_table (window kernels; grove_flash_attn_params.rel_table, see rel_table_images): the rel-pos terms are made inside the kernel; rel_out
    (bf16 [B*H, L, 32], optional) receives the score-domain operand the backward needs."""
    if rel_table is not None:
        assert rel is None and rel_table.dtype == bf16 and rel_table.numel() == 2 * 64 * 80 and rel_table.is_contiguous()
        assert rel_out is None or (rel_out.dtype == bf16 and rel_out.shape == (B * H, L, 32) and rel_out.is_contiguous())
    dev = qkv.device
    ld = qkv.stride(0)
    if out is None:
        out = torch.empty((o_rows, H * hs_valid) if o_map is not None else (B * L, H * hs), dtype=bf16, device=dev)
    lse = torch.empty((B * H, L), dtype=torch.float32, device=dev) if want_lse else None
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o = _p(qkv[:, q_off:]), _p(qkv[:, k_off:]), _p(qkv[:, v_off:]), _p(out)
    p.lse, p.kv_len, p.rel = _p(lse), _p(kv_len), _p(rel)
    p.sq = p.sk = p.sv = L * ld
    p.so = L * out.stride(0)
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, L, L, hs
    p.ld_q = p.ld_k = p.ld_v = ld
    p.ld_o = out.stride(0)
    p.causal, p.rel_kh, p.rel_kw, p.alpha = int(causal), rel_hw[0], rel_hw[1], alpha
    p.rel_ld = rel.shape[-1] if rel is not None else 0
    if rel_table is not None:
        p.rel_table, p.rel, p.rel_ld = _p(rel_table), _p(rel_out), 32
    p.hs_valid = hs_valid
    p.q_valid = _p(q_valid)
    if o_map is not None:
        assert q_valid is not None and hs_valid and o_map.numel() == B * L
        p.o_map, p.o_hs = _p(o_map), hs_valid
    if pad_row is not None:  # the qkv row of a padded position (same column layout as a row of qkv)
        assert q_valid is not None and pad_row.numel() == qkv.shape[1]
        p.pad_k, p.pad_v = _p(pad_row.view(-1)[k_off:]), _p(pad_row.view(-1)[v_off:])
    _lib.check(_lib.lib().grove_flash_attn_fwd(C.byref(p), _stream()), "grove_flash_attn_fwd")
    return out, lse



_GEMV_SPLIT_NORM = os.environ.get("GROVE_GEMV_SPLIT_NORM", "1") != "0"  # A/B arm of the batched decode step


def gemv(x, w, bias=None, *, act=ACT_NONE, residual=None, out_dtype=bf16, out=None, rms_weight=None, eps=0.0, swiglu=False, batch_invariant=False,
         norm_out=None, norm_in=None):
    """y = act(x' @ w.T + bias) + residual for 1..8 rows of x (the cached decode step): the weight-streaming kernel.
    x' = x, or rmsnorm(x) * rms_weight (rms_weight given), or silu(gate) * up of a fused [M, 2K] row (swiglu=True).
    x and residual may be fp32 (the decode step's fp32 residual stream): norm statistics then run on the fp32 values.
    batch_invariant: the matrix-core kernel for every M (grove_gemv_params.force_mfma): a row's bits do not depend on how many other
    sequences share the launch.
    Deferred RMSNorm (grove_gemv_params.xs_out / ssq_out / ssq_in; matrix-core kernel, plain x): norm_out = (next norm's weight [N]) makes this
    launch also return (xs bf16 [M, N] = bf16(y * weight), ssq f32 [ceil(N / 16), 8]) — `out` becomes (y, xs, ssq); norm_in = (ssq, eps) of the
    producer makes this launch multiply row m of its product by rsqrt(sum(ssq[:, m]) / K + eps): x must then be the producer's xs."""
    _chk_dev(x, w)
    M = x.shape[0]
    N, K = w.shape
    assert x.shape[1] == (2 * K if swiglu else K) and x.stride(1) == 1 and w.stride(1) == 1
    assert x.dtype in (bf16, torch.float32) and (residual is None or residual.dtype in (bf16, torch.float32))
    mx = 1 if M == 1 else 2 if M == 2 else 4 if M <= 4 else 8
    # which kernel the library will run is the LIBRARY's decision (its knob, its LDS bound): asked, not re-derived
    q = _lib.GemvParams()
    q.M, q.N, q.K, q.act = M, N, K, act
    q.x_mode = 2 if swiglu else 1 if rms_weight is not None else 0
    q.x_f32 = int(x.dtype == torch.float32)
    q.force_mfma = int(batch_invariant)
    mfma = bool(_lib.lib().grove_gemv_uses_mfma(C.byref(q)))
    if mfma and rms_weight is not None and not swiglu and _GEMV_SPLIT_NORM:
        # 3..8 sequences on the matrix-core kernel: a block owns 16 output rows — 128 KB of weights at K = 4096 — and a folded RMSNorm
        # makes every one of its N / 16 blocks re-read and re-normalise the M x K fp32 rows (another 128 KB, plus the block reductions)
        # before its first product: in the batched decode step the folded form ran 57 us per launch against 24-44 us for the same
        # GEMV on a plain x (round 5, profile of bench.py --mode infer_iground). So the rows are normalised ONCE, by the norm kernel,
        # into M x K bf16 — the rounding the kernel's LDS staging applies anyway — and the GEMV reads them from L2.
        xn = rmsnorm(None, rms_weight, eps, res=x) if x.dtype == torch.float32 else rmsnorm(x, rms_weight, eps)
        return gemv(xn, w, bias, act=act, residual=residual, out_dtype=out_dtype, out=out, batch_invariant=batch_invariant)
    if mx * K * 2 > 159 * 1024 and M > 4 and not mfma:  # (the matrix-core kernel of 3..8 sequences reads a plain x straight from global memory)
        # the kernel keeps its mx rows of x in LDS (bf16): 8 rows of LLaMA-7B's down-projection input (K = 11008) are 176 KB. Two
        # launches of <= 4 rows each (88 KB): the weight matrix is streamed twice for the 5..8 sequences instead of once — still one
        # stream per FOUR sequences (round 5: the clip-batched decode of infer_iground runs through here at B = 8)
        if out is None:
            out = torch.empty((M, N // 2 if act == ACT_SWIGLU_PAIR else N), dtype=out_dtype, device=x.device)
        for lo in range(0, M, 4):
            hi = min(M, lo + 4)
            gemv(x[lo:hi], w, bias, act=act, residual=(residual[lo:hi] if residual is not None else None), out_dtype=out_dtype,
                 out=out[lo:hi], rms_weight=rms_weight, eps=eps, swiglu=swiglu, batch_invariant=batch_invariant)
        return out
    if out is None:  # ACT_SWIGLU_PAIR: w rows interleaved [4 gate, 4 up] (swiglu_interleave), the result is silu(gate) * up: N / 2 columns
        out = torch.empty((M, N // 2 if act == ACT_SWIGLU_PAIR else N), dtype=out_dtype, device=x.device)
    p = _lib.GemvParams()
    p.x, p.W, p.y, p.bias, p.residual = _p(x), _p(w), _p(out), _p(bias), _p(residual)
    p.M, p.N, p.K = M, N, K
    p.ldx, p.ldw, p.ldy = x.stride(0), w.stride(0), out.stride(0)
    p.ldr = residual.stride(0) if residual is not None else 0
    p.act = act
    p.y_dtype = F32 if out.dtype == torch.float32 else BF16
    p.norm_weight, p.eps = _p(rms_weight), float(eps)
    p.x_mode = 2 if swiglu else 1 if rms_weight is not None else 0
    p.x_f32 = int(x.dtype == torch.float32)
    p.res_f32 = int(residual is not None and residual.dtype == torch.float32)
    p.force_mfma = int(batch_invariant)
    xs = ssq = None
    if norm_out is not None:
        xs = torch.empty((M, N), dtype=bf16, device=x.device)
        ssq = torch.empty(((N + 15) // 16, 8), dtype=torch.float32, device=x.device)
        p.xs_out, p.xs_weight, p.ssq_out, p.ld_xs = _p(xs), _p(norm_out), _p(ssq), xs.stride(0)
    if norm_in is not None:
        assert rms_weight is None and not swiglu
        p.ssq_in, p.ssq_in_blocks, p.eps = _p(norm_in[0]), norm_in[0].shape[0], float(norm_in[1])
    _lib.check(_lib.lib().grove_gemv_bf16(C.byref(p), _stream()), "grove_gemv_bf16")
    return (out, xs, ssq) if norm_out is not None else out


def greedy_pick(logits, V, finished, tok, pos, ids_out, eos, pad, pos0, hidden=None, hid_out=None, hidden_f32=None, hid_out_f32=None):
    """HF's greedy bookkeeping between two decoder steps in one launch (grove_greedy_pick): everything in place on device tensors."""
    B = logits.shape[0]
    assert logits.dtype == torch.float32 and finished.dtype == torch.bool and tok.dtype == torch.int32 and pos.dtype == torch.int32
    assert ids_out.dtype == torch.int64 and ids_out.stride(1) == 1
    p = _lib.GreedyPickParams()
    p.logits, p.ld_logits, p.finished, p.tok, p.pos = _p(logits), logits.stride(0), _p(finished), _p(tok), _p(pos)
    p.ids_out, p.ld_ids = _p(ids_out), ids_out.stride(0)
    p.hidden, p.hid_out, p.hidden_f32, p.hid_out_f32 = _p(hidden), _p(hid_out), _p(hidden_f32), _p(hid_out_f32)
    p.B, p.V, p.H = B, int(V), int(hidden.shape[-1]) if hidden is not None else (int(hidden_f32.shape[-1]) if hidden_f32 is not None else 0)
    p.eos, p.pad, p.pos0, p.max_steps = int(eos), int(pad), int(pos0), int(ids_out.shape[1])
    _lib.check(_lib.lib().grove_greedy_pick(C.byref(p), _stream()), "grove_greedy_pick")


_decode_partial = {}  # (device index, floats) -> f32 scratch of the split decode attention (launches of a stream are ordered: shared)


def decode_attn(qkv, cache, pos, H, hd, theta, alpha, out=None, n_split=None):
    """RoPE(q, k) + cache append + one-query attention for the new token of every sequence (grove_decode_attn).
    cache: bf16 [B, 2, H, S_max, hd] (keys / values planes, head-major), contiguous."""
    assert cache.dim() == 5 and cache.shape[1] == 2 and cache.shape[2] == H and cache.shape[4] == hd and cache.is_contiguous()
    B, S_max = cache.shape[0], cache.shape[3]
    if out is None:
        out = torch.empty((B, H * hd), dtype=bf16, device=qkv.device)
    p = _lib.DecodeAttnParams()
    p.qkv, p.cache, p.out, p.pos = _p(qkv), _p(cache), _p(out), _p(pos)
    p.B, p.H, p.hd, p.S_max, p.ld_qkv = B, H, hd, S_max, qkv.stride(0)
    p.theta, p.alpha = float(theta), float(alpha)
    if n_split is None:  # one CU streams ~23 GB/s: spread a head over up to 8 blocks while the grid stays within one wave of CUs
        n_split = max(1, min(8, 256 // (B * H)))
    if n_split > 1:
        n = B * H * n_split * (hd + 2)
        key = (qkv.device.index, torch.cuda.current_stream().cuda_stream)
        buf = _decode_partial.get(key)
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1 << 16), dtype=torch.float32, device=qkv.device)
            if not torch.cuda.is_current_stream_capturing():
                _decode_partial[key] = buf
        p.partial, p.n_split = _p(buf), n_split
        p._keep = buf
    _lib.check(_lib.lib().grove_decode_attn(C.byref(p), _stream()), "grove_decode_attn")
    return out


def flash_attn_kv(q, k, v, B, H, Lq, Lk, hs, alpha, *, sq, sk, sv, ld_q, ld_k, ld_v, causal=False, kv_len=None, out=None):
    """Fused attention with separate q / k / v views (decode: one query row per sequence against the KV cache).
    q: [B*Lq rows]; k, v: views whose batch stride is sk / sv elements; returns out [B*Lq, H*hs]."""
    if out is None:
        out = torch.empty((B * Lq, H * hs), dtype=bf16, device=q.device)
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o = _p(q), _p(k), _p(v), _p(out)
    p.kv_len = _p(kv_len)
    p.sq, p.sk, p.sv = sq, sk, sv
    p.so = Lq * out.stride(0)
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, Lq, Lk, hs
    p.ld_q, p.ld_k, p.ld_v = ld_q, ld_k, ld_v
    p.ld_o = out.stride(0)
    p.causal, p.alpha = int(causal), alpha
    _lib.check(_lib.lib().grove_flash_attn_fwd(C.byref(p), _stream()), "grove_flash_attn_fwd")
    return out

def rope_table(hd, theta, positions, device):
    """f32 [positions, hd] = cos[hd / 2] | sin[hd / 2] of the rotate-half RoPE angles pos * theta^(-2 i / hd) (HF LlamaRotaryEmbedding:
    inv_freq in fp32, angles in fp32): the table the backward attention kernels un-rotate dq / dk with (grove_flash_attn_params.rope)."""
    # evaluated on the HOST, where the reference evaluates it (same fp32 formula, same libm): the table then holds the reference's own
    # cos / sin bit for bit; made once per sequence-length high-water mark (llama._rope_table), so the upload is not on the step
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(positions, dtype=torch.float32)[:, None] * inv_freq[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], 1).contiguous().to(device)


def flash_attn_bwd(qkv, out, d_out, lse, dqkv, B, L, H, hs, q_off, k_off, v_off, alpha, *, causal=False, kv_len=None, rel=None,
                   rel_hw=(0, 0), want_drel=False, hs_valid=0, q_valid=None, pad_row=None, o_map=None, rope=None, grads_tok=False, rel_table=None):
    """grads_tok (window kernels with o_map): dqkv is in TOKEN order with compact heads — [tokens, 3 * H * hs_valid], q | k | v blocks of
    H * hs_valid columns — instead of the layout of qkv (grove_flash_attn_params.g_tok).
    rel_table (window kernels): rel is the operand tensor the forward left (rel_out); dq leaves complete, no drel."""
    dev = qkv.device
    ld, ldd = qkv.stride(0), dqkv.stride(0)
    delta = torch.empty((B * H, L), dtype=torch.float32, device=dev)
    assert rel_table is None or (rel is not None and not want_drel)
    drel = torch.empty_like(rel) if want_drel else None
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o, p.d_o = _p(qkv[:, q_off:]), _p(qkv[:, k_off:]), _p(qkv[:, v_off:]), _p(out), _p(d_out)
    if grads_tok:
        assert o_map is not None and pad_row is not None and hs_valid and dqkv.shape[1] == 3 * H * hs_valid
        p.dq, p.dk, p.dv = _p(dqkv), _p(dqkv[:, H * hs_valid:]), _p(dqkv[:, 2 * H * hs_valid:])
        p.g_tok = 1
    else:
        p.dq, p.dk, p.dv = _p(dqkv[:, q_off:]), _p(dqkv[:, k_off:]), _p(dqkv[:, v_off:])
    p.lse, p.delta, p.kv_len, p.rel, p.drel = _p(lse), _p(delta), _p(kv_len), _p(rel), _p(drel)
    p.sq = p.sk = p.sv = L * ld
    p.so, p.sdo = L * out.stride(0), L * d_out.stride(0)
    p.sdq = p.sdk = p.sdv = L * ldd
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, L, L, hs
    p.ld_q = p.ld_k = p.ld_v = ld
    p.ld_o, p.ld_do = out.stride(0), d_out.stride(0)
    p.ld_dq = p.ld_dk = p.ld_dv = ldd
    p.causal, p.rel_kh, p.rel_kw, p.alpha = int(causal), rel_hw[0], rel_hw[1], alpha
    p.rel_ld = rel.shape[-1] if rel is not None else 0
    p.hs_valid = hs_valid
    p.q_valid = _p(q_valid)
    if o_map is not None:
        assert q_valid is not None and hs_valid and o_map.numel() == B * L
        p.o_map, p.o_hs = _p(o_map), hs_valid
    if pad_row is not None:
        assert q_valid is not None and pad_row.numel() == qkv.shape[1]
        p.pad_k, p.pad_v = _p(pad_row.view(-1)[k_off:]), _p(pad_row.view(-1)[v_off:])
    if rope is not None:  # fused inverse RoPE of dq / dk: f32 [>= L, hs]
        assert rope.dtype == torch.float32 and rope.shape[1] == hs and rope.shape[0] >= L and rope.is_contiguous()
        p.rope = _p(rope)
    p.rel_table = _p(rel_table)
    _lib.check(_lib.lib().grove_flash_attn_bwd(C.byref(p), _stream()), "grove_flash_attn_bwd")
    return drel


def rel_table_images(rel_pos_h, rel_pos_w, size, alpha):
    """grove_flash_attn_params.rel_table for size x size windows: T [64, 80] = [rel_pos_h reversed ; rel_pos_w reversed ; 0] / alpha (row r of a
    part = the embedding of key - query = r - (size - 1): image_encoder.py:387-417 get_rel_pos with q_size == k_size, :420-458) followed by
    T^T [80, 64], one bf16 buffer [2 * 5120]. Init-time only."""
    n = 2 * size - 1
    assert rel_pos_h.shape == (n, 80) and rel_pos_w.shape == (n, 80) and 2 * n <= 64
    T = torch.zeros((64, 80), dtype=torch.float32, device=rel_pos_h.device)
    T[:n] = rel_pos_h.float().flip(0) / alpha
    T[n:2 * n] = rel_pos_w.float().flip(0) / alpha
    return torch.cat([T.to(bf16).reshape(-1), T.to(bf16).t().contiguous().reshape(-1)]).contiguous()


def flash_attn_tail(q, kv, B, Lq, Lk, H, hs, alpha, *, kv_len=None, want_lse=False):
    """Causal attention of the LAST Lq positions of every sequence against all Lk keys (bottom-right aligned causal mask: query i is
    position Lk - Lq + i): q bf16 [B*Lq, >= H*hs] (head h at column h*hs), kv bf16 [B*Lk, >= 2*H*hs] (keys at column 0, values at
    H*hs). Returns (out [B*Lq, H*hs], lse [B*H, Lq] or None). The general fused kernels with Lq != Lk (flash_attn.hip)."""
    dev = q.device
    out = torch.empty((B * Lq, H * hs), dtype=bf16, device=dev)
    lse = torch.empty((B * H, Lq), dtype=torch.float32, device=dev) if want_lse else None
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o = _p(q), _p(kv), _p(kv[:, H * hs:]), _p(out)
    p.lse, p.kv_len = _p(lse), _p(kv_len)
    p.sq, p.sk, p.sv, p.so = Lq * q.stride(0), Lk * kv.stride(0), Lk * kv.stride(0), Lq * out.stride(0)
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, Lq, Lk, hs
    p.ld_q, p.ld_k, p.ld_v, p.ld_o = q.stride(0), kv.stride(0), kv.stride(0), out.stride(0)
    p.causal, p.alpha = 1, alpha
    _lib.check(_lib.lib().grove_flash_attn_fwd(C.byref(p), _stream()), "grove_flash_attn_fwd")
    return out, lse


def flash_attn_tail_bwd(q, kv, out, d_out, lse, dq, dkv, B, Lq, Lk, H, hs, alpha, *, kv_len=None, rope=None):
    """Backward of flash_attn_tail: dq [B*Lq, >= H*hs] and dkv [B*Lk, >= 2*H*hs] (d keys | d values) are overwritten.
    rope (f32 [>= Lk, hs], ops.rope_table): dq / dk leave the kernels un-rotated (fused inverse RoPE)."""
    dev = q.device
    delta = torch.empty((B * H, Lq), dtype=torch.float32, device=dev)
    p = _lib.FlashAttnParams()
    p.q, p.k, p.v, p.o, p.d_o = _p(q), _p(kv), _p(kv[:, H * hs:]), _p(out), _p(d_out)
    p.dq, p.dk, p.dv = _p(dq), _p(dkv), _p(dkv[:, H * hs:])
    p.lse, p.delta, p.kv_len = _p(lse), _p(delta), _p(kv_len)
    p.sq, p.sk, p.sv = Lq * q.stride(0), Lk * kv.stride(0), Lk * kv.stride(0)
    p.so, p.sdo = Lq * out.stride(0), Lq * d_out.stride(0)
    p.sdq, p.sdk, p.sdv = Lq * dq.stride(0), Lk * dkv.stride(0), Lk * dkv.stride(0)
    p.B, p.H, p.Lq, p.Lk, p.hs = B, H, Lq, Lk, hs
    p.ld_q, p.ld_k, p.ld_v = q.stride(0), kv.stride(0), kv.stride(0)
    p.ld_o, p.ld_do = out.stride(0), d_out.stride(0)
    p.ld_dq, p.ld_dk, p.ld_dv = dq.stride(0), dkv.stride(0), dkv.stride(0)
    p.causal, p.alpha = 1, alpha
    if rope is not None:
        assert rope.dtype == torch.float32 and rope.shape[1] == hs and rope.shape[0] >= Lk and rope.is_contiguous()
        p.rope = _p(rope)
    _lib.check(_lib.lib().grove_flash_attn_bwd(C.byref(p), _stream()), "grove_flash_attn_bwd")


def rel_bias_applicable(nh, hp, rel_ld):
    """Shapes the matrix-core rel-pos streams take (grove_rel_bias_*); others go through the batched GEMM."""
    if os.environ.get("GROVE_REL_BIAS_STREAMS", "1") == "0":  # A/B arm of whole-program runs: the batched GEMMs
        return False
    return nh <= 16 and hp % 32 == 0 and hp <= 128 and rel_ld in (32, 64)


def rel_bias_fwd(q, rcat, nb, nh, L, hp, hd, *, out=None, q_valid=None, kw=0):
    """rel'[(b h), q, :] = q[(b q), h, :] . rcat[q]^T: q bf16 [nb*L, ld] with head h at column h*hp, rcat bf16 [L, rel_ld, hp]."""
    _chk_dev(q, rcat)
    rel_ld = rcat.shape[1]
    assert rcat.shape == (L, rel_ld, hp) and rcat.is_contiguous() and q.stride(1) == 1
    if out is None:
        out = torch.empty((nb * nh, L, rel_ld), dtype=bf16, device=q.device)
    p = _lib.RelBiasParams()
    p.q, p.table, p.rel, p.dq = _p(q), _p(rcat), _p(out), None
    p.nb, p.nh, p.L, p.hp, p.hd, p.rel_ld, p.ld_q, p.ld_dq = nb, nh, L, hp, hd, rel_ld, q.stride(0), 0
    p.q_valid, p.kw = _p(q_valid), kw
    _lib.check(_lib.lib().grove_rel_bias_fwd(C.byref(p), _stream()), "grove_rel_bias_fwd")
    return out


def rel_bias_bwd(drel, rcat_t, dq, nb, nh, L, hp, hd, *, q_valid=None, kw=0, dq_map=None):
    """dq[(b q), h, :] += d rel'[(b h), q, :] . rcat[q] in place: rcat_t bf16 [L, hp, rel_ld], dq bf16 [nb*L, ld] — or, with dq_map
    (int32 [nb * L]: (window, position) -> token row), dq in token order with compact heads of hd columns."""
    _chk_dev(drel, rcat_t, dq)
    rel_ld = rcat_t.shape[2]
    assert rcat_t.shape == (L, hp, rel_ld) and rcat_t.is_contiguous() and drel.is_contiguous() and dq.stride(1) == 1
    assert drel.shape == (nb * nh, L, rel_ld)
    p = _lib.RelBiasParams()
    p.q, p.table, p.rel, p.dq = None, _p(rcat_t), _p(drel), _p(dq)
    p.nb, p.nh, p.L, p.hp, p.hd, p.rel_ld, p.ld_q, p.ld_dq = nb, nh, L, hp, hd, rel_ld, 0, dq.stride(0)
    p.q_valid, p.kw = _p(q_valid), kw
    if dq_map is not None:
        assert q_valid is not None and dq_map.numel() == nb * L and dq_map.dtype == torch.int32
        p.dq_map, p.dq_hs = _p(dq_map), hd
    _lib.check(_lib.lib().grove_rel_bias_bwd(C.byref(p), _stream()), "grove_rel_bias_bwd")
    return dq


def relpos(q, Rh, Rw, batch, heads, qhw, khw, hd, hd_stride, ld_q, *, rel=None, dq=None, backward=False):
    p = _lib.RelposParams()
    L = qhw[0] * qhw[1]
    if rel is None:
        rel = torch.empty((batch * heads, L, khw[0] + khw[1]), dtype=torch.float32, device=q.device)
    p.q, p.Rh, p.Rw, p.rel, p.dq = _p(q), _p(Rh), _p(Rw), _p(rel), _p(dq)
    p.batch, p.heads, p.qh, p.qw, p.kh, p.kw = batch, heads, qhw[0], qhw[1], khw[0], khw[1]
    p.hd, p.hd_stride, p.ld_q = hd, hd_stride, ld_q
    fn = _lib.lib().grove_relpos_bwd if backward else _lib.lib().grove_relpos_fwd
    _lib.check(fn(C.byref(p), _stream()), "grove_relpos")
    return rel


def rope_(x, pos, col0, nheads, hd, theta, inverse=False, table=None):
    """table (ops.rope_table, f32 [positions, hd]): read cos | sin instead of evaluating them per thread."""
    p = _lib.RopeParams()
    p.x, p.pos = _p(x), _p(pos)
    if table is not None:
        assert table.dtype == torch.float32 and table.shape[1] == hd and table.is_contiguous()
        p.table = _p(table)
    p.rows, p.ld, p.col0, p.nheads, p.hd = x.shape[0], x.stride(0), col0, nheads, hd
    p.inverse, p.theta = int(inverse), theta
    _lib.check(_lib.lib().grove_rope_inplace(C.byref(p), _stream()), "grove_rope_inplace")
    return x


def swiglu_interleave(wgu):
    """[gate; up] rows ([2I, K]) -> rows interleaved [4 gate, 4 up] per 8, the B operand of the ACT_SWIGLU_PAIR GEMM."""
    I = wgu.shape[0] // 2
    assert I % 4 == 0
    g, u = wgu[:I].view(I // 4, 4, -1), wgu[I:].view(I // 4, 4, -1)
    return torch.cat([g, u], 1).reshape(2 * I, -1).contiguous()


def swiglu(gu, I):
    rows = gu.shape[0]
    y = torch.empty((rows, I), dtype=bf16, device=gu.device)
    _lib.check(_lib.lib().grove_swiglu_fwd(_p(gu), _p(y), rows, I, _stream()), "grove_swiglu_fwd")
    return y


def swiglu_bwd(gu, dy, I):
    rows = gu.shape[0]
    dgu = torch.empty_like(gu)
    _lib.check(_lib.lib().grove_swiglu_bwd(_p(gu), _p(dy), _p(dgu), rows, I, _stream()), "grove_swiglu_bwd")
    return dgu


def act_bwd(pre, dy, act, out=None):
    if out is None:
        out = torch.empty_like(dy)
    _lib.check(_lib.lib().grove_act_bwd(_p(pre), _p(dy), _p(out), C.c_int64(dy.numel()), act, _stream()), "grove_act_bwd")
    return out


def act_fwd(x, act, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().grove_act_fwd(_p(x), _p(out), C.c_int64(x.numel()), act, _stream()), "grove_act_fwd")
    return out


def resize_bilinear(src, H, W, crop=None):
    """fp32 [..., h, w] -> [..., H, W], bilinear with align_corners=False; crop=(h_use, w_use) samples only that corner of the source."""
    assert src.dtype == torch.float32 and src.is_contiguous()
    h, w = src.shape[-2], src.shape[-1]
    hu, wu = crop if crop is not None else (h, w)
    planes = src.numel() // (h * w)
    out = torch.empty(src.shape[:-2] + (H, W), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().grove_resize_bilinear_f32(_p(src), _p(out), planes, h, w, hu, wu, H, W, _stream()), "grove_resize_bilinear_f32")
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.lib().grove_add_bf16(_p(a), _p(b), _p(out), C.c_int64(a.numel()), _stream()), "grove_add_bf16")
    return out


def add_bcast_rows(a, b, period, out=None):
    rows, Cc = a.shape
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.lib().grove_add_bcast_rows(_p(a), _p(b), _p(out), rows, Cc, period, _stream()), "grove_add_bcast_rows")
    return out


def copy_rows(src, dst, rows, Cc, *, idx_src=None, idx_dst=None, accumulate=False):
    p = _lib.RowsParams()
    p.src, p.dst, p.idx_src, p.idx_dst = _p(src), _p(dst), _p(idx_src), _p(idx_dst)
    p.rows, p.C, p.ld_src, p.ld_dst, p.accumulate = rows, Cc, src.stride(-2), dst.stride(-2), int(accumulate)
    _lib.check(_lib.lib().grove_copy_rows(C.byref(p), _stream()), "grove_copy_rows")
    return dst


def scatter_add_f32(src, dst, idx, rows, Cc):
    _lib.check(_lib.lib().grove_scatter_add_f32(_p(src), _p(dst), _p(idx), rows, Cc, src.stride(0), dst.stride(0), _stream()),
               "grove_scatter_add_f32")
    return dst


def segment_sum_rows(src, dst, seg_ptr, rows_per_seg):
    """dst (fp32 [nseg * rows_per_seg, C])[s * rows_per_seg + r] += sum of src (bf16 [members * rows_per_seg, C])[i * rows_per_seg + r] over the
    members seg_ptr[s] <= i < seg_ptr[s + 1] (int32 [nseg + 1]) — grove_segment_sum_rows: the atomics-free form of a scatter-add over groups."""
    _chk_dev(src, dst, seg_ptr)
    nseg = seg_ptr.numel() - 1
    assert src.dtype == bf16 and dst.dtype == torch.float32 and seg_ptr.dtype == torch.int32 and dst.shape[0] == nseg * rows_per_seg
    assert src.shape[0] % rows_per_seg == 0 and src.shape[1] == dst.shape[1]
    _lib.check(_lib.lib().grove_segment_sum_rows(_p(src), _p(dst), _p(seg_ptr), nseg, rows_per_seg, src.shape[1], src.stride(0), dst.stride(0), _stream()),
               "grove_segment_sum_rows")
    return dst


def scatter_add_rows_f32(src, dst, idx):
    """dst (fp32 [*, C])[idx[r]] += src (fp32 [rows, C])[r]; idx -1 skips the row."""
    rows, Cc = src.shape
    _lib.check(_lib.lib().grove_scatter_add_rows_f32(_p(src), _p(dst), _p(idx), rows, Cc, src.stride(0), dst.stride(0), _stream()),
               "grove_scatter_add_rows_f32")
    return dst


def colsum(x, out=None, accumulate=False):
    rows, Cc = x.shape
    if out is None:
        out = torch.empty(Cc, dtype=torch.float32, device=x.device)
        accumulate = False
    _lib.check(_lib.lib().grove_colsum_f32(_p(x), _p(out), rows, Cc, x.stride(0), int(accumulate), _stream()), "grove_colsum_f32")
    return out


def dot(a, b, out, scale_ptr=None, mode=0):
    _lib.check(_lib.lib().grove_dot_bf16(_p(a), _p(b), _p(out), C.c_int64(a.numel()), _p(scale_ptr), mode, _stream()), "grove_dot_bf16")
    return out


def axpy(y, x, scale_ptr=None, mode=0):
    _lib.check(_lib.lib().grove_axpy_f32(_p(y), _p(x), C.c_int64(y.numel()), _p(scale_ptr), mode, _stream()), "grove_axpy_f32")
    return y


def to_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=bf16, device=x.device)
    _lib.check(_lib.lib().grove_cast_f32_to_bf16(_p(x), _p(out), C.c_int64(x.numel()), _stream()), "grove_cast_f32_to_bf16")
    return out


def to_f32(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().grove_cast_bf16_to_f32(_p(x), _p(out), C.c_int64(x.numel()), _stream()), "grove_cast_bf16_to_f32")
    return out


def im2col_patch(img, P, ld_col):
    B, Cc, T, H, W = img.shape
    assert img.is_contiguous()
    rows = B * T * (H // P) * (W // P)
    col = torch.empty((rows, ld_col), dtype=bf16, device=img.device)
    _lib.check(_lib.lib().grove_im2col_patch(_p(img), _p(col), B, Cc, T, H, W, P, ld_col, _stream()), "grove_im2col_patch")
    return col


def clip_pool(x, G):
    Cc = x.shape[-1]
    y = torch.empty((G, 576, Cc), dtype=bf16, device=x.device)
    _lib.check(_lib.lib().grove_clip_pool(_p(x), _p(y), G, Cc, _stream()), "grove_clip_pool")
    return y


def cross_entropy(logits, labels, V, *, loss_sum=None, dlogits=None, grad_scale=None):
    R, ld = logits.shape[0], logits.stride(0)
    if loss_sum is None:
        loss_sum = torch.zeros(1, dtype=torch.float32, device=logits.device)
    _lib.check(_lib.lib().grove_cross_entropy(_p(logits), _p(labels), _p(loss_sum), _p(dlogits), _p(grad_scale), R, V, ld, _stream()),
               "grove_cross_entropy")
    return loss_sum


def small_attn(q, k, v, inst, heads, d, Lq, Lk, *, out=None, out_dtype=bf16):
    """q / k,v / out may each be bf16 or fp32 (k and v together): the decoder's fp32 token path mixes them."""
    if out is None:
        out = torch.empty((inst * Lq, heads * d), dtype=out_dtype, device=q.device)
    assert k.dtype == v.dtype
    p = _lib.SmallAttnParams()
    p.q, p.k, p.v, p.o = _p(q), _p(k), _p(v), _p(out)
    p.inst, p.heads, p.d, p.Lq, p.Lk = inst, heads, d, Lq, Lk
    p.ld_q, p.ld_k, p.ld_v, p.ld_o = q.stride(-2), k.stride(-2), v.stride(-2), out.stride(-2)
    p.q_f32, p.kv_f32, p.o_f32 = int(q.dtype == torch.float32), int(k.dtype == torch.float32), int(out.dtype == torch.float32)
    _lib.check(_lib.lib().grove_small_attn_fwd(C.byref(p), _stream()), "grove_small_attn_fwd")
    return out


def small_attn_bwd(q, k, v, o, d_o, inst, heads, d, Lq, Lk, bf16_grads=False):
    """Returns (dq, dk, dv): fp32 — or, with bf16_grads, bf16 where the kernel that runs stores every gradient element once
    (grove_small_attn_bwd_stores_bf16; the others return fp32 and the cast is made here)."""
    dev = q.device
    p = _lib.SmallAttnParams()
    p.q, p.k, p.v, p.o, p.d_o = _p(q), _p(k), _p(v), _p(o), _p(d_o)
    p.inst, p.heads, p.d, p.Lq, p.Lk = inst, heads, d, Lq, Lk
    p.ld_q, p.ld_k, p.ld_v, p.ld_o = q.stride(-2), k.stride(-2), v.stride(-2), o.stride(-2)
    assert d_o.stride(-2) == o.stride(-2)
    direct = bool(bf16_grads and _lib.lib().grove_small_attn_bwd_stores_bf16(C.byref(p)))
    gdt = bf16 if direct else torch.float32
    dq = torch.empty((inst * Lq, heads * d), dtype=gdt, device=dev)
    dk = torch.empty((inst * Lk, heads * d), dtype=gdt, device=dev)
    dv = torch.empty((inst * Lk, heads * d), dtype=gdt, device=dev)
    p.dq, p.dk, p.dv = _p(dq), _p(dk), _p(dv)
    p.grad_bf16 = int(direct)
    _lib.check(_lib.lib().grove_small_attn_bwd(C.byref(p), _stream()), "grove_small_attn_bwd")
    if bf16_grads and not direct:
        return to_bf16(dq), to_bf16(dk), to_bf16(dv)
    return dq, dk, dv


def linear_f32(x, w, bias=None, *, act=ACT_NONE, residual=None, out=None, out_bf16=None):
    """y (fp32) = act(x (fp32) @ w.T (bf16) + bias) + residual (fp32) with exact fp32 products (grove_gemm_f32)."""
    _chk_dev(x, w)
    assert x.dtype == torch.float32 and w.dtype == bf16 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    p = _lib.GemmF32Params()
    p.A, p.W, p.bias, p.residual, p.C, p.C_bf16 = _p(x), _p(w), _p(bias), _p(residual), _p(out), _p(out_bf16)
    p.M, p.N, p.K, p.lda, p.ldw, p.ldc = M, N, K, x.stride(0), w.stride(0), out.stride(0)
    p.ldr = residual.stride(0) if residual is not None else 0
    p.act = act
    assert out_bf16 is None or out_bf16.stride(0) == out.stride(0)
    _lib.check(_lib.lib().grove_gemm_f32(C.byref(p), _stream()), "grove_gemm_f32")
    return out


def quant_fp8_rows(x, act=ACT_NONE):
    """bf16 [rows, K] -> (e4m3 codes uint8 [rows, K], fp32 scale [rows]): per-row amax / 448 scaling (grove_quant_fp8_rows) of act(x)."""
    _chk_dev(x)
    rows, K = x.shape
    assert x.dtype == bf16 and x.stride(1) == 1
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    if act != ACT_NONE:
        _lib.check(_lib.lib().grove_quant_fp8_rows_act(_p(x), _p(q), _p(sc), rows, K, x.stride(0), q.stride(0), act, _stream()), "grove_quant_fp8_rows_act")
    else:
        _lib.check(_lib.lib().grove_quant_fp8_rows(_p(x), _p(q), _p(sc), rows, K, x.stride(0), q.stride(0), _stream()), "grove_quant_fp8_rows")
    return q, sc


def linear_fp8(x, wq, w_scale, bias=None, *, act=ACT_NONE, residual=None, out=None, xq=None):
    """y (bf16) = act(x @ w.T + bias) + residual with both operands in fp8: wq / w_scale = quant_fp8_rows(w) (made once), the bf16
    activation x is quantised per row on the fly (or pass xq = (codes, scales) to reuse a quantisation)."""
    if xq is None:
        xq = quant_fp8_rows(x)
    aq, a_scale = xq
    M, K = aq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and wq.dtype == torch.uint8
    if out is None:
        out = torch.empty((M, N), dtype=bf16, device=aq.device)
    p = _lib.GemmFp8Params()
    p.A, p.B, p.C, p.scale_a, p.scale_b, p.bias, p.residual = _p(aq), _p(wq), _p(out), _p(a_scale), _p(w_scale), _p(bias), _p(residual)
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, aq.stride(0), wq.stride(0), out.stride(0)
    p.ldr = residual.stride(0) if residual is not None else 0
    p.act = act
    lib = _lib.lib()
    st = _stream()
    plan = _lib.GemmPlan()
    _lib.check(lib.grove_gemm_fp8_make_plan(C.byref(p), C.byref(plan)), "grove_gemm_fp8_make_plan")
    w = gemm_workspace(plan, lambda buf, n: _lib.check(lib.grove_gemm_fp8_plan_image(C.byref(p), buf, n), "grove_gemm_fp8_plan_image"), out.device)
    _lib.check(lib.grove_gemm_fp8(C.byref(p), C.byref(w) if w is not None else None, st), "grove_gemm_fp8")
    return out


def add_f32(a, b):
    """a + b for fp32 tensors of one shape (new tensor): cast-free copy + grove_axpy_f32."""
    out = a.clone()
    return axpy(out.view(-1), b.reshape(-1)).view(a.shape)


def box_head(x, W1, b1, W2, b2, Wo, bo):
    N, D = x.shape
    dev = x.device
    hidden = torch.empty((N, D), dtype=torch.float32, device=dev)
    box = torch.empty((N, 4), dtype=torch.float32, device=dev)
    obj = torch.empty(N, dtype=torch.float32, device=dev) if Wo is not None else None
    p = _lib.BoxHeadParams()
    p.x, p.W1, p.b1, p.W2, p.b2, p.Wo, p.bo = _p(x), _p(W1), _p(b1), _p(W2), _p(b2), _p(Wo), _p(bo)
    p.hidden, p.box, p.obj, p.N, p.D = _p(hidden), _p(box), _p(obj), N, D
    _lib.check(_lib.lib().grove_box_head_fwd(C.byref(p), _stream()), "grove_box_head_fwd")
    return box, obj, hidden


def box_head_bwd(x, W1, W2, Wo, hidden, box, dbox, dobj, grads):
    """grads: dict of f32 accumulators dW1, db1, dW2, db2, dWo, dbo. Returns dx f32 [N, D]."""
    N, D = x.shape
    dx = torch.empty((N, D), dtype=torch.float32, device=x.device)
    p = _lib.BoxHeadBwdParams()
    p.x, p.W1, p.W2, p.Wo, p.hidden, p.box = _p(x), _p(W1), _p(W2), _p(Wo), _p(hidden), _p(box)
    p.dbox, p.dobj, p.dx = _p(dbox), _p(dobj), _p(dx)
    p.dW1, p.db1, p.dW2, p.db2 = _p(grads["dW1"]), _p(grads["db1"]), _p(grads["dW2"]), _p(grads["db2"])
    p.dWo, p.dbo = _p(grads.get("dWo")), _p(grads.get("dbo"))
    p.N, p.D = N, D
    _lib.check(_lib.lib().grove_box_head_bwd(C.byref(p), _stream()), "grove_box_head_bwd")
    return dx


def box_losses(pred_box, obj_logit, gt_box, visible, w_box_over_ngt, w_obj_over_n, want_grad=True):
    N = pred_box.shape[0]
    dev = pred_box.device
    sums = torch.zeros(3, dtype=torch.float32, device=dev)
    dbox = torch.empty((N, 4), dtype=torch.float32, device=dev) if want_grad else None
    dobj = torch.empty(N, dtype=torch.float32, device=dev) if (want_grad and obj_logit is not None) else None
    _lib.check(_lib.lib().grove_box_losses(_p(pred_box), _p(obj_logit), _p(gt_box), _p(visible), _p(sums), _p(dbox), _p(dobj), N,
                                           C.c_float(w_box_over_ngt), C.c_float(w_obj_over_n), _stream()), "grove_box_losses")
    return sums, dbox, dobj


def adamw_step(master, model_bf16, grad, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    _lib.check(_lib.lib().grove_adamw_step(_p(master), _p(model_bf16), _p(grad), _p(m), _p(v), C.c_int64(master.numel()),
                                           C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                                           C.c_float(weight_decay), C.c_float(grad_scale), int(step), _stream()), "grove_adamw_step")


def adamw_step_multi(master, grad, m, v, seg_off, seg_len, model_ptrs, lr, beta1, beta2, eps, weight_decay, grad_scale, step,
                     sumsq=None, clip=0.0, norm_out=None):
    """One launch over every trainable tensor: seg_off / seg_len int64 and model_ptrs int64 (data_ptr of each bf16 tensor) device tensors.
    sumsq (device fp32 scalar = sum of squares of grad): clip the global norm of grad * grad_scale at `clip` inside the kernel."""
    _lib.check(_lib.lib().grove_adamw_step_multi(_p(master), _p(grad), _p(m), _p(v), _p(seg_off), _p(seg_len), _p(model_ptrs),
                                                 int(seg_off.numel()), C.c_int64(master.numel()), C.c_float(lr), C.c_float(beta1),
                                                 C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), C.c_float(grad_scale),
                                                 int(step), _p(sumsq), C.c_float(clip), _p(norm_out), _stream()), "grove_adamw_step_multi")


def sumsq(x, out=None):
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().grove_sumsq_f32(_p(x), _p(out), C.c_int64(x.numel()), _stream()), "grove_sumsq_f32")
    return out
