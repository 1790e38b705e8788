"""Frame preprocessing on the device: uint8 RGB frames [F, H, W, 3] -> the two encoder inputs, replacing
the per-clip host work of HowTo100M.py:309-313:

  global_enc_images    = CLIPImageProcessor.preprocess(frames)      resize shortest edge -> 336 (bicubic), centre crop 336,
                                                                    x / 255, (x - mean) / std          -> [3, F, 336, 336]
  grounding_enc_images = grounding_enc_processor(apply_image(f))    resize longest side -> 512 (bilinear, transforms.py:27-34),
                                                                    (x - mean) / std in pixel units, zero pad right / bottom
                                                                                                       -> [3, F, 512, 512]

Both resizes are Pillow's 8-bit resampler (through torchvision / transformers): two separable passes with 22-bit fixed-point
coefficients and a uint8 round-and-clamp between them. `pil_coeffs` rebuilds Pillow's coefficient / bound tables exactly
(ImagingResample: precompute_coeffs + normalize_coeffs_8bpc); the kernels of csrc/preprocess.hip apply them, so the device
result equals PIL's bit for bit (tests/test_preprocess*.py).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

BILINEAR, BICUBIC = 2, 3  # PIL.Image.Resampling values
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
SAM_MEAN = (123.675, 116.28, 103.53)  # HowTo100M.py:24-25 (pixel units)
SAM_STD = (58.395, 57.12, 57.375)
PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def pil_coeffs(in_size, out_size, resample):
    """Pillow's precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc -> (kk int32 [out, ksize], bounds int32 [out, 2])."""
    filt, fsupport = _FILTERS[resample]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)  # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


_tables = {}


def _dev_tables(in_size, out_size, resample, device):
    key = (in_size, out_size, resample, str(device))
    if key not in _tables:
        kk, b = pil_coeffs(in_size, out_size, resample)
        _tables[key] = (torch.from_numpy(kk).to(device), torch.from_numpy(b).to(device), kk.shape[1])
    return _tables[key]


def resize_u8(frames, out_hw, resample):
    """PIL `Image.resize((w, h), resample, reducing_gap=None)` of every frame: uint8 [F, H, W, 3] -> uint8 [F, h, w, 3]."""
    assert frames.dtype == torch.uint8 and frames.is_cuda and frames.dim() == 4 and frames.shape[-1] == 3
    frames = frames.contiguous()
    F, H, W, _ = frames.shape
    h, w = out_hw
    x = frames
    if w != W:  # horizontal pass first, as ImagingResample does
        kk, b, ks = _dev_tables(W, w, resample, frames.device)
        y = torch.empty((F, H, w, 3), dtype=torch.uint8, device=frames.device)
        p = _lib.ResampleParams()
        p.src, p.dst, p.kk, p.bounds = _p(x), _p(y), _p(kk), _p(b)
        p.F, p.Hin, p.Win, p.Hout, p.Wout, p.ksize, p.axis = F, H, W, H, w, ks, 0
        _lib.check(_lib.lib().grove_resample_u8(C.byref(p), _stream()), "grove_resample_u8")
        x = y
    if h != H:
        kk, b, ks = _dev_tables(H, h, resample, frames.device)
        y = torch.empty((F, h, x.shape[2], 3), dtype=torch.uint8, device=frames.device)
        p = _lib.ResampleParams()
        p.src, p.dst, p.kk, p.bounds = _p(x), _p(y), _p(kk), _p(b)
        p.F, p.Hin, p.Win, p.Hout, p.Wout, p.ksize, p.axis = F, H, x.shape[2], h, x.shape[2], ks, 1
        _lib.check(_lib.lib().grove_resample_u8(C.byref(p), _stream()), "grove_resample_u8")
        x = y
    return x


def normalize_pack(frames, out_hw, top, left, rescale, mean, std, dtype=torch.bfloat16):
    """uint8 [F, H, W, 3] -> dtype [3, F, Ho, Wo], (x * rescale - mean) / std inside the window, 0 outside."""
    F, H, W, _ = frames.shape
    out = torch.empty((3, F, out_hw[0], out_hw[1]), dtype=dtype, device=frames.device)
    p = _lib.NormalizeParams()
    p.src, p.dst = _p(frames.contiguous()), _p(out)
    p.F, p.H, p.W, p.Ho, p.Wo, p.top, p.left = F, H, W, out_hw[0], out_hw[1], top, left
    p.out_dtype = _lib.F32 if dtype == torch.float32 else _lib.BF16
    p.rescale = rescale
    for c in range(3):
        p.mean[c], p.std[c] = mean[c], std[c]
    _lib.check(_lib.lib().grove_normalize_pack(C.byref(p), _stream()), "grove_normalize_pack")
    return out


def clip_resize_shape(H, W, size=336):
    """transformers get_resize_output_image_size(shortest_edge=size, default_to_square=False)."""
    short, long_ = (W, H) if W <= H else (H, W)
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if W <= H else (new_short, new_long)


def sam_resize_shape(H, W, long_side=512):
    """ResizeLongestSide.get_preprocess_shape (transforms.py:102-113)."""
    scale = long_side * 1.0 / max(H, W)
    return int(H * scale + 0.5), int(W * scale + 0.5)


def preprocess_clip(frames, size=336, dtype=torch.bfloat16):
    """CLIPImageProcessor.preprocess of a clip's frames -> [3, F, size, size] (HowTo100M.py:309-310)."""
    F, H, W, _ = frames.shape
    h, w = clip_resize_shape(H, W, size)
    x = resize_u8(frames, (h, w), BICUBIC)
    return normalize_pack(x, (size, size), (h - size) // 2, (w - size) // 2, 1.0 / 255.0, CLIP_MEAN, CLIP_STD, dtype)


def preprocess_sam(frames, size=512, dtype=torch.bfloat16):
    """ResizeLongestSide(size).apply_image + grounding_enc_processor -> [3, F, size, size] (HowTo100M.py:312-313, 168-178)."""
    F, H, W, _ = frames.shape
    x = resize_u8(frames, sam_resize_shape(H, W, size), BILINEAR)
    return normalize_pack(x, (size, size), 0, 0, 1.0, SAM_MEAN, SAM_STD, dtype)
