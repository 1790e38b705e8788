"""CPU oracle of the frame preprocessing (test infrastructure only — never imported by the product path).

Runs the very libraries the reference runs: Pillow's `Image.resize` (what torchvision.transforms.functional.resize of a PIL
image and transformers' image_transforms.resize both call) plus the numpy/torch arithmetic of
  - CLIPImageProcessor.preprocess (transformers 4.46.3 image_processing_clip.py: resize shortest edge 336 bicubic ->
    center_crop 336 -> rescale 1/255 -> normalize; called at HowTo100M.py:309), restated here because the installed
    transformers 5.x defaults to a torchvision-tensor "fast" processor with different resampling;
  - ResizeLongestSide.apply_image (model/SAM/utils/transforms.py:27-34, 102-113) and grounding_enc_processor
    (HowTo100M.py:168-178).
"""
import numpy as np
from PIL import Image

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073])
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711])
SAM_MEAN = np.array([123.675, 116.28, 103.53], dtype=np.float32)
SAM_STD = np.array([58.395, 57.12, 57.375], dtype=np.float32)


def pil_resize(frame, h, w, resample):
    return np.array(Image.fromarray(frame).resize((w, h), resample=resample, reducing_gap=None))


def clip_preprocess(frames, size=336):
    """uint8 [F, H, W, 3] -> float32 [3, F, size, size]."""
    out = []
    for f in frames:
        H, W = f.shape[:2]
        short, long_ = (W, H) if W <= H else (H, W)
        new_short, new_long = size, int(size * long_ / short)
        h, w = (new_long, new_short) if W <= H else (new_short, new_long)
        r = pil_resize(f, h, w, Image.BICUBIC)
        top, left = (h - size) // 2, (w - size) // 2
        r = r[top:top + size, left:left + size]
        x = (r * (1 / 255)).astype(np.float32)                       # rescale(): image * scale in float64, then float32
        x = (x - CLIP_MEAN.astype(np.float32)) / CLIP_STD.astype(np.float32)  # normalize() in the image dtype
        out.append(x.transpose(2, 0, 1))
    return np.stack(out, 1)


def sam_preprocess(frames, size=512):
    """uint8 [F, H, W, 3] -> float32 [3, F, size, size]."""
    out = []
    for f in frames:
        H, W = f.shape[:2]
        scale = size * 1.0 / max(H, W)
        h, w = int(H * scale + 0.5), int(W * scale + 0.5)
        r = pil_resize(f, h, w, Image.BILINEAR).astype(np.float32)
        x = (r - SAM_MEAN) / SAM_STD
        pad = np.zeros((size, size, 3), dtype=np.float32)
        pad[:h, :w] = x
        out.append(pad.transpose(2, 0, 1))
    return np.stack(out, 1)
