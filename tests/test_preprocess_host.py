"""Host side of the device preprocessing: the coefficient tables rebuilt from Pillow's algorithm, applied by a numpy
emulation of the two 8-bit passes, reproduce PIL's resize bit for bit (so the HIP kernels only have to apply tables)."""
import numpy as np
import pytest
from PIL import Image

from grove_amd.preprocess import BICUBIC, BILINEAR, clip_resize_shape, pil_coeffs, sam_resize_shape


def resample_np(img, out_hw, resample):
    h, w = out_hw
    x = img.astype(np.int64)
    for axis, n_out in ((1, w), (0, h)):
        if x.shape[axis] == n_out:
            continue
        kk, b = pil_coeffs(x.shape[axis], n_out, resample)
        x = np.moveaxis(x, axis, 0)
        y = np.empty((n_out,) + x.shape[1:], dtype=np.int64)
        for o in range(n_out):
            lo, n = b[o]
            acc = (1 << 21) + np.tensordot(kk[o, :n].astype(np.int64), x[lo:lo + n], axes=(0, 0))
            y[o] = np.clip(acc >> 22, 0, 255)
        x = np.moveaxis(y, 0, axis)
    return x.astype(np.uint8)


@pytest.mark.parametrize("H,W", [(360, 640), (480, 640), (240, 426), (720, 1280), (336, 336), (500, 375)])
def test_tables_reproduce_pillow(H, W):
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[: H // 3, : W // 3] = 255  # saturated block: exercises the clamp between the passes (bicubic overshoot)
    img[H // 2:, W // 2:] = 0
    for resample, shape in ((BILINEAR, sam_resize_shape(H, W)), (BICUBIC, clip_resize_shape(H, W))):
        h, w = shape
        ref = np.array(Image.fromarray(img).resize((w, h), resample=resample, reducing_gap=None))
        got = resample_np(img, (h, w), resample)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"{H}x{W} -> {h}x{w} resample {resample}: {np.abs(got.astype(int) - ref).max()}"


def test_shapes_follow_the_reference():
    assert sam_resize_shape(360, 640) == (288, 512)        # transforms.py:102-113
    assert clip_resize_shape(360, 640) == (336, 597)       # shortest edge 336, int(336 * 640 / 360)
    assert clip_resize_shape(640, 360) == (597, 336)
