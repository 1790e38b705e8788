"""Device preprocessing (csrc/preprocess.hip through the C-ABI) against the CPU oracle that runs Pillow itself."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def frames(F, H, W, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    x[:, : H // 3, : W // 3] = 255
    x[:, H // 2:, W // 2:] = 0
    return x


@pytest.mark.parametrize("H,W", [(360, 640), (480, 640), (640, 360), (336, 336)])
def test_resize_is_pillow_bit_exact(dev, H, W):
    from grove_amd.preprocess import BICUBIC, BILINEAR, clip_resize_shape, resize_u8, sam_resize_shape
    x = frames(3, H, W, 5)
    xd = torch.from_numpy(x).to(dev)
    for resample, (h, w) in ((BILINEAR, sam_resize_shape(H, W)), (BICUBIC, clip_resize_shape(H, W))):
        got = resize_u8(xd, (h, w), resample).cpu().numpy()
        for f in range(x.shape[0]):
            ref = np.array(Image.fromarray(x[f]).resize((w, h), resample=resample, reducing_gap=None))
            assert np.array_equal(got[f], ref), f"frame {f}: max diff {np.abs(got[f].astype(int) - ref).max()}"


def test_encoder_inputs_match_oracle(dev):
    from grove_amd.preprocess import preprocess_clip, preprocess_sam
    from oracle import preprocess_oracle as O
    x = frames(8, 360, 640, 9)
    xd = torch.from_numpy(x).to(dev)
    for fn, ref in ((preprocess_clip, O.clip_preprocess(x)), (preprocess_sam, O.sam_preprocess(x))):
        got32 = fn(xd, dtype=torch.float32).cpu().numpy()
        assert got32.shape == ref.shape
        assert np.abs(got32 - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), np.abs(got32 - ref).max()
        got16 = fn(xd).float().cpu()
        ref16 = torch.from_numpy(ref).to(torch.bfloat16).float()
        # same value rounded to bf16: equal except where an fp32 last-bit difference straddles a rounding boundary
        assert (got16 - ref16).abs().max().item() <= 2 ** -7 * max(1.0, ref16.abs().max().item())
        assert (got16 != ref16).float().mean().item() < 1e-3
    # SAM letter-box: rows below the resized image are exactly zero (padding is applied after normalisation)
    s = preprocess_sam(xd, dtype=torch.float32)
    assert s.shape == (3, 8, 512, 512) and float(s[:, :, 288:].abs().max()) == 0.0
