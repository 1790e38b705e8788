"""Deterministic synthetic weights and inputs (there are no checkpoints or datasets offline).

Every tensor is a pure function of (name, shape): a 32-bit integer hash of the element index,
mapped to a zero-mean uniform value of the requested standard deviation. The arithmetic stays
below 2^63, so CPU and GPU, numpy and torch all produce the same bits; golden fixtures therefore
only need to store OUTPUTS (tests/golden/), and bench.py can build LLaMA-7B sized weights directly
in HBM. Shapes follow SURVEY.md §8(d).
"""
import math
import zlib
from dataclasses import dataclass, field, replace
from typing import List, Tuple

import torch


def _hash32(idx, seed):
    h = (idx ^ seed) & 0xFFFFFFFF
    h = (h * 0x45D9F3B) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x45D9F3B) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x119DE1F3) & 0xFFFFFFFF
    h = h ^ (h >> 15)
    return h


def det_uniform01(name: str, shape, device="cpu"):
    """float64-free uniform [0,1) tensor (fp32) keyed by name; element i depends only on (name, i)."""
    n = 1
    for s in shape:
        n *= int(s)
    seed = zlib.crc32(name.encode()) & 0xFFFFFFFF
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = 1 << 24
    for s0 in range(0, n, chunk):
        s1 = min(n, s0 + chunk)
        idx = torch.arange(s0, s1, dtype=torch.int64, device=device)
        h = _hash32(idx, seed)
        out[s0:s1] = (h >> 8).to(torch.float32) * (1.0 / (1 << 24))
    return out.reshape(tuple(shape))


def det_tensor(name: str, shape, std=0.02, mean=0.0, device="cpu", dtype=torch.float32):
    u = det_uniform01(name, shape, device)
    return ((u - 0.5) * (2.0 * math.sqrt(3.0) * std) + mean).to(dtype)


# ----------------------------------------------------------------------------------------------
@dataclass
class GroveDims:
    """All architecture dimensions of the hot path (defaults = the real GROVE model)."""
    # LLaMA (LLaVA-1.5 / Vicuna-7B geometry)
    hidden: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    mlp: int = 11008
    vocab: int = 32008
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    # CLIP ViT-L/14-336 (modeling_clip.py)
    clip_dim: int = 1024
    clip_layers: int = 24
    clip_heads: int = 16
    clip_mlp: int = 4096
    clip_image: int = 336
    clip_patch: int = 14
    clip_eps: float = 1e-5
    # SAM ViT-H @512 (build_sam.py:15-23,57-107)
    sam_dim: int = 1280
    sam_depth: int = 32
    sam_heads: int = 16
    sam_global: Tuple[int, ...] = (7, 15, 23, 31)
    sam_window: int = 14
    sam_image: int = 512
    sam_patch: int = 16
    sam_out: int = 256
    # decoder (build_sam.py:84-100)
    dec_dim: int = 256
    dec_heads: int = 8
    dec_mlp: int = 2048
    dec_depth: int = 2
    out_dim: int = 256
    num_frames: int = 8
    # special token ids (synthetic tokenizer)
    det_token_idx: int = 32007
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 2

    @property
    def head_dim(self):
        return self.hidden // self.n_heads

    @property
    def clip_tokens(self):
        return (self.clip_image // self.clip_patch) ** 2 + 1

    @property
    def sam_grid(self):
        return self.sam_image // self.sam_patch


FULL = GroveDims()
TINY = GroveDims(hidden=128, n_layers=2, n_heads=4, mlp=256, vocab=320, clip_dim=64, clip_layers=6, clip_heads=2,
                 clip_mlp=128, sam_dim=64, sam_depth=4, sam_heads=4, sam_global=(1, 3), det_token_idx=319)


def param_shapes(d: GroveDims, with_region_encoder: bool = False):
    """Ordered {state-dict name: shape} of the hot-path parameters, with the reference's key names
    (SURVEY.md §8(b) "State-dict names"). The never-executed region encoder is omitted unless asked."""
    P = {}
    H = d.hidden
    P["model.embed_tokens.weight"] = (d.vocab, H)
    for i in range(d.n_layers):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            P[p + f"self_attn.{n}.weight"] = (H, H)
        P[p + "mlp.gate_proj.weight"] = (d.mlp, H)
        P[p + "mlp.up_proj.weight"] = (d.mlp, H)
        P[p + "mlp.down_proj.weight"] = (H, d.mlp)
        P[p + "input_layernorm.weight"] = (H,)
        P[p + "post_attention_layernorm.weight"] = (H,)
    P["model.norm.weight"] = (H,)
    P["lm_head.weight"] = (d.vocab, H)
    # CLIP vision tower
    v = "model.vision_tower.vision_tower.vision_model."
    Cd = d.clip_dim
    P[v + "embeddings.class_embedding"] = (Cd,)
    P[v + "embeddings.patch_embedding.weight"] = (Cd, 3, d.clip_patch, d.clip_patch)
    P[v + "embeddings.position_embedding.weight"] = (d.clip_tokens, Cd)
    P[v + "pre_layrnorm.weight"] = (Cd,)
    P[v + "pre_layrnorm.bias"] = (Cd,)
    for i in range(d.clip_layers):
        p = v + f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            P[p + f"self_attn.{n}.weight"] = (Cd, Cd)
            P[p + f"self_attn.{n}.bias"] = (Cd,)
        P[p + "layer_norm1.weight"] = (Cd,)
        P[p + "layer_norm1.bias"] = (Cd,)
        P[p + "mlp.fc1.weight"] = (d.clip_mlp, Cd)
        P[p + "mlp.fc1.bias"] = (d.clip_mlp,)
        P[p + "mlp.fc2.weight"] = (Cd, d.clip_mlp)
        P[p + "mlp.fc2.bias"] = (Cd,)
        P[p + "layer_norm2.weight"] = (Cd,)
        P[p + "layer_norm2.bias"] = (Cd,)
    for i in range(d.clip_layers // 3):
        p = v + f"encoder.adapters.{i}."
        P[p + "conv3d.weight"] = (Cd, Cd, 3, 3, 3)
        P[p + "conv3d.bias"] = (Cd,)
        P[p + "alpha"] = (1,)
    P[v + "post_layernorm.weight"] = (Cd,)
    P[v + "post_layernorm.bias"] = (Cd,)
    P["model.mm_projector.0.weight"] = (H, Cd)
    P["model.mm_projector.0.bias"] = (H,)
    P["model.mm_projector.2.weight"] = (H, H)
    P["model.mm_projector.2.bias"] = (H,)
    P["model.text_hidden_fcs.0.0.weight"] = (H, H)
    P["model.text_hidden_fcs.0.0.bias"] = (H,)
    P["model.text_hidden_fcs.0.2.weight"] = (d.out_dim, H)
    P["model.text_hidden_fcs.0.2.bias"] = (d.out_dim,)
    # SAM image encoder
    s = "model.grounding_encoder.image_encoder."
    Sd, g = d.sam_dim, d.sam_grid
    hd = Sd // d.sam_heads
    P[s + "pos_embed"] = (1, g, g, Sd)
    P[s + "patch_embed.proj.weight"] = (Sd, 3, d.sam_patch, d.sam_patch)
    P[s + "patch_embed.proj.bias"] = (Sd,)
    for i in range(d.sam_depth):
        p = s + f"blocks.{i}."
        size = g if i in d.sam_global else d.sam_window
        P[p + "norm1.weight"] = (Sd,)
        P[p + "norm1.bias"] = (Sd,)
        P[p + "attn.rel_pos_h"] = (2 * size - 1, hd)
        P[p + "attn.rel_pos_w"] = (2 * size - 1, hd)
        P[p + "attn.qkv.weight"] = (3 * Sd, Sd)
        P[p + "attn.qkv.bias"] = (3 * Sd,)
        P[p + "attn.proj.weight"] = (Sd, Sd)
        P[p + "attn.proj.bias"] = (Sd,)
        P[p + "norm2.weight"] = (Sd,)
        P[p + "norm2.bias"] = (Sd,)
        P[p + "mlp.lin1.weight"] = (4 * Sd, Sd)
        P[p + "mlp.lin1.bias"] = (4 * Sd,)
        P[p + "mlp.lin2.weight"] = (Sd, 4 * Sd)
        P[p + "mlp.lin2.bias"] = (Sd,)
    for i in range(len(d.sam_global)):
        p = s + f"adapters.{i}."
        P[p + "conv3d.weight"] = (Sd, Sd, 3, 3, 3)
        P[p + "conv3d.bias"] = (Sd,)
        P[p + "alpha"] = (1,)
    P[s + "neck.0.weight"] = (d.sam_out, Sd, 1, 1)
    P[s + "neck.1.weight"] = (d.sam_out,)
    P[s + "neck.1.bias"] = (d.sam_out,)
    P[s + "neck.2.weight"] = (d.sam_out, d.sam_out, 3, 3)
    P[s + "neck.3.weight"] = (d.sam_out,)
    P[s + "neck.3.bias"] = (d.sam_out,)
    # prompt encoder (only what the text-prompt path reads)
    pe = "model.grounding_encoder.prompt_encoder."
    P[pe + "pe_layer.positional_encoding_gaussian_matrix"] = (2, d.dec_dim // 2)
    P[pe + "no_mask_embed.weight"] = (1, d.dec_dim)
    # mask decoder, query branch
    m = "model.grounding_encoder.mask_decoder."
    D = d.dec_dim
    P[m + "iou_token.weight"] = (1, D)
    P[m + "mask_tokens.weight"] = (4, D)

    def attn(prefix, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            P[prefix + f"{n}.weight"] = (internal, D)
            P[prefix + f"{n}.bias"] = (internal,)
        P[prefix + "out_proj.weight"] = (D, internal)
        P[prefix + "out_proj.bias"] = (D,)
    for i in range(d.dec_depth):
        p = m + f"transformer.layers.{i}."
        attn(p + "self_attn.", D)
        attn(p + "cross_attn_token_to_image.", D // 2)
        attn(p + "cross_attn_image_to_token.", D // 2)
        for k in (1, 2, 3, 4):
            P[p + f"norm{k}.weight"] = (D,)
            P[p + f"norm{k}.bias"] = (D,)
        P[p + "mlp.lin1.weight"] = (d.dec_mlp, D)
        P[p + "mlp.lin1.bias"] = (d.dec_mlp,)
        P[p + "mlp.lin2.weight"] = (D, d.dec_mlp)
        P[p + "mlp.lin2.bias"] = (D,)
    attn(m + "transformer.final_attn_token_to_image.", D // 2)
    P[m + "transformer.norm_final_attn.weight"] = (D,)
    P[m + "transformer.norm_final_attn.bias"] = (D,)
    P[m + "bbox_prediction_head.0.weight"] = (D, D)
    P[m + "bbox_prediction_head.0.bias"] = (D,)
    P[m + "bbox_prediction_head.2.weight"] = (4, D)
    P[m + "bbox_prediction_head.2.bias"] = (4,)
    P[m + "temporal_objectness_head.weight"] = (1, D)
    P[m + "temporal_objectness_head.bias"] = (1,)
    # mask branch (mask_decoder.py:56-78, 206-227): dormant under GROVE's decoding_type "query", present in every checkpoint
    P[m + "output_upscaling.0.weight"] = (D, D // 4, 2, 2)
    P[m + "output_upscaling.0.bias"] = (D // 4,)
    P[m + "output_upscaling.1.weight"] = (D // 4,)
    P[m + "output_upscaling.1.bias"] = (D // 4,)
    P[m + "output_upscaling.3.weight"] = (D // 4, D // 8, 2, 2)
    P[m + "output_upscaling.3.bias"] = (D // 8,)
    for i in range(4):
        for j, (o, k) in enumerate(((D, D), (D, D), (D // 8, D))):
            P[m + f"output_hypernetworks_mlps.{i}.layers.{j}.weight"] = (o, k)
            P[m + f"output_hypernetworks_mlps.{i}.layers.{j}.bias"] = (o,)
    for j, (o, k) in enumerate(((256, D), (256, 256), (4, 256))):
        P[m + f"iou_prediction_head.layers.{j}.weight"] = (o, k)
        P[m + f"iou_prediction_head.layers.{j}.bias"] = (o,)
    return P


MASK_BRANCH = ("output_upscaling.", "output_hypernetworks_mlps.", "iou_prediction_head.")


def is_mask_branch(name: str) -> bool:
    """Parameters only the (dormant) mask branch of the SAM decoder reads: no gradient reaches them on GROVE's box path."""
    return any(k in name for k in MASK_BRANCH)


def init_spec(name: str, shape, d: GroveDims):
    """(mean, std) of the synthetic initialisation of one parameter (N(0, 0.02)-like,
    norm weights 1, CLIP adapter alpha 0, SAM adapter alpha 0.1). Projection weights are scaled like
    1/sqrt(fan_in) so that activations stay O(1) through 32 layers of random weights."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "alpha":
        return (0.0, 0.0) if "vision_tower" in name else (0.1, 0.0)
    if "norm" in name and leaf == "weight" and len(shape) == 1:
        return (1.0, 0.05)
    if "output_upscaling.1." in name:  # LayerNorm2d
        return ((1.0, 0.05) if leaf == "weight" else (0.0, 0.02))
    if "neck.1." in name or "neck.3." in name:
        return ((1.0, 0.05) if leaf == "weight" else (0.0, 0.02))
    if name.endswith("temporal_objectness_head.bias"):
        return (1.9, 0.0)  # centres the synthetic objectness logits on the 0.5 threshold
    if leaf == "bias":
        return (0.0, 0.02)
    if "positional_encoding_gaussian_matrix" in name:
        return (0.0, 1.0)
    if "rel_pos" in name:
        return (0.0, 0.02)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if "embed" in name or "token" in name or "pos_embed" in name:
            return (0.0, 0.02 if "embed_tokens" not in name else 0.5)
        return (0.0, 1.0 / math.sqrt(fan_in))
    return (0.0, 0.02)


def outlier_channels(d: GroveDims):
    """The hidden channels that carry the synthetic "massive activations" (three fixed ones, like LLaMA-2-7B's 1415 / 2533)."""
    H = d.hidden
    return (H // 8 + 3, H // 2 + 5, (3 * H) // 4 + 1)


def synthetic_param(name: str, shape, d: GroveDims, device="cpu", outliers: float = 0.0):
    """One synthetic parameter (fp32). outliers = F > 0: the rows / columns that WRITE three fixed hidden channels
    of the LLaMA residual stream are scaled by F — embed_tokens' columns (every text token starts with them), o_proj of layer 0 and
    down_proj of layers 0 and 1 (rows = output channels) — so that from the first layers on the stream carries activations F times
    the typical magnitude in those channels at every position, as real LLaMA checkpoints do (N(0, 0.02) weights have none, so a
    parity figure on them says nothing about outlier handling). Everything else is unchanged."""
    mean, std = init_spec(name, shape, d)
    t = det_tensor(name, shape, std=std, mean=mean, device=device, dtype=torch.float32)
    if outliers and outliers > 0:
        ch = list(outlier_channels(d))
        if name == "model.embed_tokens.weight":
            t[:, ch] *= outliers
        elif name in ("model.layers.0.self_attn.o_proj.weight", "model.layers.0.mlp.down_proj.weight", "model.layers.1.mlp.down_proj.weight"):
            t[ch, :] *= outliers
    return t


def synthetic_state_dict(d: GroveDims, device="cpu", dtype=torch.float32, names=None, outliers: float = 0.0):
    sd = {}
    for name, shape in param_shapes(d).items():
        if names is not None and name not in names:
            continue
        sd[name] = synthetic_param(name, shape, d, device, outliers).to(dtype)
    return sd


# ----------------------------------------------------------------------------------------------
@dataclass
class SyntheticBatch:
    global_enc_images: torch.Tensor     # [B, 3, T, 336, 336]
    grounding_enc_images: torch.Tensor  # [B, 3, T, 512, 512]
    input_ids: torch.Tensor             # [B, L] int64, one -200
    labels: torch.Tensor                # [B, L] int64
    attention_masks: torch.Tensor       # [B, L] bool
    offset: torch.Tensor
    bboxes_list: List[List[torch.Tensor]]
    temp_objectness_labels_list: List[List[torch.Tensor]]
    original_size_list: List[Tuple[int, int]]

    def as_kwargs(self, inference=False):
        return dict(global_enc_images=self.global_enc_images, grounding_enc_images=self.grounding_enc_images,
                    bboxes_region=None, input_ids=self.input_ids, labels=self.labels,
                    attention_masks=self.attention_masks, offset=self.offset, bboxes_list=self.bboxes_list,
                    temp_objectness_labels_list=self.temp_objectness_labels_list,
                    original_size_list=self.original_size_list, inference=inference)


IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100


def synthetic_batch(d: GroveDims, B=1, T=8, L=64, n_det=2, seed=0, ragged=False, device="cpu", dtype=torch.float32):
    """The collate dict of dataset/dataset.py:9-70 with synthetic content.
    ids: BOS, prompt ids, -200 (the clip), prompt ids, answer with n_det [DET] tokens, right padded."""
    tag = f"batch{seed}"
    g_img = det_tensor(tag + ".global", (B, 3, T, d.clip_image, d.clip_image), std=1.0, device=device, dtype=dtype)
    s_img = det_tensor(tag + ".grounding", (B, 3, T, d.sam_image, d.sam_image), std=1.0, device=device)
    # letter-box: zero band at the bottom/right like SAM preprocessing of a 640x360 frame
    band = int(d.sam_image * 360 / 640)
    s_img[..., band:, :] = 0
    s_img = s_img.to(dtype)
    ids = torch.full((B, L), d.pad_token_id, dtype=torch.int64)
    labels = torch.full((B, L), IGNORE_INDEX, dtype=torch.int64)
    mask = torch.zeros((B, L), dtype=torch.bool)
    usable = d.vocab - 8
    for b in range(B):
        Lb = L - (3 * b if ragged else 0)
        u = det_uniform01(f"{tag}.ids{b}", (Lb,))
        row = (u * (usable - 3)).long() + 3
        row[0] = d.bos_token_id
        n_prompt = Lb // 2
        row[n_prompt // 2] = IMAGE_TOKEN_INDEX
        ans0 = n_prompt
        span = (Lb - ans0 - 1) // max(n_det, 1)
        for k in range(n_det):
            row[ans0 + (k + 1) * span - 1] = d.det_token_idx
        row[Lb - 1] = d.eos_token_id
        ids[b, :Lb] = row
        labels[b, ans0:Lb] = row[ans0:Lb]
        mask[b, :Lb] = True
    boxes, vis = [], []
    for b in range(B):
        bl, vl = [], []
        for t in range(T):
            v = (det_uniform01(f"{tag}.vis{b}.{t}", (n_det,)) < 0.7).float()
            if t == 0 and n_det > 0 and v.sum() == 0:
                v[0] = 1.0
            k = int(v.sum().item())
            bx = det_uniform01(f"{tag}.box{b}.{t}", (k, 4)) * 0.5 + 0.2
            bl.append(bx.to(device))
            vl.append(v.to(device))
        boxes.append(bl)
        vis.append(vl)
    return SyntheticBatch(g_img, s_img, ids.to(device), labels.to(device), mask.to(device),
                          torch.arange(B + 1, dtype=torch.int64, device=device), boxes, vis, [(640, 360)] * B)
