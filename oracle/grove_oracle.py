"""CPU oracle: a plain-PyTorch fp32 restatement of GROVE's per-clip forward/backward hot path.

TEST INFRASTRUCTURE ONLY. Imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg — never by grove_amd (the product path has no CPU fallback).

Parity status: PINNED against the reference itself. oracle/refgen/make_goldens.py imports
/root/reference in the build container, runs model/GROVE.py::GROVEForCausalLM on deterministic
synthetic weights/inputs (grove_amd/synthetic.py) and stores its outputs under tests/golden/;
tests/test_oracle_golden.py checks every function below against those vectors. Two third-party
pieces are NOT pinned by any reference-side test or fixture ("parity unpinned"):
  * the LLaMA decoder arithmetic (transformers 4.46.3 `LlamaModel`, un-vendored; the goldens come
    from the installed transformers 5.x eager `LlamaModel`, same maths);
  * torchvision 0.20.1 `generalized_box_iou_loss` (not installed; restated from its published
    formula both here and in the golden generator).

Everything is functional: `sd` is a state dict with the reference's key names
(grove_amd.synthetic.param_shapes), `d` a GroveDims. Each function cites the reference lines it
restates (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200  # utils/utils.py:9
IGNORE_INDEX = -100       # utils/utils.py:10

V = "model.vision_tower.vision_tower.vision_model."
S = "model.grounding_encoder.image_encoder."
M = "model.grounding_encoder.mask_decoder."
PE = "model.grounding_encoder.prompt_encoder."


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


# --------------------------------------------------------------------------------------------
# CLIP vision tower (model/llava/model/multimodal_encoder/modeling_clip.py)
# --------------------------------------------------------------------------------------------
def clip_embeddings(sd, d, frames):
    """CLIPVisionEmbeddings.forward :187-196 + pre_layrnorm :915. frames [F,3,336,336] -> [F,577,C]."""
    pe = F.conv2d(frames, sd[V + "embeddings.patch_embedding.weight"], stride=d.clip_patch)
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd[V + "embeddings.class_embedding"].expand(frames.shape[0], 1, -1)
    x = torch.cat([cls, pe], 1) + sd[V + "embeddings.position_embedding.weight"][None]
    return _ln(sd, V + "pre_layrnorm", x, d.clip_eps)


def clip_attention(sd, d, p, x):
    """CLIPAttention.forward :257-333 (q pre-scaled :269, softmax over keys :305)."""
    Fb, L, C = x.shape
    H, hd = d.clip_heads, C // d.clip_heads
    q = _lin(sd, p + "q_proj", x) * hd ** -0.5
    k, v = _lin(sd, p + "k_proj", x), _lin(sd, p + "v_proj", x)
    sh = lambda t: t.view(Fb, L, H, hd).transpose(1, 2)  # noqa: E731
    att = torch.softmax(sh(q) @ sh(k).transpose(-1, -2), -1)
    o = (att @ sh(v)).transpose(1, 2).reshape(Fb, L, C)
    return _lin(sd, p + "out_proj", o)


def clip_layer(sd, d, i, x):
    """CLIPEncoderLayer.forward :377-398; CLIPMLP quick_gelu :344-348."""
    p = V + f"encoder.layers.{i}."
    x = x + clip_attention(sd, d, p + "self_attn.", _ln(sd, p + "layer_norm1", x, d.clip_eps))
    h = _lin(sd, p + "mlp.fc1", _ln(sd, p + "layer_norm2", x, d.clip_eps))
    h = h * torch.sigmoid(1.702 * h)
    return x + _lin(sd, p + "mlp.fc2", h)


def conv_adapter(x5, w, b, alpha):
    """tanh(alpha) * relu(Conv3d 3x3x3 'same') + x on [B,C,T,H,W] (modeling_clip.py:607, image_encoder.py:54)."""
    return torch.tanh(alpha) * F.relu(F.conv3d(x5, w, b, padding=1)) + x5


def clip_adapter(sd, d, j, x):
    """SpatioTemporalConvAdapter.forward :599-611: CLS split off, '(b t) (h w) c -> b c t h w' with
    t=8, h=16 (so the 24x24 grid is reinterpreted as 16x36, quirk Q2)."""
    p = V + f"encoder.adapters.{j}."
    cls, seq = x[:, :1], x[:, 1:]
    Fb, HW, C = seq.shape
    h, w = 16, HW // 16
    s5 = seq.reshape(Fb // 8, 8, h, w, C).permute(0, 4, 1, 2, 3)
    s5 = conv_adapter(s5, sd[p + "conv3d.weight"], sd[p + "conv3d.bias"], sd[p + "alpha"])
    seq = s5.permute(0, 2, 3, 4, 1).reshape(Fb, HW, C)
    return torch.cat([cls, seq], 1)


def clip_hidden_states(sd, d, frames, n_layers=None):
    """CLIPEncoder.forward :665-722: adapter after layers 0,3,6,... (:705-707). Returns the tuple of
    hidden states (input of every layer + final output)."""
    x = clip_embeddings(sd, d, frames)
    hs = [x]
    n = d.clip_layers if n_layers is None else n_layers
    for i in range(n):
        x = clip_layer(sd, d, i, x)
        if i % 3 == 0:
            x = clip_adapter(sd, d, i // 3, x)
        hs.append(x)
    return hs


def clip_pool(x):
    """AdaptiveAvgPooling3D.forward pooling.py:15-25: [(b 8),576,C] -> [b,576,C] via AdaptiveAvgPool3d((8,8,9))."""
    Fb, HW, C = x.shape
    h = w = int(HW ** 0.5)
    x5 = x.reshape(Fb // 8, 8, h, w, C).permute(0, 4, 1, 2, 3)
    x5 = F.adaptive_avg_pool3d(x5, (8, 8, 9))
    return x5.permute(0, 2, 3, 4, 1).reshape(Fb // 8, 576, C)


def clip_vision_tower(sd, d, images):
    """CLIPVisionTower.forward clip_encoder.py:55-82: 'b c t h w -> (b t) c h w', hidden_states[-2][:,1:],
    pool. Layer `clip_layers-1` is never consumed (quirk Q6) so it is not run."""
    B, Cc, T, H, W = images.shape
    frames = images.permute(0, 2, 1, 3, 4).reshape(B * T, Cc, H, W)
    hs = clip_hidden_states(sd, d, frames, n_layers=d.clip_layers - 1)
    return clip_pool(hs[-1][:, 1:]), hs


def encode_images(sd, d, images):
    """LlavaMetaForCausalLM.encode_images llava_with_region_arch.py:79-82 (mm_projector :16-19)."""
    feats, hs = clip_vision_tower(sd, d, images)
    h = F.gelu(_lin(sd, "model.mm_projector.0", feats))
    return _lin(sd, "model.mm_projector.2", h), hs


# --------------------------------------------------------------------------------------------
# multimodal splice (llava_with_region_arch.py:84-440, mm_use_im_start_end branch :212-253)
# --------------------------------------------------------------------------------------------
def splice(sd, input_ids, labels, attention_mask, image_features, token_embeddings=None, literal_T=True):
    """Replaces the single -200 of every row by the 576 projected visual tokens of
    image_features[cur_image_idx] (:156 — row b takes feature row b, quirk Q1), IGNORE labels over the
    visual span (:241-249), right-pads embeds with zeros / labels with IGNORE (:354-390), extends the
    attention mask with True on the left and False on the right (:392-418, :425-437)."""
    table = token_embeddings if token_embeddings is not None else sd["model.embed_tokens.weight"]
    embeds, new_labels, lens = [], [], []
    for b in range(input_ids.shape[0]):
        ids = input_ids[b]
        pos = torch.where(ids == IMAGE_TOKEN_INDEX)[0]
        if pos.numel() == 0:
            embeds.append(table[ids])
            if labels is not None:
                new_labels.append(labels[b])
            continue
        s = int(pos[0])
        feats = image_features[b]
        embeds.append(torch.cat([table[ids[:s]], feats, table[ids[s + 1:]]], 0))
        if labels is not None:
            ign = torch.full((feats.shape[0],), IGNORE_INDEX, dtype=labels.dtype)
            new_labels.append(torch.cat([labels[b, :s], ign, labels[b, s + 1:]], 0))
    max_len = max(e.shape[0] for e in embeds)
    L = input_ids.shape[1]
    out_e = torch.zeros(len(embeds), max_len, embeds[0].shape[1], dtype=embeds[0].dtype)
    out_l = torch.full((len(embeds), max_len), IGNORE_INDEX, dtype=torch.int64) if labels is not None else None
    out_m = None
    if attention_mask is not None:
        out_m = torch.zeros(len(embeds), max_len, dtype=torch.bool)
    for b, e in enumerate(embeds):
        n = e.shape[0]
        out_e[b, :n] = e
        if labels is not None:
            out_l[b, :n] = new_labels[b]
        if attention_mask is not None:
            left = n - L
            out_m[b, :left] = True
            out_m[b, left:n] = attention_mask[b]
    return out_e, out_l, out_m


# --------------------------------------------------------------------------------------------
# LLaMA decoder (HF transformers LlamaModel, called at llava_llama.py:100-109) — third party
# --------------------------------------------------------------------------------------------
def _rope_cos_sin(d, positions):
    inv = 1.0 / (d.rope_theta ** (torch.arange(0, d.head_dim, 2, dtype=torch.float32) / d.head_dim))
    ang = positions.float()[:, None] * inv[None]
    emb = torch.cat([ang, ang], -1)
    return emb.cos(), emb.sin()


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def rms_norm(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def llama_layer(sd, d, i, x, cos, sin, add_mask):
    p = f"model.layers.{i}."
    B, S_, H = x.shape
    nh, hd = d.n_heads, d.head_dim
    h = rms_norm(x, sd[p + "input_layernorm.weight"], d.rms_eps)
    sh = lambda t: t.view(B, S_, nh, hd).transpose(1, 2)  # noqa: E731
    q, k, v = (sh(_lin(sd, p + f"self_attn.{n}_proj", h)) for n in "qkv")
    q = q * cos + _rotate_half(q) * sin
    k = k * cos + _rotate_half(k) * sin
    att = q @ k.transpose(-1, -2) * hd ** -0.5 + add_mask
    att = torch.softmax(att.float(), -1).to(q.dtype)
    o = (att @ v).transpose(1, 2).reshape(B, S_, H)
    x = x + _lin(sd, p + "self_attn.o_proj", o)
    h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], d.rms_eps)
    h = F.silu(_lin(sd, p + "mlp.gate_proj", h)) * _lin(sd, p + "mlp.up_proj", h)
    return x + _lin(sd, p + "mlp.down_proj", h)


def llama_forward(sd, d, embeds, attention_mask=None):
    """Full causal forward with a [B,S] key-padding mask; positions = arange(S). Returns the post-norm
    last hidden state (== hidden_states[-1], which GROVE.py:249 consumes)."""
    B, S_, _ = embeds.shape
    cos, sin = _rope_cos_sin(d, torch.arange(S_))
    neg = torch.finfo(torch.float32).min
    add = torch.full((S_, S_), neg).triu(1)[None, None].expand(B, 1, S_, S_).clone()
    if attention_mask is not None:
        add = add.masked_fill(~attention_mask[:, None, None, :], neg)
    x = embeds
    for i in range(d.n_layers):
        x = llama_layer(sd, d, i, x, cos, sin, add)
    return rms_norm(x, sd["model.norm.weight"], d.rms_eps)


def llama_forward_cached(sd, d, embeds_new, cache):
    """The same stack on NEW positions only, against a key / value cache of the earlier ones (HF `use_cache=True`, the form
    `GenerationMixin.generate` drives: GROVE.py:418-422, llava_llama.py:144-180): embeds_new [B, n, H] are positions P .. P + n - 1
    where P = the cache length; cache = list over layers of (K, V) [B, heads, P, hd] (RoPE already applied to K), [] before the
    prefill. No padding (quirk Q9: every caller feeds un-padded, equal-length prompts). Returns the post-norm hidden states of the new
    positions; the cache is extended in place. Cached and uncached streams agree (tests/test_oracle_golden.py)."""
    B, n, H = embeds_new.shape
    nh, hd = d.n_heads, d.head_dim
    P = cache[0][0].shape[2] if cache else 0
    cos, sin = (t.to(embeds_new.dtype) for t in _rope_cos_sin(d, torch.arange(P, P + n)))  # (computed in fp32, cast to the activation dtype: HF :113-127)
    neg = torch.finfo(torch.float32).min
    add = torch.cat([torch.zeros(n, P), torch.full((n, n), neg).triu(1)], 1)[None, None]  # everything cached + causal inside the new block
    x = embeds_new
    for i in range(d.n_layers):
        p = f"model.layers.{i}."
        h = rms_norm(x, sd[p + "input_layernorm.weight"], d.rms_eps)
        sh = lambda t: t.view(B, n, nh, hd).transpose(1, 2)  # noqa: E731
        q, k, v = (sh(_lin(sd, p + f"self_attn.{m}_proj", h)) for m in "qkv")
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        if len(cache) > i:
            k, v = torch.cat([cache[i][0], k], 2), torch.cat([cache[i][1], v], 2)
            cache[i] = (k, v)
        else:
            cache.append((k, v))
        att = torch.softmax((q @ k.transpose(-1, -2) * hd ** -0.5 + add).float(), -1).to(q.dtype)
        o = (att @ v).transpose(1, 2).reshape(B, n, H)
        x = x + _lin(sd, p + "self_attn.o_proj", o)
        h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], d.rms_eps)
        h = F.silu(_lin(sd, p + "mlp.gate_proj", h)) * _lin(sd, p + "mlp.up_proj", h)
        x = x + _lin(sd, p + "mlp.down_proj", h)
    return rms_norm(x, sd["model.norm.weight"], d.rms_eps)


def lm_loss(sd, hidden, labels):
    """lm_head + shifted CrossEntropyLoss (mean over labels != -100), llava_llama.py:111-125."""
    logits = F.linear(hidden, sd["lm_head.weight"])
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), labels[:, 1:].reshape(-1),
                           ignore_index=IGNORE_INDEX), logits


# --------------------------------------------------------------------------------------------
# SAM image encoder (model/SAM/modeling/image_encoder.py)
# --------------------------------------------------------------------------------------------
def _get_rel_pos(size, rel_pos):
    """get_rel_pos :387-417 with q_size == k_size == size and an already-resized table (2*size-1 rows)."""
    assert rel_pos.shape[0] == 2 * size - 1
    c = torch.arange(size)
    return rel_pos[(c[:, None] - c[None, :]) + (size - 1)]


def sam_attention(sd, d, p, x):
    """Attention.forward :301-326 + add_decomposed_rel_pos :420-458 (rel-pos uses the UNSCALED q)."""
    B, H, W, C = x.shape
    nh = d.sam_heads
    qkv = _lin(sd, p + "qkv", x).reshape(B, H * W, 3, nh, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * nh, H * W, -1).unbind(0)
    att = (q * (C // nh) ** -0.5) @ k.transpose(-2, -1)
    Rh, Rw = _get_rel_pos(H, sd[p + "rel_pos_h"]), _get_rel_pos(W, sd[p + "rel_pos_w"])
    rq = q.reshape(B * nh, H, W, -1)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    att = (att.view(B * nh, H, W, H, W) + rel_h[..., :, None] + rel_w[..., None, :]).view(B * nh, H * W, H * W)
    att = att.softmax(-1)
    o = (att @ v).view(B, nh, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return _lin(sd, p + "proj", o)


def sam_block(sd, d, i, x):
    """Block.forward :243-259 with window_partition/unpartition :329-384: zero padding is applied AFTER
    norm1, so pad tokens act as bias-only keys (quirk Q4)."""
    p = S + f"blocks.{i}."
    ws = 0 if i in d.sam_global else d.sam_window
    sc = x
    x = _ln(sd, p + "norm1", x, 1e-6)
    if ws > 0:
        B, H, W, C = x.shape
        ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    x = sam_attention(sd, d, p + "attn.", x)
    if ws > 0:
        x = x.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, :H, :W]
    x = sc + x
    h = F.gelu(_lin(sd, p + "mlp.lin1", _ln(sd, p + "norm2", x, 1e-6)))
    return x + _lin(sd, p + "mlp.lin2", h)


def sam_adapter(sd, d, j, x):
    """SpatioTemporalConvAdapter.forward :48-59: '(b t) h w c -> b c t h w', t=8."""
    p = S + f"adapters.{j}."
    Fb, H, W, C = x.shape
    x5 = x.reshape(Fb // 8, 8, H, W, C).permute(0, 4, 1, 2, 3)
    x5 = conv_adapter(x5, sd[p + "conv3d.weight"], sd[p + "conv3d.bias"], sd[p + "alpha"])
    return x5.permute(0, 2, 3, 4, 1).reshape(Fb, H, W, C)


def _ln2d(x, w, b, eps=1e-6):
    """LayerNorm2d common.py:32-43 over the channel dim of NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def sam_image_encoder(sd, d, images, upto=None):
    """ImageEncoderViT.forward :172-191. images [B,3,T,512,512] -> [B*T,256,32,32]."""
    B, Cc, T, H, W = images.shape
    x = images.permute(0, 2, 1, 3, 4).reshape(B * T, Cc, H, W)
    x = F.conv2d(x, sd[S + "patch_embed.proj.weight"], sd[S + "patch_embed.proj.bias"], stride=d.sam_patch)
    x = x.permute(0, 2, 3, 1) + sd[S + "pos_embed"]
    for i in range(d.sam_depth if upto is None else upto):
        x = sam_block(sd, d, i, x)
        if i in d.sam_global:
            x = sam_adapter(sd, d, d.sam_global.index(i), x)
    if upto is not None:
        return x
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[S + "neck.0.weight"])
    x = _ln2d(x, sd[S + "neck.1.weight"], sd[S + "neck.1.bias"])
    x = F.conv2d(x, sd[S + "neck.2.weight"], padding=1)
    return _ln2d(x, sd[S + "neck.3.weight"], sd[S + "neck.3.bias"])


# --------------------------------------------------------------------------------------------
# prompt encoder + two-way decoder + heads (prompt_encoder.py, transformer.py, mask_decoder.py)
# --------------------------------------------------------------------------------------------
def dense_pe(sd, d, dtype=torch.float32):
    """PromptEncoder.get_dense_pe :67-76 / PositionEmbeddingRandom.forward :216-229, computed in `dtype`
    (the reference computes it in the model dtype, quirk Q10)."""
    G = sd[PE + "pe_layer.positional_encoding_gaussian_matrix"].to(dtype)
    g = d.sam_grid
    grid = torch.ones((g, g), dtype=dtype)
    y = (grid.cumsum(0) - 0.5) / g
    x = (grid.cumsum(1) - 0.5) / g
    c = 2 * torch.stack([x, y], -1) - 1
    c = 2 * math.pi * (c @ G)
    return torch.cat([torch.sin(c), torch.cos(c)], -1).permute(2, 0, 1).unsqueeze(0)


def _dec_attention(sd, p, q, k, v, nh):
    """transformer.py Attention.forward :220-242."""
    q, k, v = _lin(sd, p + "q_proj", q), _lin(sd, p + "k_proj", k), _lin(sd, p + "v_proj", v)
    sep = lambda t: t.reshape(t.shape[0], t.shape[1], nh, -1).transpose(1, 2)  # noqa: E731
    q, k, v = sep(q), sep(k), sep(v)
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), -1)
    o = (att @ v).transpose(1, 2)
    return _lin(sd, p + "out_proj", o.reshape(o.shape[0], o.shape[1], -1))


def two_way_transformer(sd, d, src, pos, tokens):
    """TwoWayTransformer.forward :62-106, TwoWayAttentionBlock.forward :151-182."""
    t = M + "transformer."
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos.flatten(2).permute(0, 2, 1)
    queries, nh = tokens, d.dec_heads
    for i in range(d.dec_depth):
        p = t + f"layers.{i}."
        if i == 0:
            queries = _dec_attention(sd, p + "self_attn.", queries, queries, queries, nh)
        else:
            q = queries + tokens
            queries = queries + _dec_attention(sd, p + "self_attn.", q, q, queries, nh)
        queries = _ln(sd, p + "norm1", queries, 1e-5)
        q, k = queries + tokens, keys + key_pe
        queries = _ln(sd, p + "norm2", queries + _dec_attention(sd, p + "cross_attn_token_to_image.", q, k, keys, nh), 1e-5)
        mlp = _lin(sd, p + "mlp.lin2", F.relu(_lin(sd, p + "mlp.lin1", queries)))
        queries = _ln(sd, p + "norm3", queries + mlp, 1e-5)
        q, k = queries + tokens, keys + key_pe
        keys = _ln(sd, p + "norm4", keys + _dec_attention(sd, p + "cross_attn_image_to_token.", k, q, queries, nh), 1e-5)
    q, k = queries + tokens, keys + key_pe
    queries = queries + _dec_attention(sd, t + "final_attn_token_to_image.", q, k, keys, nh)
    return _ln(sd, t + "norm_final_attn", queries, 1e-5), keys


def mask_decoder_query(sd, d, image_embeddings, image_pe, text_embeds, reps):
    """PromptEncoder.forward :140-186 (text passthrough + no_mask_embed) and MaskDecoder.predict_masks
    "query" branch :164-205. image_embeddings [F,256,g,g]; text_embeds [N,1,256]; reps[f] = number of
    instances that attend to frame f. Returns boxes [N,4] (cxcywh, sigmoid) and objectness logits [N]."""
    N = text_embeds.shape[0]
    out_tok = torch.cat([sd[M + "iou_token.weight"], sd[M + "mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(N, -1, -1), text_embeds], 1)
    idx = torch.repeat_interleave(torch.arange(image_embeddings.shape[0]), torch.tensor(reps))
    g = d.sam_grid
    dense = sd[PE + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(N, -1, g, g)
    src = image_embeddings[idx] + dense
    pos = image_pe.repeat_interleave(N, 0)
    hs, _ = two_way_transformer(sd, d, src, pos, tokens)
    qo = hs[:, 5:, :]
    box = torch.sigmoid(_lin(sd, M + "bbox_prediction_head.2", F.relu(_lin(sd, M + "bbox_prediction_head.0", qo)))).squeeze(1)
    if M + "temporal_objectness_head.weight" in sd:
        obj = _lin(sd, M + "temporal_objectness_head", qo).squeeze(-1).squeeze(-1)
    else:  # use_temp_objectness=False: the decoder has no such head (mask_decoder.py:83-87, 200-205)
        obj = torch.zeros(qo.shape[0])
    return box, obj


def _mlp3(sd, prefix, x):
    """mask_decoder.py MLP.forward :247-252 with three layers (ReLU between)."""
    x = F.relu(_lin(sd, prefix + "layers.0", x))
    x = F.relu(_lin(sd, prefix + "layers.1", x))
    return _lin(sd, prefix + "layers.2", x)


def mask_decoder_masks(sd, d, image_embeddings, image_pe, text_embeds, reps, multimask_output=False):
    """MaskDecoder.forward :87-153 / predict_masks :155-227 with the MASK branch (decoding_type != "query"; dormant in GROVE,
    SURVEY.md section 8(f) 3): the same token / image transformer, then output_upscaling (ConvTranspose2d 256->64 k2 s2,
    LayerNorm2d eps 1e-6, GELU, ConvTranspose2d 64->32 k2 s2, GELU) on the image side, one 3-layer hyper-network MLP per mask
    token, masks = hyper_in @ upscaled, and the IoU head on the iou token. Returns (low-res mask logits [N, 1 or 3, 4g, 4g],
    iou predictions [N, 1 or 3])."""
    N = text_embeds.shape[0]
    out_tok = torch.cat([sd[M + "iou_token.weight"], sd[M + "mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(N, -1, -1), text_embeds], 1)
    idx = torch.repeat_interleave(torch.arange(image_embeddings.shape[0]), torch.tensor(reps))
    g = d.sam_grid
    dense = sd[PE + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(N, -1, g, g)
    src = image_embeddings[idx] + dense
    pos = image_pe.repeat_interleave(N, 0)
    hs, keys = two_way_transformer(sd, d, src, pos, tokens)
    iou_token_out, mask_tokens_out = hs[:, 0, :], hs[:, 1:5, :]
    x = keys.transpose(1, 2).reshape(N, -1, g, g)
    x = F.conv_transpose2d(x, sd[M + "output_upscaling.0.weight"], sd[M + "output_upscaling.0.bias"], stride=2)
    x = F.gelu(_ln2d(x, sd[M + "output_upscaling.1.weight"], sd[M + "output_upscaling.1.bias"]))
    up = F.gelu(F.conv_transpose2d(x, sd[M + "output_upscaling.3.weight"], sd[M + "output_upscaling.3.bias"], stride=2))
    hyper_in = torch.stack([_mlp3(sd, M + f"output_hypernetworks_mlps.{i}.", mask_tokens_out[:, i, :]) for i in range(4)], 1)
    b, c, h, w = up.shape
    masks = (hyper_in @ up.reshape(b, c, h * w)).reshape(b, 4, h, w)
    iou = _mlp3(sd, M + "iou_prediction_head.", iou_token_out)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl], iou[:, sl]


def postprocess_masks(masks, img_size, input_size, original_size):
    """Sam.postprocess_masks sam.py:137-172: bilinear (align_corners=False) to the encoder's square input, crop the padding away,
    bilinear to the original frame size. masks [N, C, h, w] -> [N, C, H_orig, W_orig] (logits; binarise with > 0)."""
    m = F.interpolate(masks.float(), (img_size, img_size), mode="bilinear", align_corners=False)
    m = m[..., : input_size[0], : input_size[1]]
    return F.interpolate(m, original_size, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------------
# GROVE glue (model/GROVE.py)
# --------------------------------------------------------------------------------------------
def det_token_mask(d, input_ids, trailing_pad=True):
    """_create_det_token_mask :200-205 (training/teacher-forced) and evaluate :427-430 (no trailing pad):
    575 zeros | ids[:,1:] == DET | one zero."""
    m = input_ids[:, 1:] == d.det_token_idx
    parts = [torch.zeros((m.shape[0], 575), dtype=torch.bool), m]
    if trailing_pad:
        parts.append(torch.zeros((m.shape[0], 1), dtype=torch.bool))
    return torch.cat(parts, 1)


def text_hidden_fcs(sd, h):
    """GROVE.py:75-79: Linear -> ReLU -> Linear(out_dim)."""
    return _lin(sd, "model.text_hidden_fcs.0.2", F.relu(_lin(sd, "model.text_hidden_fcs.0.0", h)))


def pred_embeddings(sd, d, hidden, mask):
    """_process_hidden_states :248-268: project, repeat per frame, gather DET rows; returns the list of
    B*T tensors [n_det_b, out_dim]."""
    h = text_hidden_fcs(sd, hidden).repeat_interleave(d.num_frames, 0)
    m = mask.repeat_interleave(d.num_frames, 0)
    return [h[i][m[i]] for i in range(h.shape[0])]


def box_cxcywh_to_xyxy(b):
    """utils/bbox_utils.py:46-62."""
    cx, cy, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), -1)


def decode_boxes(sd, d, pred_emb_list, image_embeddings, orig_sizes, pe, infer, thr=0.5, use_temp_objectness=True):
    """_generate_and_postprocess_masks :270-331. use_temp_objectness=False (GROVE.py:282-289, 313-317; mask_decoder.py:83-87, 200-205:
    the decoder has no objectness head): every box of a frame is kept at inference; the logits this function still returns are the
    head's output on whatever weights `sd` holds and are NOT part of the reference's result in that mode (callers drop them)."""
    reps = [e.shape[0] for e in pred_emb_list]
    text = torch.cat(pred_emb_list, 0).unsqueeze(1)
    if text.shape[0] == 0:
        box, obj = torch.zeros(0, 4), torch.zeros(0)
    else:
        box, obj = mask_decoder_query(sd, d, image_embeddings, pe, text, reps)
    T = d.num_frames
    boxes, logits, start = [], [], 0
    for i in range(0, len(reps), T):
        bs, ls = [], []
        for j in range(T):
            n = reps[i + j]
            b, l_ = box[start:start + n], obj[start:start + n]
            if infer:
                W, H = orig_sizes[i // T]
                ub = torch.stack([b[:, 0] * W, b[:, 1] * H, b[:, 2] * W, b[:, 3] * H], -1)  # bbox_utils.py:25-44
                ub = box_cxcywh_to_xyxy(ub)
                bs.append(ub[torch.sigmoid(l_) > thr] if use_temp_objectness else ub)
            else:
                bs.append(b)
            ls.append(l_)
            start += n
        boxes.append(bs)
        logits.append(ls)
    return boxes, logits, box, obj


def giou_loss_sum(b1, b2, eps=1e-7):
    """torchvision.ops.generalized_box_iou_loss(reduction='sum'), restated (parity unpinned)."""
    x1, y1, x2, y2 = b1.unbind(-1)
    x1g, y1g, x2g, y2g = b2.unbind(-1)
    xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.where((yk2 > yk1) & (xk2 > xk1), (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    ac = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
    return (1 - (iou - (ac - union) / (ac + eps))).sum()


def loss_components(ce_loss, boxes, logits, gt_boxes, gt_vis, w_ce=1.0, w_box=1.0, w_obj=1.0, use_temp_objectness=True):
    """_compute_loss_components_video :339-408 (everything in fp32). use_temp_objectness=False = the second branch (:383-408): GIoU + L1
    on the rows the GROUND-TRUTH objectness marks visible, no BCE term, four keys."""
    giou = torch.zeros(())
    l1 = torch.zeros(())
    bce = torch.zeros(())
    n_gt, n_pred = 0, 0
    for b in range(len(boxes)):
        for t in range(len(boxes[b])):
            gb, gv = gt_boxes[b][t].float(), gt_vis[b][t].float()
            pb, pl = boxes[b][t].float(), logits[b][t].float()
            if gb.shape[0] != 0:
                sel = pb[gv.bool()]
                giou = giou + giou_loss_sum(box_cxcywh_to_xyxy(sel), box_cxcywh_to_xyxy(gb))
                l1 = l1 + (sel - gb).abs().sum()
            if use_temp_objectness:
                bce = bce + F.binary_cross_entropy_with_logits(pl, gv, reduction="sum")
            n_gt += gb.shape[0]
            n_pred += pb.shape[0]
    ce = ce_loss * w_ce
    giou = w_box * giou / (n_gt + 1e-8)
    l1 = w_box * l1 / (n_gt + 1e-8)
    bce = w_obj * bce / (n_pred + 1e-8)
    if not use_temp_objectness:
        return {"loss": ce + giou + l1, "ce_loss": ce, "giou_loss": giou, "l1_loss": l1}
    return {"loss": ce + giou + l1 + bce, "ce_loss": ce, "giou_loss": giou, "l1_loss": l1, "temp_objectness_loss": bce}


def model_forward(sd, d, global_enc_images, grounding_enc_images, input_ids, labels, attention_masks, bboxes_list=None,
                  temp_objectness_labels_list=None, original_size_list=None, inference=False, pe_dtype=torch.float32,
                  use_temp_objectness=True, token_embeddings=None, loss_weights=(1.0, 1.0, 1.0), **_):
    """GROVEForCausalLM.model_forward GROVE.py:156-198 (training: losses; inference: boxes + logits).
    pe_dtype: dtype the dense positional encoding is evaluated in (bf16 reproduces the reference under model.to(bf16), Q10).
    use_temp_objectness=False: GROVE.py:183-195 — inference returns every box and `logits_temp_objectness` None; training returns four
    loss keys. token_embeddings: the dumped embedding table of embed_tokens.py (llava_with_region_arch.py:134-137). loss_weights =
    (ce, giou [also weighs L1, GROVE.py:375], temp_objectness)."""
    image_embeddings = sam_image_encoder(sd, d, grounding_enc_images)
    mask = det_token_mask(d, input_ids)
    feats, _ = encode_images(sd, d, global_enc_images)
    if inference:
        embeds, _, _ = splice(sd, input_ids, None, None, feats, token_embeddings=token_embeddings)
        hidden = llama_forward(sd, d, embeds, None)
        ce = None
    else:
        embeds, new_labels, new_mask = splice(sd, input_ids, labels, attention_masks, feats, token_embeddings=token_embeddings)
        hidden = llama_forward(sd, d, embeds, new_mask)
        ce, _ = lm_loss(sd, hidden, new_labels)
    emb = pred_embeddings(sd, d, hidden, mask)
    pe = dense_pe(sd, d, dtype=pe_dtype).float()
    boxes, logits, flat_box, flat_obj = decode_boxes(sd, d, emb, image_embeddings, original_size_list, pe, inference,
                                                     use_temp_objectness=use_temp_objectness)
    if inference:
        return {"pred_bboxes": boxes, "logits_temp_objectness": logits if use_temp_objectness else None, "flat_boxes": flat_box,
                "flat_logits": flat_obj, "hidden": hidden, "image_embeddings": image_embeddings}
    out = loss_components(ce, boxes, logits, bboxes_list, temp_objectness_labels_list, *loss_weights, use_temp_objectness=use_temp_objectness)
    out.update({"flat_boxes": flat_box, "flat_logits": flat_obj, "hidden": hidden})
    return out


def evaluate(sd, d, image_features, image_embeddings, input_ids, original_size_list, max_tokens_new=32, pe=None):
    """GROVEForCausalLM.evaluate GROVE.py:412-451 with HF greedy decoding (num_beams=1, do_sample=False):
    rows finish at eos and are padded with pad_token_id; the hidden state of the last generated token is
    never needed (quirk Q3). Uncached recompute — cached and uncached streams agree."""
    B = input_ids.shape[0]
    ids = input_ids.clone()
    finished = torch.zeros(B, dtype=torch.bool)
    hidden = None
    for _ in range(max_tokens_new):
        embeds, _, _ = splice(sd, ids, None, None, image_features)
        hidden = llama_forward(sd, d, embeds, None)
        nxt = F.linear(hidden[:, -1], sd["lm_head.weight"]).argmax(-1)
        nxt = torch.where(finished, torch.full_like(nxt, d.pad_token_id), nxt)
        ids = torch.cat([ids, nxt[:, None]], 1)
        finished = finished | (nxt == d.eos_token_id)
        if finished.all():
            break
    mask = det_token_mask(d, ids, trailing_pad=False)
    emb = pred_embeddings(sd, d, hidden, mask)
    if pe is None:
        pe = dense_pe(sd, d)
    boxes, logits, flat_box, flat_obj = decode_boxes(sd, d, emb, image_embeddings, original_size_list, pe, True)
    return ids, boxes, logits, flat_box, flat_obj
