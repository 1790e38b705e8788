"""Reference-checkpoint import / export — host logic only.

The reference moves weights around as FLAT state dicts with its own key names (SURVEY.md section 8(b) "State-dict names"):
  * a fine-tuned GROVE checkpoint is the DeepSpeed run consolidated by `zero_to_fp32.py ./ pytorch_model.bin`
    (infer_eval_scripts/infer_eval_iground.sh:11-15) and read back with `torch.load` + `load_state_dict(strict=False)`
    (infer_iground.py:526-528, train.py:621-624); with LoRA the keys gain a `base_model.model.` prefix (infer_iground.py:530-535);
  * the pre-trained base is a HuggingFace directory (`from_pretrained`, train.py:207-218): `pytorch_model.bin`, or shards
    listed in `pytorch_model.bin.index.json` / `model.safetensors.index.json`;
  * SAM's absolute and global-block relative position tables are resized from the 1024-pixel geometry to 512 when the model
    is built (`interpolate_positional_embeddings`, train.py:503-576).
This module reads all of those into the model's `load_state_dict`, and writes the consolidated `pytorch_model.bin` the
reference's inference scripts expect, without DeepSpeed. Tensors stay on the host until `load_state_dict` copies them.
"""
import json
import os

import torch
import torch.nn.functional as F

_PREFIXES = ("module.", "base_model.model.")  # DistributedDataParallel / DeepSpeed engine wrapper, peft LoRA wrapper


def _strip(name):
    changed = True
    while changed:
        changed = False
        for p in _PREFIXES:
            if name.startswith(p):
                name, changed = name[len(p):], True
    return name


def _read_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "module" in obj and isinstance(obj["module"], dict):  # a DeepSpeed mp_rank_*_model_states.pt / our .pt
        obj = obj["module"]
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{path}: not a flat state dict of tensors")
    return obj


def read_state_dict(path):
    """A flat {reference key name: CPU tensor} from a file (.bin / .pt / .pth / .safetensors) or a HuggingFace-style directory
    (single file or index.json + shards). Wrapper prefixes (`module.`, `base_model.model.`) are removed."""
    files = []
    if os.path.isdir(path):
        for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
            ip = os.path.join(path, index)
            if os.path.exists(ip):
                with open(ip) as fh:
                    files = sorted({os.path.join(path, f) for f in json.load(fh)["weight_map"].values()})
                break
        if not files:
            for single in ("model.safetensors", "pytorch_model.bin"):
                if os.path.exists(os.path.join(path, single)):
                    files = [os.path.join(path, single)]
                    break
        if not files:
            raise FileNotFoundError(f"{path}: no pytorch_model.bin / model.safetensors (or their index.json) inside")
    else:
        files = [path]
    sd = {}
    for f in files:
        for k, v in _read_file(f).items():
            sd[_strip(k)] = v
    return sd


def resize_abs_pos_embedding(pos_embed, target_size, patch_size):
    """train.py:503-529: [1, g, g, C] -> [1, target/patch, target/patch, C], bicubic, align_corners=False."""
    n = target_size // patch_size
    x = pos_embed.float().permute(0, 3, 1, 2)
    return F.interpolate(x, size=(n, n), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).to(pos_embed.dtype)


def resize_rel_pos_embedding(rel_pos_h, rel_pos_w, target_size, patch_size):
    """train.py:532-558: [2g-1, d] -> [2 * target/patch - 1, d] for both tables, bicubic along the position axis with
    align_corners=True (the reference interpolates an [n, 1] / [1, n] image; the unit axis is left alone)."""
    n = 2 * (target_size // patch_size) - 1
    h = rel_pos_h.float().unsqueeze(0).unsqueeze(0).permute(0, 3, 2, 1)
    w = rel_pos_w.float().unsqueeze(0).unsqueeze(0).permute(0, 3, 1, 2)
    h = F.interpolate(h, size=(n, 1), mode="bicubic", align_corners=True).permute(0, 3, 2, 1).squeeze(0).squeeze(0)
    w = F.interpolate(w, size=(1, n), mode="bicubic", align_corners=True).permute(0, 2, 3, 1).squeeze(0).squeeze(0)
    return h.to(rel_pos_h.dtype), w.to(rel_pos_w.dtype)


def interpolate_positional_embeddings(sd, img_size, patch_size, global_blocks, prefix="model.grounding_encoder.image_encoder."):
    """train.py:561-576 on a state dict: resize SAM's `pos_embed` and the global-attention blocks' `rel_pos_h/w` to the
    `img_size` geometry when they were saved for another one (the pre-trained 1024-pixel SAM). Returns the changed keys."""
    changed = []
    g = img_size // patch_size
    k = prefix + "pos_embed"
    if k in sd and sd[k].shape[1] != g:
        sd[k] = resize_abs_pos_embedding(sd[k], img_size, patch_size).contiguous()
        changed.append(k)
    for i in global_blocks:
        kh, kw = prefix + f"blocks.{i}.attn.rel_pos_h", prefix + f"blocks.{i}.attn.rel_pos_w"
        if kh in sd and kw in sd and sd[kh].shape[0] != 2 * g - 1:
            sd[kh], sd[kw] = (t.contiguous() for t in resize_rel_pos_embedding(sd[kh], sd[kw], img_size, patch_size))
            changed += [kh, kw]
    return changed


def dims_from_checkpoint(path, sd=None, base=None):
    """Architecture dimensions of a checkpoint: the HuggingFace `config.json` beside it when there is one (LlamaConfig names:
    hidden_size, num_hidden_layers, num_attention_heads, intermediate_size, rms_norm_eps, rope_theta — what
    `from_pretrained` reads at train.py:207-218), the vocabulary size from `model.embed_tokens.weight` itself (the reference
    resizes it after adding its special tokens, train.py:330), everything else the real GROVE geometry."""
    from dataclasses import replace
    from .synthetic import GroveDims
    d = base if base is not None else GroveDims()
    cfg_path = os.path.join(path if os.path.isdir(path) else os.path.dirname(os.path.abspath(path)), "config.json")
    upd = {}
    if os.path.exists(cfg_path):
        with open(cfg_path) as fh:
            cfg = json.load(fh)
        for ours, theirs in (("hidden", "hidden_size"), ("n_layers", "num_hidden_layers"), ("n_heads", "num_attention_heads"),
                             ("mlp", "intermediate_size"), ("rms_eps", "rms_norm_eps"), ("rope_theta", "rope_theta"),
                             ("vocab", "vocab_size"), ("bos_token_id", "bos_token_id"), ("eos_token_id", "eos_token_id"),
                             ("pad_token_id", "pad_token_id")):
            if cfg.get(theirs) is not None:
                upd[ours] = cfg[theirs]
    if sd is not None and "model.embed_tokens.weight" in sd:
        upd["vocab"] = int(sd["model.embed_tokens.weight"].shape[0])
    return replace(d, **upd)


def constructor_init(name, shape):
    """The value a parameter has in the reference when NO checkpoint provides it: PyTorch's constructor defaults, which is what
    `initialize_custom_layers_in_model` (train.py:160-191) and GROVEForCausalLM.__init__ (GROVE.py:75-79) leave in the custom layers
    of a model started from a base LLaVA checkpoint — nn.Linear / nn.Conv3d: weight and bias U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    (kaiming_uniform with a = sqrt(5)); adapter `alpha` zeros (image_encoder.py:45); LayerNorm 1 / 0; nn.Embedding N(0, 1).
    Deterministic in (name, element index) so that every rank builds the same tensor."""
    import math
    from .synthetic import det_uniform01
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "alpha":
        return torch.zeros(shape)
    if "norm" in name and len(shape) == 1:
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    if name.endswith(("iou_token.weight", "mask_tokens.weight", "no_mask_embed.weight", "embed_tokens.weight")):
        u1, u2 = det_uniform01(name + "#1", shape), det_uniform01(name + "#2", shape)
        return torch.sqrt(-2.0 * torch.log(u1.clamp_min(1e-7))) * torch.cos(2.0 * math.pi * u2)  # Box-Muller: N(0, 1)
    if leaf == "bias":
        return ("bias", name[: -len("bias")] + "weight")  # bound comes from the fan-in of the `weight` beside it
    fan_in = 1
    for s_ in shape[1:]:
        fan_in *= int(s_)
    bound = 1.0 / math.sqrt(max(fan_in, 1))
    return (det_uniform01(name, shape) * 2.0 - 1.0) * bound


def init_missing_trainable(model, missing):
    """{name: tensor} for the TRAINABLE parameters a checkpoint does not provide (they used to stay all-zero — a dead
    ReLU MLP for `text_hidden_fcs` / the box head, silently). Frozen ones are left to the caller to report."""
    import math
    from .model.GROVE import trainable_names
    from .synthetic import det_uniform01, param_shapes
    shapes = param_shapes(model.dims)
    train = set(model.trainable) if getattr(model, "trainable", None) is not None else set(trainable_names(model.dims))
    out = {}
    for n in missing:
        if n not in train:
            continue
        v = constructor_init(n, shapes[n])
        if isinstance(v, tuple):  # bias: U(-1/sqrt(fan_in of its weight), +)
            wshape = shapes.get(v[1])
            fan_in = 1
            for s_ in (wshape[1:] if wshape is not None else shapes[n]):
                fan_in *= int(s_)
            v = (det_uniform01(n, shapes[n]) * 2.0 - 1.0) / math.sqrt(max(fan_in, 1))
        out[n] = v
    return out


def load_grove_weights(model, path, strict=False, sd=None, log=None):
    """infer_iground.py:526-535 / train.py:621-624: read `path`, fit SAM's position tables to the model's image size, and
    `load_state_dict` (non-strict by default, as the reference). Shape mismatches raise — a silently skipped tensor would
    leave synthetic weights in place. Keys the checkpoint lacks: trainable ones (a base LLaVA checkpoint has no `text_hidden_fcs`,
    SAM adapters or box heads, train.py:207-218) get the reference's constructor initialisation (`constructor_init`), as the
    reference's own modules would hold; frozen ones stay zero and are reported with a warning. Returns the load report
    (missing_keys, unexpected_keys) plus `resized`, `initialised`, `missing_frozen`."""
    import warnings
    sd = read_state_dict(path) if sd is None else sd
    d = model.dims
    resized = interpolate_positional_embeddings(sd, d.sam_image, d.sam_patch, d.sam_global)
    want = model.state_dict()
    bad = [(k, tuple(v.shape), tuple(want[k].shape)) for k, v in sd.items() if k in want and tuple(v.shape) != tuple(want[k].shape)]
    if bad:
        raise RuntimeError(f"{path}: {len(bad)} tensors do not fit the model, e.g. {bad[:3]}")
    missing = [n for n in want if n not in sd]
    init = init_missing_trainable(model, missing)
    if init:
        sd = dict(sd)
        sd.update(init)
    rep = model.load_state_dict(sd, strict=strict)
    # tensors the checkpoint carries for modules that are not on this path (the never-executed region encoder, the prompt encoder's
    # point / mask tables — DESIGN section 8): kept on the host and written back by consolidated_state_dict, so that a checkpoint
    # saved here has the reference model's FULL key set again (infer_anet.py:556 loads with strict=True)
    model._passthrough = {k: v.detach().cpu() for k, v in sd.items() if k not in want}
    rep.missing_keys = missing
    rep.resized = resized
    rep.initialised = sorted(init)
    rep.missing_frozen = [n for n in missing if n not in init]
    say = log if log is not None else (lambda m: warnings.warn(m, stacklevel=2))
    if rep.initialised:
        say(f"{path}: {len(rep.initialised)} trainable tensors absent from the checkpoint were given the reference's constructor "
            f"initialisation (e.g. {rep.initialised[:3]})")
    if rep.missing_frozen:
        say(f"{path}: {len(rep.missing_frozen)} FROZEN tensors are absent from the checkpoint and stay zero (e.g. {rep.missing_frozen[:3]})")
    return rep


def consolidated_state_dict(model, engine=None, dtype=torch.float32):
    """What `zero_to_fp32.py ./ pytorch_model.bin` produces from a DeepSpeed run (infer_eval_iground.sh:13): every parameter
    under its reference key name in fp32 — trainable ones from the optimizer's fp32 master copy when an engine is given
    (bit-exact resume), the frozen ones widened from the model's bf16."""
    out = {k: v.detach().to("cpu", dtype) for k, v in model.state_dict().items()}
    if engine is not None:
        flat = engine.master.detach().cpu()
        for name in model.trainable:
            off = model._grad_off[name]  # the master copy shares the flat gradient buffer's layout
            ref_shape = out[name].shape
            t = flat[off:off + out[name].numel()]
            if name.endswith("conv3d.weight"):  # master holds the packed [Co, 27 * Ci] layout the kernels train in
                co, ci = ref_shape[0], ref_shape[1]
                t = t.view(co, 27, ci).permute(0, 2, 1).reshape(ref_shape)
            out[name] = t.reshape(ref_shape).to(dtype).clone()
    obj = "model.grounding_encoder.mask_decoder.temporal_objectness_head."
    for k, v in (getattr(model, "_passthrough", None) or {}).items():
        if k not in out and not (k.startswith(obj) and not getattr(getattr(model, "config", None), "use_temp_objectness", True)):
            out[k] = v.to(dtype) if v.is_floating_point() else v  # (an integer buffer — older HF CLIP's position_ids — keeps its dtype: ADVICE r5)
    return out


def save_grove_weights(model, path, engine=None, dtype=torch.float32):
    """Write the consolidated checkpoint (`pytorch_model.bin` by convention) that infer_iground.py --grove_weights reads."""
    sd = consolidated_state_dict(model, engine, dtype)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(sd, path)
    return sd
