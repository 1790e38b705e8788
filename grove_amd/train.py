"""Data-parallel training entry points — the mirror of the reference's train.py for the hot path.

The reference drives the model through a DeepSpeed ZeRO-2 engine (train.py:466-486, 761-782). ZeRO is a
memory trick for 40-80 GB GPUs; at 288 GB per MI355X the build keeps a full replica per GPU and
exchanges gradients with plain RCCL all-reduces over xGMI (one process per GPU, torch.distributed
backend "nccl" == RCCL). GroveEngine exposes the engine methods train.py uses: __call__, backward(loss),
step(), train()/eval(), save_checkpoint(dir), load_checkpoint(dir).

Gradient exchange: the trainable set (~482 M parameters that really receive gradients)
lives in ONE flat fp32 buffer; it is all-reduced in a few large buckets on a side stream (xGMI is
point-to-point, large messages amortise the per-link latency), then a fused AdamW kernel updates the
fp32 master copy and the bf16 model weights in one pass.
"""
import argparse
import math
import os
import time

import torch
import torch.distributed as dist

from . import ops
from .model.GROVE import GROVEForCausalLM, trainable_names
from .synthetic import GroveDims


# Reference flags that only feed subsystems outside the hot path (datasets, tokenizer files, LoRA, logging). They are accepted with
# the reference's names and defaults so that the shipped launch lines parse unchanged; what each one does here is listed.
IGNORED_REFERENCE_FLAGS = {
    "vision_pretrained": "SAM checkpoint path: the SAM weights come with the GROVE checkpoint (--version / --grove_weights)",
    "vision_tower": "CLIP hub id: the CLIP ViT-L/14-336 geometry is fixed (GroveDims); weights come with the checkpoint",
    "conv_type": "conversation template: text plumbing, the loader hands over token ids",
    "tune_mm_mlp_adapter": "LLaVA stage-1 switch, unused by the reference's own code path",
    "freeze_mm_mlp_adapter": "LLaVA stage-1 switch, unused by the reference's own code path",
    "mm_use_im_start_end": "decides the prompt text the dataset builds; the splice handles either form",
    "image_size": "must be 512 (SAM geometry the kernels are built for); checked",
    "model_max_length": "tokenizer truncation length",
    "lora_target_modules": "LoRA only (rejected: --lora_r must be 0)", "lora_alpha": "LoRA only", "lora_dropout": "LoRA only",
    "with_region": "region encoder: constructed but never executed by GROVE (SURVEY.md section 2 row 10)",
    "mm_vision_select_layer": "must be -2 (hidden_states[-2]; CLIP layer 24 is skipped on that ground); checked",
    "pretrain_mm_mlp_adapter": "LLaVA stage-1 projector file", "video_dir": "dataset", "train_ann_dir": "dataset", "val_ann_dir": "dataset",
    "train_keys": "dataset", "val_keys": "dataset", "frame_timestamps": "dataset", "num_classes_per_sample": "dataset",
    "weight": "unused by the reference", "val_batch_size": "validation loader batch (synthetic loader: --batch_size)",
    "workers": "DataLoader workers", "gradient_checkpointing": "activations are kept (288 GB HBM): no recompute, same gradients",
    "use_mm_start_end": "tokenizer: adds <vid_start>/<vid_end>", "val_dataset": "GLaMM leftover, unused", "no_eval": "unused by the reference",
}


def parse_args(argv=None):
    """train.py:40-112 — every flag of the reference with its name and default (so `train_scripts/*.sh` launch lines parse
    unchanged), plus this build's own switches in a separate group. Flags that feed out-of-scope subsystems are accepted and
    listed in IGNORED_REFERENCE_FLAGS; `--lora_r > 0`, `--precision != bf16` and a missing `--pretrained` are rejected by
    `check_supported(args)` (called from main / initialize_model), not silently ignored."""
    p = argparse.ArgumentParser(description="GROVE Model Training (MI355X)")
    # Model-specific settings (train.py:43-58)
    p.add_argument("--version", default="MBZUAI/GLaMM-GCG")
    p.add_argument("--vision_pretrained", default="./checkpoints/sam_vit_h_4b8939.pth", type=str)
    p.add_argument("--vision-tower", default="openai/clip-vit-large-patch14-336", type=str)
    p.add_argument("--conv_type", default="llava_v1", type=str, choices=["llava_v1", "llava_llama_2"])
    p.add_argument("--tune_mm_mlp_adapter", action="store_true")
    p.add_argument("--freeze_mm_mlp_adapter", action="store_true")
    p.add_argument("--mm_use_im_start_end", action="store_true", default=True)
    p.add_argument("--out_dim", default=256, type=int)
    p.add_argument("--image_size", default=512, type=int, help="Image size for grounding image encoder")
    p.add_argument("--model_max_length", default=1536, type=int)
    p.add_argument("--lora_target_modules", default="q_proj,v_proj", type=str)
    p.add_argument("--with_region", action="store_true", default=True)
    p.add_argument("--mm_vision_select_layer", default=-2, type=int)
    p.add_argument("--pretrain_mm_mlp_adapter", default="", type=str)
    p.add_argument("--precision", default="bf16", type=str)
    # Dataset settings (train.py:60-69)
    p.add_argument("--dataset", default="HowToGround", choices=["HowToGround", "ActivityNetEntities", "VidSTG"], type=str)
    p.add_argument("--video_dir", default="/home/HowTo100M_small", type=str)
    p.add_argument("--train_ann_dir", default="/home/train_annotations/", type=str)
    p.add_argument("--val_ann_dir", default="/home/val_annotations/", type=str)
    p.add_argument("--train_keys", default="/home/train_keys.pkl", type=str)
    p.add_argument("--val_keys", default="/home/val_keys.pkl", type=str)
    p.add_argument("--frame_timestamps", default="/home/ActivityNetEntities/timestamps_metadata.json")
    p.add_argument("--num_classes_per_sample", default=3, type=int)
    p.add_argument("--num_frames", default=8, type=int)
    # Training settings (train.py:71-98)
    p.add_argument("--pretrained", action="store_true")
    p.add_argument("--grove_weights", default=None, type=str)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--auto_resume", action="store_true")
    p.add_argument("--weight", default="", type=str)
    p.add_argument("--lr", default=0.0003, type=float)
    p.add_argument("--wd", default=0.0, type=float)
    p.add_argument("--epochs", default=10, type=int)
    p.add_argument("--steps_per_epoch", default=500, type=int)
    p.add_argument("--batch_size", default=1, type=int, help="batch size per device per step")
    p.add_argument("--grad_accumulation_steps", default=1, type=int)
    p.add_argument("--val_batch_size", default=1, type=int)
    p.add_argument("--workers", default=0, type=int)
    p.add_argument("--lora_r", default=8, type=int)
    p.add_argument("--lora_alpha", default=16, type=int)
    p.add_argument("--lora_dropout", default=0.05, type=float)
    p.add_argument("--ce_loss_weight", default=1.0, type=float)
    p.add_argument("--giou_loss_weight", default=1.0, type=float)
    p.add_argument("--temp_objectness_loss_weight", default=1.0, type=float)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--beta2", default=0.95, type=float)
    p.add_argument("--gradient_checkpointing", action="store_true", default=True)
    p.add_argument("--train_mask_decoder", action="store_true", default=False)
    p.add_argument("--use_mm_start_end", action="store_true", default=True)
    p.add_argument("--print_freq", default=1, type=int)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int, help="node rank")
    # Evaluation settings (train.py:100-106)
    p.add_argument("--val_dataset", default="RefCOCOgRegVal", type=str)
    p.add_argument("--bbox_validation", action="store_true")
    p.add_argument("--no_eval", action="store_true")
    p.add_argument("--eval_only", action="store_true")
    # Experiment settings (train.py:108-110)
    p.add_argument("--log_base_dir", default="/home/grove_checkpoints", type=str)
    p.add_argument("--exp_name", default="iGround", type=str)
    # ---- this build's own switches (not in the reference)
    g = p.add_argument_group("grove_amd")
    g.add_argument("--log_dir", default=None, type=str, help="overrides <log_base_dir>/<exp_name> (train.py:116)")
    g.add_argument("--val_batches", default=2, type=int, help="validation batches per epoch (synthetic loader)")
    g.add_argument("--dims", default="full", choices=["full", "tiny"], help="architecture size when no checkpoint gives it")
    g.add_argument("--text_len", default=128, type=int, help="synthetic loader: text ids per sample")
    g.add_argument("--n_det", default=3, type=int, help="synthetic loader: [DET] tokens per sample")
    g.add_argument("--exchange", default="allreduce", choices=["allreduce", "rs_ag", "a2a_f32"], help="N > 1 gradient exchange form")
    g.add_argument("--dense_embed", action="store_true", help="N > 1: embed_tokens' gradient as the dense slice, not touched rows")
    g.add_argument("--no_comm_overlap", action="store_true", help="N > 1: exchange after the backward, not from inside it")
    args = p.parse_args(argv)
    if args.log_dir is None:
        args.log_dir = os.path.join(args.log_base_dir, args.exp_name)  # initialize_environment, train.py:116
    return args


def shipped_args(extra=()):
    """parse_args on the switches every shipped launch line passes (train_scripts/*.sh: `--lora_r 0 --pretrained
    --train_mask_decoder`) — the configuration the build implements; tests and bench.py start from it."""
    return parse_args(["--lora_r", "0", "--pretrained", "--train_mask_decoder"] + list(extra))


def check_supported(args):
    """Reject, loudly, the configurations of the reference's command line that this build does not implement."""
    if getattr(args, "lora_r", 0) > 0:
        raise NotImplementedError(
            f"--lora_r {args.lora_r}: LoRA (peft, train.py:268-271) is out of scope — every shipped launch line passes --lora_r 0 "
            "(train_scripts/*.sh); pass --lora_r 0")
    if not getattr(args, "pretrained", True):
        raise NotImplementedError(
            "without --pretrained the reference rebuilds the whole SAM grounding encoder as trainable modules "
            "(initialize_grove_model, train.py:273-274; GROVE.py:53-59): not a shipped configuration; pass --pretrained")
    if getattr(args, "precision", "bf16") != "bf16":
        raise NotImplementedError(f"--precision {args.precision}: the kernels compute in bf16 with fp32 accumulation (train.py:58 default)")
    if getattr(args, "image_size", 512) != 512:
        raise NotImplementedError(f"--image_size {args.image_size}: the grounding encoder is SAM ViT-H at 512 pixels (train.py:51 default)")
    if getattr(args, "mm_vision_select_layer", -2) != -2:
        raise NotImplementedError("--mm_vision_select_layer must be -2 (train.py:56 default): CLIP layer 24 is never computed")
    if getattr(args, "bbox_validation", False):
        raise NotImplementedError("--bbox_validation: that branch of validate_model_performance cannot run in the reference either "
                                  "(train.py:816-819 reads keys model_forward no longer returns; SURVEY.md quirk Q8)")


class SyntheticTokenizer:
    """Stand-in for the LLaMA sentencepiece tokenizer when no tokenizer files are installed (there is no network): only the
    surface train.py / infer_iground.py touch — len(), add_tokens(), unk/pad/eos/bos ids and tokenizer(text).input_ids for the
    special-token lookups of setup_tokenizer_and_special_tokens (train.py:154-157). Base vocabulary 32000 (Vicuna-7B-v1.5) plus
    the added tokens a GLaMM-GranD-Pretrained tokenizer already carries (non-special added tokens, which the slow LlamaTokenizer
    emits behind a "▁" piece — hence the reference's `.input_ids[1]` for <bbox>, <p>, </p> and `.input_ids[0]` for [DET])."""
    SPIECE_UNDERLINE_ID = 29871
    PRETRAINED_ADDED = ("<im_start>", "<im_end>", "<bbox>", "<point>", "<p>", "</p>")

    def __init__(self, base_vocab=32000, pretrained=True, model_max_length=1536):
        self.base_vocab, self.model_max_length = base_vocab, model_max_length
        self.unk_token, self.bos_token, self.eos_token = "<unk>", "<s>", "</s>"
        self.unk_token_id, self.bos_token_id, self.eos_token_id = 0, 1, 2
        self.pad_token = None
        self.padding_side = "right"
        self._added, self._special = {}, set()
        if pretrained:
            for t in self.PRETRAINED_ADDED:
                self._added[t] = base_vocab + len(self._added)

    @property
    def pad_token_id(self):
        return {self.unk_token: self.unk_token_id, self.eos_token: self.eos_token_id}.get(self.pad_token)

    def __len__(self):
        return self.base_vocab + len(self._added)

    def add_tokens(self, tokens, special_tokens=False):
        n = 0
        for t in ([tokens] if isinstance(tokens, str) else tokens):
            if t not in self._added:
                self._added[t] = self.base_vocab + len(self._added)
                n += 1
            if special_tokens:
                self._special.add(t)
        return n

    def convert_tokens_to_ids(self, t):
        return self._added.get(t, self.unk_token_id)

    def __call__(self, text, add_special_tokens=True):
        from types import SimpleNamespace
        ids = [self.bos_token_id] if add_special_tokens else []
        if text in self._added:
            ids += ([] if text in self._special else [self.SPIECE_UNDERLINE_ID]) + [self._added[text]]
        else:  # no sentencepiece model offline: ordinary text maps to hashed ids (shape-only stand-in)
            import zlib
            ids += [3 + zlib.crc32(w.encode()) % (self.base_vocab - 3) for w in text.split()]
        return SimpleNamespace(input_ids=ids)


DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN = "<vid_start>", "<vid_end>"  # utils/utils.py


def setup_tokenizer_and_special_tokens(args, tokenizer=None):
    """train.py:124-159: load the tokenizer of `args.version` (when its files exist locally; a SyntheticTokenizer otherwise or when
    one is passed in), pad = unk, add <vid_start>/<vid_end> (+ the region / phrase tokens unless --pretrained) and [DET], and
    record `bbox_token_idx / det_token_idx / bop_token_idx / eop_token_idx` on `args` with the reference's index choices."""
    if tokenizer is None:
        if os.path.isdir(str(args.version)):
            import transformers
            tokenizer = transformers.AutoTokenizer.from_pretrained(args.version, model_max_length=args.model_max_length,
                                                                   padding_side="right", use_fast=False)
        else:
            tokenizer = SyntheticTokenizer(pretrained=getattr(args, "pretrained", True), model_max_length=args.model_max_length)
    tokenizer.pad_token = tokenizer.unk_token
    if args.use_mm_start_end:
        tokenizer.add_tokens([DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN], special_tokens=True)
    if not args.pretrained:
        tokenizer.add_tokens(["<bbox>", "<point>"] + ["[DET]"] + ["<p>", "</p>"], special_tokens=True)
    else:
        tokenizer.add_tokens(["[DET]"], special_tokens=True)

    def second(text):  # `.input_ids[1]` (train.py:154,156,157): the id behind the "▁" piece; a one-piece result has no [1]
        ids = tokenizer(text, add_special_tokens=False).input_ids
        return ids[1] if len(ids) > 1 else ids[0]
    args.bbox_token_idx = second("<bbox>")
    args.det_token_idx = tokenizer("[DET]", add_special_tokens=False).input_ids[0]
    args.bop_token_idx = second("<p>")
    args.eop_token_idx = second("</p>")
    return tokenizer


def _reinit(model, names):
    """Give `names` the value the reference's freshly constructed modules hold (checkpoint.constructor_init) and rebuild the
    model's derived state."""
    from .checkpoint import constructor_init
    from .synthetic import det_uniform01, param_shapes
    shapes = param_shapes(model.dims)
    upd = {}
    for n in names:
        v = constructor_init(n, shapes[n])
        if isinstance(v, tuple):
            wshape = shapes.get(v[1], shapes[n])
            fan_in = 1
            for s_ in wshape[1:]:
                fan_in *= int(s_)
            v = (det_uniform01(n, shapes[n]) * 2.0 - 1.0) / math.sqrt(max(fan_in, 1))
        upd[n] = v
    if upd:
        model.load_state_dict(upd, strict=False)
    return sorted(upd)


def initialize_custom_layers_in_model(model):
    """train.py:162-191: fresh SAM spatio-temporal adapters (Conv3d default init, alpha 0), box head (Linear-ReLU-Linear) and —
    with use_temp_objectness — temporal-objectness head. Returns the re-initialised names."""
    from .model.decoder import M_
    from .model.sam import S as SAM_PREFIX
    from .synthetic import param_shapes
    names = [n for n in param_shapes(model.dims) if n.startswith(SAM_PREFIX + "adapters.") or n.startswith(M_ + "bbox_prediction_head.")
             or (model.config.use_temp_objectness and n.startswith(M_ + "temporal_objectness_head."))]
    return _reinit(model, names)


def initialize_custom_layers_in_global_encoder(vision_tower):
    """train.py:222-230: fresh CLIP spatio-temporal adapters (alpha 0: an exact identity). Takes the model (this build has no
    separate vision-tower module object; `model.get_vision_tower()` returns the model itself for this call)."""
    from .synthetic import param_shapes
    model = getattr(vision_tower, "_grove_model", vision_tower)
    names = [n for n in param_shapes(model.dims) if ".vision_model.encoder.adapters." in n]
    return _reinit(model, names)


def setup_lora_config(model, args):
    """train.py:336-359. LoRA / peft is out of scope (every shipped script passes --lora_r 0): importable, refuses to run."""
    raise NotImplementedError("setup_lora_config: LoRA (peft) is out of scope for the MI355X hot path; run with --lora_r 0 "
                              "(train_scripts/*.sh) — merged LoRA checkpoints load through checkpoint.read_state_dict")


def interpolate_positional_embeddings(ds_model, *a, **k):
    """train.py:561-576: SAM's absolute and global-block relative position tables from the 1024-pixel geometry to 512. Two forms:
    `(state_dict, img_size, patch_size, global_blocks)` resizes a checkpoint's tables (checkpoint.interpolate_positional_embeddings:
    what load_grove_weights calls); `(model)` — the reference's call — checks that the model's tables have the 512 geometry its
    kernels are built for (they are allocated at that size and every checkpoint is resized while loading) and returns the keys it
    had to change: none."""
    if isinstance(ds_model, dict):
        from .checkpoint import interpolate_positional_embeddings as on_state_dict
        return on_state_dict(ds_model, *a, **k)
    from .model.sam import S as SAM_PREFIX
    d = ds_model.dims
    g = d.sam_image // d.sam_patch
    sd = ds_model._sd
    assert sd[SAM_PREFIX + "pos_embed"].shape[1] == g, "SAM pos_embed is not at the model's geometry"
    for i in d.sam_global:
        assert sd[SAM_PREFIX + f"blocks.{i}.attn.rel_pos_h"].shape[0] == 2 * g - 1
    return []


def initialize_model(args, tokenizer=None, dims=None, state_dict=None, device=None):
    """train.py:194-218 `initialize_model(args, tokenizer)`: GROVEForCausalLM in bf16 with the loss weights / token ids of `args`
    (`from_pretrained(args.version)` when that is a local checkpoint directory; deterministic synthetic weights otherwise — there
    are no checkpoints offline), custom layers re-initialised, token ids of the tokenizer on the config. The vocabulary is
    len(tokenizer) (the reference resizes the embeddings afterwards, train.py:330; here the tables are allocated at that size).
    `dims= / state_dict= / device=` are this build's additions (tests and bench.py build tiny / synthetic models)."""
    from dataclasses import replace
    from .synthetic import FULL, TINY
    if isinstance(tokenizer, GroveDims):  # round-3 call form initialize_model(args, dims, ...)
        tokenizer, dims = None, tokenizer
    check_supported(args)
    device = device or torch.device("cuda", args.local_rank)
    if dims is None:
        dims = FULL if getattr(args, "dims", "full") == "full" else TINY
    if tokenizer is not None and state_dict is None and getattr(args, "dims", "full") == "full":
        dims = replace(dims, vocab=len(tokenizer), det_token_idx=args.det_token_idx,
                       bos_token_id=tokenizer.bos_token_id, eos_token_id=tokenizer.eos_token_id, pad_token_id=tokenizer.pad_token_id)
    kw = dict(device=device, train=True, det_token_idx=getattr(args, "det_token_idx", dims.det_token_idx), num_frames=args.num_frames,
              out_dim=args.out_dim, ce_loss_weight=args.ce_loss_weight, giou_loss_weight=args.giou_loss_weight,
              temp_objectness_loss_weight=args.temp_objectness_loss_weight, train_mask_decoder=args.train_mask_decoder,
              use_temp_objectness=getattr(args, "dataset", "HowToGround") == "HowToGround",   # train.py:203
              bbox_token_idx=getattr(args, "bbox_token_idx", None))
    if getattr(args, "stream_dtype", None) is not None:  # (this build's addition: the residual streams of a training model, bf16 unless asked)
        kw["stream_dtype"] = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(args.stream_dtype, args.stream_dtype)
    if state_dict is None and os.path.isdir(str(getattr(args, "version", ""))):
        model = GROVEForCausalLM.from_pretrained(args.version, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, dims=dims, **kw)
        initialize_custom_layers_in_model(model)
    else:
        if state_dict is None:
            from .synthetic import synthetic_state_dict
            state_dict = synthetic_state_dict(dims, device=device, dtype=torch.bfloat16)
        model = GROVEForCausalLM(dims=dims, state_dict=state_dict, **kw)
    if tokenizer is not None:
        model.config.eos_token_id, model.config.bos_token_id, model.config.pad_token_id = (
            tokenizer.eos_token_id, tokenizer.bos_token_id, tokenizer.pad_token_id)
    return model


def prepare_model_for_training(model, tokenizer=None, args=None):
    """train.py:234-333: the freeze policy. This build fixes the trainable set when the model is allocated (flat gradient buffer),
    so the call VALIDATES that the model was built for the policy `args` asks for — --lora_r 0, --pretrained, the
    --train_mask_decoder choice, len(tokenizer) rows in embed_tokens / lm_head (train.py:330 resize_token_embeddings) — and
    returns the names that train. Gradient checkpointing (train.py:237) is not applied: activations are kept."""
    if args is not None:
        check_supported(args)
        if bool(args.train_mask_decoder) != bool(model.config.train_mask_decoder):
            raise RuntimeError("model was built with train_mask_decoder=%s, args say %s" % (model.config.train_mask_decoder, args.train_mask_decoder))
    if tokenizer is not None and len(tokenizer) != model.dims.vocab and getattr(args, "dims", "full") == "full":
        raise RuntimeError(f"len(tokenizer) = {len(tokenizer)} but the model's embedding tables have {model.dims.vocab} rows: build it with "
                           "initialize_model(args, tokenizer)")
    return list(model.trainable) if getattr(model, "trainable", None) is not None else trainable_names(
        model.dims, model.config.train_mask_decoder, model.config.use_temp_objectness)


class WarmupDecayLR:
    """DeepSpeed WarmupDecayLR (train.py:471-474): linear warm-up 0 -> lr over warmup steps, then linear decay to 0."""

    def __init__(self, lr, total_steps, warmup_steps=100):
        self.lr, self.total, self.warm = lr, max(total_steps, 1), max(2, warmup_steps)  # (WarmupLR.__init__: warmup_num_steps = max(2, n))
        self.last = 0.0

    def for_update(self, k):
        """Learning rate of the k-th optimizer update (k = 1, 2, ...) under DeepSpeed's calling order: the scheduler is built with
        last_batch_iteration = -1, which writes warmup_min_lr (0, train.py:472) into the optimizer — the value update 1 uses —
        and the engine steps the scheduler AFTER every optimizer update (iteration 0 after update 1, gamma(0) = 0 for update 2,
        gamma(1) for update 3, ...). So update k runs with gamma(k - 2): the first two updates have lr = 0.
        Pinned (round 4) against oracle/ds_lr_schedule.py — DeepSpeed 0.15.1's `WarmupLR` / `WarmupDecayLR` classes and the engine's
        optimizer-then-scheduler order restated from the published source and simulated update by update
        (tests/test_distributed_cpu.py::test_lr_of_every_update_follows_deepspeeds_calling_order). DeepSpeed itself is not installed
        offline, so no lr trace of the reference STACK exists: the restatement is flagged "parity unpinned" in its header. (Older
        DeepSpeed releases left the optimizer's base lr for update 1; the pin is the reference's requirements.txt:6.)"""
        return self.get(max(k - 2, 0))

    def get(self, step):
        if step < self.warm:
            g = step / max(1, self.warm)
        else:
            g = max(0.0, (self.total - step) / max(1.0, self.total - self.warm))
        self.last = self.lr * g
        return self.last

    def get_last_lr(self):
        return [self.last]


def allreduce_buckets(flat, bucket_elems, comm_stream=None, comm_buf=None):
    """SUM all-reduce of one flat gradient buffer in buckets of `bucket_elems` elements. On GPU the
    collectives are issued on `comm_stream` (RCCL over xGMI) after the producing stream's work, and the
    compute stream waits for them; on CPU (gloo, used by the tests) the same bucketing runs inline.
    `comm_buf` (bf16, same length): exchange in bf16 — DeepSpeed's `communication_data_type` default when bf16 is enabled
    (the reference's config sets none, train.py:466-478), half the xGMI bytes of the fp32 buffer: the gradients are rounded
    into comm_buf, summed there, and widened back into `flat`."""
    n = flat.numel()
    wire = flat
    if comm_buf is not None:
        assert comm_buf.numel() == n and comm_buf.dtype == torch.bfloat16
        if flat.is_cuda:
            ops.to_bf16(flat, out=comm_buf)
        else:
            comm_buf.copy_(flat)  # host tensors exist only in the gloo tests
        wire = comm_buf
    if comm_stream is None:
        handles = [dist.all_reduce(wire[s0:s0 + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                   for s0 in range(0, n, bucket_elems)]
        for h in handles:
            h.wait()
    else:
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            handles = [dist.all_reduce(wire[s0:s0 + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                       for s0 in range(0, n, bucket_elems)]
            for h in handles:
                h.wait()
        torch.cuda.current_stream().wait_stream(comm_stream)
    if comm_buf is not None:
        if flat.is_cuda:
            ops.to_f32(comm_buf, out=flat)
        else:
            flat.copy_(comm_buf)


class GradExchange:
    """Gradient exchange of the flat fp32 gradient buffer, overlapped with the backward.

    The trainable parameters lie in the flat buffer in GROUPS whose gradients complete at different points of the backward
    (reverse-autograd order: box decoder -> lm_head -> text_hidden_fcs -> SAM adapters 3..0 on the SAM stream -> embed_tokens +
    mm_projector at the very end). `ready(lo, hi, stream)` is called by the model as soon as a group's slice [lo, hi) is final:
    the slice is rounded into the bf16 wire buffer (DeepSpeed's communication dtype under bf16; fp32 optional) and exchanged on the
    communication stream in buckets, while the rest of the backward keeps the CUs busy. `finish()` (engine.step) makes the compute
    stream wait for the collectives and widens the wire buffer back into the fp32 buffer the optimizer reads.
      mode "allreduce": one SUM all-reduce per bucket (RCCL accumulates in the wire dtype).
      mode "rs_ag":     reduce-scatter + all-gather per bucket (the two halves of a ring all-reduce as separate RCCL calls: the form
                        a sharded optimizer update slots between; here the full replica updates everything, so the result is the same).
      mode "a2a_f32":   FP32 ACCUMULATION of a bf16 wire: per bucket an all-to-all hands rank r chunk r of
                        every rank's bucket (xGMI is point-to-point and fully connected: each chunk crosses ONE link, no ring hops),
                        the rank sums its `world` copies in fp32 (grove_colsum_f32), rounds ONCE, and an all-gather distributes the
                        shard sums — the same bytes per rank as reduce-scatter + all-gather, one rounding instead of world - 1.
    `sparse_rows(...)`: the embedding table's gradient has at most B * L non-zero rows per rank (the text tokens of the batch) out of
    32 K — it travels as an all-gather of (row ids, rows) and is summed in fp32 into the dense slice by every rank
    (<= 512 rows x 8 KB per rank instead of 262 MB; SURVEY.md section 8(a) a7 "sparse rows"). The optimizer stays dense.
    On CPU tensors (gloo: the tests) the same code runs inline without streams."""

    # CUs left to RCCL while gradient buckets are in flight (the rule behind `grove_gemm_set_persistent_blocks`):
    # a persistent GEMM block owns a whole CU (128 KB of LDS, 8 waves x 256 VGPRs), so a channel block of an RCCL kernel can only
    # start on a CU that a GEMM block has left, and the NEXT persistent launch then finds that CU taken: the block dealt to it waits
    # for another block's whole share — every GEMM launch overlapping the collective takes up to two rounds instead of one. With the
    # grid capped at CUs - RESERVED_CUS both fit (the GEMMs lose RESERVED_CUS / CUs = 6 % for the few ms a bucket is in flight). 16 =
    # the channel cap init_distributed() gives RCCL (NCCL_MAX_NCHANNELS, one block per channel): 16 CUs stream ~0.4 TB/s of HBM
    # (23 GB/s per CU, measured on the decode kernels), above what seven xGMI links take (7 x 50 GB/s per direction).
    RESERVED_CUS = 16

    def __init__(self, flat, world, bucket_elems, comm_dtype=torch.bfloat16, mode="allreduce", comm_stream=None):
        assert mode in ("allreduce", "rs_ag", "a2a_f32")
        if mode == "a2a_f32" and comm_dtype != torch.bfloat16 and flat.is_cuda:
            # (grove_colsum_f32 reads bf16 rows; an fp32 wire summed by a plain all-reduce IS fp32 accumulation already)
            raise ValueError("exchange mode a2a_f32 = fp32 accumulation of a BF16 wire; with comm_dtype float32 use 'allreduce' or 'rs_ag'")
        self.flat, self.world, self.mode = flat, world, mode
        self.wire = torch.empty(flat.numel(), dtype=comm_dtype, device=flat.device) if comm_dtype != torch.float32 else None
        # buckets are multiples of the world size so that reduce-scatter shards are equal
        self.bucket = max(world, bucket_elems // world * world)
        self.stream = comm_stream
        self.pending = []     # [lo, hi) ranges already handed to the communication stream this step
        self.sparse_done = [] # [lo, hi) ranges whose SUM already sits in `flat` (sparse_rows): not widened from the wire
        self.handles = []
        self._keep = []       # staging tensors of collectives in flight (freed at finish)
        self._kmax = None
        self._inflight = []   # what the CU reservation waits for: Work handles / events of the buckets handed to RCCL
        self._user_blocks = None
        # OPT-IN (round 5): no 8-GPU A/B exists for this rule, so the default is 0 (the GEMMs keep every CU). Ask for it with
        # GROVE_RCCL_RESERVED_CUS=16 or `engine.exchange.reserve_cus = 16`; `bench.py --gpus N` measures both settings in its
        # calibration pass before the timed region and runs the faster one (the line prints every arm).
        self.reserve_cus = int(os.environ.get("GROVE_RCCL_RESERVED_CUS", "0")) if flat.is_cuda else 0
        self.reserved_launch_polls = 0  # (diagnostic: how many GEMM launches ran under the cap in the last step)

    # ---- CU reservation for the collectives in flight
    def _reserve(self):
        if not self.flat.is_cuda or self.reserve_cus <= 0 or self._user_blocks is not None:
            return
        cus = torch.cuda.get_device_properties(self.flat.device).multi_processor_count
        self._user_blocks = ops.gemm_set_persistent_blocks(max(cus - self.reserve_cus, 8))
        ops._pre_gemm_hook = self._poll

    def _release(self):
        if self._user_blocks is not None:
            ops.gemm_set_persistent_blocks(self._user_blocks)
            self._user_blocks = None
        if ops._pre_gemm_hook == self._poll:
            ops._pre_gemm_hook = None
        self._inflight = []

    def _poll(self):
        """Before a persistent-GEMM launch while reserved: lift the cap once the host sees every handed-over collective complete."""
        self.reserved_launch_polls += 1
        for w in self._inflight:
            done = w.is_completed() if hasattr(w, "is_completed") else w.query()
            if not done:
                return
        self._release()

    def _mark_inflight(self, n_handles_before):
        """Called after collectives were queued on the communication stream: reserve CUs until they have completed."""
        if not self.flat.is_cuda or self.reserve_cus <= 0:
            return
        new = self.handles[n_handles_before:]
        if new:
            self._inflight.extend(new)
        else:  # collectives issued without a handle (stream-ordered on the communication stream): an event behind them
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._inflight.append(ev)
        self._reserve()

    def _sum_copies(self, recv, out):
        """out (wire dtype) [k] = round(sum over the `world` rows of recv [world, k]) with the sum in fp32."""
        if recv.is_cuda:
            acc = ops.colsum(recv)                       # fp32 [k]
            if out.dtype == torch.float32:
                out.copy_(acc)
            else:
                ops.to_bf16(acc, out=out)
        else:
            out.copy_(recv.float().sum(0).to(out.dtype))

    def _exchange(self, buf):
        n = buf.numel()
        for s0 in range(0, n, self.bucket):
            b = buf[s0:s0 + self.bucket]
            k = b.numel() // self.world
            # shards: equal, and (a2a_f32: grove_colsum_f32 works on bf16 PAIRS) even — a ragged tail bucket takes the plain all-reduce;
            # the test depends on sizes only, so every rank takes the same branch
            if self.mode in ("rs_ag", "a2a_f32") and b.numel() % self.world == 0 and (self.mode == "rs_ag" or k % 2 == 0):
                r = dist.get_rank()
                shard = b[r * k:(r + 1) * k]  # in place: RCCL reduces into / gathers from the rank's own slice of the bucket
                if self.mode == "a2a_f32":
                    recv = torch.empty((self.world, k), dtype=b.dtype, device=b.device)
                    dist.all_to_all_single(recv.view(-1), b)  # (stream-ordered on the communication stream; blocking on gloo)
                    self._sum_copies(recv, shard)
                    if b.is_cuda:
                        self._keep.append(recv)
                        self.handles.append(dist.all_gather_into_tensor(b, shard, async_op=True))
                    else:
                        dist.all_gather_into_tensor(b, shard.clone())
                elif b.is_cuda:  # stream-ordered on the communication stream
                    self.handles.append(dist.reduce_scatter_tensor(shard, b, op=dist.ReduceOp.SUM, async_op=True))
                    self.handles.append(dist.all_gather_into_tensor(b, shard, async_op=True))
                else:          # gloo (tests): no in-place aliasing, no ordering between queued operations
                    tmp = torch.empty_like(shard)
                    dist.reduce_scatter_tensor(tmp, b.clone(), op=dist.ReduceOp.SUM)
                    dist.all_gather_into_tensor(b, tmp)
            else:  # ragged tail of a group (not a multiple of the world size): plain all-reduce
                self.handles.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True))

    def ready(self, lo, hi, producer_stream=None, event=None):
        """Gradients of flat[lo:hi] are final once `producer_stream` reaches this point (or once `event`, recorded earlier by the
        caller, has passed): start their exchange. Collectives are issued in call order — identical on every rank."""
        if hi <= lo:
            return
        src = self.flat[lo:hi]
        if not self.flat.is_cuda:
            buf = src
            if self.wire is not None:
                self.wire[lo:hi].copy_(src)
                buf = self.wire[lo:hi]
            self._exchange(buf)
            self.pending.append((lo, hi))
            return
        ev = event
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(producer_stream if producer_stream is not None else torch.cuda.current_stream(self.flat.device))
        n0 = len(self.handles)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            buf = src
            if self.wire is not None:
                buf = self.wire[lo:hi]
                ops.to_bf16(src, out=buf)
            self._exchange(buf)
            self._mark_inflight(n0)
        self.pending.append((lo, hi))

    # ---- sparse rows (embed_tokens)
    def sparse_begin(self, count):
        """Forward time: this rank will contribute `count` distinct rows. The padded row count every rank uses is the MAX over
        ranks; its tiny all-reduce is issued now (first in the communication queue of the step) and read at the end of the
        backward, when it has long completed — no host stall in the step."""
        t = torch.tensor([int(count)], dtype=torch.int32)
        if not self.flat.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self._kmax = (t, None)
            return
        if getattr(self, "_kmax_host", None) is None:  # one pinned word and one device word for the life of the exchange
            self._kmax_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self._kmax_dev = torch.empty(1, dtype=torch.int32, device=self.flat.device)
        host, td = self._kmax_host, self._kmax_dev
        with torch.cuda.stream(self.stream):
            td.fill_(int(count))
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            host.copy_(td, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._kmax = (host, ev, td)

    def sparse_kmax(self):
        assert self._kmax is not None, "sparse_begin() was not called in this step's forward"
        if self._kmax[1] is not None:
            self._kmax[1].synchronize()
        return int(self._kmax[0][0])

    def sparse_rows(self, ids, rows, lo, hi, ld, producer_stream=None, event=None):
        """ids int32 [K] (distinct row numbers of this rank, -1 = padding), rows [K, ld] fp32 (this rank's gradient rows in the
        order of ids; padding rows ignored), K = the same on every rank (sparse_kmax()). The dense slice flat[lo:hi] viewed as
        [*, ld] must be ZERO on entry; on return (stream-ordered) it holds the sum over all ranks. All-gather of ids and of the rows
        in the wire dtype; the sum itself is fp32, added RANK BY RANK in rank order (one scatter-add per rank's block: ids are distinct
        inside a block, so nothing collides inside a launch, and the launches are stream-ordered) — a row touched by three or more
        ranks is summed in the same order on every replica, so the replicas' embed_tokens gradients stay bit-identical (one
        atomic scatter-add over all blocks summed such rows in a rank-dependent order)."""
        K = ids.numel()
        dense = self.flat[lo:hi].view(-1, ld)
        wdt = self.wire.dtype if self.wire is not None else torch.float32
        if not self.flat.is_cuda:
            wr = rows.to(wdt)
            all_ids = torch.empty(self.world * K, dtype=torch.int32)
            all_rows = torch.empty((self.world * K, ld), dtype=wdt)
            dist.all_gather_into_tensor(all_ids, ids)
            dist.all_gather_into_tensor(all_rows.view(-1), wr.reshape(-1))
            for r in range(self.world):
                blk_ids, blk = all_ids[r * K:(r + 1) * K], all_rows[r * K:(r + 1) * K]
                keep = blk_ids >= 0
                dense.index_add_(0, blk_ids[keep].long(), blk[keep].float())
        else:
            ev = event
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(producer_stream if producer_stream is not None else torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                wr = rows if wdt == torch.float32 else ops.to_bf16(rows)
                all_ids = torch.empty(self.world * K, dtype=torch.int32, device=ids.device)
                all_rows = torch.empty((self.world * K, ld), dtype=wdt, device=ids.device)
                dist.all_gather_into_tensor(all_ids, ids)
                dist.all_gather_into_tensor(all_rows.view(-1), wr.view(-1))
                for r in range(self.world):
                    blk_ids, blk = all_ids[r * K:(r + 1) * K], all_rows[r * K:(r + 1) * K]
                    if wdt == torch.float32:
                        ops.scatter_add_rows_f32(blk, dense, blk_ids)
                    else:
                        ops.scatter_add_f32(blk, dense, blk_ids, K, ld)
                for t in (ids, rows, wr, all_ids, all_rows):
                    t.record_stream(self.stream)
                self._mark_inflight(len(self.handles))
        self.pending.append((lo, hi))
        self.sparse_done.append((lo, hi))
        self._kmax = None

    def finish(self):
        """Everything not handed over by ready() is exchanged now; then wait and widen. Returns the exchanged ranges."""
        covered = sorted(self.pending)
        pos, gaps = 0, []
        for lo, hi in covered:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.flat.numel():
            gaps.append((pos, self.flat.numel()))
        for lo, hi in gaps:
            self.ready(lo, hi)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.flat.is_cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
            self._release()  # the compute stream now waits for the collectives: whatever is queued next runs after them
        self._keep = []
        if self.wire is not None:
            dense = [r for r in self.pending if r not in self.sparse_done]
            if not self.sparse_done:
                dense = [(0, self.flat.numel())]
            for lo, hi in dense:  # (a sparse slice already holds its fp32 sum: the wire never carried it)
                if self.flat.is_cuda:
                    ops.to_f32(self.wire[lo:hi], out=self.flat[lo:hi])
                else:
                    self.flat[lo:hi].copy_(self.wire[lo:hi])
        done, self.pending, self.sparse_done = self.pending, [], []
        return done


EXCHANGE_ARMS = [(m, r) for m in ("allreduce", "rs_ag", "a2a_f32") for r in (0, GradExchange.RESERVED_CUS)]


def calibrate_exchange(engine, step_fn, steps=3, arms=None, barrier=None, on_arm=None, done_arms=None):
    """The first N > 1 run on real hardware is ONE shot with default arguments, so that run measures its own
    choices: every arm {exchange mode} x {CUs reserved for RCCL while buckets are in flight} is driven for `steps` steps of the real
    workload (`step_fn` = engine(**batch) / backward / step) between two barriers, timed by the wall clock (max over ranks) with the
    exposed-communication events of its last step. Returns (table, best) — `table` one dict per arm in the order run, `best` the arm
    with the smallest ms/step — and leaves the engine on `best` unless `GROVE_EXCHANGE_KEEP=1`. Every rank runs the same arms in the
    same order (the collectives of an arm must match across ranks); the decision is taken on the all-reduced maxima, so every rank
    decides alike. RCCL's channel count is NOT an arm: NCCL_MAX_NCHANNELS is read once when the communicator is created.
    `on_arm(mode, reserve)` is called before an arm starts and finished arms are appended to the list `done_arms` (bench.py's stall
    watchdog reports where a wedged run stood)."""
    import time as _time
    ex = engine.exchange
    if ex is None:
        return [], None
    arms = list(arms if arms is not None else EXCHANGE_ARMS)
    dev = engine.dev
    on_gpu = torch.device(dev).type == "cuda"
    sync = (lambda: torch.cuda.synchronize(dev)) if on_gpu else (lambda: None)
    bar = barrier if barrier is not None else (dist.barrier if (dist.is_initialized() and dist.get_world_size() > 1) else (lambda: None))
    start = (ex.mode, ex.reserve_cus)
    table = []
    for mode, reserve in arms:
        if mode == "a2a_f32" and ex.wire is None:  # (fp32 wire: the all-to-all arm is defined for a bf16 wire only)
            continue
        ex.mode, ex.reserve_cus = mode, (reserve if on_gpu else 0)
        if on_arm is not None:
            on_arm(mode, reserve)
        step_fn()  # one untimed step per arm: RCCL builds its plan for a new collective / message size on first use
        sync()
        bar()
        t0 = _time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync()
        bar()
        dt = (_time.perf_counter() - t0) / steps * 1e3
        exposed = engine.exposed_comm_ms() if on_gpu else None
        vals = torch.tensor([dt, exposed if exposed is not None else 0.0], dtype=torch.float64, device=dev if on_gpu else "cpu")
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        table.append({"exchange": mode, "reserved_cus": ex.reserve_cus, "ms_per_step": round(float(vals[0]), 3),
                      "exposed_comm_ms": (round(float(vals[1]), 3) if exposed is not None else None),
                      "gemm_launches_under_cap": ex.reserved_launch_polls})
        ex.reserved_launch_polls = 0
        if done_arms is not None:
            done_arms.append(f"{mode}/{ex.reserve_cus}: {table[-1]['ms_per_step']} ms")
    best = min(table, key=lambda a: a["ms_per_step"]) if table else None
    if best is not None and os.environ.get("GROVE_EXCHANGE_KEEP") != "1":
        ex.mode, ex.reserve_cus = best["exchange"], best["reserved_cus"]
    else:
        ex.mode, ex.reserve_cus = start
    return table, best


def shard_clips(n_clips, rank, world):
    """DistributedSampler partition of train.py:453 (no shuffle, padded by wrap-around): clip indices of `rank`."""
    per = (n_clips + world - 1) // world
    idx = list(range(n_clips)) + list(range(per * world - n_clips))
    return idx[rank:per * world:world]


class GroveEngine:
    """Replica-per-GPU data-parallel engine with the DeepSpeed-engine surface train.py relies on."""

    def __init__(self, model: GROVEForCausalLM, args, total_steps=None, bucket_bytes=128 << 20, comm_dtype=torch.bfloat16,
                 exchange="allreduce", overlap=True, sparse_embed=True, force_exchange=False):
        """force_exchange: build and run the gradient exchange on a ONE-rank group too (every collective of the N > 1 step is then
        issued against the real backend with identity results: tests/test_train_gpu.py drives RCCL this way on a one-GPU box)."""
        self.module = model
        self.args = args
        self.dev = model.dev
        torch.cuda.set_device(self.dev)  # what deepspeed.initialize() did for the reference: this rank's launches go to ITS GPU
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.names = model.trainable
        g = model._flat_grad
        self.master = torch.zeros_like(g)
        self.m = torch.zeros_like(g)
        self.v = torch.zeros_like(g)
        self.slices = []
        for n in self.names:
            off = model._grad_off[n]
            k = model._sd[n].numel()
            w = model._sd[n]
            if n.endswith("conv3d.weight"):
                w = w.permute(0, 2, 3, 4, 1)  # the contiguous tap-major storage behind the canonical view
            assert w.is_contiguous(), n
            ops.to_f32(w.reshape(-1), out=self.master[off:off + k])
            self.slices.append((n, off, k, w))
        # device tables of the multi-tensor optimizer step (the slices keep their tensors alive, so the pointers stay valid)
        self._seg_off = torch.tensor([off for _, off, _, _ in self.slices], dtype=torch.int64).to(g.device)
        self._seg_len = torch.tensor([k for _, _, k, _ in self.slices], dtype=torch.int64).to(g.device)
        self._seg_ptr = torch.tensor([w.data_ptr() for _, _, _, w in self.slices], dtype=torch.int64).to(g.device)
        total = total_steps if total_steps is not None else args.epochs * args.steps_per_epoch
        self.scheduler = WarmupDecayLR(args.lr, total, 100)
        self.global_step = 0
        self.micro = 0
        self.clip = 1.0  # "gradient_clipping": 1.0 (train.py:475)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=g.device)
        self._norm = torch.zeros(1, dtype=torch.float32, device=g.device)
        # gradient exchange: bf16 on the wire like DeepSpeed under bf16 (engine.communication_data_type) or torch.float32; bucketed,
        # launched from inside the backward as each parameter group's gradients complete (GradExchange)
        comm = self.world > 1 or (force_exchange and dist.is_initialized())
        self.comm_stream = torch.cuda.Stream(device=self.dev) if comm else None
        self.exchange = None
        if comm:
            self.exchange = GradExchange(g, self.world, bucket_bytes // (2 if comm_dtype == torch.bfloat16 else 4), comm_dtype, exchange,
                                         self.comm_stream)
        self.overlap = overlap
        # embed_tokens' gradient as touched rows (all-gather of (ids, rows), fp32 sum) instead of the dense 131 M-element slice
        self.sparse_embed = sparse_embed
        self.exposed_comm_events = None  # (start, end) events around the wait for the collectives in the last step (N > 1)
        self.opt_stream = torch.cuda.Stream(device=self.dev)
        self.overlap_optimizer = True  # False: the compute stream waits for the update inside step() (A/B arm)
        self._grads_cleared = False
        self.training = True
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Replicas start from rank 0's values (DeepSpeed broadcasts the module's parameters at initialize() and after a
        checkpoint load): the fp32 master copy and the bf16 trainable tensors; frozen tensors come from the same checkpoint /
        initialiser on every rank and are not sent."""
        if self.exchange is None:
            return
        dist.broadcast(self.master, src=0)
        for _, _, _, w in self.slices:
            dist.broadcast(w, src=0)
        # values derived from trainable tensors at build time (the SAM adapters' fp32 alpha scalars) follow the new weights;
        # every other trainable tensor is read in place
        self.module.sam.refresh_adapter_scalars()

    # ---- DeepSpeed-engine surface
    def __call__(self, **batch):
        if self.micro == 0 and not self._grads_cleared:
            self.module.zero_grad()
        self._grads_cleared = False
        last_micro = self.micro + 1 >= self.args.grad_accumulation_steps
        # (with gradient accumulation the touched rows are a union over micro-steps: the dense exchange serves that case)
        self.module._sparse_embed = self.exchange if (self.exchange is not None and self.sparse_embed and last_micro and
                                                      self.args.grad_accumulation_steps == 1 and self.module._train_mode) else None
        return self.module(**batch)

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def backward(self, loss):
        last_micro = self.micro + 1 >= self.args.grad_accumulation_steps
        # on the micro-step that completes the accumulation the model reports every finished gradient group to the exchange
        self.module._grad_ready_cb = self.exchange.ready if (self.exchange is not None and self.overlap and last_micro) else None
        try:
            self.module.backward(loss)
        except BaseException:
            # a backward that raises never reaches step(): without this the persistent-GEMM grid cap and the pre-launch poll hook that
            # GradExchange installed for the buckets already in flight would stay on for the rest of the process
            if self.exchange is not None:
                self.exchange._release()
            raise
        finally:
            self.module._grad_ready_cb = None
        self.micro += 1

    def _allreduce(self):
        """Wait for (and, for groups not handed over from inside the backward, start) the gradient collectives. The two events
        bracket what the compute stream WAITS here = the exposed communication of the step (bench.py prints it for N > 1)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.exchanged_ranges = self.exchange.finish()
        e1.record()
        self.exposed_comm_events = (e0, e1)

    def exposed_comm_ms(self):
        """Device time the compute stream spent waiting for gradient collectives (+ widening the wire buffer) in the last step."""
        if self.exposed_comm_events is None:
            return None
        self.exposed_comm_events[1].synchronize()
        return self.exposed_comm_events[0].elapsed_time(self.exposed_comm_events[1])

    def step(self):
        a = self.args
        if self.micro < a.grad_accumulation_steps:
            return
        self.micro = 0
        if self.exchange is not None:
            self._allreduce()
        g = self.module._flat_grad
        scale = 1.0 / (self.world * a.grad_accumulation_steps)
        self.global_step += 1
        lr = self.scheduler.for_update(self.global_step)
        # The update runs on its own stream. It is HBM-bound (5.8 GB of optimizer state, ~3 ms) and touches only the trainable
        # tensors, while the next step starts with the frozen CLIP tower and SAM blocks 0-7: the model's forward waits for
        # `weights_ready` exactly where it first reads a trainable tensor (projector after the CLIP tower, first SAM adapter), so
        # the update hides under the next step's first GEMMs. The gradient buffer is cleared on the same stream, behind the update.
        main = torch.cuda.current_stream(self.dev)
        self.opt_stream.wait_stream(main)
        with torch.cuda.stream(self.opt_stream):
            # global-norm clipping at 1.0: the sum of squares stays on the device and the AdamW kernel derives the clip factor from
            # it (no host read-back in the step: the host keeps queueing the next step's launches while this one runs)
            self._sumsq.zero_()
            ops.sumsq(g, out=self._sumsq)
            # one multi-tensor launch (DeepSpeed's FusedAdam does the same): 112 per-tensor launches left 0.9 ms of gaps per step
            ops.adamw_step_multi(self.master, g, self.m, self.v, self._seg_off, self._seg_len, self._seg_ptr, lr, a.beta1, a.beta2, 1e-8,
                                 a.wd, scale, self.global_step, sumsq=self._sumsq, clip=self.clip, norm_out=self._norm)
            self.module.sam.refresh_adapter_scalars()
            g.zero_()
            self._grads_cleared = True
        ev = torch.cuda.Event()
        ev.record(self.opt_stream)
        self.module.set_weights_event(ev)
        if not self.overlap_optimizer:
            main.wait_event(ev)

    @property
    def last_grad_norm(self):
        """Pre-clip global gradient norm of the last step (a device scalar; reading it synchronises)."""
        self.opt_stream.synchronize()
        return float(self._norm[0])

    def save_checkpoint(self, save_dir, tag=None, consolidated=True):
        """Engine state for resume (`<tag>.pt` + `latest`, like DeepSpeed's tag directory) AND, beside it, the consolidated fp32
        `pytorch_model.bin` under the reference's key names — what `zero_to_fp32.py ./ pytorch_model.bin` makes out of a DeepSpeed
        checkpoint (infer_eval_iground.sh:11-15), so the reference's inference scripts read this directory without that step."""
        self.module.wait_weights()  # the last update may still be running on the optimizer stream
        if self.rank == 0:
            os.makedirs(save_dir, exist_ok=True)
            tag = tag or f"global_step{self.global_step}"
            torch.save({"module": {k: v.cpu() for k, v in self.module.state_dict().items()}, "master": self.master.cpu(),
                        "exp_avg": self.m.cpu(), "exp_avg_sq": self.v.cpu(), "global_step": self.global_step,
                        "trainable": list(self.names), "world": self.world,
                        # tensors of the reference checkpoint for modules off this path (checkpoint.load_grove_weights keeps them on the model):
                        # saved with the engine state so that a RESUMED run still writes the reference's full key set
                        "passthrough": dict(getattr(self.module, "_passthrough", None) or {})},
                       os.path.join(save_dir, tag + ".pt"))
            if consolidated:
                from .checkpoint import save_grove_weights
                save_grove_weights(self.module, os.path.join(save_dir, "pytorch_model.bin"), engine=self)
            with open(os.path.join(save_dir, "latest"), "w") as f:
                f.write(tag)
        if dist.is_initialized():
            dist.barrier()

    def load_checkpoint(self, load_dir):
        """Resume: every rank reads the file (one node, page cache), validates it against THIS engine's layout, and the replicas
        are re-synchronised from rank 0 afterwards."""
        with open(os.path.join(load_dir, "latest")) as f:
            tag = f.readlines()[0].strip()
        ck = torch.load(os.path.join(load_dir, tag + ".pt"), map_location="cpu", weights_only=True)
        if ck["master"].numel() != self.master.numel() or list(ck.get("trainable", self.names)) != list(self.names):
            raise RuntimeError(f"{load_dir}/{tag}.pt was written for another trainable set / architecture "
                               f"({ck['master'].numel()} vs {self.master.numel()} optimizer elements)")
        self.module.wait_weights()
        self.module.load_state_dict(ck["module"])
        self.master.copy_(ck["master"])
        self.m.copy_(ck["exp_avg"])
        self.v.copy_(ck["exp_avg_sq"])
        self.global_step = int(ck["global_step"])
        if ck.get("passthrough") and not getattr(self.module, "_passthrough", None):
            self.module._passthrough = dict(ck["passthrough"])
        self.broadcast_parameters()
        if dist.is_initialized():
            dist.barrier()
        return load_dir, {"global_step": self.global_step}


class AverageMeter:
    """utils/utils.py:22-77 (sum/count meter; all_reduce folds every meter of a log step into ONE collective
    in train() below instead of one blocking all-reduce per meter)."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


LOSS_KEYS = ("loss", "ce_loss", "giou_loss", "l1_loss", "temp_objectness_loss")


class ScalarWriter:
    """Stand-in for tensorboard's SummaryWriter (train.py:113-121; tensorboard is not installed): `add_scalar` appends one JSON
    line to <log_dir>/scalars.jsonl."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")

    def add_scalar(self, tag, value, step):
        import json
        with open(self.path, "a") as fh:
            fh.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")


def initialize_environment(args):
    """train.py:113-121: log directory <log_base_dir>/<exp_name> (or --log_dir) and, on rank 0, the scalar writer."""
    if not getattr(args, "log_dir", None):
        args.log_dir = os.path.join(args.log_base_dir, args.exp_name)
    if not dist.is_initialized() or dist.get_rank() == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter
            return SummaryWriter(args.log_dir)
        except Exception:
            return ScalarWriter(args.log_dir)
    return None


def _to_device_batch(batch, dev):
    """dict_to_cuda + `.bfloat16()` of the two image tensors (train.py:751-753); tensors already in HBM pass through."""
    out = dict(batch)
    for k, v in batch.items():
        if torch.is_tensor(v) and v.device != dev:
            out[k] = v.to(dev, non_blocking=True)
    for k in ("global_enc_images", "grounding_enc_images"):
        if k in out and torch.is_tensor(out[k]) and out[k].dtype != torch.bfloat16:
            out[k] = out[k].bfloat16()
    return out


def train(data_loader, model, epoch, *rest, log=None):
    """train.py:704-793, the hot loop, under BOTH call forms:
      reference: train(data_loader, model_engine, epoch, scheduler, writer, dataset_iter, args, logger) -> dataset_iter
                 (the iterator restarts from data_loader on StopIteration, train.py:707-713; lr goes to writer as "train/lr");
      short:     train(data_iter, engine, epoch, args, log=print) -> data_iter  (an endless iterator: the synthetic loader).
    Per step: grad_accumulation_steps micro-batches of engine(**batch) / engine.backward(loss) / engine.step(); every print_freq
    steps the meters of all ranks are folded in ONE all-reduce (the reference issues one blocking all-reduce per meter,
    utils/utils.py:72) and rank 0 logs the reference's line."""
    engine = model
    if len(rest) == 5:
        scheduler, writer, dataset_iter, args, logger = rest
        say = logger.info if logger is not None and hasattr(logger, "info") else (logger or log or print)
    else:
        args = rest[0]
        say = rest[1] if len(rest) > 1 else (log or print)
        scheduler, writer, dataset_iter, data_loader = engine.scheduler, None, data_loader, None

    def next_input(it):
        try:
            return next(it), it
        except StopIteration:
            if data_loader is None:
                raise
            it = iter(data_loader)
            return next(it), it

    trackers = {k: AverageMeter(k) for k in LOSS_KEYS}
    batch_time = AverageMeter("Time")
    engine.train()
    # Loss terms are read back ONE micro-step late (round 5): the reference's `loss.item()`-style update of the
    # meters right after the forward (train.py:733-742) is a stream sync BETWEEN forward and backward — the host would queue the
    # backward's ~900 launches onto an idle GPU. Here the terms of micro-step k are copied into a pinned slot behind the forward
    # (async D2H + an event) and folded into the meters while micro-step k + 1 runs; the slots still pending are drained before the
    # meters are printed / all-reduced, so every logged average covers exactly the steps the reference's would.
    on_gpu = torch.device(engine.dev).type == "cuda"
    ring = [torch.empty(len(LOSS_KEYS), dtype=torch.float32, pin_memory=True) for _ in range(2)] if on_gpu else None
    pending = []  # (slot tensor, event, keys present)

    def drain(keep=0):
        while len(pending) > keep:
            slot, ev, keys = pending.pop(0)
            if ev is not None:
                ev.synchronize()
            for k, v in zip(keys, slot[:len(keys)].tolist()):
                trackers[k].update(v, 1)

    micro = 0
    end = time.time()
    for global_step in range(args.steps_per_epoch):
        for _ in range(args.grad_accumulation_steps):
            batch, dataset_iter = next_input(dataset_iter)
            out = engine(**_to_device_batch(batch, engine.dev))
            keys = [k for k in trackers if k in out]
            vals = torch.stack([out[k].float() for k in keys])
            if on_gpu:
                drain(keep=1)  # the slot about to be reused belongs to micro-step k - 2: long finished
                slot = ring[micro % 2]
                slot[:len(keys)].copy_(vals, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((slot, ev, keys))
            else:
                pending.append((vals.cpu(), None, keys))
            micro += 1
            engine.backward(out["loss"])
            engine.step()
            drain(keep=1 if on_gpu else 0)  # micro-step k - 1's terms: their event completed during this micro-step's forward
        batch_time.update(time.time() - end)
        end = time.time()
        if global_step % args.print_freq == 0:
            drain()
            if engine.world > 1:
                t = torch.tensor([x for tr in trackers.values() for x in (tr.sum, tr.count)], dtype=torch.float32, device=engine.dev)
                dist.all_reduce(t)
                t = t.tolist()
                for i, tr in enumerate(trackers.values()):
                    tr.sum, tr.count = t[2 * i], t[2 * i + 1]
                    tr.avg = tr.sum / (tr.count + 1e-5)
            if engine.rank == 0:
                say(f"Epoch: [{epoch}][{global_step + 1}/{args.steps_per_epoch}] time {batch_time.avg:.3f} " +
                    " ".join(f"{k} {tr.avg:.4f}" for k, tr in trackers.items()))
                if writer is not None:
                    for k, tr in trackers.items():
                        writer.add_scalar(f"train/{k}", tr.avg, global_step)
                    writer.add_scalar("metrics/total_secs_per_batch", batch_time.avg, global_step)
            for tr in trackers.values():
                tr.reset()
        if global_step != 0 and writer is not None and engine.rank == 0:
            writer.add_scalar("train/lr", scheduler.get_last_lr()[0], global_step)
    drain()
    return dataset_iter


@torch.no_grad()
def validate_model_performance(validation_loader, training_model, *rest):
    """train.py:796-916, the live loss-validation branch (876-916; the bbox branch cannot run in the reference, quirk Q8), under
    both call forms:
      reference: validate_model_performance(val_loader, model_engine, epoch, writer, args) -> avg_val_loss, the mean of the
                 giou, l1 (and, for HowToGround, temp_objectness) meters (train.py:905-908); every batch of the loader is used;
      short:     validate_model_performance(val_iter, engine, n_batches, args) -> {loss key: mean} incl. "val_loss" = the
                 reference's figure (an endless iterator: n_batches are drawn)."""
    engine = training_model
    ref_form = len(rest) == 3
    if ref_form:
        epoch, writer, args = rest
        n_batches = None
    else:
        n_batches, args = rest
        epoch, writer = 0, None
    if getattr(args, "bbox_validation", False):
        check_supported(args)
    use_obj = engine.module.config.use_temp_objectness
    keys = [k for k in LOSS_KEYS if use_obj or k != "temp_objectness_loss"]
    meters = {k: AverageMeter(k) for k in keys}
    engine.eval()
    was = engine.module._train_mode
    engine.module._train_mode = False  # loss only: no tape, no saved activations, no dlogits
    try:
        it = iter(validation_loader)
        i = 0
        while n_batches is None or i < n_batches:
            try:
                batch = next(it)
            except StopIteration:
                break
            out = engine.module(**_to_device_batch(batch, engine.dev))
            for k, m in meters.items():
                if k in out:
                    m.update(float(out[k]), 1)
            i += 1
    finally:
        engine.module._ctx = None
        engine.module._train_mode = was
    if engine.world > 1:  # the reference all-reduces every meter (utils/utils.py:72); here all of them in ONE collective
        t = torch.tensor([x for m in meters.values() for x in (m.sum, m.count)], dtype=torch.float32, device=engine.dev)
        dist.all_reduce(t)
        t = t.tolist()
        for i, m in enumerate(meters.values()):
            m.sum, m.count = t[2 * i], t[2 * i + 1]
            m.avg = m.sum / max(m.count, 1e-5)
    parts = ["giou_loss", "l1_loss"] + (["temp_objectness_loss"] if use_obj else [])
    avg_val_loss = sum(meters[k].avg for k in parts) / len(parts)      # train.py:905-908
    if writer is not None and engine.rank == 0:
        for k, m in meters.items():
            writer.add_scalar(f"val/{k}", m.avg, epoch)
    if ref_form:
        return avg_val_loss
    res = {k: m.avg for k, m in meters.items()}
    res["val_loss"] = avg_val_loss
    return res


def save_checkpoint(model_engine, *rest):
    """train.py:685-701: only improving checkpoints are kept, in <log_dir>/ckpt_model_best, with the reference's marker file
    `epoch_<e>_val_<metric>_<value>.pth`. Call forms: the reference's (model_engine, tokenizer, args, epoch, metric_name,
    metric_value, is_best) and the short one without the tokenizer."""
    if len(rest) == 6:
        _tokenizer, args, epoch, metric_name, metric_value, is_best = rest
    else:
        args, epoch, metric_name, metric_value, is_best = rest
    if not is_best:
        return None
    save_dir = os.path.join(args.log_dir, "ckpt_model_best")
    if model_engine.rank == 0:
        os.makedirs(save_dir, exist_ok=True)
        torch.save({"epoch": epoch, f"val_{metric_name}": metric_value}, os.path.join(save_dir, f"epoch_{epoch}_val_{metric_name}_{metric_value}.pth"))
    if dist.is_initialized():
        dist.barrier()
    model_engine.save_checkpoint(save_dir)
    return save_dir


def synthetic_loader(dims, args, device, rank, world, seed0=0):
    """Stand-in for the DataLoader over HowToGround / iGround with a DistributedSampler (train.py:440-463): an endless iterator of
    collate dicts (dataset/dataset.py:64-70) whose clip indices are sharded over the ranks like `shard_clips`. Datasets, video
    decoding and the tokenizer are out of scope; the content is synthetic, the shapes are the real ones."""
    from .synthetic import synthetic_batch
    step = 0
    while True:
        clip0 = (step * world + rank) * args.batch_size
        b = synthetic_batch(dims, B=args.batch_size, T=args.num_frames, L=args.text_len, n_det=args.n_det, seed=seed0 + clip0,
                            device=device, dtype=torch.bfloat16)
        yield b.as_kwargs(inference=False)
        step += 1


def resume_training_from_checkpoint(engine, args, log=print):
    """train.py:489-500: --auto_resume picks `<log_dir>/ckpt_model` (the reference's name), else `ckpt_model_last_epoch` /
    `ckpt_model_best` (the directories save_checkpoint writes, train.py:688-690) when present, --resume names a directory; the
    epoch to continue from is derived from the restored step count (the reference parses it out of the `latest` tag)."""
    path = args.resume
    if not path and args.auto_resume:
        for name in ("ckpt_model", "ckpt_model_last_epoch", "ckpt_model_best"):
            cand = os.path.join(args.log_dir, name)
            if os.path.exists(os.path.join(cand, "latest")):
                path = cand
                break
    if path:
        engine.load_checkpoint(path)
        args.start_epoch = engine.global_step // max(args.steps_per_epoch, 1)
        log(f"Resume training from {path}, start from epoch {args.start_epoch}")


def init_distributed(local_rank, timeout_s=1800, backend=None):
    """deepspeed.init_distributed() (train.py:932): one process per GPU, RCCL ("nccl" IS RCCL on ROCm) bound to this rank's device,
    with a FINITE timeout so that a wedged collective ends the job with RCCL's own report of the stuck operation instead of hanging
    it (the reference's inference scripts pass 2-4 h, infer_iground.py:495). RCCL's channel count is left to RCCL (round 5: the
    cap of 16 that rounds 3-4 set by default could throttle an 8-GPU xGMI all-reduce and was never measured; the variable is read
    once at communicator creation, so it cannot be an arm of a calibration inside one run — set NCCL_MAX_NCHANNELS=16 in the
    environment, together with GROVE_RCCL_RESERVED_CUS=16, to get the rounds 3-4 behaviour)."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("GROVE_BACKEND", "nccl")  # gloo only for one-GPU rehearsals
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return world


def main(args, dims=None, log=print):
    """train.py:609-680 for the hot path, in the reference's order: tokenizer -> initialize_model -> prepare_model_for_training ->
    interpolate_positional_embeddings -> (optional --grove_weights, non-strict) -> engine -> resume -> epochs of train() + loss
    validation + keep-the-best checkpointing. One process per GPU (launch with `python -m torch.distributed.run --nproc-per-node N
    -m grove_amd.train ...`, or one plain process for N = 1). Datasets are out of scope: the loaders are synthetic."""
    from .synthetic import FULL, TINY
    check_supported(args)
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    world = init_distributed(args.local_rank)
    rank = dist.get_rank() if dist.is_initialized() else 0
    tiny = dims is not None or args.dims == "tiny"
    if dims is None:
        dims = FULL if args.dims == "full" else TINY
    tokenizer = setup_tokenizer_and_special_tokens(args)
    if tiny:  # tiny test geometry: its own small vocabulary / [DET] id (no tokenizer is that small)
        args.det_token_idx = dims.det_token_idx
    if args.grove_weights:
        from .checkpoint import dims_from_checkpoint, load_grove_weights, read_state_dict
        sd = read_state_dict(args.grove_weights)
        dims = dims_from_checkpoint(args.grove_weights, sd, base=dims)
        if not tiny and args.det_token_idx >= dims.vocab:
            # the reference resizes the embeddings to len(tokenizer) before it loads (train.py:330), so a GROVE checkpoint always holds a
            # row for [DET]; a table too small for the tokenizer's [DET] id is not such a checkpoint — aliasing another token's row would
            # train silently on the wrong embedding
            raise ValueError(f"--grove_weights {args.grove_weights}: its embedding table has {dims.vocab} rows but the tokenizer places [DET] at id "
                             f"{args.det_token_idx}; the checkpoint was not written after the tokenizer's special tokens were added (train.py:330)")
        model = GROVEForCausalLM(dims=dims, device=device, state_dict=None, train=True, det_token_idx=args.det_token_idx,
                                 num_frames=args.num_frames, out_dim=args.out_dim, ce_loss_weight=args.ce_loss_weight,
                                 giou_loss_weight=args.giou_loss_weight, temp_objectness_loss_weight=args.temp_objectness_loss_weight,
                                 train_mask_decoder=args.train_mask_decoder, use_temp_objectness=args.dataset == "HowToGround")
    else:
        model = initialize_model(args, None if tiny else tokenizer, dims=dims, device=device)
    prepare_model_for_training(model, None if (tiny or args.grove_weights) else tokenizer, args)
    interpolate_positional_embeddings(model)
    if args.grove_weights:
        log(f"Fine-tuning using GROVE weights from {args.grove_weights}.")
        rep = load_grove_weights(model, args.grove_weights, sd=sd, log=log if rank == 0 else (lambda m: None))
        if rank == 0:
            log(f"missing keys: {len(rep.missing_keys)} ({len(rep.initialised)} trainable ones initialised like the reference's "
                f"constructors, {len(rep.missing_frozen)} frozen left zero), unexpected keys: {len(rep.unexpected_keys)}")
        del sd
    engine = GroveEngine(model, args, exchange=args.exchange, overlap=not args.no_comm_overlap, sparse_embed=not args.dense_embed)
    resume_training_from_checkpoint(engine, args, log)
    train_iter = synthetic_loader(model.dims, args, device, rank, world, seed0=0)
    val_iter = synthetic_loader(model.dims, args, device, rank, world, seed0=10 ** 6)
    writer = initialize_environment(args) if rank == 0 else None
    if args.eval_only:
        val = validate_model_performance(val_iter, engine, args.val_batches, args)
        if rank == 0:
            log(f"Validation: {val}")
        return val
    best_val_loss = float("inf")
    val = None
    for epoch in range(args.start_epoch, args.epochs):
        train_iter = train(None, engine, epoch, engine.scheduler, writer, train_iter, args, log)
        if dist.is_initialized():
            dist.barrier()
        val = validate_model_performance(val_iter, engine, args.val_batches, args)
        cur = val["val_loss"]
        is_best = cur < best_val_loss
        best_val_loss = min(cur, best_val_loss)
        if rank == 0:
            log(f"Epoch: {epoch}, Current Validation Loss: {cur:.4f}, Best Validation Loss: {best_val_loss:}")
        save_checkpoint(engine, tokenizer, args, epoch, "loss", f"{cur:.4f}", is_best)
    return {"best_val_loss": best_val_loss, "last_val": val, "global_step": engine.global_step}


def set_seed(seed):
    """train.py:918-927."""
    import random
    import numpy as np
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


if __name__ == "__main__":
    import sys
    _args = parse_args(sys.argv[1:])
    _args.local_rank = int(os.environ.get("LOCAL_RANK", _args.local_rank))
    set_seed(42)
    main(_args)
    if dist.is_initialized():
        dist.destroy_process_group()
