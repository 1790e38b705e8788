"""GROVEForCausalLM — the drop-in boundary of the hot path (mirror of model/GROVE.py:101-451).

Same construction kwargs, the same kwargs-dispatching forward(**kwargs) (GROVE.py:138-154), the same
return values and loss-dict keys, and a state dict with the reference's parameter names
(SURVEY.md §8(b)). Every tensor op underneath is a hand-written gfx950 kernel reached through
include/grove_hip.h; this file is host logic (index tables, dispatch, list plumbing) only.

Frames per sequence: the reference is self-consistent only at T = 8 (quirk Q1). For T = 8*G this
implementation defaults to the reference's own sliding-window semantics (infer_iground.py:245-259):
every 8-frame group of a clip is an independent window with the same text. `literal_T=True`
reproduces the reference's row indexing at T != 8 bit-for-bit in structure (sample b reads pooled
feature row b).
"""
import math
from types import SimpleNamespace

import os

import contextlib

import torch

from .. import ops
from ..synthetic import GroveDims, IGNORE_INDEX, IMAGE_TOKEN_INDEX, is_mask_branch, param_shapes
from .clip import ClipTower
from .decoder import BoxDecoder, M_ as DEC_PREFIX
from .llama import LlamaStack
from .sam import SamEncoder, S as SAM_PREFIX
from .tape import Param, Tape, Var

bf = torch.bfloat16


OBJ_HEAD = DEC_PREFIX + "temporal_objectness_head."


def trainable_names(d: GroveDims, train_mask_decoder=True, use_temp_objectness=True):
    """The parameters that receive gradients under the shipped freeze policy
    (train.py::prepare_model_for_training :234-333 with --lora_r 0 --pretrained --train_mask_decoder).
    CLIP adapters are flagged trainable there too but the tower runs under no_grad (clip_encoder.py:55),
    so they never get a gradient and are excluded. Without --train_mask_decoder (train.py:280-283 not taken)
    only the box head and the temporal-objectness head of the decoder train (train.py:284-288). Without use_temp_objectness
    (ANet / VidSTG, train.py:203) the reference's decoder HAS no objectness head (mask_decoder.py:83-87): not trainable, not in any
    state dict."""
    names = []
    heads = (DEC_PREFIX + "bbox_prediction_head.", OBJ_HEAD)
    for n in param_shapes(d):
        if n.startswith(OBJ_HEAD) and not use_temp_objectness:
            continue
        if is_mask_branch(n):  # flagged trainable by --train_mask_decoder, but no gradient reaches them on the box ("query") path
            continue
        if n.startswith(DEC_PREFIX) and not train_mask_decoder and not n.startswith(heads):
            continue
        if n in ("model.embed_tokens.weight", "lm_head.weight") or n.startswith("model.mm_projector.") \
                or n.startswith("model.text_hidden_fcs.") or n.startswith(DEC_PREFIX) \
                or n.startswith(SAM_PREFIX + "adapters."):
            names.append(n)
    return names


class KVCache:
    """`past_key_values` of the cached LM step: per layer one bf16 [B, 2, heads, capacity, head_dim] tensor — plane 0 the rotated
    keys, plane 1 the values of positions 0..length-1, head-major so that the positions of one head are contiguous rows (what
    grove_decode_attn streams); zero beyond `length` (the decode kernel reads ahead of the position and masks). Falsy while
    empty, like the `past_key_values` test in prepare_inputs_for_generation (llava_llama.py:158-159). `rows(layer, lo, hi)` gives
    the position-major [B, hi-lo, 2*hidden] keys | values view HF's cache tensors correspond to."""

    def __init__(self, layers, length=0):
        self.layers, self.length = layers, length

    @property
    def capacity(self):
        return self.layers[0].shape[3] if self.layers else 0

    def rows(self, layer, lo, hi):
        t = self.layers[layer][:, :, :, lo:hi]                      # [B, 2, heads, n, hd]
        B, _, nh, n, hd = t.shape
        return t.permute(0, 3, 1, 2, 4).reshape(B, n, 2 * nh * hd)  # keys | values per position

    def __len__(self):
        return len(self.layers) if self.length > 0 else 0

    def get_seq_length(self, layer_idx=0):
        return self.length

    def reserve(self, n):
        """Room for n positions (grows by reallocation; a captured decode graph keeps the old buffers, so generate() sizes
        the cache for its whole run up front)."""
        if n <= self.capacity:
            return
        cap = max(n, self.capacity + 256)
        for i, t in enumerate(self.layers):
            g = torch.zeros(t.shape[:3] + (cap, t.shape[4]), dtype=t.dtype, device=t.device)
            g[:, :, :, :self.length].copy_(t[:, :, :, :self.length])
            self.layers[i] = g


class LMOutput(SimpleNamespace):
    """Result of lm_forward (the fields of HF's CausalLMOutputWithPast that generation reads). `hidden_states_f32` — this build's extra:
    the final-norm hidden states in fp32 from the fp32 residual stream of an inference model, what evaluate()'s box path reads — is
    produced ON FIRST ACCESS (one more norm launch and an fp32 [B*S, H] tensor): the teacher-forced forward and the use_cache=False
    loop never read it."""

    def __init__(self, f32fn=None, f32shape=None, **kw):
        super().__init__(**kw)
        self._f32fn, self._f32shape, self._f32 = f32fn, f32shape, None

    @property
    def hidden_states_f32(self):
        if self._f32 is None and self._f32fn is not None:
            self._f32 = self._f32fn().view(self._f32shape)
            self._f32fn = None  # (releases the residual stream the closure holds)
        return self._f32


class GROVEForCausalLM(torch.nn.Module):
    def __init__(self, config=None, dims: GroveDims = None, device="cuda", state_dict=None, train=False, **kwargs):
        super().__init__()
        self.det_token_idx = kwargs.pop("det_token_idx")                        # GROVE.py:120 (required)
        self.ce_loss_weight = kwargs.pop("ce_loss_weight", 1.0)                 # GROVE.py:122-125
        self.giou_loss_weight = kwargs.pop("giou_loss_weight", 1.0)
        self.temp_objectness_loss_weight = kwargs.pop("temp_objectness_loss_weight", 1.0)
        d = dims if dims is not None else GroveDims()
        d = d.__class__(**{**d.__dict__, "det_token_idx": self.det_token_idx, "out_dim": kwargs.get("out_dim", d.out_dim)})
        self.dims = d
        self.config = config if config is not None else SimpleNamespace()
        self.config.num_frames = kwargs.get("num_frames", 8) or 8              # GROVE.py:117
        self.config.temp_objectness_threshold = kwargs.get("temp_objectness_threshold", 0.5)
        self.config.use_temp_objectness = kwargs.get("use_temp_objectness", True)
        # (the reference's constructor default is False, GROVE.py:45; every caller passes it from args, and every shipped launch line
        # sets --train_mask_decoder: a directly constructed model gets the shipped policy)
        self.config.train_mask_decoder = kwargs.get("train_mask_decoder", True)
        self.config.bbox_token_idx = kwargs.get("bbox_token_idx", 32002)      # GROVE.py:115 (region prompts: never read on this path)
        self.config.out_dim = d.out_dim
        self.literal_T = kwargs.get("literal_T", False)
        # last LLaMA layer on the rows a training step reads only (labelled + [DET] rows: the answer's tail); False = every row (A/B, tests)
        self.llama_tail = kwargs.get("llama_tail", __import__("os").environ.get("GROVE_LLAMA_TAIL", "1") != "0")  # (env: whole-program A/B arm)
        # dense positional encoding dtype: bf16 reproduces the reference under model.to(bf16) (quirk Q10)
        self.pe_dtype = kwargs.get("pe_dtype", torch.bfloat16)
        self.stream_dtype = kwargs.get("stream_dtype", None)  # None: fp32 for inference models, bf16 for training models
        # "bf16" (default) or "fp8": the linear layers of the CLIP tower and the LLaMA stack on the e4m3 MFMA GEMM (config 5; inference)
        self.gemm_dtype = kwargs.get("gemm_dtype", "bf16")
        # which GEMMs run in e4m3 under gemm_dtype="fp8". Default (round 6): "sam_mlp" — the SAM tower's mlp.lin1 / lin2 only (the policy
        # that keeps the boxes inside 1e-3; see _build_engines). The LLaMA policies ("all", "det16_kv16": [DET] rows + k / v projections in
        # bf16, "..._clip16": the CLIP tower in bf16 too) carry their own 5e-3 .. 1e-2 tolerance and stay as fenced options.
        self.fp8_policy = kwargs.get("fp8_policy", "sam_mlp")
        assert self.fp8_policy in ("sam_mlp", "all", "det16_kv16", "all_clip16", "det16_kv16_clip16"), self.fp8_policy
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("grove_amd runs on MI355X only: there is no CPU path (use oracle/ for a CPU check)")
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        # one process per GPU: the C-ABI launches go to the calling thread's current HIP device / torch's current stream there
        # (DeepSpeed's initialize() did this for the reference, train.py:480-486); ops refuses tensors of any other device
        torch.cuda.set_device(self.dev)
        self._train_mode = train
        self._sd = {}
        self._grad = {}
        self._flat_grad = None
        self._plan_stream = None
        self._sam_stream = None
        self.tower_overlap = True  # SAM tower on its own stream beside CLIP -> LLaMA (forward and backward)
        self._alloc_params(state_dict)
        self._build_engines()
        self._ctx = None
        self._grad_ready_cb = None  # set by GroveEngine: called with (lo, hi, stream) when flat-gradient slice [lo, hi) is final
        self._sparse_embed = None   # set by GroveEngine (N > 1): the GradExchange that moves embed_tokens' gradient as touched rows
        self._weights_event = None  # set by GroveEngine.step: recorded behind the optimizer update on its own stream

    def set_weights_event(self, ev):
        self._weights_event = ev

    def wait_weights(self, stream=None):
        """Make `stream` (default: the current one) wait for the optimizer update that may still be writing the trainable tensors
        (GroveEngine.step runs it on its own stream so that it hides under the next forward's frozen layers). Every first read of a
        trainable tensor in a forward, and every export of the weights, goes through here; a completed event costs nothing."""
        if self._weights_event is not None:
            (stream if stream is not None else torch.cuda.current_stream(self.dev)).wait_event(self._weights_event)

    def _grads_final(self, prefixes, stream=None, event=None):
        """Tell the gradient exchange that every trainable tensor whose name starts with one of `prefixes` has its final gradient
        (a prefix's tensors are contiguous in the flat buffer: trainable_names keeps groups together)."""
        if self._grad_ready_cb is None:
            return
        for pre in prefixes:  # one contiguous range per prefix
            names = [n for n in self.trainable if n.startswith(pre)]
            if names:
                lo = min(self._grad_off[n] for n in names)
                hi = max(self._grad_off[n] + self._grad_slot[n] for n in names)
                self._grad_ready_cb(lo, hi, stream, event)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, dims=None, device="cuda", train=False, torch_dtype=None,
                        low_cpu_mem_usage=None, **kwargs):
        """`GROVEForCausalLM.from_pretrained(path, torch_dtype=bf16, low_cpu_mem_usage=True, **model_args)` (train.py:207-218,
        infer_iground.py:511-528): read a HuggingFace directory / consolidated `pytorch_model.bin` / DeepSpeed module file with
        the reference's key names, take the LLaMA geometry from its `config.json` (vocabulary from the embedding table itself),
        fit SAM's position tables to the 512-pixel geometry (train.py:503-576) and load non-strictly. bf16 only."""
        from ..checkpoint import dims_from_checkpoint, load_grove_weights, read_state_dict
        if torch_dtype not in (None, torch.bfloat16):
            raise ValueError("grove_amd computes in bf16 (precision 'bf16', train.py:58): torch_dtype must be torch.bfloat16")
        sd = read_state_dict(pretrained_model_name_or_path)
        d = dims_from_checkpoint(pretrained_model_name_or_path, sd, base=dims)
        model = cls(*model_args, dims=d, device=device, train=train, **kwargs)
        model.load_report = load_grove_weights(model, pretrained_model_name_or_path, sd=sd)
        return model

    # ------------------------------------------------------------------ parameters / state dict
    def _alloc_params(self, state_dict):
        d = self.dims
        for name, shape in param_shapes(d).items():
            src = state_dict[name] if state_dict is not None and name in state_dict else None
            if name.startswith(SAM_PREFIX + "adapters.") and name.endswith("conv3d.weight"):
                # stored tap-major [Co, kt, kh, kw, Ci]; the canonical Conv3d layout is a permuted VIEW of it
                Co, Ci = shape[0], shape[1]
                packed = torch.zeros((Co, 3, 3, 3, Ci), dtype=bf, device=self.dev)
                if src is not None:
                    packed.copy_(src.to(self.dev).permute(0, 2, 3, 4, 1))
                self._sd[name] = packed.permute(0, 4, 1, 2, 3)
            elif name == "lm_head.weight" and shape[0] % 8:
                # the real vocabulary (32000 + the added special tokens, train.py:132-152, 330) is not a multiple of 8, which the
                # weight-gradient GEMM's output rows must be: the storage (and, below, the gradient slot) carries zero rows up to the
                # next multiple; the parameter everyone sees is the [:V] view of it
                full = torch.zeros((ops.pad_to(shape[0], 8), shape[1]), dtype=bf, device=self.dev)
                if src is not None:
                    full[:shape[0]].copy_(src.to(self.dev))
                self._lm_head_full = full
                self._sd[name] = full[:shape[0]]
            else:
                t = torch.zeros(shape, dtype=bf, device=self.dev)
                if src is not None:
                    t.copy_(src.to(self.dev))
                self._sd[name] = t
        if self._train_mode:
            names = trainable_names(d, self.config.train_mask_decoder, self.config.use_temp_objectness)
            # every parameter starts on a 16-byte boundary of the flat fp32 buffer (vector / atomic epilogues)
            offs, off = {}, 0
            slot = lambda n: (ops.pad_to(self._sd[n].shape[0], 8) * self._sd[n].shape[1] if n == "lm_head.weight" else self._sd[n].numel())
            for n in names:
                offs[n] = off
                off += (slot(n) + 3) // 4 * 4
            self._flat_grad = torch.zeros(off, dtype=torch.float32, device=self.dev)
            self._grad_off = offs
            self._grad_slot = {n: (slot(n) + 3) // 4 * 4 for n in names}  # elements of the flat buffer a parameter owns (padding included)
            for n in names:
                k = self._sd[n].numel()
                shape = tuple(self._sd[n].shape)
                if n.endswith("conv3d.weight"):
                    shape = (shape[0], 27 * shape[1])  # gradient lives in the packed (tap-major) layout
                self._grad[n] = self._flat_grad[offs[n]:offs[n] + k].view(shape)
            if "lm_head.weight" in offs:  # the weight-gradient GEMM's target: all pad_to(V, 8) rows (the pad rows only ever receive zeros)
                n = "lm_head.weight"
                self._lm_head_grad_full = self._flat_grad[offs[n]:offs[n] + slot(n)].view(-1, self._sd[n].shape[1])
            self.trainable = names

    def _visible(self, n):
        """The reference model built with use_temp_objectness=False has no `temporal_objectness_head` (mask_decoder.py:83-87). The
        fused box-head kernel still takes the head's pointers, so the (zero) storage stays in `_sd`, but it is no parameter of the
        model: not in state_dict() / named_parameters() / a consolidated checkpoint (infer_anet.py:556 loads with strict=True)."""
        return self.config.use_temp_objectness or not n.startswith(OBJ_HEAD)

    def state_dict(self, *a, **k):
        self.wait_weights()
        return {n: v for n, v in self._sd.items() if self._visible(n)}

    def load_state_dict(self, sd, strict=False, **k):
        missing = [n for n in self._sd if n not in sd and self._visible(n)]
        unexpected = [n for n in sd if n not in self._sd or not self._visible(n)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing {missing[:3]} unexpected {unexpected[:3]}")
        self.wait_weights()
        for n, v in sd.items():
            if n in self._sd and self._visible(n):
                self._sd[n].copy_(v.to(self.dev))
        self._build_engines()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def named_parameters(self, *a, **k):
        return iter((n, v) for n, v in self._sd.items() if self._visible(n))

    def parameters(self, *a, **k):
        return iter(v for n, v in self._sd.items() if self._visible(n))

    def _build_engines(self):
        d, sd, dev, tr = self.dims, self._sd, self.dev, self._train_mode
        # residual streams of the three towers: FP32 for models built for inference (accuracy: full-depth box L1 vs the fp32 oracle
        # 1.5e-3 -> 6.6e-4 together with the fp32 box path), bf16 — what the reference stores — for models built for training
        # (the losses are insensitive to it; ~6 ms of norm traffic per step). stream_dtype= overrides.
        f32s = (not tr) if self.stream_dtype is None else (self.stream_dtype == torch.float32)
        fp8 = self.gemm_dtype == "fp8"
        if fp8 and tr:
            raise ValueError("gemm_dtype='fp8' is an inference configuration (BASELINE config 5): build the model with train=False")
        # "<policy>_clip16" keeps the CLIP tower's GEMMs in bf16 (round 4: with massive-activation channels in the LLaMA stream the e4m3
        # error of the VISUAL TOKENS is what reaches the boxes — tools/fp8_policy_study.py --outliers, profiles/r04_fp8_outlier_policy_study_*)
        # "sam_mlp" (round 6; the fp8 DEFAULT): only the SAM tower's mlp.lin1 / lin2 in e4m3, CLIP and LLaMA in bf16 — the one policy whose
        # boxes stay inside the bf16 path's 1e-3 (tools/fp8_policy_study.py --sam: 9.0e-4 against 7.1e-4 for bf16)
        sam_mlp = fp8 and self.fp8_policy == "sam_mlp"
        clip16 = self.fp8_policy.endswith("_clip16")
        self.clip = ClipTower(sd, d, dev, fp32_stream=f32s, fp8=fp8 and not clip16 and not sam_mlp)
        self.llama = LlamaStack(sd, d, dev, train=tr, fp32_stream=f32s, fp8=fp8 and not sam_mlp,
                                fp8_policy="det16_kv16" if (sam_mlp or not fp8) else (self.fp8_policy[:-7] if clip16 else self.fp8_policy))
        self.sam = SamEncoder(sd, d, dev, train=tr, grads=self._grad, fp32_stream=f32s, fp8_mlp=sam_mlp)
        self.decoder = BoxDecoder(sd, d, dev, grads=self._grad, pe_dtype=self.pe_dtype)
        # round 6b: the box decoder's ~40 weight gradients and ~40 bias column sums (small, latency-bound launches: 3.6 ms of a serial
        # chain in front of the SAM backward) leave the critical path — a side stream, joined before the group is handed to the exchange /
        # the optimizer (GROVE_DEC_WGRAD_STREAM=0: in line, the A/B arm)
        self.decoder.wgrad_stream = (torch.cuda.Stream(device=self.dev) if (tr and os.environ.get("GROVE_DEC_WGRAD_STREAM", "1") != "0") else None)

    def P(self, name):
        return Param(self._sd[name], self._grad.get(name))

    @contextlib.contextmanager
    def batch_invariant_mode(self, on=True):
        """Inference arithmetic in which a clip's bits do not depend on which — or how many — other clips share its launches (round 6,
        VERDICT r5 next #6: the clip-batched form of infer_iground.py:150-288). Two things vary with the batch otherwise: the
        persistent GEMMs cut a partial last round of tiles into K ranges by the TILE COUNT (a different fp32 sum order for the same
        row), and the cached decode step picks its GEMV kernel and its K / V split count by the number of sequences. Here: whole tiles
        only (every output element is ONE sequential sum over K whatever the tile shape), the matrix-core GEMV for every M, eight
        cache splits per head for every M. Norms, attention and the element-wise kernels work per row / per (sequence, head) already."""
        if not on:
            yield
            return
        prev_sk = ops.gemm_set_stream_k(0)
        prev_bi, self.llama.batch_invariant = self.llama.batch_invariant, True
        try:
            yield
        finally:
            self.llama.batch_invariant = prev_bi
            ops.gemm_set_stream_k(prev_sk)

    # ------------------------------------------------------------------ modes (GROVE.py:138-154)
    def forward(self, **kwargs):
        if "past_key_values" in kwargs:  # GROVE.py:138-140 -> LlavaLlamaForCausalLM.forward (one LM step of generate())
            return self.lm_forward(**kwargs)
        mode = kwargs.get("mode")
        if mode == "encode_images":
            return self.encode_images(kwargs["images"])
        if mode == "get_grounding_encoder_embs":
            return self.get_grounding_encoder_embs(kwargs["images"])
        if mode == "get_dense_pe":
            return self.decoder.pe_nchw
        if mode == "evaluate":
            return self.evaluate(kwargs["image_features"], kwargs.get("image_forward_outs"), kwargs.get("images_dtype"),
                                 kwargs["image_embeddings"], kwargs["input_ids"], kwargs["original_size_list"],
                                 max_tokens_new=kwargs["max_tokens_new"], bboxes=kwargs.get("bboxes"),
                                 token_embeddings=kwargs.get("token_embeddings"), dense_pe=kwargs.get("dense_pe"),
                                 device=kwargs.get("device"), use_cache=kwargs.get("use_cache", True))
        return self.model_forward(**kwargs)

    def encode_images(self, images, tape=None):
        """llava_with_region_arch.py:79-82 -> (image_features [B*T/8, 576, hidden], hidden_states[-2])."""
        pooled, hs = self.clip.forward(images.to(bf))  # frozen tower: may run beside a pending optimizer update
        self.wait_weights()                             # the projector is trainable
        G = pooled.shape[0]
        tp = tape if tape is not None else Tape(enabled=False)
        x = Var(pooled.view(G * 576, -1), needs_grad=False)
        h = tp.linear(x, self.P("model.mm_projector.0.weight"), self.P("model.mm_projector.0.bias"), act=ops.ACT_GELU)
        y = tp.linear(h, self.P("model.mm_projector.2.weight"), self.P("model.mm_projector.2.bias"))
        if tape is not None:
            return y, hs
        return y.data.view(G, 576, -1), SimpleNamespace(hidden_states=(hs,))

    @torch.no_grad()
    def predict_masks(self, image_embeddings, text_embeds, frame_of_instance, input_size=None, original_size=None, multimask_output=False):
        """The SAM mask output north_star names (dormant in the reference: its MaskDecoder is built with decoding_type "query",
        GROVE.py:39-51): `mask_decoder(image_embeddings, dense_pe, sparse=text_embeds, dense=no_mask, multimask_output, reps)` with
        the mask branch (mask_decoder.py:206-227) + `postprocess_masks` (sam.py:137-172). image_embeddings: channels-last rows
        [F, g*g, 256] (as model_forward(inference=True) returns) or NCHW [F, 256, g, g]; text_embeds [N, 256] = the [DET] embeddings
        (text_hidden_fcs output); frame_of_instance int32 [N]. Returns dict(low_res_masks [N, C, 4g, 4g], iou_predictions [N, C],
        boxes, objectness, and masks [N, C, H, W] logits when the sizes are given; binarise with > 0)."""
        d = self.dims
        g2 = d.sam_grid ** 2
        self.wait_weights()
        emb = image_embeddings
        if emb.dim() == 4:  # NCHW -> channels-last rows
            F = emb.shape[0]
            rows = torch.empty((F, g2, d.sam_out), dtype=bf, device=self.dev)
            ops.transpose(emb.contiguous(), d.sam_out, g2, g2, rows, d.sam_out, batch=(F, 1), s_in=(d.sam_out * g2, 0), s_out=(g2 * d.sam_out, 0))
            emb = rows
        box, obj, low, iou = self.decoder.forward_f32(emb.reshape(-1, d.sam_out), text_embeds, frame_of_instance.to(self.dev).to(torch.int32),
                                                      want_masks=True, multimask_output=multimask_output)
        out = {"low_res_masks": low, "iou_predictions": iou, "boxes": box, "objectness": obj}
        if input_size is not None and original_size is not None:
            out["masks"] = self.decoder.postprocess_masks(low, input_size, original_size)
        return out

    def get_grounding_encoder_embs(self, images):
        """GROVE.py:134-136 -> [B*T, 256, g, g] (NCHW like the reference; internally channels-last)."""
        rows, _ = self.sam.forward(images.to(bf), before_adapters=self.wait_weights)
        F, g = rows.shape[0], self.dims.sam_grid
        out = torch.empty((F, rows.shape[2], g * g), dtype=bf, device=self.dev)
        ops.transpose(rows, g * g, rows.shape[2], rows.shape[2], out, g * g, batch=(F, 1), s_in=(g * g * rows.shape[2], 0),
                      s_out=(rows.shape[2] * g * g, 0))
        return out.view(F, rows.shape[2], g, g)

    # ------------------------------------------------------------------ host-side index construction
    def _splice_plan(self, input_ids, labels, attention_masks, feat_row_of_seq):
        """llava_with_region_arch.py:127-438 as index tables: for every row of the padded [B, S] embedding
        matrix, which embedding-table row or which projected visual token it is."""
        ids = input_ids.cpu()
        B, L = ids.shape
        lab = labels.cpu() if labels is not None else None
        am = attention_masks.cpu() if attention_masks is not None else None
        seq_tok, seq_lab, seq_mask, vis_pos = [], [], [], []
        for b in range(B):
            row = ids[b]
            pos = (row == IMAGE_TOKEN_INDEX).nonzero().flatten()
            if pos.numel() == 0:
                seq_tok.append(row.clone())
                seq_lab.append(lab[b].clone() if lab is not None else None)
                seq_mask.append(am[b].clone() if am is not None else None)
                vis_pos.append(-1)
                continue
            s = int(pos[0])
            seq_tok.append(torch.cat([row[:s], torch.full((576,), -1, dtype=row.dtype), row[s + 1:]]))
            if lab is not None:
                seq_lab.append(torch.cat([lab[b, :s], torch.full((576,), IGNORE_INDEX, dtype=lab.dtype), lab[b, s + 1:]]))
            if am is not None:
                seq_mask.append(torch.cat([torch.ones(575, dtype=torch.bool), am[b]]))
            vis_pos.append(s)
        S = max(t.numel() for t in seq_tok)
        tok = torch.full((B, S), -1, dtype=torch.int32)
        new_lab = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64)
        kv = torch.full((B,), S, dtype=torch.int32)
        vis_dst, vis_src = [], []
        for b in range(B):
            n = seq_tok[b].numel()
            tok[b, :n] = seq_tok[b].to(torch.int32)
            if lab is not None:
                new_lab[b, :n] = seq_lab[b]
            if am is not None:
                # right padding => valid length = last True + 1 (mask is a prefix in every caller, dataset.py:22-35)
                m = torch.zeros(S, dtype=torch.bool)
                m[:n] = seq_mask[b]
                nz = m.nonzero().flatten()
                kv[b] = int(nz[-1]) + 1 if nz.numel() else 1
            if vis_pos[b] >= 0:
                vis_dst.append(b * S + vis_pos[b] + torch.arange(576))
                vis_src.append(int(feat_row_of_seq[b]) * 576 + torch.arange(576))
        plan = SimpleNamespace(B=B, S=S, L=L)
        plan.tok = tok.reshape(-1).to(self.dev)
        plan.labels = new_lab if lab is not None else None
        plan.kv_len = kv.to(self.dev) if am is not None else None
        plan.vis_dst = torch.cat(vis_dst).to(torch.int32).to(self.dev) if vis_dst else None
        plan.vis_src = torch.cat(vis_src).to(torch.int32).to(self.dev) if vis_src else None
        return plan

    def _embed(self, plan, feats_rows, token_embeddings=None):
        table = token_embeddings if token_embeddings is not None else self._sd["model.embed_tokens.weight"]
        H = self.dims.hidden
        x = torch.empty((plan.B * plan.S, H), dtype=bf, device=self.dev)
        ops.copy_rows(table, x, plan.B * plan.S, H, idx_src=plan.tok)
        if plan.vis_dst is not None:
            ops.copy_rows(feats_rows, x, plan.vis_dst.numel(), H, idx_src=plan.vis_src, idx_dst=plan.vis_dst)
        return x

    def _det_rows(self, ids_cpu, S, trailing_pad=True):
        """GROVE.py:200-205 / :427-430: hidden row (b, 575 + j) for every j with ids[b, j+1] == [DET]."""
        B = ids_cpu.shape[0]
        rows, counts = [], []
        for b in range(B):
            j = (ids_cpu[b, 1:] == self.det_token_idx).nonzero().flatten()
            rows.append(b * S + 575 + j)
            counts.append(int(j.numel()))
        return torch.cat(rows).to(torch.int32), counts

    # ------------------------------------------------------------------ the per-clip hot path
    def _windows(self, global_enc_images, grounding_enc_images, input_ids, labels, attention_masks, lists):
        """T = 8*G -> B*G independent 8-frame windows (default semantics for T != 8)."""
        B, C, T, H, W = global_enc_images.shape
        G = T // 8
        if G == 1 or self.literal_T:
            return global_enc_images, grounding_enc_images, input_ids, labels, attention_masks, lists, T, 1
        # The images stay [B, C, T, H, W]: both towers work on the frame-major rows (b, t, patch) the patch im2col writes, and window
        # (b, g) = frames 8g .. 8g+7 of clip b is exactly frames 8 (b G + g) .. of that order — regrouping the pixel tensors into
        # [B*G, C, 8, H, W] (two strided copies, 1.6 ms per step) would produce the same rows.
        gi, si = global_enc_images, grounding_enc_images
        rep = lambda t: t.repeat_interleave(G, 0) if t is not None else None  # noqa: E731
        new_lists = []
        for lst in lists:
            if lst is None:
                new_lists.append(None)
            else:
                new_lists.append([lst[b][g * 8:(g + 1) * 8] for b in range(B) for g in range(G)])
        return gi, si, rep(input_ids), rep(labels), rep(attention_masks), new_lists, 8, G

    def _host_plan(self, ids, labs, amask, boxes_l, vis_l, B, Tseq, inference):
        """Host-side bookkeeping of one forward, from the inputs alone (runs under a side stream, see model_forward):
        the splice plan (llava_with_region_arch.py:84-440), the labelled rows of the shifted CE (llava_llama.py:111-125), the
        [DET] rows and the (sequence, frame, det) instance order (GROVE.py:200-205, 248-268), and the ground truth per instance
        (GROVE.py:339-360). Everything the device needs is uploaded here; `device_tensors` lists those uploads."""
        hp = SimpleNamespace()
        hp.tail_start, hp.det_rows_hidden = 0, None
        feat_row = list(range(B))  # image_features[cur_image_idx] (llava_with_region_arch.py:156): row b
        hp.plan = plan = self._splice_plan(ids, None if inference else labs, None if inference else amask, feat_row)
        S = plan.S
        det_rows, hp.counts = self._det_rows(ids.cpu(), S)
        hp.det_rows = det_rows.to(self.dev)
        dev_t = [hp.det_rows] + [t for t in (plan.tok, plan.kv_len, plan.vis_dst, plan.vis_src) if t is not None]
        # instance order = (sequence b, frame t, det k)  (repeat_interleave + boolean gather, GROVE.py:254-257)
        inst_det, inst_frame, frame_ptr, base = [], [], [0], 0
        for b in range(B):
            for t in range(Tseq):
                inst_det += list(range(base, base + hp.counts[b]))
                inst_frame += [b * Tseq + t] * hp.counts[b]
                frame_ptr.append(len(inst_frame))  # (instances come grouped by frame: the decoder's backward sums them per frame)
            base += hp.counts[b]
        hp.N = N = len(inst_det)
        hp.inst_det_t = hp.inst_frame_t = hp.inst_frame_ptr_t = None
        if int(det_rows.numel()):
            hp.inst_det_t = torch.tensor(inst_det, dtype=torch.int32).to(self.dev)
            hp.inst_frame_t = torch.tensor(inst_frame, dtype=torch.int32).to(self.dev)
            hp.inst_frame_ptr_t = torch.tensor(frame_ptr, dtype=torch.int32).to(self.dev)
            dev_t += [hp.inst_det_t, hp.inst_frame_t, hp.inst_frame_ptr_t]
        if not inference:
            lab = plan.labels
            valid = (lab[:, 1:] != IGNORE_INDEX)
            bi, ti = valid.nonzero(as_tuple=True)
            hp.rows = (bi * S + ti).to(torch.int32).to(self.dev)
            hp.tgt = lab[bi, ti + 1].to(torch.int32).to(self.dev)
            # LAST-LAYER TAIL: the only hidden states anything reads in a training step are the labelled rows of the shifted CE and the
            # [DET] rows — all in the answer, behind the visual tokens and the prompt. s0 = the first such position over the batch; when
            # the tail is at most half the sequence, LlamaStack runs its last layer's queries / MLP on positions >= s0 only and returns
            # the compact [B * (S - s0), H] rows; the row tables below index that layout.
            hp.tail_start = 0
            if self.llama_tail and not self.llama.fp32_stream and not self.llama.fp8 and (int(ti.numel()) or int(det_rows.numel())):
                cand = ([int(ti.min())] if int(ti.numel()) else []) + ([int((det_rows % S).min())] if int(det_rows.numel()) else [])
                s0 = min(cand)
                if s0 > 0 and (S - s0) * 2 <= S:
                    Lq = S - s0
                    hp.tail_start = s0
                    hp.rows = (bi * Lq + (ti - s0)).to(torch.int32).to(self.dev)
                    hp.det_rows_hidden = ((det_rows // S) * Lq + (det_rows % S - s0)).to(torch.int32).to(self.dev)
                    dev_t.append(hp.det_rows_hidden)
            # the per-frame GT tensors come to the host in TWO transfers
            v_all = torch.cat([vis_l[b][t].reshape(-1).float() for b in range(B) for t in range(Tseq)]).cpu()
            gb_rows = [boxes_l[b][t].reshape(-1, 4).shape[0] for b in range(B) for t in range(Tseq)]
            gb_all = torch.cat([boxes_l[b][t].reshape(-1, 4).float() for b in range(B) for t in range(Tseq)]).cpu()
            gt = torch.zeros((max(N, 1), 4), dtype=torch.float32)
            vis = torch.zeros((max(N, 1),), dtype=torch.float32)
            n_gt = off = v_off = gb_off = 0
            for b in range(B):
                for t in range(Tseq):
                    nv, ng = vis_l[b][t].numel(), gb_rows[b * Tseq + t]
                    v, gb = v_all[v_off:v_off + nv], gb_all[gb_off:gb_off + ng]
                    v_off, gb_off = v_off + nv, gb_off + ng
                    assert gb.shape[0] == int(v.sum()), "Number of ground truth bboxes and objectness labels do not match"
                    n = hp.counts[b]
                    vis[off:off + n] = v
                    if gb.shape[0]:
                        gt[off:off + n][v.bool()] = gb
                    n_gt += gb.shape[0]
                    off += n
            hp.n_gt = n_gt
            hp.gt, hp.vis = gt.to(self.dev), vis.to(self.dev)
            dev_t += [hp.rows, hp.tgt, hp.gt, hp.vis]
        if self._sparse_embed is not None and not inference and self._train_mode:  # (a loss-only validation forward has no backward to serve)
            # embed_tokens' gradient touches only the text-token rows of this batch: the distinct ids (sorted) and, per hidden row, the
            # position of its id in that list (-1 for visual rows) — the sparse gradient exchange sends (ids, rows) instead of the table
            tok_c = plan.tok.cpu()
            ids_u = torch.unique(tok_c[tok_c >= 0])
            comp = torch.full_like(tok_c, -1)
            comp[tok_c >= 0] = torch.searchsorted(ids_u, tok_c[tok_c >= 0]).to(torch.int32)
            hp.embed_ids, hp.embed_compact = ids_u.to(torch.int32), comp.to(self.dev)
            dev_t.append(hp.embed_compact)
        hp.device_tensors = dev_t
        return hp

    def model_forward(self, global_enc_images, grounding_enc_images, bboxes_region=None, input_ids=None, labels=None,
                      attention_masks=None, offset=None, bboxes_list=None, temp_objectness_labels_list=None,
                      original_size_list=None, inference=False, **kwargs):
        """GROVE.py:156-198. Training returns the loss dict (keys :380-381) and keeps what backward() needs;
        inference returns {"pred_bboxes", "logits_temp_objectness"}."""
        assert bboxes_region is None, "the region encoder is dead on GROVE's path (SURVEY.md §2 row 10)"
        d = self.dims
        B0, T0 = global_enc_images.shape[0], global_enc_images.shape[2]
        (gimg, simg, ids, labs, amask, (boxes_l, vis_l), Tseq, G) = self._windows(
            global_enc_images, grounding_enc_images, input_ids, labels, attention_masks,
            [bboxes_list, temp_objectness_labels_list])
        B = ids.shape[0]
        train = (not inference) and self._train_mode
        tp = Tape(enabled=train)
        # Queue order (the host queues ~1900 launches per step at ~30 us each, so WHEN a tower is queued decides when it can start):
        #   CLIP tower (main stream; needs only the images) -> host planning (side stream; its read-backs wait for the previous
        #   step, by which time the GPU already has the CLIP tower to run) -> splice + LLaMA (main) -> SAM tower (its own stream).
        # CLIP -> LLaMA is the longer chain, so it goes first and the SAM tower fills in beside it: both finish together and
        # overlap for the whole forward (with SAM queued first it finished 40 ms before the LLaMA stack did).
        main = torch.cuda.current_stream(self.dev)
        if self._plan_stream is None:
            self._plan_stream = torch.cuda.Stream(device=self.dev)
        side = self._plan_stream
        start = torch.cuda.Event()
        start.record(main)  # the inputs are complete where the main stream stands now (before this step's kernels)
        side.wait_event(start)
        if not self.tower_overlap:  # serial order (clean per-kernel timings): grounding encoder first, as the reference (GROVE.py:162)
            emb_rows, sam_ctx = self.sam.forward(simg.to(bf), save=train, before_adapters=self.wait_weights)
        # 2. global encoder + projector (GROVE.py:170-176)
        feats, _ = self.encode_images(gimg, tape=tp)
        # 0. everything the host derives from the INPUTS — splice plan, labelled rows, [DET] rows / instances, ground truth —
        # on a side stream while the GPU runs the CLIP tower. Each read-back (.cpu()) and each upload from pageable memory is a
        # stream sync; on the main stream, in the middle of the step, they left the GPU idle 6-9 ms per step (rocprofv3 kernel
        # trace, tools/step_gaps.py); the reference does 2 * B * T of them per step for the ground truth alone (GROVE.py:352-353).
        with torch.cuda.stream(side):
            hp = self._host_plan(ids, labs, amask, boxes_l, vis_l, B, Tseq, inference)
        main.wait_stream(side)
        for t_ in hp.device_tensors:
            t_.record_stream(main)
        if getattr(hp, "embed_ids", None) is not None:
            self._sparse_embed.sparse_begin(int(hp.embed_ids.numel()))  # padded row count = MAX over ranks, resolved by backward time
        plan, S, det_rows, counts = hp.plan, hp.plan.S, hp.det_rows, hp.counts
        # 3. splice, LLaMA (llava_llama.py:88-109)
        self.wait_weights()
        x = self._embed(plan, feats.data)
        hidden, llama_ctx = self.llama.forward(x, plan.B, plan.S, kv_len=plan.kv_len, save=train,
                                               precise_rows=det_rows if self.llama.fp8 else None,
                                               tail_start=hp.tail_start or None)
        tail = hidden.shape[0] != plan.B * plan.S  # the last layer ran on the tail rows only: `hidden` is [B * (S - s0), H]
        assert tail == bool(hp.tail_start), (tail, hp.tail_start)
        det_rows_h = hp.det_rows_hidden if tail else det_rows  # where the [DET] rows sit in `hidden`
        # 1. grounding encoder (GROVE.py:162). It shares nothing with the CLIP -> LLaMA tower until the decoder, so it runs on its
        # own stream: the two kernel sequences interleave on the CUs and fill each other's partial rounds and tails
        # (same-box A/B: -11 ms per step, forward and backward). `tower_overlap = False` serialises them.
        if self.tower_overlap:
            if self._sam_stream is None:
                # (GROVE_SAM_STREAM_PRIORITY: A/B knob — -1 = high, 0 = default; the persistent GEMMs of both towers want every CU, the
                # priority decides whose workgroups are dispatched first when both have some pending)
                self._sam_stream = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("GROVE_SAM_STREAM_PRIORITY", "0")))
            self._sam_stream.wait_event(start)
            with torch.cuda.stream(self._sam_stream):
                emb_rows, sam_ctx = self.sam.forward(simg.to(bf), save=train, before_adapters=self.wait_weights)
        F = emb_rows.shape[0]
        emb_rows2 = emb_rows.view(F * d.sam_grid ** 2, -1)
        if self.tower_overlap:
            # (joining later — the CE head and the [DET] rows' text_hidden_fcs do not read the embeddings — was measured: +0.45 ms, same box,
            # GROVE_LATE_JOIN arm of round 6b: those launches then share the device with the SAM tower's last GEMMs and slow them)
            main.wait_stream(self._sam_stream)
            emb_rows.record_stream(main)  # allocated under the SAM stream, read by the decoder on this one
        H = d.hidden
        out = {}
        ce_state = None
        if not inference:
            # 4. lm_head + shifted CE on the labelled rows only (llava_llama.py:111-125)
            rows, tgt = hp.rows, hp.tgt
            R = int(rows.numel())
            hrows = torch.empty((max(R, 1), H), dtype=bf, device=self.dev)
            ops.copy_rows(hidden, hrows, R, H, idx_src=rows)
            hv = Var(hrows[:R])
            Vp = ops.pad_to(d.vocab, 8)
            logits = torch.empty((R, Vp), dtype=bf, device=self.dev)
            gs = torch.full((1,), self.ce_loss_weight / max(R, 1), dtype=torch.float32, device=self.dev)
            dlogits = torch.zeros_like(logits) if train else None
            if R > 0:
                ops.linear(hv.data, self._sd["lm_head.weight"], out=logits)
                loss_sum = ops.cross_entropy(logits, tgt, d.vocab, dlogits=dlogits, grad_scale=gs)
            else:
                loss_sum = torch.zeros(1, dtype=torch.float32, device=self.dev)
            ce_loss = loss_sum[0] * (self.ce_loss_weight / max(R, 1))
            ce_state = (hv, dlogits, rows, R)
        # 5. [DET] rows -> text_hidden_fcs -> per-frame instances (GROVE.py:248-268)
        n_det = int(det_rows.numel())
        pred_boxes, pred_logits = [], []
        if n_det == 0:
            box = torch.zeros((0, 4), dtype=torch.float32, device=self.dev)
            obj = torch.zeros((0,), dtype=torch.float32, device=self.dev)
            dec_state = None
            N = 0
        else:
            inst_det_t, inst_frame_t, N = hp.inst_det_t, hp.inst_frame_t, hp.N
            if train or not self.decoder.precise:
                drows = torch.empty((n_det, H), dtype=bf, device=self.dev)
                ops.copy_rows(hidden, drows, n_det, H, idx_src=det_rows_h)
                dv = Var(drows)
                h1 = tp.linear(dv, self.P("model.text_hidden_fcs.0.0.weight"), self.P("model.text_hidden_fcs.0.0.bias"), act=ops.ACT_RELU)
                te = tp.linear(h1, self.P("model.text_hidden_fcs.0.2.weight"), self.P("model.text_hidden_fcs.0.2.bias"))
                text = torch.empty((N, d.out_dim), dtype=bf, device=self.dev)
                ops.copy_rows(te.data, text, N, d.out_dim, idx_src=inst_det_t)
            else:
                # no backward to serve: the [DET] rows stay in fp32 from the LLaMA residual stream to the box head — final RMSNorm of
                # the fp32 stream rows, text_hidden_fcs on the exact-fp32 MFMA GEMM, fp32 token side of the decoder (row gathers are
                # index selection)
                dv = te = None
                srows = self.llama.last_stream.index_select(0, det_rows_h.long()).float()  # (already fp32 with the fp32 stream)
                hn = ops.rmsnorm(None, self._sd["model.norm.weight"], d.rms_eps, res=srows, out_dtype=torch.float32)
                h1 = ops.linear_f32(hn, self._sd["model.text_hidden_fcs.0.0.weight"], self._sd["model.text_hidden_fcs.0.0.bias"], act=ops.ACT_RELU)
                te32 = ops.linear_f32(h1, self._sd["model.text_hidden_fcs.0.2.weight"], self._sd["model.text_hidden_fcs.0.2.bias"])
                text = te32.index_select(0, inst_det_t.long())
                self._last_text = text  # the ([DET], frame) embeddings of this forward: predict_masks(...) takes them as prompts
            text_var = Var(text)
            box, obj, dec_state = self.decoder.forward(emb_rows2, text_var, inst_frame_t, train=train, frame_ptr=hp.inst_frame_ptr_t)
        # 6. split per clip / frame (GROVE.py:297-331)
        flat_box, flat_obj = box, obj
        off = 0
        box_c, obj_c = box.cpu() if inference else None, obj.cpu() if inference else None
        thr = self.config.temp_objectness_threshold
        for b in range(B):
            bl, ll = [], []
            for t in range(Tseq):
                n = counts[b]
                if inference:
                    Wd, Hd = original_size_list[b // G]
                    bb = box_c[off:off + n]
                    ub = torch.stack([bb[:, 0] * Wd, bb[:, 1] * Hd, bb[:, 2] * Wd, bb[:, 3] * Hd], -1)
                    xy = torch.stack([ub[:, 0] - ub[:, 2] / 2, ub[:, 1] - ub[:, 3] / 2, ub[:, 0] + ub[:, 2] / 2, ub[:, 1] + ub[:, 3] / 2], -1)
                    lo = obj_c[off:off + n]
                    bl.append(xy[torch.sigmoid(lo) > thr] if self.config.use_temp_objectness else xy)
                    ll.append(lo)
                else:
                    bl.append(box[off:off + n])
                    ll.append(obj[off:off + n])
                off += n
            pred_boxes.append(bl)
            pred_logits.append(ll)
        # regroup windows into clips
        if G > 1:
            pred_boxes = [sum((pred_boxes[b * G + g] for g in range(G)), []) for b in range(B0)]
            pred_logits = [sum((pred_logits[b * G + g] for g in range(G)), []) for b in range(B0)]
        if inference:
            return {"pred_bboxes": pred_boxes, "logits_temp_objectness": pred_logits if self.config.use_temp_objectness else None,
                    "flat_boxes": flat_box, "flat_logits": flat_obj, "hidden": hidden.view(B, S, H),
                    "image_embeddings": emb_rows}
        # 7. losses (GROVE.py:339-381) in fp32 on device (ground truth laid out per instance by _host_plan)
        n_gt = hp.n_gt
        wb = self.giou_loss_weight / (n_gt + 1e-8)
        wo = self.temp_objectness_loss_weight / (N + 1e-8)
        if N > 0:
            sums, dbox, dobj = ops.box_losses(box, obj if self.config.use_temp_objectness else None, hp.gt, hp.vis,
                                              wb, wo, want_grad=train)
        else:
            sums, dbox, dobj = torch.zeros(3, device=self.dev), None, None
        giou, l1, bce = sums[0] * wb, sums[1] * wb, sums[2] * wo
        out = {"loss": ce_loss + giou + l1 + bce, "ce_loss": ce_loss, "giou_loss": giou, "l1_loss": l1,
               "temp_objectness_loss": bce}
        if not self.config.use_temp_objectness:
            out.pop("temp_objectness_loss")
            out["loss"] = ce_loss + giou + l1
        out["flat_boxes"], out["flat_logits"] = flat_box, flat_obj
        if train:
            self._ctx = SimpleNamespace(tp=tp, sam_ctx=sam_ctx, llama_ctx=llama_ctx, plan=plan, feats=feats, ce_state=ce_state,
                                        det=(dv, te, inst_det_t, det_rows_h, n_det) if n_det else None, dec_state=dec_state,
                                        dbox=dbox, dobj=dobj, N=N, F=F, hidden_shape=(hidden.shape[0], H),
                                        embed=(hp.embed_ids, hp.embed_compact) if getattr(hp, "embed_ids", None) is not None else None)
        return out

    # ------------------------------------------------------------------ backward of the last training forward
    def zero_grad(self, set_to_none=False):
        if self._flat_grad is not None:
            self._flat_grad.zero_()

    def backward(self, loss=None):
        """Back-propagates the loss of the last training forward (the reference's model_engine.backward(loss),
        train.py:770). Gradients are ACCUMULATED into the flat fp32 buffer (self._flat_grad)."""
        c = self._ctx
        assert c is not None, "backward() needs a preceding training forward"
        d = self.dims
        H = d.hidden
        g2 = d.sam_grid ** 2
        rowsN, _ = c.hidden_shape
        d_hidden = torch.zeros((rowsN, H), dtype=bf, device=self.dev)
        d_emb = torch.zeros((c.F * g2, d.sam_out), dtype=torch.float32, device=self.dev)
        # decoder + heads
        if c.dec_state is not None:
            self.decoder.backward(c.dec_state, c.dbox, c.dobj, d_emb)
            dv, te, inst_det_t, det_rows, n_det = c.det
            dte = torch.zeros((n_det, d.out_dim), dtype=torch.float32, device=self.dev)
            ops.scatter_add_f32(c.dec_state["text"].grad, dte, inst_det_t, c.N, d.out_dim)
            te.grad = ops.to_bf16(dte)
        dec_side = getattr(self.decoder, "wgrad_stream", None)
        dec_side_done = None
        if dec_side is not None and c.dec_state is not None:
            # the decoder's weight / bias gradients ride a side stream (tape.py): the group is final when that stream has seen both its own
            # launches and the norm / head gradients written on this one
            dec_side.wait_stream(torch.cuda.current_stream(self.dev))
            dec_side_done = torch.cuda.Event()
            dec_side_done.record(dec_side)
            self._grads_final([DEC_PREFIX], event=dec_side_done)
        else:
            self._grads_final([DEC_PREFIX])  # box decoder + heads: first group to finish
        # SAM (adapters' weight gradients, dgrad through blocks 31..8) needs d_emb only: it runs on the SAM stream beside the
        # lm_head / LLaMA / projector backward (disjoint slices of the flat gradient buffer), queued AFTER that longer chain
        main = torch.cuda.current_stream(self.dev)
        dec_done = torch.cuda.Event()
        dec_done.record(main)
        # lm_head: wgrad + dgrad on the labelled rows
        hv, dlogits, rows, R = c.ce_state
        if R > 0:
            Vv = d.vocab
            ops.wgrad(dlogits, hv.data, self._lm_head_grad_full, K=R)  # dlogits' pad columns are zero
            # dgrad: d h[R, H] = dlogits[R, V] . W[V, H]. W is K-major for this product, so it runs as the TN form on the weight
            # as stored — (d h)^T[H, R] = W^T . dlogits^T, K = V split over the chip by the kernel's own split-K — instead of an NT
            # GEMM on a transposed copy of the 262 MB matrix (the copy alone cost more than this whole sequence).
            Rp = ops.pad_to(R, 8)
            dlT = torch.empty((Vv, Rp), dtype=bf, device=self.dev)  # (transpose zero-fills the pad columns)
            ops.transpose(dlogits, R, Vv, dlogits.stride(0), dlT, Rp, pad_to_cols=Rp)
            dhT = torch.zeros((H, Rp), dtype=torch.float32, device=self.dev)
            # (6 K ranges: beyond that the fp32 atomics on the small [H, R] output outweigh the extra blocks — 235 us at the
            # automatic 12, 168 at 6, tools/bench_general_gemm.py)
            ops.wgrad(self._sd["lm_head.weight"], dlT, dhT, K=Vv, split_k=6 if Vv >= 4096 else 0)
            dh = torch.empty((Rp, H), dtype=bf, device=self.dev)
            ops.transpose(ops.to_bf16(dhT), H, Rp, Rp, dh, H)
            ops.copy_rows(dh, d_hidden, R, H, idx_dst=rows, accumulate=True)
        self._grads_final(["lm_head."])
        # text_hidden_fcs (tape) -> d hidden at the DET rows;  projector closures run later with feats.grad set
        c.tp_fns = c.tp.fns
        if c.det is not None:
            dv, te, inst_det_t, det_rows, n_det = c.det
            # run only the text_hidden_fcs closures now (they were pushed last)
            fns = c.tp.fns[-2:]
            c.tp.fns = c.tp.fns[:-2]
            for fn in reversed(fns):
                fn()
            ops.copy_rows(dv.grad, d_hidden, n_det, H, idx_dst=det_rows, accumulate=True)
        self._grads_final(["model.text_hidden_fcs."])
        # LLaMA (dgrad only)
        dx = self.llama.backward(c.llama_ctx, d_hidden)
        # splice backward: visual rows -> projector; text rows -> embed_tokens
        plan = c.plan
        if plan.vis_dst is not None:
            nv = plan.vis_dst.numel()
            dfe = torch.zeros_like(c.feats.data)
            ops.copy_rows(dx, dfe, nv, H, idx_src=plan.vis_dst, idx_dst=plan.vis_src)
            c.feats.grad = dfe
        sparse = None
        if c.embed is not None and self._sparse_embed is not None:
            # N > 1: this rank's rows go into a compact [K, H] fp32 block (K = max distinct rows over the ranks); the exchange
            # all-gathers (ids, rows) and every rank sums all blocks into the (still zero) dense slice — see GradExchange.sparse_rows
            ex = self._sparse_embed
            ids_u, comp = c.embed
            K = max(ex.sparse_kmax(), 1)
            loc = torch.zeros((K, H), dtype=torch.float32, device=self.dev)
            ops.scatter_add_f32(dx, loc, comp, plan.B * plan.S, H)
            ids_pad = torch.full((K,), -1, dtype=torch.int32)
            ids_pad[:ids_u.numel()] = ids_u
            sparse = (ids_pad.to(self.dev), loc)
        else:
            ops.scatter_add_f32(dx, self._grad["model.embed_tokens.weight"], plan.tok, plan.B * plan.S, H)
        c.tp.backward()  # mm_projector
        # embed_tokens + projector are final HERE on the main stream, but their exchange is queued behind the SAM adapters' (below):
        # the communication stream is FIFO, and the adapters finish on the SAM stream while the LLaMA dgrad above is still running —
        # handed over first, this last group would hold them back until the end of the LLaMA sweep
        main_grads_done = None
        if self._grad_ready_cb is not None or sparse is not None:
            main_grads_done = torch.cuda.Event()
            main_grads_done.record(main)

        def last_group():
            if sparse is not None:
                n = "model.embed_tokens.weight"
                lo = self._grad_off[n]
                self._sparse_embed.sparse_rows(sparse[0], sparse[1], lo, lo + self._sd[n].numel(), H, event=main_grads_done)
                self._grads_final(["model.mm_projector."], event=main_grads_done)
            else:
                self._grads_final(["model.embed_tokens.", "model.mm_projector."], event=main_grads_done)

        def adapter_done(j, stream):  # the SAM tower's own stream finishes adapter j's weight / bias / alpha gradients
            self._grads_final([SAM_PREFIX + f"adapters.{j}."], stream)
        if self.tower_overlap and self._sam_stream is not None:
            self._sam_stream.wait_event(dec_done)
            with torch.cuda.stream(self._sam_stream):
                d_emb.record_stream(self._sam_stream)
                self.sam.backward(c.sam_ctx, ops.to_bf16(d_emb), on_adapter_done=lambda j: adapter_done(j, self._sam_stream))
            last_group()
            main.wait_stream(self._sam_stream)
        else:
            self.sam.backward(c.sam_ctx, ops.to_bf16(d_emb), on_adapter_done=lambda j: adapter_done(j, None))
            last_group()
        if dec_side_done is not None:
            main.wait_event(dec_side_done)  # (the optimizer reads the decoder's gradient views on this stream)
        self._ctx = None

    # ------------------------------------------------------------------ generation (GROVE.py:412-451, llava_llama.py:57-180)
    def new_kv_cache(self, B, S_max):
        """An empty `past_key_values` with room for S_max positions."""
        return KVCache(self.llama.new_kv_cache(B, S_max), 0)

    @torch.no_grad()
    def lm_forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
                   output_attentions=None, output_hidden_states=None, images=None, image_features=None, image_forward_outs=None,
                   images_dtype=None, token_embeddings=None, bboxes=None, return_dict=None, last_logits_only=False, **_):
        """`LlavaLlamaForCausalLM.forward` as generation drives it (llava_llama.py:57-141; reached through
        `forward(past_key_values=...)`, GROVE.py:138-140). Two cases, as in `prepare_inputs_for_generation` (:144-180):
          * `past_key_values` empty / None: the prompt step — splice `image_features[b]` at the -200 slot
            (llava_with_region_arch.py:84-440), run the stack over the whole [B, L+575] sequence and, with `use_cache`, leave the
            rotated keys | values of every layer in the returned cache;
          * `past_key_values` holding t > 0 positions: ONE new token per sequence (the last column of `input_ids`) at position t
            through the weight-streaming GEMV path, appended to the cache in place.
        Returns `.logits` ([B, S, V] bf16 for the prompt step — only the last position with `last_logits_only` —, [B, 1, V] fp32
        for a cached step), `.hidden_states` = the final-norm hidden tensor (what the reference returns outside training mode,
        llava_llama.py:130-133) and `.past_key_values`. Un-padded equal-length prompts only (quirk Q9: the reference's
        attention mask is wrong for anything else), so `attention_mask` is not read."""
        self.wait_weights()
        if labels is not None or inputs_embeds is not None or output_attentions:
            raise NotImplementedError("lm_forward serves generation only: the training CE runs inside model_forward; "
                                      "inputs_embeds / attentions are not produced on this path")
        assert bboxes is None or len(bboxes) == 0, "the region encoder is dead on GROVE's path (SURVEY.md section 2 row 10)"
        d = self.dims
        H = d.hidden
        B = input_ids.shape[0]
        lm_head = self._sd["lm_head.weight"]
        embed = token_embeddings if token_embeddings is not None else self._sd["model.embed_tokens.weight"]
        cache = past_key_values if isinstance(past_key_values, KVCache) else None
        if past_key_values is not None and cache is None and len(past_key_values):
            raise TypeError("past_key_values must be the KVCache a previous lm_forward / generate returned")
        if cache is not None and cache.length > 0:
            if B > 8:
                raise NotImplementedError("cached decode serves 1-8 sequences per step (the GEMV path)")
            t = cache.length
            cache.reserve(t + 1)
            nxt = input_ids[:, -1].to(self.dev).to(torch.int32)
            xt = torch.empty((B, H), dtype=bf, device=self.dev)
            ops.copy_rows(embed, xt, B, H, idx_src=nxt)
            hidden, logits = self.llama.decode_step(xt, t, cache.layers, lm_head)
            cache.length = t + 1
            h32 = self.llama.last_decode_hidden_f32
            return SimpleNamespace(loss=None, logits=logits.view(B, 1, -1), past_key_values=cache, hidden_states=hidden.view(B, 1, H),
                                   hidden_states_f32=h32.view(B, 1, H) if h32 is not None else None, attentions=None)
        if image_features is None and images is not None:
            image_features, image_forward_outs = self.encode_images(images)
        plan = self._splice_plan(input_ids, None, None, list(range(B)))
        feats = image_features.reshape(-1, image_features.shape[-1]) if image_features is not None else None
        x = self._embed(plan, feats, token_embeddings)
        S = plan.S
        if use_cache:
            if cache is None:
                cache = self.new_kv_cache(B, S + 64)
            cache.reserve(S)
        else:
            cache = None
        h0, _ = self.llama.forward(x, B, S, kv_cache=cache.layers if cache is not None else None)
        if cache is not None:
            cache.length = S
        if last_logits_only:
            last = torch.empty((B, H), dtype=bf, device=self.dev)
            ops.copy_rows(h0, last, B, H, idx_src=(torch.arange(B, dtype=torch.int32, device=self.dev) * S + S - 1))
            logits = (ops.gemv(last, lm_head, out_dtype=torch.float32) if B <= 8
                      else ops.linear(last, lm_head, out_dtype=torch.float32)).view(B, 1, -1)
        else:
            logits = ops.linear(h0, lm_head).view(B, S, -1)
        # (extra to HF: the same hidden states in fp32, normalised from the fp32 residual stream of an inference model — evaluate()'s box path)
        return LMOutput(f32fn=getattr(self.llama, "_final_norm_f32", None), f32shape=(B, S, H), loss=None, logits=logits,
                        past_key_values=cache, hidden_states=h0.view(B, S, H), attentions=None)

    @torch.no_grad()
    def generate(self, images=None, input_ids=None, bboxes=None, image_features=None, image_forward_outs=None, images_dtype=None,
                 token_embeddings=None, max_new_tokens=32, num_beams=1, output_hidden_states=False, return_dict_in_generate=False,
                 do_sample=False, use_cache=True, synced_gpus=False, eos_token_id=None, pad_token_id=None, use_graph=True, **kwargs):
        """HF `GenerationMixin.generate` in the one configuration GROVE ever uses (GROVE.py:418-422; infer_iground.py:192):
        greedy (`num_beams=1, do_sample=False`), `use_cache=True`, extra kwargs threaded to every LM step as
        `prepare_inputs_for_generation` does (llava_llama.py:144-180). Rows finish at `eos_token_id` and are padded with
        `pad_token_id` afterwards; the loop stops when every row is finished or after `max_new_tokens`; the last generated
        token is never fed back. With `return_dict_in_generate` the result carries `.sequences` [B, L+new] (still containing
        -200) and, with `output_hidden_states`, `.hidden_states` = one tensor per LM step ([B, L+575, H] for the prompt step,
        [B, 1, H] after), which `evaluate` concatenates on dim 1 (GROVE.py:423-426); otherwise the sequences alone.
        Steps after the first replay ONE captured HIP graph of the cached step (`use_graph`); each of them is exactly
        `forward(past_key_values=cache, input_ids=ids[:, -1:], ...)`. `use_cache=False` recomputes the whole sequence per step
        (second implementation for the tests); its hidden_states are split the same way so that the concatenation is unchanged."""
        if num_beams != 1 or do_sample:
            raise NotImplementedError("only greedy decoding (num_beams=1, do_sample=False) exists on GROVE's path (GROVE.py:418-422)")
        d = self.dims
        H = d.hidden
        eos = d.eos_token_id if eos_token_id is None else eos_token_id
        pad = d.pad_token_id if pad_token_id is None else pad_token_id
        if image_features is None and images is not None:
            image_features, image_forward_outs = self.encode_images(images)
        ids = input_ids.clone()
        B = ids.shape[0]
        finished = torch.zeros(B, dtype=torch.bool, device=ids.device)
        lm_head = self._sd["lm_head.weight"]
        embed = token_embeddings if token_embeddings is not None else self._sd["model.embed_tokens.weight"]

        def pick(logits):
            nxt = logits.reshape(B, -1).argmax(-1).to(ids.device)  # argmax over one [B, V] row block (index selection, not arithmetic)
            return torch.where(finished, torch.full_like(nxt, pad), nxt)

        hiddens32 = []  # parallel to `hiddens` when the model keeps fp32 residual streams (None entries otherwise)

        def result(seqs, hiddens, cache):
            if not return_dict_in_generate:
                return seqs
            ok32 = output_hidden_states and len(hiddens32) == len(hiddens) and all(h is not None for h in hiddens32)
            return SimpleNamespace(sequences=seqs, hidden_states=tuple(hiddens) if output_hidden_states else None,
                                   hidden_states_f32=tuple(hiddens32) if ok32 else None,
                                   past_key_values=cache, scores=None, attentions=None)

        if not use_cache or B > 8:
            hidden, hidden32, S, S0 = None, None, 0, None
            for _ in range(max_new_tokens):
                out = self.lm_forward(input_ids=ids, image_features=image_features, token_embeddings=token_embeddings,
                                      use_cache=False, last_logits_only=True)
                hidden = out.hidden_states
                S = hidden.shape[1]
                S0 = S if S0 is None else S0
                nxt = pick(out.logits)
                ids = torch.cat([ids, nxt[:, None]], 1)
                finished = finished | (nxt == eos)
                if bool(finished.all()):
                    break
            hidden32 = out.hidden_states_f32 if output_hidden_states else None  # (lazy: only the LAST step's rows are ever read)
            if hidden32 is not None:
                hiddens32 += [hidden32[:, :S0]] + [hidden32[:, j:j + 1] for j in range(S0, S)]
            return result(ids, [hidden[:, :S0]] + [hidden[:, j:j + 1] for j in range(S0, S)], None)

        out = self.lm_forward(input_ids=ids, image_features=image_features, token_embeddings=token_embeddings, use_cache=True,
                              past_key_values=self.new_kv_cache(B, ids.shape[1] + 575 + max_new_tokens), last_logits_only=True)
        cache = out.past_key_values
        hiddens = [out.hidden_states]
        hiddens32.append(out.hidden_states_f32)
        logits = out.logits
        if use_graph and max_new_tokens > 1:
            # first token from the prompt step's logits (one host read), every later step on the device: see LlamaStack.greedy_graph
            nxt = pick(logits)
            ids = torch.cat([ids, nxt[:, None]], 1)
            finished = finished | (nxt == eos)
            if bool(finished.all()):
                return result(ids, hiddens, cache)
            n_steps = max_new_tokens - 1
            replay, st = self.llama.greedy_graph(B, cache.layers, embed, lm_head, nxt.to(self.dev), cache.length, n_steps, d.vocab, eos, pad,
                                                 finished.to(self.dev))
            done, CHECK = 0, 8  # the host looks at `finished` every CHECK steps only; steps run past the stop are cut off below
            while done < n_steps:
                for _ in range(min(CHECK, n_steps - done)):
                    replay()
                    done += 1
                if done < n_steps and bool(st["finished"].all()):
                    break
            new_ids = st["ids_out"][:, :done].to(ids.device)
            # HF's stop rule, applied after the fact: the loop ends with the first step after which every row is finished
            fin = finished[:, None].to(new_ids.device) | ((new_ids == eos).cumsum(1) > 0)
            allfin = fin.all(0).nonzero().flatten()
            used = int(allfin[0]) + 1 if allfin.numel() else done
            ids = torch.cat([ids, new_ids[:, :used]], 1)
            hid = st["hid_out"][:used]
            hiddens += [hid[j].view(B, 1, H).clone() for j in range(used)]
            hiddens32 += [st["hid_out_f32"][j].view(B, 1, H).clone() if st["hid_out_f32"] is not None else None for j in range(used)]
            cache.length += used
            return result(ids, hiddens, cache)
        for step in range(max_new_tokens):
            nxt = pick(logits)
            ids = torch.cat([ids, nxt[:, None]], 1)
            finished = finished | (nxt == eos)
            if bool(finished.all()) or step == max_new_tokens - 1:
                break  # the last generated token is never fed back (quirk Q3)
            out = self.forward(past_key_values=cache, input_ids=ids[:, -1:], image_features=image_features,
                               token_embeddings=token_embeddings, use_cache=True)
            logits = out.logits
            hiddens.append(out.hidden_states)
            hiddens32.append(out.hidden_states_f32)
        return result(ids, hiddens, cache)

    def generate_greedy(self, image_features, input_ids, max_new_tokens, token_embeddings=None, eos_token_id=None, pad_token_id=None,
                        use_cache=True, use_graph=True):
        """(sequences, hidden of every fed position [B, L+575+new-1, H]) — `generate` with the per-step hidden states concatenated."""
        out = self.generate(input_ids=input_ids, image_features=image_features, token_embeddings=token_embeddings,
                            max_new_tokens=max_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id, use_cache=use_cache,
                            use_graph=use_graph, output_hidden_states=True, return_dict_in_generate=True)
        return out.sequences, torch.cat(out.hidden_states, 1)

    @torch.no_grad()
    def evaluate(self, image_features, image_forward_outs, images_dtype, image_embeddings, input_ids, orig_sizes,
                 max_tokens_new=32, bboxes=None, token_embeddings=None, dense_pe=None, device=None, use_cache=True):
        d = self.dims
        generation_outputs = self.generate(
            images=None, input_ids=input_ids, bboxes=bboxes, image_features=image_features, image_forward_outs=image_forward_outs,
            images_dtype=images_dtype, token_embeddings=token_embeddings, max_new_tokens=max_tokens_new, num_beams=1,
            output_hidden_states=True, return_dict_in_generate=True, do_sample=False, use_cache=use_cache, synced_gpus=False)
        ids = generation_outputs.sequences
        hidden = torch.cat(generation_outputs.hidden_states, dim=1).contiguous()  # GROVE.py:423-426
        h32 = getattr(generation_outputs, "hidden_states_f32", None)
        hidden32 = torch.cat(h32, dim=1).contiguous() if (h32 is not None and self.decoder.precise) else None
        B, S, H = hidden.shape
        det_rows, counts = self._det_rows(ids.cpu(), S, trailing_pad=False)
        Tseq = self.config.num_frames
        # image_embeddings arrive NCHW [F,256,g,g] (mode get_grounding_encoder_embs) -> channels-last rows
        F = image_embeddings.shape[0]
        g2 = d.sam_grid ** 2
        emb_rows = torch.empty((F, g2, d.sam_out), dtype=bf, device=self.dev)
        ops.transpose(image_embeddings.contiguous(), d.sam_out, g2, g2, emb_rows, d.sam_out, batch=(F, 1), s_in=(d.sam_out * g2, 0),
                      s_out=(g2 * d.sam_out, 0))
        n_det = int(det_rows.numel())
        boxes, logits = [[torch.zeros(0, 4) for _ in range(Tseq)] for _ in range(B)], [[torch.zeros(0) for _ in range(Tseq)] for _ in range(B)]
        if n_det:
            det_rows = det_rows.to(self.dev)
            drows = torch.empty((n_det, H), dtype=bf, device=self.dev)
            ops.copy_rows(hidden.view(B * S, H), drows, n_det, H, idx_src=det_rows)
            if self.decoder.precise:  # the box path in fp32, as model_forward(inference=True) runs it: from the fp32 hidden rows of the
                # fp32 residual streams (prefill AND cached decode steps) when the model keeps them, else from the bf16 rows
                rows32 = hidden32.view(B * S, H).index_select(0, det_rows.long()) if hidden32 is not None else ops.to_f32(drows)
                h1 = ops.linear_f32(rows32, self._sd["model.text_hidden_fcs.0.0.weight"], self._sd["model.text_hidden_fcs.0.0.bias"], act=ops.ACT_RELU)
                te = ops.linear_f32(h1, self._sd["model.text_hidden_fcs.0.2.weight"], self._sd["model.text_hidden_fcs.0.2.bias"])
            else:
                h1 = ops.linear(drows, self._sd["model.text_hidden_fcs.0.0.weight"], self._sd["model.text_hidden_fcs.0.0.bias"], act=ops.ACT_RELU)
                te = ops.linear(h1, self._sd["model.text_hidden_fcs.0.2.weight"], self._sd["model.text_hidden_fcs.0.2.bias"])
            inst_det, inst_frame, base = [], [], 0
            for b in range(B):
                for t in range(Tseq):
                    inst_det += list(range(base, base + counts[b]))
                    inst_frame += [b * Tseq + t] * counts[b]
                base += counts[b]
            N = len(inst_det)
            if te.dtype == torch.float32:
                text = te.index_select(0, torch.tensor(inst_det, dtype=torch.int64, device=self.dev))  # (row selection)
            else:
                text = torch.empty((N, d.out_dim), dtype=bf, device=self.dev)
                ops.copy_rows(te, text, N, d.out_dim, idx_src=torch.tensor(inst_det, dtype=torch.int32, device=self.dev))
            box, obj, _ = self.decoder.forward(emb_rows.view(F * g2, -1), Var(text), torch.tensor(inst_frame, dtype=torch.int32, device=self.dev))
            box_c, obj_c, off = box.cpu(), obj.cpu(), 0
            thr = self.config.temp_objectness_threshold
            for b in range(B):
                for t in range(Tseq):
                    n = counts[b]
                    Wd, Hd = orig_sizes[b]
                    bb = box_c[off:off + n]
                    ub = torch.stack([bb[:, 0] * Wd, bb[:, 1] * Hd, bb[:, 2] * Wd, bb[:, 3] * Hd], -1)
                    xy = torch.stack([ub[:, 0] - ub[:, 2] / 2, ub[:, 1] - ub[:, 3] / 2, ub[:, 0] + ub[:, 2] / 2, ub[:, 1] + ub[:, 3] / 2], -1)
                    lo = obj_c[off:off + n]
                    boxes[b][t] = xy[torch.sigmoid(lo) > thr] if self.config.use_temp_objectness else xy
                    logits[b][t] = lo
                    off += n
        if self.config.use_temp_objectness:
            return ids, boxes, logits
        return ids, boxes
