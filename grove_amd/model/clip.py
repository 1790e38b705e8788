"""CLIP ViT-L/14-336 frame encoder with spatio-temporal conv adapters on the HIP kernels.

Host-side mirror of model/llava/model/multimodal_encoder/{modeling_clip,clip_encoder,pooling}.py.
The tower runs under @torch.no_grad in the reference (clip_encoder.py:55), so there is no backward.
Data layout in HBM: tokens [F, 577, C] bf16 row-major (F = B*T frames, CLS at token 0); fused qkv
activations [F*577, 3C]; the patch stem is an im2col + GEMM whose epilogue adds the position
embedding and scatters straight into rows 1..576 of each frame.
"""
import os

import torch

from .. import ops
from .attention import attention_fwd
from .indexing import conv3d_gather_index, frame_rows_index

V = "model.vision_tower.vision_tower.vision_model."


class ClipTower:
    def __init__(self, sd, d, device, fp32_stream=True, fp8=False):
        self.d = d
        self.dev = device
        self.fp32_stream = fp32_stream and os.environ.get("GROVE_CLIP_STREAM", "fp32") != "bf16"
        C, P = d.clip_dim, d.clip_patch
        bf = torch.bfloat16
        self.kpad = ops.pad_to(3 * P * P, 32)
        w = sd[V + "embeddings.patch_embedding.weight"].reshape(C, 3 * P * P)
        self.w_patch = torch.zeros((C, self.kpad), dtype=bf, device=device)
        self.w_patch[:, :3 * P * P] = w
        pos = sd[V + "embeddings.position_embedding.weight"]
        self.pos_patch = pos[1:].contiguous()
        self.cls_row = (sd[V + "embeddings.class_embedding"].float() + pos[0].float()).to(bf).reshape(1, C).contiguous()
        self.pre_ln = (sd[V + "pre_layrnorm.weight"], sd[V + "pre_layrnorm.bias"])
        self.layers = []
        for i in range(d.clip_layers - 1):  # layer clip_layers-1 is never consumed (hidden_states[-2])
            p = V + f"encoder.layers.{i}."
            L = {
                "ln1": (sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"]),
                "ln2": (sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"]),
                "wqkv": torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).contiguous(),
                "bqkv": torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0).contiguous(),
                "wo": sd[p + "self_attn.out_proj.weight"], "bo": sd[p + "self_attn.out_proj.bias"],
                "w1": sd[p + "mlp.fc1.weight"], "b1": sd[p + "mlp.fc1.bias"],
                "w2": sd[p + "mlp.fc2.weight"], "b2": sd[p + "mlp.fc2.bias"],
            }
            if fp8:  # e4m3 copies of the four projections (per-output-channel scales); K must fit the fp8 GEMM's 128-byte K tile
                for k in ("wqkv", "wo", "w1", "w2"):
                    if L[k].shape[1] % 128 == 0:
                        L[k + "_q"] = ops.quant_fp8_rows(L[k].contiguous())
            self.layers.append(L)
        self.adapters = []
        for j in range(d.clip_layers // 3):
            p = V + f"encoder.adapters.{j}."
            alpha = sd[p + "alpha"]
            A = {"alpha": alpha.float().contiguous(), "active": bool((alpha.float() != 0).any().item())}
            if A["active"]:
                # Conv3d weight [Co, Ci, kt, kh, kw] -> implicit-GEMM B operand [Co, (kt kh kw), Ci]
                A["w"] = sd[p + "conv3d.weight"].permute(0, 2, 3, 4, 1).reshape(C, 27 * C).contiguous()
                A["b"] = sd[p + "conv3d.bias"]
            self.adapters.append(A)
        # round 6: active adapters (alpha != 0: trained checkpoints) in Winograd F(2x2x2, 3x3x3) form like SAM's (csrc/winograd.hip): the
        # 16 x 36 grid of a frame's 576 patch rows behind its CLS row, groups of 8 frames; the tower is frozen, so the transformed weights
        # are made once. GROVE_CLIP_WINOGRAD=0: the 27-tap implicit GEMM.
        self.wino = os.environ.get("GROVE_CLIP_WINOGRAD", "1") != "0" and C % 64 == 0 and (d.clip_tokens - 1) % 32 == 0
        self._idx = {}

    def _indices(self, F):
        if F not in self._idx:
            n = self.d.clip_tokens - 1
            patch_rows = frame_rows_index(F, n, n + 1, 1).to(self.dev)
            pos_rows = (torch.arange(F * n, dtype=torch.int32) % n).to(self.dev)
            cls_dst = (torch.arange(F, dtype=torch.int32) * (n + 1)).to(self.dev)
            cls_src = torch.zeros(F, dtype=torch.int32, device=self.dev)
            # adapter: '(b t) (h w) c -> b c t h w' with t=8, h=16 => w = n/16 (modeling_clip.py:604)
            conv = conv3d_gather_index(F // 8, 8, 16, n // 16, frame_rows=n + 1, row_offset=1).to(self.dev)
            self._idx[F] = (patch_rows, pos_rows, cls_dst, cls_src, conv)
        return self._idx[F]

    @staticmethod
    def _lin(L, k, x, bias, act=ops.ACT_NONE, residual=None, out=None):
        if (k + "_q") in L:
            return ops.linear_fp8(x, L[k + "_q"][0], L[k + "_q"][1], bias, act=act, residual=residual, out=out)
        return ops.linear(x, L[k], bias, act=act, residual=residual, out=out)

    def hidden_states(self, images, upto=None, taps=None):
        """images bf16 [B, 3, T, H, W] -> hidden state after `upto` layers ([F*577, C]).
        taps: optional dict {layer_count: None} filled with copies of intermediate hidden states."""
        d = self.d
        B, _, T, _, _ = images.shape
        F = B * T
        C, n = d.clip_dim, d.clip_tokens - 1
        H, hd = d.clip_heads, d.clip_dim // d.clip_heads
        patch_rows, pos_rows, cls_dst, cls_src, conv_idx = self._indices(F)
        col = ops.im2col_patch(images.contiguous(), d.clip_patch, self.kpad)
        x0 = torch.empty((F * (n + 1), C), dtype=torch.bfloat16, device=self.dev)
        ops.linear(col, self.w_patch, out=x0, c_idx=patch_rows, residual=self.pos_patch, r_idx=pos_rows)
        ops.copy_rows(self.cls_row, x0, F, C, idx_src=cls_src, idx_dst=cls_dst)
        del col
        x, _, _ = ops.layernorm(x0, self.pre_ln[0], self.pre_ln[1], d.clip_eps)
        del x0
        if taps is not None and 0 in taps:
            taps[0] = x.clone()
        if not self.fp32_stream:
            return self._hidden_states_bf16_stream(x, F, upto, taps)
        # The residual stream lives in FP32 (`res`); a branch output `t` (bf16, from the GEMM) is added to it inside the LayerNorm
        # kernel that follows (residual-stream form of grove_layernorm_fwd): 46 residual adds without a bf16 rounding in between.
        res = ops.to_f32(x)
        t = None
        nl = len(self.layers) if upto is None else upto
        h = torch.empty_like(x)
        for i in range(nl):
            L = self.layers[i]
            ops.layernorm(t, L["ln1"][0], L["ln1"][1], d.clip_eps, out=h, res=res)
            qkv = self._lin(L, "wqkv", h, L["bqkv"])
            o, _ = attention_fwd(qkv, F, n + 1, H, hd, 0, C, 2 * C, hd ** -0.5)
            del qkv
            t = self._lin(L, "wo", o, L["bo"])
            ops.layernorm(t, L["ln2"][0], L["ln2"][1], d.clip_eps, out=h, res=res)
            f = self._lin(L, "w1", h, L["b1"], act=ops.ACT_QUICKGELU)
            t = self._lin(L, "w2", f, L["b2"])
            del f, o
            if i % 3 == 0:
                A = self.adapters[i // 3]
                if A["active"]:  # tanh(0) * relu(conv) + x == x exactly, so alpha == 0 skips the conv
                    # the Conv3d reads the stream itself (as the implicit GEMM's gathered A rows): materialise its bf16 rounding, then
                    # the adapter's own contribution tanh(alpha) * relu(conv + b) becomes the next pending branch output (CLS rows: 0)
                    ops.stream_add(res, t, res_bf16=x)
                    t = torch.zeros_like(x)
                    if self.wino:
                        self._adapter_wino(A, x, t, F, n, None)
                    else:
                        ops.linear(x, A["w"], A["b"], act=ops.ACT_RELU, scale_ptr=A["alpha"], scale_tanh=True, a_idx=conv_idx,
                                   a_taps=27, M=F * n, c_idx=patch_rows, out=t)
            if taps is not None and (i + 1) in taps:
                ops.stream_add(res, t, res_bf16=x)
                t = None
                taps[i + 1] = x.clone()
        if t is not None:
            ops.stream_add(res, t, res_bf16=x)
        else:
            ops.stream_add(res, None, res_bf16=x)
        return x

    def _adapter_wino(self, A, x, out, F, n, residual):
        """tanh(alpha) relu(Conv3d(x) + b) (+ residual) on the patch rows of `out` (modeling_clip.py:599-611: CLS split off,
        '(b t) (h w) c -> b c t h w' with t = 8, h = 16, w = n / 16); the CLS rows of `out` are left as they are."""
        if "U" not in A:
            A["U"] = ops.wino3d_transform_weight(A["w"])
        if not hasattr(self, "_wino_ws"):
            self._wino_ws = {}
        # groups of 8 frames are independent: chunks of 8 groups on one persistent pair of temporaries (see SamEncoder._adapter)
        fr = 8 * (n + 1)
        for g0 in range(0, F // 8, 8):
            gc = min(8, F // 8 - g0)
            r0, r1 = g0 * fr, (g0 + gc) * fr
            ops.wino3d_conv(x[r0:r1], A["U"], (gc, 8, 16, n // 16), out[r0:r1], bias=A["b"], act=ops.ACT_RELU, scale_ptr=A["alpha"], scale_tanh=True,
                            residual=residual[r0:r1] if residual is not None else None, frames=(n + 1, 1), ws=self._wino_ws)

    def _hidden_states_bf16_stream(self, x, F, upto, taps):
        """The same layers with the residual stream in bf16 (what the reference stores): the residual add rides in the epilogue of the
        out_proj / fc2 GEMMs. 2.5x further from the fp32 oracle at layer 23 than the fp32 stream (0.9 % vs 0.36 % rms), 6 bytes per
        element and norm less HBM traffic; selected with ClipTower.fp32_stream = False (GROVE_CLIP_STREAM=bf16)."""
        d = self.d
        C, n = d.clip_dim, d.clip_tokens - 1
        H, hd = d.clip_heads, d.clip_dim // d.clip_heads
        patch_rows, pos_rows, cls_dst, cls_src, conv_idx = self._indices(F)
        nl = len(self.layers) if upto is None else upto
        for i in range(nl):
            L = self.layers[i]
            h, _, _ = ops.layernorm(x, L["ln1"][0], L["ln1"][1], d.clip_eps)
            qkv = self._lin(L, "wqkv", h, L["bqkv"])
            o, _ = attention_fwd(qkv, F, n + 1, H, hd, 0, C, 2 * C, hd ** -0.5)
            del qkv
            self._lin(L, "wo", o, L["bo"], residual=x, out=x)
            ops.layernorm(x, L["ln2"][0], L["ln2"][1], d.clip_eps, out=h)
            f = self._lin(L, "w1", h, L["b1"], act=ops.ACT_QUICKGELU)
            self._lin(L, "w2", f, L["b2"], residual=x, out=x)
            del f, h, o
            if i % 3 == 0:
                A = self.adapters[i // 3]
                if A["active"]:
                    y = torch.empty_like(x)
                    if self.wino:
                        self._adapter_wino(A, x, y, F, n, x)
                    else:
                        ops.linear(x, A["w"], A["b"], act=ops.ACT_RELU, scale_ptr=A["alpha"], scale_tanh=True, a_idx=conv_idx,
                                   a_taps=27, M=F * n, c_idx=patch_rows, residual=x, out=y)
                    ops.copy_rows(x, y, F, C, idx_src=cls_dst, idx_dst=cls_dst)
                    x = y
            if taps is not None and (i + 1) in taps:
                taps[i + 1] = x.clone()
        return x

    def forward(self, images, taps=None):
        """CLIPVisionTower.forward: returns pooled tokens [B*T/8, 576, C] and the hidden state
        hidden_states[-2] ([F, 577, C])."""
        B, _, T, _, _ = images.shape
        assert (B * T) % 8 == 0, "adapters and pooling treat frames as groups of 8 (modeling_clip.py:604, pooling.py:6)"
        x = self.hidden_states(images, taps=taps)
        G = B * T // 8
        pooled = ops.clip_pool(x, G)
        return pooled, x.view(B * T, self.d.clip_tokens, self.d.clip_dim)
