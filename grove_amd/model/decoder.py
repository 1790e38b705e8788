"""Prompt encoder + two-way transformer + box / temporal-objectness heads on the HIP kernels.

Host-side mirror of model/SAM/modeling/{prompt_encoder,transformer,mask_decoder}.py for the "query"
branch GROVE uses (mask_decoder.py:164-205). All N = sum(reps) (frame, [DET]) instances are batched:
tokens [N*6, 256], image keys [N*g*g, 256] (channels-last rows gathered per instance straight from
the SAM embeddings — the reference's index_select + dense-embedding add at :181-186). Projections are
MFMA GEMMs; the 6-token attentions are wavefront-reduced kernels; the last LayerNorm, the box MLP,
the sigmoid and the objectness logit stay in fp32 (SURVEY.md §7 vii).
"""
import math

import torch

from .. import ops
from .tape import Param, Tape, Var

M_ = "model.grounding_encoder.mask_decoder."
PE_ = "model.grounding_encoder.prompt_encoder."
bf = torch.bfloat16


def dense_pe_rows(gauss, g, dtype):
    """PositionEmbeddingRandom.forward (prompt_encoder.py:216-229) evaluated once with the identical torch
    op sequence in the model dtype (quirk Q10: the reference computes it in bf16 under model.to(bf16));
    input-independent, so it is an init-time constant, returned as token rows [g*g, 256]."""
    G = gauss.to(dtype)
    grid = torch.ones((g, g), device=G.device, dtype=dtype)
    y = (grid.cumsum(dim=0) - 0.5) / g
    x = (grid.cumsum(dim=1) - 0.5) / g
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * math.pi * (c @ G)
    pe = torch.cat([torch.sin(c), torch.cos(c)], dim=-1)  # [g, g, 256]
    return pe.reshape(g * g, -1)


class BoxDecoder:
    def __init__(self, sd, d, device, grads=None, pe_dtype=torch.bfloat16):
        self.d, self.dev = d, device
        self.grads = grads or {}
        self.sd = sd
        self.precise = True  # inference: token side of the decoder in fp32 (forward_f32); False = the bf16 path everywhere
        g = d.sam_grid
        self.key_pe = dense_pe_rows(sd[PE_ + "pe_layer.positional_encoding_gaussian_matrix"], g, pe_dtype).to(bf).contiguous()
        self.pe_nchw = self.key_pe.float().reshape(g, g, -1).permute(2, 0, 1).unsqueeze(0).to(pe_dtype)

    def P(self, name):
        return Param(self.sd[name], self.grads.get(name))

    def _attn(self, tp, prefix, q, k, v, inst, Lq, Lk, internal, residual=None):
        d = self.d
        qp = tp.linear(q, self.P(prefix + "q_proj.weight"), self.P(prefix + "q_proj.bias"))
        kp = tp.linear(k, self.P(prefix + "k_proj.weight"), self.P(prefix + "k_proj.bias"))
        vp = tp.linear(v, self.P(prefix + "v_proj.weight"), self.P(prefix + "v_proj.bias"))
        o = tp.small_attn(qp, kp, vp, inst, d.dec_heads, internal // d.dec_heads, Lq, Lk)
        return tp.linear(o, self.P(prefix + "out_proj.weight"), self.P(prefix + "out_proj.bias"), residual=residual)

    # ------------------------------------------------------------------ inference: the token side in fp32
    def forward_f32(self, image_emb_rows, text, frame_of_instance, want_masks=False, multimask_output=False):
        """The two-way decoder with its TOKEN side in fp32 (inference): the 6 tokens per instance — their residual / post-norm
        chain, every projection that reads them (exact-fp32 MFMA GEMM, grove_gemm_f32), the 6-token attentions' q or k / v, and the
        outputs of the token -> image attentions — carry no bf16 rounding from the [DET] embedding to the box head. The image side
        (N x 1024 keys: k / v projections, image -> token attention, norm4) stays on the bf16 MFMA kernels: a measured budget of
        the box error against the fp32 oracle put 6.5e-4 of the 9.3e-4 mean L1 on the token side's bf16 activations and 7e-6 on the
        whole SAM tower's (tools/box_error_budget.py). text: fp32 or bf16 [N, 256]. Returns (box f32 [N, 4], obj f32 [N]) and, with
        want_masks, the mask branch's (low-res mask logits [N, 1 or 3, 4g, 4g], iou predictions [N, 1 or 3]) as well."""
        d = self.d
        D, g2, nh = d.dec_dim, d.sam_grid ** 2, d.dec_heads
        N = text.shape[0]
        sd = self.sd
        f32 = torch.float32

        def lin32(x, pre, act=ops.ACT_NONE, residual=None):
            return ops.linear_f32(x, sd[pre + ".weight"], sd[pre + ".bias"], act=act, residual=residual)

        def ln32(x, pre):
            return ops.layernorm(None, sd[pre + ".weight"], sd[pre + ".bias"], 1e-5, res=x, out_dtype=f32)[0]

        tokens = torch.empty((N, 6, D), dtype=f32, device=self.dev)  # [iou | 4 mask | text] (mask_decoder.py:166-173): exact widening copies
        tokens[:, 0] = sd[M_ + "iou_token.weight"][0]
        tokens[:, 1:5] = sd[M_ + "mask_tokens.weight"]
        tokens[:, 5] = text
        tokens = tokens.view(N * 6, D)
        key_src = (frame_of_instance.to(torch.int64)[:, None] * g2 + torch.arange(g2, device=self.dev)[None]).reshape(-1).to(torch.int32)
        keys = torch.empty((N * g2, D), dtype=bf, device=self.dev)
        ops.copy_rows(image_emb_rows, keys, N * g2, D, idx_src=key_src)
        ops.add_bcast_rows(keys, sd[PE_ + "no_mask_embed.weight"], 1, out=keys)
        queries = tokens
        t = M_ + "transformer."

        def token_to_image(pre, queries, keys):
            q_in = ops.add_f32(queries, tokens)
            k_img = ops.add_bcast_rows(keys, self.key_pe, g2)
            qp = lin32(q_in, pre + "q_proj")
            kp = ops.linear(k_img, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"])
            vp = ops.linear(keys, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])
            hd = (D // 2) // nh
            o = ops.small_attn(qp, kp, vp, N, nh, hd, 6, g2, out_dtype=f32)
            return lin32(o, pre + "out_proj", residual=queries), k_img

        for i in range(d.dec_depth):
            p = t + f"layers.{i}."
            sa = p + "self_attn."
            qk_in = queries if i == 0 else ops.add_f32(queries, tokens)  # skip_first_layer_pe (transformer.py:153-155)
            qp, kp, vp = lin32(qk_in, sa + "q_proj"), lin32(qk_in, sa + "k_proj"), lin32(queries, sa + "v_proj")
            o = ops.small_attn(qp, kp, vp, N, nh, D // nh, 6, 6, out_dtype=f32)
            queries = lin32(o, sa + "out_proj", residual=None if i == 0 else queries)
            queries = ln32(queries, p + "norm1")
            queries, k_img = token_to_image(p + "cross_attn_token_to_image.", queries, keys)
            queries = ln32(queries, p + "norm2")
            h = lin32(queries, p + "mlp.lin1", act=ops.ACT_RELU)
            queries = lin32(h, p + "mlp.lin2", residual=queries)
            queries = ln32(queries, p + "norm3")
            ia = p + "cross_attn_image_to_token."
            q_in = ops.add_f32(queries, tokens)
            qp = ops.linear(k_img, sd[ia + "q_proj.weight"], sd[ia + "q_proj.bias"])
            kp, vp = lin32(q_in, ia + "k_proj"), lin32(queries, ia + "v_proj")
            o = ops.small_attn(qp, kp, vp, N, nh, (D // 2) // nh, g2, 6)
            keys = ops.linear(o, sd[ia + "out_proj.weight"], sd[ia + "out_proj.bias"], residual=keys)
            keys = ops.layernorm(keys, sd[p + "norm4.weight"], sd[p + "norm4.bias"], 1e-5)[0]
        queries, _ = token_to_image(t + "final_attn_token_to_image.", queries, keys)
        if want_masks:
            hs_all = ln32(queries, t + "norm_final_attn").view(N, 6, D)  # hs of all six tokens (transformer.py:99-106)
            hs = hs_all[:, 5].contiguous()
        else:
            hs_all = None
            hs = ln32(queries.view(N, 6, D)[:, 5].contiguous(), t + "norm_final_attn")
        hp = M_ + "bbox_prediction_head."
        box, obj, _ = ops.box_head(hs, sd[hp + "0.weight"], sd[hp + "0.bias"], sd[hp + "2.weight"], sd[hp + "2.bias"],
                                   sd[M_ + "temporal_objectness_head.weight"], sd[M_ + "temporal_objectness_head.bias"])
        if not want_masks:
            return box, obj
        low, iou = self._mask_branch(hs_all, keys, N)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return box, obj, low[:, sl], iou[:, sl]

    def _mask_branch(self, hs_all, keys, N):
        """MaskDecoder.predict_masks, mask branch (mask_decoder.py:206-227; dormant under GROVE's decoding_type "query" — SURVEY.md
        section 8(f) 3). hs_all fp32 [N, 6, D] = the decoder's output tokens, keys bf16 [N*g*g, D] = its output image tokens
        (channels-last: `src.transpose(1, 2).view(b, c, h, w)` is a view of these rows).
          output_upscaling: a ConvTranspose2d with kernel 2 / stride 2 writes every input pixel to its own 2 x 2 output block, so it IS
          a linear layer per pixel with the 4 sub-pixels stacked on the output channels — two MFMA GEMMs (256 -> 4 x 64, then per
          sub-pixel 64 -> 4 x 32, GELU in its epilogue) around a LayerNorm2d over 64 channels (a row LayerNorm in this layout) and
          a GELU; the 16 sub-pixels are un-shuffled into image order once, at the very end, on the tiny mask tensor.
          hyper-network MLPs + IoU head: fp32 (grove_gemm_f32) on the token side; masks = up . hyper_in^T as one batched GEMM.
        Returns (mask logits fp32 [N, 4, 4g, 4g], iou fp32 [N, 4])."""
        d, sd = self.d, self.sd
        D, g, g2 = d.dec_dim, d.sam_grid, d.sam_grid ** 2
        C1, C2 = D // 4, D // 8
        if not hasattr(self, "_up"):
            w1 = sd[M_ + "output_upscaling.0.weight"]  # [Ci, Co, 2, 2] -> [(dy dx co), ci]
            w2 = sd[M_ + "output_upscaling.3.weight"]
            self._up = dict(w1=w1.permute(2, 3, 1, 0).reshape(4 * C1, D).contiguous(), b1=sd[M_ + "output_upscaling.0.bias"].repeat(4).contiguous(),
                            w2=w2.permute(2, 3, 1, 0).reshape(4 * C2, C1).contiguous(), b2=sd[M_ + "output_upscaling.3.bias"].repeat(4).contiguous())
        U = self._up
        x = ops.linear(keys, U["w1"], U["b1"])                                            # [N*g2, 4*C1]: pixel-major, sub-pixel, channel
        x = x.view(N * g2 * 4, C1)
        x, _, _ = ops.layernorm(x, sd[M_ + "output_upscaling.1.weight"], sd[M_ + "output_upscaling.1.bias"], 1e-6)
        ops.act_fwd(x, ops.ACT_GELU, out=x)
        up = ops.linear(x, U["w2"], U["b2"], act=ops.ACT_GELU)                            # [N*g2*4, 4*C2]
        f32 = torch.float32

        def mlp3(pre, v):
            v = ops.linear_f32(v, sd[pre + "layers.0.weight"], sd[pre + "layers.0.bias"], act=ops.ACT_RELU)
            v = ops.linear_f32(v, sd[pre + "layers.1.weight"], sd[pre + "layers.1.bias"], act=ops.ACT_RELU)
            return ops.linear_f32(v, sd[pre + "layers.2.weight"], sd[pre + "layers.2.bias"])
        hyper = torch.zeros((N, 64, C2), dtype=bf, device=self.dev)  # rows 0..3 = the four mask tokens' hyper vectors (B operand, zero padded)
        for i in range(4):
            hyper[:, i] = mlp3(M_ + f"output_hypernetworks_mlps.{i}.", hs_all[:, 1 + i].contiguous())
        iou = mlp3(M_ + "iou_prediction_head.", hs_all[:, 0].contiguous())
        P = g2 * 16                                                                       # output pixels per instance
        m = torch.empty((N, P, 8), dtype=f32, device=self.dev)
        ops.gemm_raw(up, hyper, m, P, 8, C2, C2, C2, 8, batch=(N, 1), sA=(P * C2, 0), sB=(64 * C2, 0), sC=(P * 8, 0))
        # rows of `m` run (y, x, dy, dx, dy2, dx2): image row 4y + 2dy + dy2, column 4x + 2dx + dx2 (index plumbing on the result)
        masks = m[:, :, :4].reshape(N, g, g, 2, 2, 2, 2, 4).permute(0, 7, 1, 3, 5, 2, 4, 6).reshape(N, 4, 4 * g, 4 * g).contiguous()
        return masks, iou

    def postprocess_masks(self, masks, input_size, original_size):
        """Sam.postprocess_masks (sam.py:137-172): bilinear to the encoder's square input, crop the padding, bilinear to the original
        frame — the crop folds into the second resize's source window. masks fp32 [N, C, h, w] -> [N, C, H, W] logits."""
        S = self.d.sam_image
        m = ops.resize_bilinear(masks.contiguous(), S, S)
        return ops.resize_bilinear(m, int(original_size[0]), int(original_size[1]), crop=(int(input_size[0]), int(input_size[1])))

    def prepare(self, frame_of_instance, N):
        """The index tensors of one forward (and its backward) and the learned token rows — nothing here reads a tower's output, so the
        caller can issue these ~20 tiny launches BEFORE it joins the SAM stream instead of on the serial stretch behind the join."""
        d = self.d
        g2 = d.sam_grid ** 2
        t_idx = torch.arange(N * 6, device=self.dev, dtype=torch.int32)
        neg = torch.full_like(t_idx, -1)
        src5 = torch.where(t_idx % 6 < 5, t_idx % 6, neg)
        dst5 = torch.where(t_idx % 6 < 5, t_idx, neg)
        text_dst = (torch.arange(N, device=self.dev, dtype=torch.int32) * 6 + 5)
        key_src = (frame_of_instance.to(torch.int64)[:, None] * g2 + torch.arange(g2, device=self.dev)[None]).reshape(-1).to(torch.int32)
        out_tok = torch.cat([self.sd[M_ + "iou_token.weight"], self.sd[M_ + "mask_tokens.weight"]], 0)  # [5, D] (tiny concat of weights)
        return dict(N=N, src5=src5, dst5=dst5, text_dst=text_dst, key_src=key_src, out_tok=out_tok)

    def forward(self, image_emb_rows, text_embeds, frame_of_instance, train=False, prep=None, frame_ptr=None):
        # frame_ptr (int32 [F + 1] or None): instances of frame f are frame_ptr[f] .. frame_ptr[f + 1] - 1 (the instance order groups them by
        # frame) — the backward then sums the key gradients per frame without atomics
        """image_emb_rows: bf16 [F*g*g, 256] channels-last SAM embeddings; text_embeds: Var bf16 [N, 256]
        ([DET] embeddings, one per (frame, DET) instance; fp32 allowed when not training); frame_of_instance: int32 [N] frame index.
        Returns (box f32 [N,4], obj f32 [N], state for backward). Without a backward to serve (train=False) the token side runs
        in fp32 (forward_f32); the training forward keeps the bf16 tape path whose backward kernels exist."""
        if not train and self.precise:
            box, obj = self.forward_f32(image_emb_rows, text_embeds.data, frame_of_instance)
            return box, obj, None
        d = self.d
        D, g2 = d.dec_dim, d.sam_grid ** 2
        N = text_embeds.data.shape[0]
        tp = Tape(enabled=train, side=getattr(self, "wgrad_stream", None) if train else None)
        # tokens = [iou | 4 mask | text]  (mask_decoder.py:166-173)
        if prep is None or prep["N"] != N:
            prep = self.prepare(frame_of_instance, N)
        out_tok, src5, dst5, text_dst, key_src = prep["out_tok"], prep["src5"], prep["dst5"], prep["text_dst"], prep["key_src"]
        tokens_data = torch.empty((N * 6, D), dtype=bf, device=self.dev)
        ops.copy_rows(out_tok, tokens_data, N * 6, D, idx_src=src5, idx_dst=dst5)
        ops.copy_rows(text_embeds.data, tokens_data, N, D, idx_dst=text_dst)
        tokens = Var(tokens_data)
        # keys = image_embeddings[idx] + no_mask_embed  (:181-186; prompt_encoder.py:182-184)
        keys0 = torch.empty((N * g2, D), dtype=bf, device=self.dev)
        ops.copy_rows(image_emb_rows, keys0, N * g2, D, idx_src=key_src)
        ops.add_bcast_rows(keys0, self.sd[PE_ + "no_mask_embed.weight"], 1, out=keys0)
        keys = Var(keys0)
        keys_init = keys
        queries = tokens
        t = M_ + "transformer."
        for i in range(d.dec_depth):
            p = t + f"layers.{i}."
            if i == 0:  # skip_first_layer_pe (transformer.py:153-155)
                queries = self._attn(tp, p + "self_attn.", queries, queries, queries, N, 6, 6, D)
            else:
                q = tp.add(queries, tokens)
                queries = self._attn(tp, p + "self_attn.", q, q, queries, N, 6, 6, D, residual=queries)
            queries = tp.layernorm(queries, self.P(p + "norm1.weight"), self.P(p + "norm1.bias"), 1e-5)
            q = tp.add(queries, tokens)
            k = tp.add_const_rows(keys, self.key_pe, g2)
            queries = self._attn(tp, p + "cross_attn_token_to_image.", q, k, keys, N, 6, g2, D // 2, residual=queries)
            queries = tp.layernorm(queries, self.P(p + "norm2.weight"), self.P(p + "norm2.bias"), 1e-5)
            h = tp.linear(queries, self.P(p + "mlp.lin1.weight"), self.P(p + "mlp.lin1.bias"), act=ops.ACT_RELU)
            queries = tp.linear(h, self.P(p + "mlp.lin2.weight"), self.P(p + "mlp.lin2.bias"), residual=queries)
            queries = tp.layernorm(queries, self.P(p + "norm3.weight"), self.P(p + "norm3.bias"), 1e-5)
            q = tp.add(queries, tokens)
            keys = self._attn(tp, p + "cross_attn_image_to_token.", k, q, queries, N, g2, 6, D // 2, residual=keys)
            keys = tp.layernorm(keys, self.P(p + "norm4.weight"), self.P(p + "norm4.bias"), 1e-5)
        q = tp.add(queries, tokens)
        k = tp.add_const_rows(keys, self.key_pe, g2)
        queries = self._attn(tp, t + "final_attn_token_to_image.", q, k, keys, N, 6, g2, D // 2, residual=queries)
        # last LayerNorm only matters for token 5 of every instance; gather first, then normalise in fp32
        q5 = torch.empty((N, D), dtype=bf, device=self.dev)
        ops.copy_rows(queries.data, q5, N, D, idx_src=text_dst)
        nw, nb = self.sd[t + "norm_final_attn.weight"], self.sd[t + "norm_final_attn.bias"]
        hs, mean, rstd = ops.layernorm(q5, nw, nb, 1e-5, save_stats=train, out_dtype=torch.float32)
        hp = M_ + "bbox_prediction_head."
        box, obj, hidden = ops.box_head(hs, self.sd[hp + "0.weight"], self.sd[hp + "0.bias"], self.sd[hp + "2.weight"],
                                        self.sd[hp + "2.bias"], self.sd[M_ + "temporal_objectness_head.weight"],
                                        self.sd[M_ + "temporal_objectness_head.bias"])
        state = None
        if train:
            state = dict(tp=tp, queries=queries, q5=q5, hs=hs, mean=mean, rstd=rstd, hidden=hidden, box=box, text_dst=text_dst,
                         tokens=tokens, keys_init=keys_init, key_src=key_src, N=N, text=text_embeds, src5=src5, frame_ptr=frame_ptr)
        return box, obj, state

    def backward(self, state, dbox, dobj, d_image_emb_rows):
        """dbox f32 [N,4], dobj f32 [N]. Accumulates decoder weight grads; adds the gradient w.r.t. the SAM
        embeddings into d_image_emb_rows (f32 [F*g*g, 256]) and sets state['text'].grad (bf16 [N, 256])."""
        d = self.d
        D, g2 = d.dec_dim, d.sam_grid ** 2
        N = state["N"]
        G = self.grads
        hp = M_ + "bbox_prediction_head."
        t = M_ + "transformer."
        grads = {"dW1": G[hp + "0.weight"], "db1": G[hp + "0.bias"], "dW2": G[hp + "2.weight"], "db2": G[hp + "2.bias"],
                 "dWo": G[M_ + "temporal_objectness_head.weight"].view(-1) if M_ + "temporal_objectness_head.weight" in G else None,
                 "dbo": G.get(M_ + "temporal_objectness_head.bias")}  # (no objectness head without use_temp_objectness: mask_decoder.py:83-87)
        dhs = ops.box_head_bwd(state["hs"], self.sd[hp + "0.weight"], self.sd[hp + "2.weight"],
                               self.sd[M_ + "temporal_objectness_head.weight"], state["hidden"], state["box"], dbox, dobj, grads)
        # (without --train_mask_decoder only the two heads train: the transformer's tensors have no gradient views)
        dq5 = ops.layernorm_bwd(state["q5"], self.sd[t + "norm_final_attn.weight"], ops.to_bf16(dhs), state["mean"], state["rstd"],
                                dweight=G.get(t + "norm_final_attn.weight"), dbias=G.get(t + "norm_final_attn.bias"))
        dqueries = torch.zeros((N * 6, D), dtype=bf, device=self.dev)
        ops.copy_rows(dq5, dqueries, N, D, idx_dst=state["text_dst"])
        state["queries"].grad = dqueries
        tokens = state["tokens"]
        state["tp"].backward()
        # gradient of the gathered image keys -> SAM embeddings (prompt_encoder.no_mask_embed is frozen, train.py:279-296)
        if state.get("frame_ptr") is not None and (state["frame_ptr"].numel() - 1) * g2 == d_image_emb_rows.shape[0]:
            ops.segment_sum_rows(state["keys_init"].grad, d_image_emb_rows, state["frame_ptr"], g2)
        else:
            ops.scatter_add_f32(state["keys_init"].grad, d_image_emb_rows, state["key_src"], N * g2, D)
        # tokens.grad: rows 0..4 of every instance -> iou/mask token weights; row 5 -> text embeddings
        tg = tokens.grad
        dtext = torch.empty((N, D), dtype=bf, device=self.dev)
        ops.copy_rows(tg, dtext, N, D, idx_src=state["text_dst"])
        state["text"].grad = dtext
        if G.get(M_ + "iou_token.weight") is not None:
            tok_of_row = state["src5"]  # row -> learned token 0..4, -1 for the text rows
            tokg = torch.zeros((5, D), dtype=torch.float32, device=self.dev)
            ops.scatter_add_f32(tg, tokg, tok_of_row, N * 6, D)
            ops.axpy(G[M_ + "iou_token.weight"].view(-1), tokg[0])
            ops.axpy(G[M_ + "mask_tokens.weight"].view(-1), tokg[1:5].reshape(-1))
        return None
