"""LLaMA decoder stack on the HIP kernels (the HF `LlamaModel` the reference calls at
llava_llama.py:100-109 with flash-attention-2; transformers is third party and un-vendored).

Per layer: RMSNorm -> fused q|k|v GEMM -> RoPE in place -> causal attention with right-padding
lengths -> o_proj (+residual in the epilogue) -> RMSNorm -> fused gate|up GEMM -> SwiGLU ->
down_proj (+residual). Weights are frozen in the shipped fine-tune (train.py:254-255, lora_r 0),
so the backward is dgrad only and uses pre-transposed weight copies (HBM is 288 GB; the 13.5 GB of
W^T buys NT-form GEMMs for every dgrad). Activations of all layers are kept (no recompute).
Layout: hidden [B*S, H] bf16 row-major, sequence b at rows b*S .. b*S+S-1, right padded.
"""
import os

import torch

from .. import ops
from .attention import attention_bwd, attention_fwd


class LlamaStack:
    def __init__(self, sd, d, device, train=False, fp32_stream=None, fp8=False, fp8_policy="det16_kv16"):
        self.d, self.dev, self.train = d, device, train
        self.fp32_stream = (not train) if fp32_stream is None else fp32_stream
        # fp8 (BASELINE config 5): the four projections of every layer run on the e4m3 block-scaled MFMA GEMM (grove_gemm_fp8) with
        # per-output-channel weight scales made here and per-row activation scales made on the fly; inference only (no fp8 dgrad)
        # fp8_policy (tools/fp8_policy_study.py, DESIGN section 8 — e4m3's 3 mantissa bits cost ~5 % per GEMM output whatever the scales):
        #   "all"        every projection of every layer in e4m3 (round 2);
        #   "det16_kv16" k_proj / v_proj of all rows stay bf16 (16.6 % of the stack's FLOPs: every later row attends to them) and the
        #                rows passed as `precise_rows` (the [DET] positions, whose hidden state feeds the box head) are recomputed in
        #                bf16 after each e4m3 GEMM (a few rows: one pass over the bf16 weights) — halves the box error.
        assert not (fp8 and train), "the fp8 path is inference-only"
        assert fp8_policy in ("all", "det16_kv16")
        self.fp8, self.fp8_policy = fp8, fp8_policy
        self.defer_norm = os.environ.get("GROVE_DECODE_DEFER_NORM", "1") != "0"  # A/B knob of the deferred RMSNorm in the batched decode step
        self.batch_invariant = os.environ.get("GROVE_DECODE_BATCH_INVARIANT", "0") == "1"  # (set per call by GROVEForCausalLM.generate(batch_invariant=...))
        self.layers = []
        for i in range(d.n_layers):
            p = f"model.layers.{i}."
            wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).contiguous()
            wgu = torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0).contiguous()
            L = {"ln1": sd[p + "input_layernorm.weight"], "ln2": sd[p + "post_attention_layernorm.weight"],
                 "wqkv": wqkv, "wo": sd[p + "self_attn.o_proj.weight"], "wgu": wgu, "wd": sd[p + "mlp.down_proj.weight"]}
            if d.mlp % 4 == 0:
                # gate / up rows interleaved [4 gate, 4 up]: the gate|up GEMM then applies silu(gate) * up in its epilogue
                # (ACT_SWIGLU_PAIR) and writes the product directly — no [rows, 2I] round trip, no separate SwiGLU launch
                L["wgu_sw"] = ops.swiglu_interleave(wgu)
            if train:
                for k in ("wqkv", "wo", "wgu", "wd"):
                    L[k + "_t"] = ops.transpose2d(L[k])
            if fp8:
                for k in ("wqkv", "wo", "wgu", "wd"):
                    if L[k].shape[1] % 128 == 0:
                        L[k + "_q"] = ops.quant_fp8_rows(L[k])
                if fp8_policy == "det16_kv16" and "wqkv_q" in L:
                    H = d.hidden
                    L["wq_q"] = (L["wqkv_q"][0][:H], L["wqkv_q"][1][:H])  # (row-wise scales: the q rows of the fused copy are q_proj's own)
                    L["wq"], L["wkv"] = wqkv[:H], wqkv[H:]
            self.layers.append(L)
        self.norm = sd["model.norm.weight"]

    def _lin(self, L, k, x, residual=None, precise_rows=None):
        """x @ L[k].T (+ residual): the fp8 GEMM when this stack is quantised (and K fits its 128-byte K tile), else the bf16 one.
        precise_rows (policy det16_kv16): those rows of the result are recomputed by the bf16 GEMM (gathered A rows, scattered C rows)."""
        if (k + "_q") in L:
            y = ops.linear_fp8(x, L[k + "_q"][0], L[k + "_q"][1], residual=residual)
            if precise_rows is not None:
                ops.linear(x, L[k], residual=residual, a_idx=precise_rows, c_idx=precise_rows, M=int(precise_rows.numel()), out=y)
            return y
        return ops.linear(x, L[k], residual=residual)

    def _qkv(self, L, h, precise_rows=None):
        """The fused q|k|v activation [rows, 3H]. Policy det16_kv16: q from the e4m3 GEMM, k|v from the bf16 one (two launches writing
        the two column ranges of one tensor), the precise rows of q recomputed in bf16."""
        if "wq_q" not in L:
            return self._lin(L, "wqkv", h, precise_rows=precise_rows if self.fp8_policy == "det16_kv16" else None)
        H = self.d.hidden
        qkv = torch.empty((h.shape[0], 3 * H), dtype=torch.bfloat16, device=self.dev)
        ops.linear_fp8(h, L["wq_q"][0], L["wq_q"][1], out=qkv[:, :H])
        ops.linear(h, L["wkv"], out=qkv[:, H:])
        if precise_rows is not None:
            ops.linear(h, L["wq"], a_idx=precise_rows, c_idx=precise_rows, M=int(precise_rows.numel()), out=qkv[:, :H])
        return qkv

    fuse_rope_bwd = __import__("os").environ.get("GROVE_FUSE_ROPE_BWD", "1") != "0"  # inverse RoPE inside the attention backward kernels (A/B knob)
    rope_from_table = __import__("os").environ.get("GROVE_ROPE_TABLE", "1") != "0"   # forward RoPE reads cos | sin from the table (A/B knob)
    fuse_swiglu_bwd = __import__("os").environ.get("GROVE_FUSE_SWIGLU_BWD", "1") != "0"  # d(gate | up) in the down-proj dgrad GEMM's epilogue (A/B knob)

    def _mlp_dgu(self, L, dx, gu, I):
        """d(gate | up) [rows, 2I] from d x2 [rows, H]: the down-projection's dgrad with SwiGLU's backward in its epilogue (round 4:
        one GEMM instead of GEMM -> [rows, I] -> grove_swiglu_bwd; bit-identical), or the two launches (A/B arm, K % 64 != 0)."""
        if self.fuse_swiglu_bwd and L["wd_t"].shape[1] % 64 == 0 and I % 8 == 0:
            return ops.linear(dx, L["wd_t"], act=ops.ACT_SWIGLU_BWD, residual=gu)
        return ops.swiglu_bwd(gu, ops.linear(dx, L["wd_t"]), I)

    def _rope_table(self, S):
        """cos | sin of positions 0 .. S-1 (f32 [>= S, head_dim], ops.rope_table), grown on demand."""
        t = getattr(self, "_rope_cs", None)
        if t is None or t.shape[0] < S:
            t = self._rope_cs = ops.rope_table(self.d.head_dim, self.d.rope_theta, max(S, 1024), self.dev)
        return t

    def _tail_index(self, B, S, s0):
        """int32 rows b*S + s0 + j (the last S - s0 positions of every sequence) and their positions (cached per geometry)."""
        key = (B, S, s0)
        if getattr(self, "_tail_cache", None) is None:
            self._tail_cache = {}
        if key not in self._tail_cache:
            if len(self._tail_cache) >= 64:  # real data: (S, s0) differs per batch — keep the most recent geometries only
                self._tail_cache.pop(next(iter(self._tail_cache)))
            j = torch.arange(s0, S, dtype=torch.int32)
            idx = (torch.arange(B, dtype=torch.int32)[:, None] * S + j[None]).reshape(-1)
            self._tail_cache[key] = (idx.to(self.dev), j.repeat(B).to(self.dev))
        return self._tail_cache[key]

    def _last_layer_tail(self, L, x, B, S, s0, kv_len, save):
        """The LAST layer when only the hidden states of positions >= s0 are consumed downstream (training: the labelled rows of the
        shifted CE and the [DET] rows all lie in the answer, behind the 575 visual tokens and the prompt): keys / values of every
        position, but queries, attention output, o_proj and the whole MLP for the tail rows only — the same values the full layer gives
        at those rows (HF computes all S rows of every layer, llava_llama.py:100-109; nothing reads the others of the last one). bf16
        stream form. Returns (x_out [B*Lq, H] compact tail rows, saved tuple)."""
        d = self.d
        H, nh, hd, I = d.hidden, d.n_heads, d.head_dim, d.mlp
        Lq = S - s0
        tail_idx, pos_t = self._tail_index(B, S, s0)
        pos = torch.arange(S, dtype=torch.int32, device=self.dev).repeat(B)
        h = ops.rmsnorm(x, L["ln1"], d.rms_eps)
        kv = ops.linear(h, L["wqkv"][H:])                       # k | v of all rows
        rope_tab = self._rope_table(S) if (self.rope_from_table and hd % 16 == 0) else None
        ops.rope_(kv, pos, 0, nh, hd, d.rope_theta, table=rope_tab)  # (keys only: the first nh heads of the k | v activation)
        n_t = B * Lq
        x_t = torch.empty((n_t, H), dtype=torch.bfloat16, device=self.dev)
        h_t = torch.empty((n_t, H), dtype=torch.bfloat16, device=self.dev)
        ops.copy_rows(x, x_t, n_t, H, idx_src=tail_idx)
        ops.copy_rows(h, h_t, n_t, H, idx_src=tail_idx)
        q_t = ops.linear(h_t, L["wqkv"][:H])
        ops.rope_(q_t, pos_t, 0, nh, hd, d.rope_theta, table=rope_tab)
        o_t, lse = ops.flash_attn_tail(q_t, kv, B, Lq, S, nh, hd, hd ** -0.5, kv_len=kv_len, want_lse=save)
        x1_t = ops.linear(o_t, L["wo"], residual=x_t)
        h2_t = ops.rmsnorm(x1_t, L["ln2"], d.rms_eps)
        if "wgu_sw" in L and n_t >= 1024:
            gu = torch.empty((n_t, 2 * I), dtype=torch.bfloat16, device=self.dev) if save else None
            a = ops.linear(h2_t, L["wgu_sw"], act=ops.ACT_SWIGLU_PAIR, aux=gu, ld_aux=2 * I)
        else:
            gu = ops.linear(h2_t, L["wgu"])
            a = ops.swiglu(gu, I)
        x_out = ops.linear(a, L["wd"], residual=x1_t)
        saved = ("tail", x, kv, q_t, o_t, lse, x1_t, gu, s0, kv_len) if save else None
        return x_out, saved

    def _last_layer_tail_bwd(self, L, saved, dx_t, B, S):
        """dgrad of _last_layer_tail. dx_t: bf16 [B*Lq, H] gradient of the compact tail output; returns d x [B*S, H]."""
        d = self.d
        H, nh, hd, I = d.hidden, d.n_heads, d.head_dim, d.mlp
        _, x, kv, q_t, o_t, lse, x1_t, gu, s0, kv_len = saved
        Lq = S - s0
        n_t = B * Lq
        tail_idx, pos_t = self._tail_index(B, S, s0)
        pos = torch.arange(S, dtype=torch.int32, device=self.dev).repeat(B)
        dgu = self._mlp_dgu(L, dx_t, gu, I)
        dh2 = ops.linear(dgu, L["wgu_t"])
        ops.rmsnorm_bwd(x1_t, L["ln2"], dh2, d.rms_eps, dx=dx_t, accumulate=True)    # dx_t now d x1 (tail rows)
        do = ops.linear(dx_t, L["wo_t"])
        dq_t = torch.empty_like(q_t)
        dkv = torch.empty_like(kv)
        if self.fuse_rope_bwd:
            ops.flash_attn_tail_bwd(q_t, kv, o_t, do, lse, dq_t, dkv, B, Lq, S, nh, hd, hd ** -0.5, kv_len=kv_len, rope=self._rope_table(S))
        else:
            ops.flash_attn_tail_bwd(q_t, kv, o_t, do, lse, dq_t, dkv, B, Lq, S, nh, hd, hd ** -0.5, kv_len=kv_len)
            ops.rope_(dq_t, pos_t, 0, nh, hd, d.rope_theta, inverse=True)
            ops.rope_(dkv, pos, 0, nh, hd, d.rope_theta, inverse=True)
        wt = L["wqkv_t"]                                        # [H, 3H]: W^T of the fused projection (K-major for the dgrad)
        dh = ops.linear(dkv, wt[:, H:])                          # all rows: d (k | v) . W_kv
        dh_t = ops.linear(dq_t, wt[:, :H])                       # tail rows: d q . W_q
        ops.copy_rows(dh_t, dh, n_t, H, idx_dst=tail_idx, accumulate=True)
        dx = torch.zeros((B * S, H), dtype=torch.bfloat16, device=self.dev)
        ops.copy_rows(dx_t, dx, n_t, H, idx_dst=tail_idx)        # the residual path of the tail rows
        ops.rmsnorm_bwd(x, L["ln1"], dh, d.rms_eps, dx=dx, accumulate=True)
        return dx

    def forward(self, x, B, S, kv_len=None, save=False, kv_cache=None, precise_rows=None, tail_start=None):
        """x: bf16 [B*S, H] input embeddings (consumed). kv_len: int32 [B] valid lengths or None.
        kv_cache: optional list (one per layer) of bf16 [B, 2, heads, S_max, hd] tensors (KVCache layout) that receive the rotated keys
        and the values of positions 0..S-1 (the prefill of a cached decode). Returns (final-norm hidden [B*S, H], ctx).
        Residual stream, `self.fp32_stream` (default: models built for inference):
          fp32: the stream is held in FP32 (`res`); each branch output (o_proj, down_proj; bf16 from the GEMM) is added to it inside
                the RMSNorm kernel that follows (grove_rmsnorm_fwd, residual-stream form), so it is never rounded to bf16 between the
                64 residual adds — that rounding was most of the stack's distance to the fp32 oracle at full depth (hidden state
                1.4 % -> 0.56 % rms). Costs 6 more bytes of HBM traffic per element and residual add.
          bf16: what the reference stores; the residual add rides in the o_proj / down_proj GEMM epilogues (models built for training:
                the losses are insensitive to it, the step saves ~1.2 ms)."""
        d = self.d
        H, nh, hd, I = d.hidden, d.n_heads, d.head_dim, d.mlp
        pos = torch.arange(S, dtype=torch.int32, device=self.dev).repeat(B)
        saved = []
        f32 = self.fp32_stream
        pr = precise_rows if (self.fp8 and self.fp8_policy == "det16_kv16" and precise_rows is not None and precise_rows.numel()) else None
        res = ops.to_f32(x) if f32 else None
        t = None  # fp32 stream: branch output not yet added to the stream
        use_tail = tail_start is not None and tail_start > 0 and not f32 and not self.fp8 and kv_cache is None
        rope_tab = self._rope_table(S) if (self.rope_from_table and hd % 16 == 0) else None  # cos | sin read, not evaluated per thread
        for li, L in enumerate(self.layers):
            if use_tail and li == len(self.layers) - 1:
                x, sv = self._last_layer_tail(L, x, B, S, tail_start, kv_len, save)
                if save:
                    saved.append(sv)
                continue
            if f32:
                xb = torch.empty_like(x) if save else None
                h = ops.rmsnorm(t, L["ln1"], d.rms_eps, res=res, res_bf16=xb)
            else:
                xb = x
                h = ops.rmsnorm(x, L["ln1"], d.rms_eps)
            qkv = self._qkv(L, h, pr) if self.fp8 else ops.linear(h, L["wqkv"])
            ops.rope_(qkv, pos, 0, 2 * nh, hd, d.rope_theta, table=rope_tab)
            if kv_cache is not None:
                kv_cache[li][:, :, :, :S].copy_(qkv.view(B, S, 3, nh, hd)[:, :, 1:].permute(0, 2, 3, 1, 4))  # (a strided copy: layout only)
            o, actx = attention_fwd(qkv, B, S, nh, hd, 0, H, 2 * H, hd ** -0.5, causal=True, kv_len=kv_len, save=save)
            if f32:
                t = self._lin(L, "wo", o, precise_rows=pr)
                x1b = torch.empty_like(x) if save else None
                h2 = ops.rmsnorm(t, L["ln2"], d.rms_eps, res=res, res_bf16=x1b)
            else:
                x1b = self._lin(L, "wo", o, residual=x, precise_rows=pr)
                h2 = ops.rmsnorm(x1b, L["ln2"], d.rms_eps)
            if "wgu_q" in L:
                gu = self._lin(L, "wgu", h2, precise_rows=pr)
                a = ops.swiglu(gu, I)
            elif "wgu_sw" in L and h2.shape[0] >= 1024:  # (the fused epilogue lives in the pipelined kernel: big GEMMs only)
                gu = torch.empty((h2.shape[0], 2 * I), dtype=torch.bfloat16, device=self.dev) if save else None
                a = ops.linear(h2, L["wgu_sw"], act=ops.ACT_SWIGLU_PAIR, aux=gu, ld_aux=2 * I)
            else:
                gu = ops.linear(h2, L["wgu"])
                a = ops.swiglu(gu, I)
            if f32:
                t = self._lin(L, "wd", a, precise_rows=pr)
            else:
                x = self._lin(L, "wd", a, residual=x1b, precise_rows=pr)
            if save:
                saved.append((xb, qkv, actx, x1b, gu))
        if f32:
            xl = torch.empty_like(x) if save else None
            out = ops.rmsnorm(t, self.norm, d.rms_eps, res=res, res_bf16=xl)
            self.last_stream = res  # fp32 pre-norm stream of this forward: the box path re-normalises its [DET] rows in fp32
            self._final_norm_f32 = lambda: ops.rmsnorm(None, self.norm, d.rms_eps, res=res, out_dtype=torch.float32)
        else:
            xl = x
            out = ops.rmsnorm(x, self.norm, d.rms_eps)
            self.last_stream = x
            self._final_norm_f32 = None
        ctx = (saved, xl, pos, B, S) if save else None
        return out, ctx

    def new_kv_cache(self, B, S_max):
        d = self.d
        # zero-filled: the graph-replayed step reads whole 64-key tiles and masks by kv_len; 0 * stale NaN would poison P V
        return [torch.zeros((B, 2, d.n_heads, S_max, d.head_dim), dtype=torch.bfloat16, device=self.dev) for _ in self.layers]

    def _decode_body(self, x, pos, kv_cache, lm_head=None):
        """The launches of one cached step: six per layer — RMSNorm folded into the q|k|v GEMV, RoPE + cache append + one-query
        attention over n_split blocks per head + the merge of their partial results, o_proj GEMV (+residual), RMSNorm folded into
        the gate|up GEMV whose epilogue applies the SwiGLU, down GEMV (+residual). Everything that changes from step to step (the position) is DEVICE data, so the same
        sequence can be replayed from a captured HIP graph."""
        d = self.d
        nh, hd = d.n_heads, d.head_dim
        # Residual stream of the step (self.fp32_stream, as in forward()): FP32 for inference models — x arrives as the bf16 embedding,
        # is widened once, every GEMV that reads it normalises the fp32 values, o_proj / down add into it in fp32 — so the 64
        # residual adds of a token are not rounded to bf16 one by one (generated rows: hidden state 1.18 % -> see DESIGN 7a);
        # bf16 (what the reference stores) for training models.
        f32 = self.fp32_stream
        sdt = torch.float32 if f32 else torch.bfloat16
        if f32 and x.dtype != torch.float32:
            x = ops.to_f32(x)
        # batch-invariant arithmetic (round 6, `self.batch_invariant`: the clip-batched decode of infer_iground): every GEMV on the
        # matrix-core kernel whatever the number of sequences, and a fixed number of cache splits per head — a sequence's bits do not
        # depend on which (or how many) other sequences share its step, so batched ids equal the one-at-a-time ids BY CONSTRUCTION
        bi = self.batch_invariant
        ns = 8 if bi else None
        B = x.shape[0]
        # Deferred RMSNorm (round 6): when every GEMV of the step runs on the matrix-core kernel (3..8 sequences, or any number in the
        # batch-invariant mode), o_proj / down_proj leave bf16(stream * next norm weight) and their per-workgroup sums of squares, and the
        # q|k|v / gate|up / lm_head launches scale their product by the row's rstd: 64 of the step's 67 norm launches disappear
        # (grove_gemv_params.xs_out / ssq_in; the step keeps one norm for the first layer's input and the two that PRODUCE the final hidden rows)
        deferred = (f32 and (bi or B >= 3) and self.defer_norm and d.hidden % 128 == 0 and d.mlp % 128 == 0 and (2 * d.mlp) % 16 == 0 and
                    all("wgu_sw" in L for L in self.layers))
        if deferred:
            nl = len(self.layers)
            xs = ops.rmsnorm(None, self.layers[0]["ln1"], d.rms_eps, res=x)  # layer 0's input: the one norm launch of the layers
            ssq = None
            for i, (L, kv) in enumerate(zip(self.layers, kv_cache)):
                qkv = ops.gemv(xs, L["wqkv"], batch_invariant=bi, norm_in=(ssq, d.rms_eps) if ssq is not None else None)
                o = ops.decode_attn(qkv, kv, pos, nh, hd, d.rope_theta, hd ** -0.5, n_split=ns)
                x1, xs2, ssq2 = ops.gemv(o, L["wo"], residual=x, out_dtype=sdt, batch_invariant=bi, norm_out=L["ln2"])
                a = ops.gemv(xs2, L["wgu_sw"], act=ops.ACT_SWIGLU_PAIR, batch_invariant=bi, norm_in=(ssq2, d.rms_eps))
                nxt = self.layers[i + 1]["ln1"] if i + 1 < nl else self.norm
                x, xs, ssq = ops.gemv(a, L["wd"], residual=x1, out_dtype=sdt, batch_invariant=bi, norm_out=nxt)
            out = ops.rmsnorm(None, self.norm, d.rms_eps, res=x)
            self.last_decode_hidden_f32 = ops.rmsnorm(None, self.norm, d.rms_eps, res=x, out_dtype=torch.float32)  # the box path's rows
            logits = ops.gemv(xs, lm_head, out_dtype=torch.float32, batch_invariant=bi, norm_in=(ssq, d.rms_eps)) if lm_head is not None else None
            return out, logits
        for L, kv in zip(self.layers, kv_cache):
            qkv = ops.gemv(x, L["wqkv"], rms_weight=L["ln1"], eps=d.rms_eps, batch_invariant=bi)
            o = ops.decode_attn(qkv, kv, pos, nh, hd, d.rope_theta, hd ** -0.5, n_split=ns)
            x1 = ops.gemv(o, L["wo"], residual=x, out_dtype=sdt, batch_invariant=bi)
            if "wgu_sw" in L and (2 * d.mlp) % 16 == 0:  # SwiGLU in the gate|up GEMV's epilogue (rows interleaved 4 gate / 4 up)
                a = ops.gemv(x1, L["wgu_sw"], rms_weight=L["ln2"], eps=d.rms_eps, act=ops.ACT_SWIGLU_PAIR, batch_invariant=bi)
                x = ops.gemv(a, L["wd"], residual=x1, out_dtype=sdt, batch_invariant=bi)
            else:
                gu = ops.gemv(x1, L["wgu"], rms_weight=L["ln2"], eps=d.rms_eps, batch_invariant=bi)
                x = ops.gemv(gu, L["wd"], residual=x1, swiglu=True, out_dtype=sdt, batch_invariant=bi)
        if f32:
            out = ops.rmsnorm(None, self.norm, d.rms_eps, res=x)
            self.last_decode_hidden_f32 = ops.rmsnorm(None, self.norm, d.rms_eps, res=x, out_dtype=torch.float32)  # the box path's rows
        else:
            out = ops.rmsnorm(x, self.norm, d.rms_eps)
            self.last_decode_hidden_f32 = None
        logits = ops.gemv(x, lm_head, out_dtype=torch.float32, rms_weight=self.norm, eps=d.rms_eps, batch_invariant=bi) if lm_head is not None else None
        return out, logits

    def decode_step(self, x, t, kv_cache, lm_head=None):
        """One cached step (HF use_cache=True): x bf16 [B, H] = embedding of the token at position t (the cache holds positions
        0..t-1). Every projection is the weight-streaming GEMV; attention is one query row per (sequence, head) against the
        cache. Returns (final-norm hidden [B, H], fp32 logits or None)."""
        pos = torch.full((x.shape[0],), t, dtype=torch.int32, device=self.dev)
        return self._decode_body(x, pos, kv_cache, lm_head)

    def decode_graph(self, B, kv_cache, x0, t0, lm_head=None):
        """Capture one cached step in a HIP graph (a step is ~420 launches of a few microseconds each: launch-bound from
        Python) and return step(x, t) -> (hidden, logits) that replays it.
        The warm-up before the capture is a REAL launch sequence on the live cache (first-call attribute setup of every kernel):
        it runs with the first step's own input (x0 at position t0), so the K|V row it appends is exactly the row the first
        replay rewrites — it must never run with a dummy input / position, which would overwrite a prefilled row (position 0 is
        the BOS attention sink)."""
        dev, H = self.dev, self.d.hidden
        x_in = torch.empty((B, H), dtype=torch.bfloat16, device=dev)
        x_in.copy_(x0)
        pos = torch.full((B,), int(t0), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._decode_body(x_in, pos, kv_cache, lm_head)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out, logits = self._decode_body(x_in, pos, kv_cache, lm_head)

        def step(x, t):
            x_in.copy_(x)
            pos.fill_(t)
            graph.replay()
            return out, logits
        return step

    def greedy_graph(self, B, kv_cache, embed, lm_head, first_tok, t0, max_steps, vocab, eos, pad, finished0):
        """The WHOLE greedy step on the device, captured once: embedding gather of the current token -> the cached step
        (_decode_body) -> grove_greedy_pick: argmax over the [B, V] logits, HF's finished / pad bookkeeping, the token and its
        position advance in place, the step's final-norm hidden row and the picked id land in `hid_out[step]` / `ids_out[:, step]`.
        Nothing per token comes back to the host, so a run of steps is a run of graph replays with no stream sync between them
        (round 2 read `finished.all()` and the argmax back every token: the GPU idled while the host queued the next replay).
        Returns (replay, state): state = dict(ids_out [B, max_steps] int64, hid_out [max_steps, B, H] (+ hid_out_f32 for fp32-stream
        models), finished [B] bool, pos [B] int32). The warm-up pass is the real first step (see decode_graph), so the capture starts at step 0 again with
        the state rewound — the K|V row it appended is rewritten with identical bytes."""
        dev, H = self.dev, self.d.hidden
        tok = first_tok.to(torch.int32).clone()
        pos = torch.full((B,), int(t0), dtype=torch.int32, device=dev)
        finished = finished0.clone()
        ids_out = torch.full((B, max_steps), int(pad), dtype=torch.int64, device=dev)
        hid_out = torch.zeros((max_steps, B, H), dtype=torch.bfloat16, device=dev)
        hid32 = torch.zeros((max_steps, B, H), dtype=torch.float32, device=dev) if self.fp32_stream else None
        x_in = torch.empty((B, H), dtype=torch.bfloat16, device=dev)

        def body():
            ops.copy_rows(embed, x_in, B, H, idx_src=tok)
            out, logits = self._decode_body(x_in, pos, kv_cache, lm_head)
            # argmax + finished / pad bookkeeping + filing the id and the hidden row(s) under the step number + tok / pos advance: one launch
            ops.greedy_pick(logits.view(B, -1), vocab, finished, tok, pos, ids_out, eos, pad, int(t0), hidden=out, hid_out=hid_out,
                            hidden_f32=self.last_decode_hidden_f32 if hid32 is not None else None, hid_out_f32=hid32)
        keep = (tok.clone(), finished.clone())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()                                                  # warm-up = the real step 0
        torch.cuda.current_stream(dev).wait_stream(side)
        tok.copy_(keep[0]); finished.copy_(keep[1]); pos.fill_(int(t0))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        # the graph's nodes hold raw ADDRESSES: every tensor they touch must outlive the replays (dropping tok / pos / x_in here hands
        # their memory back to the allocator while the graph still reads and writes it)
        return graph.replay, dict(ids_out=ids_out, hid_out=hid_out, hid_out_f32=hid32, finished=finished, pos=pos,
                                  _keep=(graph, tok, pos, x_in, kv_cache, embed, lm_head))

    def backward(self, ctx, d_out):
        """dgrad through the frozen stack. d_out: bf16 [B*S, H] gradient of the post-norm hidden.
        Returns the gradient w.r.t. the input embeddings."""
        d = self.d
        saved, x_last, pos, B, S = ctx
        H, nh, hd, I = d.hidden, d.n_heads, d.head_dim, d.mlp
        dx = ops.rmsnorm_bwd(x_last, self.norm, d_out, d.rms_eps)
        for L, sv in zip(reversed(self.layers), reversed(saved)):
            if isinstance(sv[0], str):  # ("tail", ...): the last layer ran on the tail rows only: d_out / dx are compact [B*Lq, H] here
                dx = self._last_layer_tail_bwd(L, sv, dx, B, S)
                continue
            x, qkv, actx, x1, gu = sv
            # x2 = x1 + down(swiglu(gu));  dx is d x2
            dgu = self._mlp_dgu(L, dx, gu, I)                    # [B*S, 2I]
            dh2 = ops.linear(dgu, L["wgu_t"])                   # [B*S, H]
            del dgu
            ops.rmsnorm_bwd(x1, L["ln2"], dh2, d.rms_eps, dx=dx, accumulate=True)   # dx now d x1
            # x1 = x + o_proj(attn(rope(qkv(rms(x)))))
            do = ops.linear(dx, L["wo_t"])                      # [B*S, H]
            dqkv = torch.empty_like(qkv)
            if actx.flash and self.fuse_rope_bwd:  # dq / dk come back un-rotated from the two backward kernels' epilogues
                attention_bwd(actx, qkv, do, dqkv, rope=self._rope_table(S))
            else:
                attention_bwd(actx, qkv, do, dqkv)
                ops.rope_(dqkv, pos, 0, 2 * nh, hd, d.rope_theta, inverse=True)
            dh = ops.linear(dqkv, L["wqkv_t"])
            del dqkv, do
            ops.rmsnorm_bwd(x, L["ln1"], dh, d.rms_eps, dx=dx, accumulate=True)     # dx now d x
        return dx
