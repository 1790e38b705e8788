"""Multi-head attention over fused qkv activations, built from the GEMM / softmax / transpose kernels.

Scores are materialised in fp32 (alpha * Q K^T), soft-maxed into zero-padded bf16 probabilities and
multiplied with V^T, all as strided-batched launches over (batch, head) — no per-head Python loops
and no .transpose().contiguous() copies of q/k. Replaces the bmm/softmax/bmm sequences at
modeling_clip.py:279-319, image_encoder.py:310-319 and HF eager LlamaAttention; the backward is the
explicit five-product form (dP, dS, dV, dQ, dK).
"""
import os

import torch

from .. import ops

# "flash" (default): fused kernels of csrc/flash_attn.hip. "materialized": the round-1 GEMM + softmax path, kept
# as an in-process A/B arm and as a second implementation for the tests (GROVE_ATTN=materialized).
MODE = os.environ.get("GROVE_ATTN", "flash")


class AttnCtx:
    __slots__ = ("probs", "B", "H", "L", "hs", "ld", "q_off", "k_off", "v_off", "alpha", "ld_p", "rel", "rel_hw",
                 "out", "lse", "causal", "kv_len", "flash", "hs_valid", "q_valid", "pad_row", "o_map", "rel_table")


def attention_fwd(qkv, B, L, H, hs, q_off, k_off, v_off, alpha, *, causal=False, kv_len=None, rel=None, rel_hw=(0, 0),
                  out=None, save=False, hs_valid=0, q_valid=None, pad_row=None, o_map=None, o_rows=0, rel_table=None):
    """qkv: bf16 [B*L, ld]; head h of q/k/v lives at columns off + h*hs (hs = padded head dim, the pad
    columns are exact zeros). Returns (out [B*L, H*hs], ctx or None).
    rel_table (ops.rel_table_images; window kernels): the rel-pos terms are made inside the attention kernels, rel must be None."""
    dev = qkv.device
    ld = qkv.stride(0)
    if MODE == "flash" or rel is not None or rel_table is not None:  # the rel-pos bias exists only in the fused kernels
        if rel_table is not None and save:  # the operand the backward reads (written by the forward kernel)
            rel = torch.empty((B * H, L, 32), dtype=torch.bfloat16, device=dev)
        out, lse = ops.flash_attn(qkv, B, L, H, hs, q_off, k_off, v_off, alpha, causal=causal, kv_len=kv_len, rel=None if rel_table is not None else rel,
                                  rel_hw=rel_hw, out=out, want_lse=save, hs_valid=hs_valid, q_valid=q_valid, pad_row=pad_row, o_map=o_map,
                                  o_rows=o_rows, rel_table=rel_table, rel_out=rel if rel_table is not None else None)
        ctx = None
        if save:
            ctx = AttnCtx()
            ctx.rel_table = rel_table
            ctx.hs_valid, ctx.q_valid, ctx.pad_row, ctx.o_map = hs_valid, q_valid, pad_row, o_map
            ctx.flash, ctx.out, ctx.lse, ctx.B, ctx.H, ctx.L, ctx.hs, ctx.ld = True, out, lse, B, H, L, hs, ld
            ctx.q_off, ctx.k_off, ctx.v_off, ctx.alpha = q_off, k_off, v_off, alpha
            ctx.rel, ctx.rel_hw, ctx.causal, ctx.kv_len = rel, rel_hw, causal, kv_len
        return out, ctx
    ld_s = ops.pad_to(L, 4)
    ld_p = ops.pad_to(L, 32)
    scores = torch.empty((B * H, L, ld_s), dtype=torch.float32, device=dev)
    ops.gemm_raw(qkv[:, q_off:], qkv[:, k_off:], scores, L, L, hs, ld, ld, ld_s, batch=(B, H),
                 sA=(L * ld, hs), sB=(L * ld, hs), sC=(H * L * ld_s, L * ld_s), alpha=alpha)
    probs = ops.softmax(scores, L, heads=H, causal=causal, kv_len=kv_len, rel=rel, rel_hw=rel_hw, ld_p=ld_p)
    del scores
    vt = torch.empty((B * H, hs, ld_p), dtype=torch.bfloat16, device=dev)
    ops.transpose(qkv[:, v_off:], L, hs, ld, vt, ld_p, pad_to_cols=ld_p, batch=(B, H), s_in=(L * ld, hs),
                  s_out=(H * hs * ld_p, hs * ld_p))
    if out is None:
        out = torch.empty((B * L, H * hs), dtype=torch.bfloat16, device=dev)
    ops.gemm_raw(probs, vt, out, L, hs, ld_p, ld_p, ld_p, H * hs, batch=(B, H), sA=(H * L * ld_p, L * ld_p),
                 sB=(H * hs * ld_p, hs * ld_p), sC=(L * H * hs, hs))
    ctx = None
    if save:
        ctx = AttnCtx()
        ctx.flash, ctx.rel_table = False, None
        ctx.probs, ctx.B, ctx.H, ctx.L, ctx.hs, ctx.ld = probs, B, H, L, hs, ld
        ctx.q_off, ctx.k_off, ctx.v_off, ctx.alpha, ctx.ld_p = q_off, k_off, v_off, alpha, ld_p
        ctx.rel, ctx.rel_hw = rel, rel_hw
    return out, ctx


def attention_bwd(ctx, qkv, d_out, dqkv, want_drel=False, rope=None, grads_tok=False):
    """d_out: bf16 [B*L, H*hs]. Writes dq/dk/dv into the matching column blocks of dqkv (bf16, same
    layout as qkv; every column of the three blocks is overwritten). Returns drel (f32) if asked.
    rope (ops.rope_table; fused kernels only — the caller checks ctx.flash): dq / dk are un-rotated inside the two backward kernels;
    with the materialised second implementation the caller runs the inverse RoPE pass itself."""
    ctx_rope = rope if ctx.flash else None
    if ctx.flash:
        return ops.flash_attn_bwd(qkv, ctx.out, d_out, ctx.lse, dqkv, ctx.B, ctx.L, ctx.H, ctx.hs, ctx.q_off, ctx.k_off, ctx.v_off,
                                  ctx.alpha, causal=ctx.causal, kv_len=ctx.kv_len, rel=ctx.rel, rel_hw=ctx.rel_hw, want_drel=want_drel,
                                  hs_valid=ctx.hs_valid, q_valid=ctx.q_valid, pad_row=ctx.pad_row, o_map=ctx.o_map, rope=ctx_rope, grads_tok=grads_tok,
                                  rel_table=ctx.rel_table)
    B, H, L, hs, ld, ld_p = ctx.B, ctx.H, ctx.L, ctx.hs, ctx.ld, ctx.ld_p
    dev = qkv.device
    bf = torch.bfloat16
    ld_s = ops.pad_to(L, 4)
    Lp = ops.pad_to(L, 32)  # == ld_p
    ldo = d_out.stride(0)
    ldd = dqkv.stride(0)
    # dP[q, key] = sum_d dO[q, d] V[key, d]
    dP = torch.empty((B * H, L, ld_s), dtype=torch.float32, device=dev)
    ops.gemm_raw(d_out, qkv[:, ctx.v_off:], dP, L, L, hs, ldo, ld, ld_s, batch=(B, H), sA=(L * ldo, hs),
                 sB=(L * ld, hs), sC=(H * L * ld_s, L * ld_s))
    drel = None
    if want_drel:
        drel = torch.empty((B * H, L, ctx.rel_hw[0] + ctx.rel_hw[1]), dtype=torch.float32, device=dev)
    dS = ops.softmax_bwd(dP, ctx.probs, L, ctx.alpha, drel=drel, rel_hw=ctx.rel_hw)
    del dP
    # dV[key, d] = sum_q P[q, key] dO[q, d]  ->  A = P^T [key, q], B = dO^T [d, q]
    pT = torch.empty((B * H, L, Lp), dtype=bf, device=dev)
    ops.transpose(ctx.probs, L, L, ld_p, pT, Lp, pad_to_cols=Lp, batch=(B * H, 1), s_in=(L * ld_p, 0), s_out=(L * Lp, 0))
    doT = torch.empty((B * H, hs, Lp), dtype=bf, device=dev)
    ops.transpose(d_out, L, hs, ldo, doT, Lp, pad_to_cols=Lp, batch=(B, H), s_in=(L * ldo, hs), s_out=(H * hs * Lp, hs * Lp))
    ops.gemm_raw(pT, doT, dqkv[:, ctx.v_off:], L, hs, Lp, Lp, Lp, ldd, batch=(B, H), sA=(H * L * Lp, L * Lp),
                 sB=(H * hs * Lp, hs * Lp), sC=(L * ldd, hs))
    del pT, doT
    # dQ[q, d] = sum_key dS[q, key] K[key, d]  ->  B = K^T [d, key]
    kT = torch.empty((B * H, hs, Lp), dtype=bf, device=dev)
    ops.transpose(qkv[:, ctx.k_off:], L, hs, ld, kT, Lp, pad_to_cols=Lp, batch=(B, H), s_in=(L * ld, hs),
                  s_out=(H * hs * Lp, hs * Lp))
    ops.gemm_raw(dS, kT, dqkv[:, ctx.q_off:], L, hs, Lp, ld_p, Lp, ldd, batch=(B, H), sA=(H * L * ld_p, L * ld_p),
                 sB=(H * hs * Lp, hs * Lp), sC=(L * ldd, hs))
    del kT
    # dK[key, d] = sum_q dS[q, key] Q[q, d]  ->  A = dS^T [key, q], B = Q^T [d, q]
    dsT = torch.empty((B * H, L, Lp), dtype=bf, device=dev)
    ops.transpose(dS, L, L, ld_p, dsT, Lp, pad_to_cols=Lp, batch=(B * H, 1), s_in=(L * ld_p, 0), s_out=(L * Lp, 0))
    qT = torch.empty((B * H, hs, Lp), dtype=bf, device=dev)
    ops.transpose(qkv[:, ctx.q_off:], L, hs, ld, qT, Lp, pad_to_cols=Lp, batch=(B, H), s_in=(L * ld, hs),
                  s_out=(H * hs * Lp, hs * Lp))
    ops.gemm_raw(dsT, qT, dqkv[:, ctx.k_off:], L, hs, Lp, Lp, Lp, ldd, batch=(B, H), sA=(H * L * Lp, L * Lp),
                 sB=(H * hs * Lp, hs * Lp), sC=(L * ldd, hs))
    return drel
