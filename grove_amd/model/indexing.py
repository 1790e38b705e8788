"""Host-side construction of the int32 gather/scatter row maps the GEMM / LayerNorm kernels take.

These replace the data-movement ops of the reference with index tables built once per shape:
  - conv3d_gather_index: 'same'-padded 3x3x3 taps over (t, h, w) tokens — the rearranges around
    Conv3d in SpatioTemporalConvAdapter (modeling_clip.py:599-611, image_encoder.py:48-59);
  - window_partition_index: SAM's pad + window_partition / window_unpartition
    (image_encoder.py:329-384).
"""
import torch


def conv3d_gather_index(G, T, H, W, kt=3, kh=3, kw=3, frame_rows=None, row_offset=0):
    """int32 [kt*kh*kw, G*T*H*W]: source row of tap (dt,dh,dw) for output position m, -1 if the
    tap falls into the zero padding. Positions are ordered (g, t, h, w); frame f = g*T + t starts
    at row f*frame_rows + row_offset of the activation (frame_rows defaults to H*W)."""
    if frame_rows is None:
        frame_rows = H * W
    g = torch.arange(G).view(G, 1, 1, 1)
    t = torch.arange(T).view(1, T, 1, 1)
    h = torch.arange(H).view(1, 1, H, 1)
    w = torch.arange(W).view(1, 1, 1, W)
    taps = []
    for dt in range(kt):
        for dh in range(kh):
            for dw in range(kw):
                tt, hh, ww = t + dt - kt // 2, h + dh - kh // 2, w + dw - kw // 2
                ok = (tt >= 0) & (tt < T) & (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
                row = (g * T + tt) * frame_rows + row_offset + hh * W + ww
                row = torch.where(ok, row, torch.full_like(row, -1))
                taps.append(row.expand(G, T, H, W).reshape(-1))
    return torch.stack(taps, 0).to(torch.int32).contiguous()


def frame_rows_index(F, rows, frame_rows, row_offset):
    """int32 [F*rows]: row f*frame_rows + row_offset + r for (f, r) — addresses the patch tokens of
    a [F, frame_rows, C] activation that carries extra leading tokens (CLIP's CLS)."""
    f = torch.arange(F).view(F, 1)
    r = torch.arange(rows).view(1, rows)
    return (f * frame_rows + row_offset + r).reshape(-1).to(torch.int32).contiguous()


def window_partition_index(F, H, W, ws):
    """SAM window partition with bottom/right zero padding.
    Returns (tok2win int32 [F*H*W], win2tok int32 [F*nw*ws*ws], n_windows_per_frame, Hp, Wp):
    tok2win[token row] = row in the windowed layout [F*nwin, ws*ws]; win2tok is the inverse with
    -1 for padding rows."""
    Hp = (H + ws - 1) // ws * ws
    Wp = (W + ws - 1) // ws * ws
    nh, nw = Hp // ws, Wp // ws
    f = torch.arange(F).view(F, 1, 1)
    y = torch.arange(H).view(1, H, 1)
    x = torch.arange(W).view(1, 1, W)
    win = (f * nh + y // ws) * nw + x // ws
    tok2win = (win * ws + y % ws) * ws + x % ws
    tok2win = tok2win.reshape(-1)
    win2tok = torch.full((F * nh * nw * ws * ws,), -1, dtype=torch.int64)
    win2tok[tok2win] = torch.arange(F * H * W)
    return tok2win.to(torch.int32).contiguous(), win2tok.to(torch.int32).contiguous(), nh * nw, Hp, Wp
