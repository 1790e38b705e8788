"""TEST INFRASTRUCTURE (oracle/): fp32 torch restatement of the Winograd F(2x2x2, 3x3x3) algebra that grove_amd/csrc/winograd.hip and the
grouped / K-batched GEMMs implement for the Conv3d adapters (reference: model/SAM/modeling/image_encoder.py:43-59,
nn.Conv3d(C, C, 3, padding=1) on '(b t) h w c -> b c t h w'). The reference itself is the DIRECT convolution (oracle/grove_oracle.py
conv_adapter, F.conv3d); this file only states the minimal-filtering identity  Y = A^T[(G g G^T) (.) (B^T d B)]A  (Lavin & Gray 2015,
applied along t, y and x) so that the tests can check every intermediate tensor of the HIP path, and tools/winograd_study.py can price
its two extra bf16 rounding points. Only tests/ and tools/ import it; the product path never does.
"""
import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])


def r16(x):
    return x.to(torch.bfloat16).float()


def t3(m, x):
    """apply matrix m [o, i] along each of the last three dims of x"""
    return torch.einsum("ai,bj,ck,...ijk->...abc", m, m, m, x)


def tiles_in(x5):
    """[B, C, T, H, W] -> overlapping 4x4x4 input tiles of the zero-padded tensor, stride 2: [B, C, nt, nh, nw, 4, 4, 4]"""
    xp = F.pad(x5, (1, 1, 1, 1, 1, 1))
    return xp.unfold(2, 4, 2).unfold(3, 4, 2).unfold(4, 4, 2)


def tiles_out(y5):
    """[B, C, T, H, W] -> disjoint 2x2x2 output tiles [B, C, nt, nh, nw, 2, 2, 2]"""
    return y5.unfold(2, 2, 2).unfold(3, 2, 2).unfold(4, 2, 2)


def untile_out(yt):
    B, C, nt, nh, nw = yt.shape[:5]
    return yt.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(B, C, 2 * nt, 2 * nh, 2 * nw)


def input_transform(x5):
    """V = (B^T (x) B^T (x) B^T) d per tile: [B, C, nt, nh, nw, 4, 4, 4] fp32"""
    return t3(BT, tiles_in(x5))


def weight_transform(w):
    """U = (G (x) G (x) G) g: [Co, Ci, 3, 3, 3] -> [Co, Ci, 4, 4, 4]"""
    return t3(G, w)


def grad_transform(g5):
    """dM = (A (x) A (x) A) dY per 2x2x2 tile -> [B, C, nt, nh, nw, 4, 4, 4]"""
    return t3(AT.t().contiguous(), tiles_out(g5))


def wino_conv(x5, w, m16=False, round_operands=True):
    """Conv3d 3x3x3 'same' (no bias) through F(2,3)^3; operands rounded to bf16 after their transforms (what the MFMA reads), the
    products optionally too (m16: a GEMM with a bf16 output). Returns (y5, V)."""
    q = r16 if round_operands else (lambda t: t)
    V = q(input_transform(x5))
    U = q(weight_transform(w))
    M = torch.einsum("bcthwxyz,ocxyz->bothwxyz", V, U)
    if m16:
        M = r16(M)
    return untile_out(t3(AT, M)), V


def wino_wgrad(V, g5, round_operands=True):
    """dW [Co, Ci, 3, 3, 3] from the transformed input V and the output gradient g5 [B, Co, T, H, W]"""
    dM = grad_transform(g5)
    if round_operands:
        dM = r16(dM)
    dU = torch.einsum("bothwxyz,bcthwxyz->ocxyz", dM, V)
    return t3(G.t().contiguous(), dU)


# ---- layouts of the HIP path (include/grove_hip.h "grove_wino3d_*")
def tokens_to_5d(x, geom):
    g, T, H, W = geom
    return x.reshape(g, T, H, W, -1).permute(0, 4, 1, 2, 3)


def to_point_major(Vt):
    """[B, C, nt, nh, nw, 4, 4, 4] -> [64, tiles, C] (point = (a 4 + b) 4 + c, tile = ((g nt + tt) nh + ty) nw + tx)"""
    B, C, nt, nh, nw = Vt.shape[:5]
    return Vt.permute(5, 6, 7, 0, 2, 3, 4, 1).reshape(64, B * nt * nh * nw, C)


def weight_to_point_major(U):
    """[Co, Ci, 4, 4, 4] -> [64, Co, Ci]"""
    return U.permute(2, 3, 4, 0, 1).reshape(64, U.shape[0], U.shape[1])
