"""The SAM Conv3d adapter (C = 1280, 4 groups of 8 frames of 32 x 32) stage by stage: the 27-tap implicit GEMMs (forward, dgrad, wgrad) against
the Winograd F(2x2x2, 3x3x3) pipeline (transforms + grouped NT / K-batched TN GEMM). HIP-event times on torch's current stream (the
launches' stream), best of 3 x 2.   python tools/bench_winograd.py [C] -> stdout (profiles/r06_winograd_bench.txt)"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
from grove_amd.model.indexing import conv3d_gather_index
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
geom = (4, 8, 32, 32)
if len(sys.argv) > 2 and sys.argv[2] == "clip":
    # the CLIP tower's adapter (modeling_clip.py:599-611) at the bench's 32 frames: forward only, CLS row per frame
    C, geom, fr = 1024, (4, 8, 16, 36), 577
    rows = 32 * fr
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(C, 27 * C, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(C, device=dev).to(torch.bfloat16)
    a = torch.tensor([0.1], device=dev)
    idx = conv3d_gather_index(4, 8, 16, 36, frame_rows=fr, row_offset=1).to(dev)
    from grove_amd.model.indexing import frame_rows_index
    patch_rows = frame_rows_index(32, 576, fr, 1).to(dev)
    y = torch.zeros_like(x)
    U = ops.wino3d_transform_weight(w)

    def t_(fn, n=3):
        best = 1e9
        for _ in range(3):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        return best
    d_ = t_(lambda: ops.linear(x, w, b, act=ops.ACT_RELU, scale_ptr=a, scale_tanh=True, a_idx=idx, a_taps=27, M=32 * 576, c_idx=patch_rows, out=y))
    w_ = t_(lambda: ops.wino3d_conv(x, U, geom, y, bias=b, act=ops.ACT_RELU, scale_ptr=a, scale_tanh=True, frames=(fr, 1)))
    print(f"CLIP adapter, 32 frames x (1 + 16 x 36) rows, C = 1024: direct {d_:.1f} us, winograd {w_:.1f} us")
    sys.exit(0)
rows = geom[0] * geom[1] * geom[2] * geom[3]
tiles = ops.wino3d_tiles(geom)
bf = torch.bfloat16


def timeit(fn, n=2):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


x = (torch.randn(rows, C, device=dev)).to(bf)
dz = (torch.randn(rows, C, device=dev)).to(bf)
w = (torch.randn(C, 27 * C, device=dev) * 0.02).to(bf)
b = torch.randn(C, device=dev).to(bf)
a = torch.tensor([0.1], device=dev)
idx = conv3d_gather_index(*geom).to(dev)
frames = (geom[2] * geom[3], geom[1])
y, pre = torch.empty_like(x), torch.empty_like(x)
gw = torch.zeros(C, 27 * C, dtype=torch.float32, device=dev)
direct_flops = 2.0 * rows * 27 * C * C
wino_flops = 2.0 * 64 * tiles * C * C
rep = {}
rep["direct fwd"] = timeit(lambda: ops.linear(x, w, b, act=ops.ACT_RELU, scale_ptr=a, scale_tanh=True, a_idx=idx, a_taps=27, M=rows, residual=x, aux=pre, a_frames=frames, out=y))
rep["direct dgrad"] = timeit(lambda: ops.linear(dz, w, a_idx=idx, a_taps=27, M=rows, residual=x, scale_ptr=a, scale_tanh=True, a_frames=frames, out=y))
rep["direct wgrad"] = timeit(lambda: ops.wgrad(dz, x, gw, b_idx=idx, b_taps=27, scale_ptr=a, scale_tanh=True, K=rows, b_frames=frames))
V = torch.empty(64, tiles, C, dtype=bf, device=dev)
dM = torch.empty_like(V)
Mh = torch.empty_like(V)
U = torch.empty(64, C, C, dtype=bf, device=dev)
dU = torch.empty(64, C, C, dtype=torch.float32, device=dev)
rep["wino input transform"] = timeit(lambda: ops.wino3d_transform_tokens(x, geom, 0, out=V))
rep["wino grad transform"] = timeit(lambda: ops.wino3d_transform_tokens(dz, geom, 1, out=dM))
rep["wino weight transform"] = timeit(lambda: ops.wino3d_transform_weight(w, out=U))
rep["wino grouped NT gemm"] = timeit(lambda: ops.gemm_raw(V, U, Mh, 64 * tiles, C, C, C, C, C, b_group=tiles))
rep["wino output transform"] = timeit(lambda: ops.wino3d_output(Mh, geom, y, bias=b, act=ops.ACT_RELU, scale_ptr=a, scale_tanh=True, residual=x, aux=pre))
rep["wino k-batched TN gemm"] = timeit(lambda: ops.wgrad(dM, V, dU, K=tiles, k_batches=64, sC_batch=C * C, overwrite=True, M=C, N=C))
rep["wino wgrad output"] = timeit(lambda: ops.wino3d_wgrad_output(dU, gw, scale_ptr=a, scale_tanh=True))
rep["wino fwd total (conv)"] = timeit(lambda: ops.wino3d_conv(x, ops.wino3d_transform_weight(w), geom, y, bias=b, act=ops.ACT_RELU, scale_ptr=a, scale_tanh=True, residual=x, aux=pre))
rep["wino wgrad total (V given)"] = timeit(lambda: ops.wino3d_wgrad(dz, V, geom, gw, scale_ptr=a, scale_tanh=True))
print(f"C = {C}, geometry {geom}: {rows} rows, {tiles} tiles; direct {direct_flops / 1e12:.2f} TFLOP, winograd {wino_flops / 1e12:.2f} TFLOP per pass")
for k, v in rep.items():
    extra = ""
    if k.startswith("direct"):
        extra = f"  {direct_flops / v / 1e6:7.1f} TF/s"
    elif "gemm" in k:
        extra = f"  {wino_flops / v / 1e6:7.1f} TF/s executed"
    elif "transform" in k or "output" in k:
        nbytes = {"wino input transform": (rows + 64 * tiles) * C * 2, "wino grad transform": (rows + 64 * tiles) * C * 2, "wino weight transform": (27 + 64) * C * C * 2,
                  "wino output transform": (64 * tiles + 3 * rows) * C * 2, "wino wgrad output": (64 * 4 + 27 * 8) * C * C}.get(k)
        if nbytes:
            extra = f"  {nbytes / v / 1e6:7.2f} TB/s algorithmic"
    print(f"{k:32s} {v:9.1f} us{extra}")
