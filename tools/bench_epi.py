"""Per-launch time of the two pipelined GEMM tiles (forced) on the step's shapes and epilogues: interleaved repeats, median.
Also prints which variant the cost model picks — the calibration data for grove_gemm_bf16's pp_cost()."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
bf = torch.bfloat16


def timed(f, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SHAPES = ((32768, 5120, 1280, "gelu_aux"), (32768, 1280, 5120, "bias_res"), (32768, 3840, 1280, "bias"), (32768, 1280, 1280, "bias_res"),
          (18464, 4096, 1024, "qgelu_aux"), (18464, 1024, 4096, "bias_res"), (18464, 3072, 1024, "bias"), (18464, 1024, 1024, "bias_res"),
          (2812, 22016, 4096, "plain"), (2812, 4096, 11008, "res"), (2812, 12288, 4096, "plain"), (2812, 4096, 4096, "res"),
          (2812, 4096, 22016, "plain"), (2812, 11008, 4096, "plain"), (32768, 1024, 128, "plain"))
print("M N K epilogue | us(256) us(192) | auto variant")
for M, N, K, epi in SHAPES:
    a = torch.randn(M, K, device=dev).to(bf)
    w = torch.randn(N, K, device=dev).to(bf)
    b = torch.randn(N, device=dev).to(bf) if epi != "plain" and epi != "res" else None
    r = torch.randn(M, N, device=dev).to(bf) if epi.endswith("res") else None
    aux = torch.empty(M, N, device=dev, dtype=bf) if epi.endswith("aux") else None
    act = ops.ACT_GELU if epi.startswith("gelu") else ops.ACT_QUICKGELU if epi.startswith("qgelu") else ops.ACT_NONE
    f = lambda: ops.linear(a, w, b, act=act, residual=r, aux=aux)
    res = {}
    for tm in (256, 193):
        L.grove_gemm_set_tile_m(tm)
        f(); torch.cuda.synchronize()
    for rep in range(5):
        for tm in (256, 193):
            L.grove_gemm_set_tile_m(tm)
            res.setdefault(tm, []).append(timed(f))
    L.grove_gemm_set_tile_m(0)
    f()
    var = L.grove_gemm_last_variant()
    m256, m192 = sorted(res[256])[2], sorted(res[193])[2]
    print(f"{M:6d} {N:5d} {K:5d} {epi:9s} | {m256:7.1f} {m192:7.1f} | {2.0 * M * N * K / min(m256, m192) / 1e6:7.1f} TF/s best | auto -> {var}", flush=True)
