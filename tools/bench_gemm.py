"""NT GEMM micro-benchmarks on the shapes of the full-size step (one script, three sub-commands; random data):
    python tools/bench_gemm.py tiles       auto kernel selection vs every forced variant (calibrates the cost model of grove_gemm_bf16)
    python tools/bench_gemm.py streamk     stream-K tail of the persistent kernels on vs off (time, TF/s, |diff| between the arms, error vs fp32)
    python tools/bench_gemm.py waves       four-wave form of the 256-row instances vs the eight-wave form (bit equality, time)
    python tools/bench_gemm.py epilogues   epilogue variants of the persistent kernel vs the 128-row kernel (plain / GELU+aux / QuickGELU+aux / residual)
(round 3: merged from bench_gemm_tiles.py, bench_gemm_streamk.py, bench_gemm_pipelined.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
bf = torch.bfloat16
STEP_SHAPES = [(32768, 5120, 1280), (32768, 1280, 5120), (32768, 3840, 1280), (32768, 1280, 3840), (32768, 1280, 1280), (2812, 22016, 4096),
               (2812, 4096, 22016), (2812, 12288, 4096), (2812, 11008, 4096), (2812, 4096, 12288), (2812, 4096, 11008), (18464, 4096, 1024),
               (2812, 4096, 4096), (18464, 1024, 4096), (18464, 3072, 1024), (18464, 1024, 1024), (32768, 4608, 1280), (32768, 1280, 4608),
               (2304, 4096, 4096), (256, 32008, 4096), (98304, 256, 128), (576, 256, 256), (1000, 520, 192), (777, 1000, 128), (300, 264, 64),
               (5000, 776, 64), (70000, 512, 128), (777, 1000, 2048), (4000, 1300, 8192)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def operands(M, N, K):
    a = torch.randn(M, K, device=dev).to(bf)
    b = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    return a, b, torch.randn(N, device=dev).to(bf), torch.empty(M, N, device=dev, dtype=bf)


def tiles():
    variants = [("auto", 0), ("192", 192), ("pp192", 193), ("pp256", 256)]
    for M, N, K in STEP_SHAPES:
        a, b, bias, out = operands(M, N, K)
        res = {v[0]: 1e9 for v in variants}
        ref = None
        for rnd_ in range(3):
            for name, tm in variants:
                L.grove_gemm_set_tile_m(tm)
                t = timed(lambda: ops.linear(a, b, bias, out=out), 5)
                if rnd_ == 0:
                    if name == "auto":
                        ref = out.clone()
                    else:  # bit-identical unless a stream-K split changed the fp32 sum order of some tiles (then: bf16 rounding)
                        err = (out.float() - ref.float()).abs().max().item()
                        assert err <= 2 ** -7 * ref.float().abs().max().item(), (name, M, N, K, err)
                res[name] = min(res[name], t)
        best = min(res[k] for k in ("192", "pp192", "pp256"))
        flag = "" if res["auto"] <= 1.04 * best else "   <-- auto misses"
        print(f"M={M} N={N} K={K}: " + "  ".join(f"{k}: {v:7.1f}us" for k, v in res.items()) + f"  ({2.0 * M * N * K / res['auto'] / 1e6:.0f} TF auto)" + flag, flush=True)
    L.grove_gemm_set_tile_m(0)


def streamk():
    tot = {0: 0.0, 1: 0.0}
    for M, N, K in STEP_SHAPES:
        a, b, bias, _ = operands(M, N, K)
        outs, res, S, var = {}, {0: 1e9, 1: 1e9}, 0, 0
        for _ in range(3):
            for on in (0, 1):
                L.grove_gemm_set_stream_k(on)
                out = torch.empty(M, N, device=dev, dtype=bf)
                res[on] = min(res[on], timed(lambda: ops.linear(a, b, bias, out=out), 10))
                if on:
                    S = L.grove_gemm_last_stream_k()
                var = L.grove_gemm_last_variant()
                outs[on] = out
        d = (outs[0].float() - outs[1].float()).abs().max().item()
        rows = slice(0, min(M, 2048))
        ref = a[rows].float() @ b.float().t() + bias.float()
        err = ((outs[1][rows].float() - ref).abs().max() / ref.abs().max()).item()
        tot[0] += res[0]
        tot[1] += res[1]
        print(f"M={M} N={N} K={K} variant {var} S={S}: off {res[0]:7.1f}us  on {res[1]:7.1f}us  ({2.0 * M * N * K / res[1] / 1e6:.0f} TF on, {res[0] / res[1]:.3f}x)  "
              f"|on-off| {d:.3g}  rel err vs fp32 {err:.2e}", flush=True)
    L.grove_gemm_set_stream_k(1)
    print(f"sum: off {tot[0]:.0f}us on {tot[1]:.0f}us")


def epilogues():
    def run(M, N, K, tm, **kw):
        L.grove_gemm_set_tile_m(tm)
        a, b, bias, out = operands(M, N, K)
        args = dict(bias=bias)
        if kw.get("aux"):
            args["aux"] = torch.empty(M, N, device=dev, dtype=bf)
        if kw.get("res"):
            args["residual"] = torch.randn(M, N, device=dev).to(bf)
        if kw.get("act"):
            args["act"] = kw["act"]
        best = min(timed(lambda: ops.linear(a, b, out=out, **args), 5) for _ in range(3))
        return best, out, args.get("aux")
    for (M, N, K) in [(32768, 5120, 1280), (18464, 4096, 1024), (32768, 1280, 5120)]:
        for name, kw in [("plain", {}), ("gelu+aux", dict(act=ops.ACT_GELU, aux=True)), ("qgelu+aux", dict(act=ops.ACT_QUICKGELU, aux=True)), ("residual", dict(res=True))]:
            torch.manual_seed(0)
            t0, o0, x0 = run(M, N, K, 128, **kw)
            torch.manual_seed(0)
            t1, o1, x1 = run(M, N, K, 0, **kw)
            d = (o0.float() - o1.float()).abs().max().item()
            dx = (x0.float() - x1.float()).abs().max().item() if x0 is not None else 0.0
            print(f"M={M} N={N} K={K} {name:10s}: 128-row {t0:7.1f}us  auto {t1:7.1f}us  maxdiff {d:.3g} aux {dx:.3g}", flush=True)
    L.grove_gemm_set_tile_m(0)


def waves():
    """gemm_nt_w4_kernel (one wave per SIMD, hand-scheduled K loop; EXPERIMENT: needs a library built with `make -C grove_amd/csrc W4=1`,
    e.g. into another file passed as GROVE_HIP_LIB) against the eight-wave kernel: same bits, time of both."""
    assert hasattr(L, "grove_gemm_set_waves"), "this library was built without the experiment: make -C grove_amd/csrc W4=1"
    shapes = [(32768, 1280, 5120), (32768, 5120, 1280), (32768, 3840, 1280), (32768, 1280, 3840), (32768, 1280, 1280), (32768, 4608, 1280),
              (32768, 1280, 4608), (2816, 22016, 4096), (2816, 4096, 22016), (18432, 4096, 1024), (18432, 1024, 4096), (256, 256, 64), (512, 768, 192),
              (4096, 4096, 4096), (8192, 8192, 8192)]
    L.grove_gemm_set_tile_m(256)
    tot = {8: 0.0, 4: 0.0}
    for M, N, K in shapes:
        a, b, bias, _ = operands(M, N, K)
        res_in = torch.randn(M, N, device=dev).to(bf)
        for label, kw in (("plain", {}), ("gelu+aux+res", dict(act=ops.ACT_GELU, residual=res_in, aux=torch.empty(M, N, device=dev, dtype=bf)))):
            outs, t = {}, {8: 1e9, 4: 1e9}
            for rnd_ in range(3):
                for w in (8, 4):
                    L.grove_gemm_set_waves(w)
                    out = torch.empty(M, N, device=dev, dtype=bf)
                    t[w] = min(t[w], timed(lambda: ops.linear(a, b, bias, out=out, **kw), 10))
                    outs[w] = out
                    if "aux" in kw:
                        outs[(w, "aux")] = kw["aux"].clone()
            same = torch.equal(outs[8], outs[4]) and ("aux" not in kw or torch.equal(outs[(8, "aux")], outs[(4, "aux")]))
            S = L.grove_gemm_last_stream_k()
            if label == "plain" and M >= 2048:
                tot[8] += t[8]
                tot[4] += t[4]
            print(f"M={M} N={N} K={K} {label:13s} 8 waves {t[8]:8.1f}us ({2.0 * M * N * K / t[8] / 1e6:6.0f} TF)  4 waves {t[4]:8.1f}us ({2.0 * M * N * K / t[4] / 1e6:6.0f} TF)"
                  f"  x{t[8] / t[4]:.3f}  stream-K S={S}  bit-equal={same}", flush=True)
            assert same or os.environ.get("W4_NO_ASSERT"), (M, N, K, label)
    L.grove_gemm_set_waves(8)
    L.grove_gemm_set_tile_m(0)
    print(f"sum over the plain large shapes: 8 waves {tot[8] / 1e3:.2f} ms, 4 waves {tot[4] / 1e3:.2f} ms")


if __name__ == "__main__":
    {"tiles": tiles, "streamk": streamk, "epilogues": epilogues, "waves": waves}[sys.argv[1] if len(sys.argv) > 1 else "tiles"]()
