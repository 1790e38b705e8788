import sys, os, torch
sys.path.insert(0, "/root/repo")
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
tag = "prev" if os.environ.get("GROVE_HIP_LIB") else "new "
for M, N, K, use_bias, use_res in [(32768, 5120, 1280, 0, 0), (32768, 5120, 1280, 1, 0), (32768, 1280, 1280, 0, 0), (32768, 1280, 1280, 1, 1), (32768, 1280, 5120, 1, 1), (32768, 3840, 1280, 1, 0), (2812, 4096, 4096, 0, 1), (18464, 4096, 1024, 1, 0), (2812, 12288, 4096, 0, 0), (2812, 4096, 11008, 0, 1)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16) if use_bias else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if use_res else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.linear(a, b, bias, residual=res, out=out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.linear(a, b, bias, residual=res, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(tag, M, N, K, "bias" if use_bias else "    ", "res" if use_res else "   ", f"{best:.1f} us", flush=True)
