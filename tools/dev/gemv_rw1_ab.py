import sys, torch
sys.path.insert(0, "/root/repo")
from grove_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for mode in (5, 1, 5, 1):  # bit 2 set = two rows per wave (the round-3 form), clear = one row per wave (default)
    L.grove_gemv_set_mfma(mode)
    for N, K in [(4096, 4096), (4096, 11008)]:
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(8)]
        x = torch.randn(1, K, device=dev).to(torch.bfloat16); out = torch.empty(1, N, device=dev, dtype=torch.bfloat16)
        for w in ws: ops.gemv(x, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for w in ws: ops.gemv(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"mode {mode} N={N} K={K}: {us:6.1f} us {N*K*2/us/1e3:7.1f} GB/s", flush=True)
L.grove_gemv_set_mfma(1)
