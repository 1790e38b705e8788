"""Does the chip hold its clock under the pipelined GEMM? Loops one GEMM shape for ~6 s per case in a child thread while the parent samples
rocm-smi (sclk, socket power) once a second; prints the samples and the achieved TFLOP/s. Cases: the whole chip, and the same kernel
with its grid limited by the problem size (a quarter of the tiles -> a quarter of the CUs busy)."""
import subprocess, sys, threading, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
dev = torch.device("cuda:0")

def smi():
    out = subprocess.run(["rocm-smi", "-c", "-P", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Power (W)", "Socket Power", "junction", "Average Graphics"))]
    return " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep)

for (M, N, K) in [(32768, 1280, 5120), (8192, 1280, 5120), (2048, 2560, 5120)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    for _ in range(3): ops.linear(a, w)
    torch.cuda.synchronize()
    stop, count = [False], [0]
    def run():
        while not stop[0]:
            for _ in range(50): ops.linear(a, w)
            torch.cuda.synchronize(); count[0] += 50
    t = threading.Thread(target=run); t0 = time.time(); t.start()
    samples = []
    for _ in range(5):
        time.sleep(1.0); samples.append(smi())
    stop[0] = True; t.join(); dt = time.time() - t0
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"({M}, {N}, {K}) {tiles} tiles: {2.0 * M * N * K * count[0] / dt / 1e12:7.1f} TF/s over {dt:.1f} s")
    for s in samples: print("    ", s)
