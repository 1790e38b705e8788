"""TN weight-gradient kernel: product build vs ablation builds (tools/dev/build_tn_abl.sh), one subprocess per library."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from grove_amd import ops, _lib
from grove_amd.model.indexing import conv3d_gather_index
dev = torch.device("cuda:0")
L = _lib.lib()
def t(fn):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 2 * 1e3)
    return best
K2, M2, Ci = 32768, 1280, 1280
a = torch.randn(K2, M2, device=dev).to(torch.bfloat16)
bp = torch.randn(K2, 27 * Ci, device=dev).to(torch.bfloat16)
o = torch.zeros(M2, 27 * Ci, dtype=torch.float32, device=dev)
us = t(lambda: ops.wgrad(a, bp, o))
print("plain    (1280, 34560, 32768): %%8.1f us %%7.1f TF" %% (us, 2.0 * M2 * 27 * Ci * K2 / us / 1e6))
bg = torch.randn(K2, Ci, device=dev).to(torch.bfloat16)
idx = conv3d_gather_index(4, 8, 32, 32).to(dev)
us = t(lambda: ops.wgrad(a, bg, o, b_idx=idx, b_taps=27, b_frames=(1024, 8)))
print("gathered conv3d, tap skip     : %%8.1f us %%7.1f TF (of the un-skipped FLOPs)" %% (us, 2.0 * M2 * 27 * Ci * K2 / us / 1e6))
''' % ROOT
import glob
names = sys.argv[1:] or ["", "_tnabl1", "_tnabl2", "_tnabl3", "_tnabl4"]
for name in names:
    name = "" if name == "product" else name
    lib = os.path.join(ROOT, "grove_amd", "csrc", "libgrove_hip%s.so" % name)
    name = "" if name == "product" else name
    print("==", name or "product", flush=True)
    subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, GROVE_HIP_LIB=lib))
