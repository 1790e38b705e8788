"""PMC target: the pipelined NT GEMM and the pipelined TN (weight-gradient) GEMM on the step's big shapes, 3 launches each.
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d out -o x --output-format csv -- python3 tools/pmc_gemm_nt_tn.py"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from grove_amd import ops
from grove_amd.model.indexing import conv3d_gather_index
dev = torch.device("cuda:0")
bf = torch.bfloat16
a = torch.randn(32768, 5120, device=dev).to(bf); w = torch.randn(1280, 5120, device=dev).to(bf)
for _ in range(3): ops.linear(a, w)
K2, M2, Ci = 32768, 1280, 1280
dy = torch.randn(K2, M2, device=dev).to(bf)
x27 = torch.randn(K2, 27 * Ci, device=dev).to(bf)
o = torch.zeros(M2, 27 * Ci, dtype=torch.float32, device=dev)
for _ in range(3): ops.wgrad(dy, x27, o)
x = torch.randn(K2, Ci, device=dev).to(bf)
idx = conv3d_gather_index(2, 16, 32, 32).to(dev)
for _ in range(3): ops.wgrad(dy, x, o, b_idx=idx, b_taps=27)
torch.cuda.synchronize()
print("done")
