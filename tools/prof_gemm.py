"""One GEMM shape, a few launches (for rocprofv3 --pmc passes): python tools/prof_gemm.py M N K [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm_nt(a, w, b, out=out)
torch.cuda.synchronize()
