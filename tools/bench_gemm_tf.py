"""GEMM variants on the teacher-forced step's decoder-stream shapes (M = 16 x 513 = 8208 rows) and a few ragged neighbours (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib, ops
dev = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 3, 4]
def run(M, N, K, iters=30):
    dt = torch.bfloat16
    a = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    res = []
    for v in variants:
        _lib.lib().acai_gemm_set_variant(v)
        ops.gemm_nt(a, w, b, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.gemm_nt(a, w, b, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append(f"v{v}: {ms*1e3:6.1f} us {2*M*N*K/ms/1e9:5.0f} TF")
    _lib.lib().acai_gemm_set_variant(0)
    print(f"M={M:6d} N={N:5d} K={K:5d}  " + " | ".join(res), flush=True)
for M in (8208, 8192, 4104, 12000, 16416, 20000):
    for (N, K) in ((1024, 1024), (1024, 4096), (3072, 1024), (4096, 1024), (2048, 1024)):
        run(M, N, K)
