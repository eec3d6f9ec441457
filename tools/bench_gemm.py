"""GEMM micro-benchmark (GPU box): TFLOP/s of acai_gemm_nt on the shapes the path uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import ops
dev = "cuda"
def run(M, N, K, dt, iters=20):
    a = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    ops.gemm_nt(a, w, b, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt(a, w, b, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{str(dt):16s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s")
for dt in (torch.bfloat16, torch.float32):
    run(32768, 2304, 768, dt); run(32768, 3072, 768, dt); run(32768, 768, 3072, dt); run(131072, 1536, 512, dt); run(131072, 3072, 512, dt); run(131072, 512, 3072, dt); run(4096, 4096, 4096, dt)
