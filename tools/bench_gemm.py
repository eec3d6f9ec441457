"""GEMM micro-benchmark (GPU box): TFLOP/s of acai_gemm_nt on the shapes the path uses, per kernel variant
(0 auto, 1 128x128, 3 256x128 three-stage, 4 persistent, 5 256x256).  python tools/bench_gemm.py [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib, ops
dev = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [0]
def run(M, N, K, dt, iters=20):
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
        res.append(f"v{v}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:6.1f} TF")
    _lib.lib().acai_gemm_set_variant(0)
    print(f"{str(dt):15s} M={M:6d} N={N:5d} K={K:5d}  " + " | ".join(res), flush=True)
for dt in (torch.bfloat16, torch.float32):
    run(32768, 2304, 768, dt); run(32768, 3072, 768, dt); run(32768, 768, 3072, dt); run(131072, 1536, 512, dt); run(131072, 2048, 512, dt); run(131072, 512, 2048, dt); run(4096, 4096, 4096, dt)
    run(8192, 8192, 8192, dt, iters=5)
