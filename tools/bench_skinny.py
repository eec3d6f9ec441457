"""Micro-benchmark of the decode GEMV configurations (run on the GPU box): time per launch inside a hipGraph of 48 launches
cycling over 12 different weight matrices (so weights stream from HBM as in a real step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import ops

dev = "cuda"
torch.manual_seed(0)
B = 8


def run(name, N, K, xdt, ln, res, ydt=torch.float32, gelu=False):
    Ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(12)]
    x = torch.randn(B, K, device=dev).to(xdt)
    bias = torch.randn(N, device=dev)
    lnp = (torch.randn(K, device=dev), torch.randn(K, device=dev)) if ln else None
    stats = torch.zeros(B, 2, device=dev)
    r = torch.randn(B, N, device=dev) if res else None
    rln = (torch.randn(N, device=dev), torch.randn(N, device=dev)) if res == 2 else None
    rst = torch.rand(B, 2, device=dev) if res == 2 else None
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f = lambda i: ops.skinny_gemm_ex(x, Ws[i % 12], bias, residual=r, gelu=gelu, round_bf16=True, out_dtype=ydt, ln=lnp, stats_out=stats if ln else None, rln=rln, rstats=rst)
        f(0)
        torch.cuda.synchronize()
        g = ops.Graph(); g.begin()
        for i in range(48):
            f(i)
        g.end()
        g.launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.launch()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 / 48 * 1e3
    print(f"{name:28s} N={N:5d} K={K:5d}  {us:6.2f} us/launch  {N*K*2/us/1e6:7.2f} TB/s(W)")


run("qkv (LN on load)", 3072, 1024, torch.float32, True, 0)
run("out (res+rLN)", 1024, 1024, torch.float32, False, 2)
run("out (plain res)", 1024, 1024, torch.float32, False, 1)
run("crossq (LN on load)", 1024, 1024, torch.float32, True, 0)
run("lin1 (LN, gelu, bf16 out)", 4096, 1024, torch.float32, True, 0, torch.bfloat16, True)
run("lin1 (LN, gelu, f32 out)", 4096, 1024, torch.float32, True, 0, torch.float32, True)
run("lin2 (bf16 x, res+rLN)", 1024, 4096, torch.bfloat16, False, 2)
run("lin2 (bf16 x, no res)", 1024, 4096, torch.bfloat16, False, 0)
run("lin2 (fp32 x, res+rLN)", 1024, 4096, torch.float32, False, 2)
run("unembed (LN)", 227, 1024, torch.float32, True, 0)
