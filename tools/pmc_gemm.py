"""HBM / L2 traffic of a few MAE GEMM shapes: run under rocprofv3 --pmc (tools/pmc_gemm.sh); each shape is launched 4 times in the order below."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import ops
dev, bf = "cuda", torch.bfloat16
SHAPES = [("dec qkv fwd", 131072, 1536, 512, False), ("dec lin1 fwd gelu", 131072, 3072, 512, True), ("dec lin2 fwd", 131072, 512, 3072, False),
          ("enc qkv fwd", 32768, 2304, 768, False)]
if __name__ == "__main__":
    for name, M, N, K, gelu in SHAPES:
        a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=bf)
        pre = torch.empty(M, N, device=dev, dtype=bf) if gelu else None
        for _ in range(4):
            ops.gemm_nt(a, w, b, out=out, gelu=gelu, round_bf16=True, pre_act=pre)
        torch.cuda.synchronize()
