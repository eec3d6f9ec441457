"""HBM / L2 traffic of a few MAE GEMM shapes: run under rocprofv3 --pmc (tools/pmc_gemm.sh); each shape is launched 4 times in the order below."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import ops
dev, bf = "cuda", torch.bfloat16
SHAPES = [("dec qkv fwd", 131072, 1536, 512, "plain"), ("dec lin1 fwd gelu", 131072, 3072, 512, "gelu"), ("dec lin2 fwd", 131072, 512, 3072, "plain"),
          ("enc qkv fwd", 32768, 2304, 768, "plain"), ("dec da gelu'", 131072, 3072, 512, "dgelu"), ("dec out +res", 131072, 512, 512, "res")]
if __name__ == "__main__":
    for name, M, N, K, form in SHAPES:
        a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if form == "res" else None
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if form == "res" else bf)
        pre = torch.empty(M, N, device=dev, dtype=bf) if form == "gelu" else None
        saved = torch.randn(M, N, device=dev).to(bf) if form == "dgelu" else None
        for _ in range(4):
            ops.gemm_nt(a, w, None if form == "dgelu" else b, residual=res, out=out, gelu=form == "gelu", round_bf16=True, pre_act=pre, gelu_grad_of=saved)
        torch.cuda.synchronize()
