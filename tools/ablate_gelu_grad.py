"""Where does the gelu' GEMM (C = round(dY W2) * gelu'(saved a), M = 131072, N = 3072, K = 512) spend its time?  ACAI_GEMM_DEBUG bits
(timing only): 4 no epilogue, 64 epilogue stores land in a 1 MB window (no HBM write stream), 128 no epilogue stores at all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib, ops
dev, bf = "cuda", torch.bfloat16
M, N, K = 131072, 3072, 512


def timed(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf)
saved = torch.randn(M, N, device=dev).to(bf)
out = torch.empty(M, N, device=dev, dtype=bf)
forms = {"gelu' (aux read + VALU + store)": lambda: ops.gemm_nt(a, w, None, out=out, round_bf16=True, gelu_grad_of=saved),
         "plain bf16 (store only)": lambda: ops.gemm_nt(a, w, None, out=out, round_bf16=True)}
for name, fn in forms.items():
    for bits, label in ((0, "full"), (64, "stores into a 1 MB window"), (128, "no stores"), (4, "no epilogue")):
        os.environ["ACAI_GEMM_DEBUG"] = str(bits)
        t = timed(fn)
        print(f"{name:34s} {label:28s} {t*1e6:8.1f} us", flush=True)
    os.environ.pop("ACAI_GEMM_DEBUG")
