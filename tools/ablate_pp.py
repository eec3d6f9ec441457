"""Timing ablations of gemm_nt_pp_kernel (variant 7) through ACAI_GEMM_DEBUG bits (results are wrong in these modes; timing only).  The bits exist in
-DACAI_GEMM_ABLATE builds only: bash tools/build_variant.sh ablpp "-DACAI_GEMM_ABLATE" gemm.hip; ACAI_OMR_LIB=.../variants/ablpp.so python tools/ablate_pp.py
Bits:
1 no counted DMA waits, 2 no LDS-DMA issue, 4 no epilogue, 8 no MFMAs, 16 no fragment reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib, ops
dev, bf = "cuda", torch.bfloat16


def timed(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


modes = [(0, "full"), (32, "start-up skew"), (4, "no epilogue"), (1, "no waits"), (5, "no waits, no epi"), (2 | 1, "no DMA"), (2 | 1 | 4, "no DMA, no epi"), (2 | 1 | 4 | 16, "MFMA + barriers only"),
         (2 | 1 | 4 | 8, "reads + barriers only"), (2 | 1 | 4 | 8 | 16, "barriers only")]
shapes = [("dec qkv nobias", 131072, 1536, 512, dict(nobias=True)), ("dec qkv", 131072, 1536, 512, {}), ("dec qkv nobias", 131072, 1536, 512, dict(nobias=True)), ("dec qkv", 131072, 1536, 512, {}), ("dec lin1 gelu", 131072, 3072, 512, dict(gelu=True)), ("enc qkv", 32768, 2304, 768, {})]
if len(sys.argv) > 1:
    modes = [(int(b), "bits " + b) for b in sys.argv[1:]]
for name, M, N, K, kw in shapes:
    a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=bf)
    pre = torch.empty(M, N, device=dev, dtype=bf) if kw.get("gelu") else None
    fn = lambda: ops.gemm_nt(a, w, None if kw.get('nobias') else b, out=out, round_bf16=True, gelu=bool(kw.get("gelu")), pre_act=pre)
    _lib.lib().acai_gemm_set_variant(6)
    t6 = timed(fn)
    _lib.lib().acai_gemm_set_variant(7)
    print(f"{name} {M}x{N}x{K}: variant 6 {t6*1e6:.1f} us", flush=True)
    for bits, label in modes:
        os.environ["ACAI_GEMM_DEBUG"] = str(bits)
        t = timed(fn)
        print(f"    {label:28s} {t*1e6:9.1f} us  {2*M*N*K/t/1e12:7.0f} TF", flush=True)
    os.environ.pop("ACAI_GEMM_DEBUG")
    _lib.lib().acai_gemm_set_variant(0)
