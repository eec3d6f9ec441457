"""Timing ablations of attn_fwd64_kernel (results are wrong in these modes): needs the -DACAI_ATTN64_ABLATE build
(bash tools/build_variant.sh abl64 "-fno-slp-vectorize -DACAI_ATTN64_ABLATE" attn_fwd64.hip; ACAI_OMR_LIB=.../variants/abl64.so).
Bits: 1 no exp2, 2 no pack, 4 no LDS fragment reads, 8 no staging / barrier, 16 no S MFMAs, 32 no P V MFMAs, 64 no row-sum MFMAs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops
dev, bf = "cuda", torch.bfloat16
B, H, S, dh = 16, 12, 4096, 64
E = H * dh
g = torch.Generator(device="cpu").manual_seed(0)
q = (torch.randn(B * S, E, generator=g) * 0.7 * ops.QSCALE(dh)).to(dev).to(bf)
k = (torch.randn(B * S, E, generator=g) * 0.7).to(dev).to(bf)
v = (torch.randn(B * S, E, generator=g) * 0.7).to(dev).to(bf)
cu = engine.cu_from_lens([S] * B, dev)
lse = torch.empty(H * B * S, device=dev)
out = torch.empty(B * S, E, device=dev, dtype=bf)


def timed(iters=10):
    fn = lambda: ops.attn_varlen(q, k, v, cu, cu, H, dh, S, lse=lse, out=out, q_prescaled=True)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


modes = [(0, "full"), (64, "no row-sum MFMA"), (1, "no exp2"), (3, "no exp2, no pack"), (4, "no LDS fragment reads"), (8, "no staging / barrier"), (12, "no LDS reads, no staging"),
         (7, "no exp / pack / LDS reads"), (15, "MFMAs only"), (79, "S and PV MFMAs only"), (16, "no S MFMAs"), (32, "no PV MFMAs"), (48, "no S / PV MFMAs"), (112, "no MFMAs at all"),
         (124, "exp2 + pack only"), (120, "exp2 + pack + LDS reads"), (76, "MFMAs + exp2 + pack"), (0, "full")]
if len(sys.argv) > 1:
    modes = [(int(x), "bits " + x) for x in sys.argv[1:]]
scores = B * H * S * S
for bits, label in modes:
    os.environ["ACAI_ATTN64_ABL"] = str(bits)
    t = timed()
    # cycles per 32-key block iteration of one wave, were it alone on its SIMD at 2.0 GHz: 1024 scores per block iteration
    print(f"{label:32s} {t*1e6:8.1f} us  {scores/t/1e12:5.2f} T scores/s   {t * 2.0e9 * 1024 / (scores / 1024):7.0f} SIMD-cycles @2GHz per 1024 scores", flush=True)
