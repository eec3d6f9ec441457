"""A/B of the GEMM variants on the MAE step's shapes: variant 0 (auto: what the step uses today) against 7 (ping-pong ring, register
epilogue), interleaved rounds in one process, median of rounds.  python tools/bench_pp.py [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib, ops
dev = "cuda"
bf = torch.bfloat16
variants = [int(v) for v in sys.argv[1:]] or [0, 7]


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def case(name, M, N, K, resid=False, gelu=False, dgelu=False, scale=False):
    a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if resid else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if resid else bf)
    pre = torch.empty(M, N, device=dev, dtype=bf) if gelu else None
    sv = torch.randn(M, N, device=dev).to(bf) if dgelu else None
    def fn():
        ops.gemm_nt(a, w, None if dgelu else b, residual=r, out=out, gelu=gelu, round_bf16=True, pre_act=pre, gelu_grad_of=sv,
                    col_scale=(N // 3, 0.18) if scale else None)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            _lib.lib().acai_gemm_set_variant(v)
            fn(); torch.cuda.synchronize()
            res[v].append(timed(fn, 6))
    _lib.lib().acai_gemm_set_variant(0)
    line = f"{name:26s} {M:7d} {N:5d} {K:5d}"
    for v in variants:
        t = sorted(res[v])[len(res[v]) // 2]
        line += f"   v{v}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:6.0f} TF"
    print(line, flush=True)


def dw(name, M, N, K):   # dW[N, K] = dY[M, N]^T X[M, K]
    dy = torch.randn(M, N, device=dev).to(bf); x = torch.randn(M, K, device=dev).to(bf)
    out = torch.zeros(N, K, device=dev)
    fn = lambda: ops.gemm(dy, x, trans_a=True, trans_w=True, out=out)
    fn(); torch.cuda.synchronize()
    ts = sorted(timed(fn, 6) for _ in range(5))
    print(f"{name:26s} {M:7d} {N:5d} {K:5d}   {ts[2]*1e6:8.1f} us {2*M*N*K/ts[2]/1e12:6.0f} TF", flush=True)


Me, Md = 32768, 131072
if "dw" in os.environ.get("ACAI_BENCH_PP", ""):
    for tag, M, d in (("enc", Me, 768), ("dec", Md, 512)):
        dw(f"{tag} dWi", M, 3 * d, d); dw(f"{tag} dWo", M, d, d); dw(f"{tag} dW1", M, 3072, d); dw(f"{tag} dW2", M, d, 3072)
    sys.exit(0)
for tag, M, d in (("enc", Me, 768), ("dec", Md, 512)):
    case(f"{tag} qkv fwd (scale)", M, 3 * d, d, scale=True)
    case(f"{tag} out fwd (+res)", M, d, d, resid=True)
    case(f"{tag} lin1 fwd (gelu)", M, 3072, d, gelu=True)
    case(f"{tag} lin2 fwd (+res)", M, d, 3072, resid=True)
    case(f"{tag} dattn = dy Wo", M, d, d)
    case(f"{tag} dx = dqkv Wi (+res)", M, d, 3 * d, resid=True)
    case(f"{tag} da = dy W2 (gelu')", M, 3072, d, dgelu=True)
    case(f"{tag} dx = da W1 (+res)", M, d, 3072, resid=True)
case("square 4096", 4096, 4096, 4096)
case("square 8192", 8192, 8192, 8192)
