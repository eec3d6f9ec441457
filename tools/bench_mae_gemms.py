"""Every GEMM shape of the MAE step (batch 32 x 512x2048: encoder on 32768 kept tokens d=768, decoder on 131072 tokens d=512) in its
epilogue form, forward / dX / dW, with the step's launch count per shape: time per launch, achieved TFLOP/s and its share of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import ops
dev = "cuda"
bf = torch.bfloat16


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


rows = []
def fwd(name, M, N, K, count, resid=False, gelu=False, out32=False, dgelu=False, scale=False):
    a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if resid else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if (resid or out32) else bf)
    pre = torch.empty(M, N, device=dev, dtype=bf) if gelu else None
    saved = torch.randn(M, N, device=dev).to(bf) if dgelu else None      # the kept pre-activation whose GELU derivative multiplies the product
    keep_pre = os.environ.get("ACAI_GELU_KEEP_PRE") == "1"   # round 3's form: the forward keeps the pre-activation, the backward evaluates gelu' (default: the forward keeps gelu', the backward multiplies)
    s = t(lambda: ops.gemm_nt(a, w, None if dgelu else b, residual=r, out=out, gelu=gelu, round_bf16=True, pre_act=pre if keep_pre else None,
                              gelu_grad_out=None if keep_pre else pre, gelu_grad_of=saved if keep_pre else None, times=None if keep_pre else saved,
                              col_scale=(N // 3, 0.18) if scale else None))
    rows.append((name, M, N, K, count, s))

def dw(name, M, N, K, count):   # dW[N, K] = dY[M, N]^T X[M, K]
    dy = torch.randn(M, N, device=dev).to(bf); x = torch.randn(M, K, device=dev).to(bf)
    s = t(lambda: ops.gemm(dy, x, trans_a=True, trans_w=True, out_dtype=torch.float32))
    rows.append((name, M, N, K, count, s))

Me, Md = 32768, 131072
for tag, M, d, L in (("enc", Me, 768, 12), ("dec", Md, 512, 8)):
    fwd(f"{tag} qkv fwd (q scale)", M, 3 * d, d, L, scale=True)
    fwd(f"{tag} out fwd (+res)", M, d, d, L, resid=True)
    fwd(f"{tag} lin1 fwd (gelu)", M, 3072, d, L, gelu=True)
    fwd(f"{tag} lin2 fwd (+res)", M, d, 3072, L, resid=True)
    fwd(f"{tag} dattn = dy Wo", M, d, d, L)
    fwd(f"{tag} dx = dqkv Wi (+res)", M, d, 3 * d, L, resid=True)
    fwd(f"{tag} da = dy W2 (gelu')", M, 3072, d, L, dgelu=True)
    fwd(f"{tag} dx = da W1 (+res)", M, d, 3072, L, resid=True)
    dw(f"{tag} dWi", M, 3 * d, d, L)
    dw(f"{tag} dWo", M, d, d, L)
    dw(f"{tag} dW1", M, 3072, d, L)
    dw(f"{tag} dW2", M, d, 3072, L)
tot = sum(c * s for *_, c, s in rows)
print(f"{'shape':28s} {'M':>7s} {'N':>5s} {'K':>5s} cnt   us/launch   TF/s   ms/step")
for name, M, N, K, c, s in rows:
    print(f"{name:28s} {M:7d} {N:5d} {K:5d} {c:3d} {s*1e6:10.1f} {2*M*N*K/s/1e12:7.0f} {c*s*1e3:8.2f}")
print(f"sum {tot*1e3:.1f} ms/step; flops {sum(2*M*N*K*c for _, M, N, K, c, s in rows)/1e12:.1f} T -> {sum(2*M*N*K*c for _, M, N, K, c, s in rows)/tot/1e12:.0f} TF/s")
