"""Teacher-forced decoder attention shapes (bf16, d_h = 64, 16 heads, batch 16, q prescaled): cross attention 513 queries x 4096 keys, causal
self attention 513 x 513, next to the encoder's 4096 x 4096 - forward and backward, T scores/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops
dev, bf = "cuda", torch.bfloat16
H, dh, B = 16, 64, 16
E = H * dh


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for name, lq, lk, causal in (("encoder self", 4096, 4096, False), ("decoder cross", 513, 4096, False), ("decoder cross (512 q)", 512, 4096, False), ("decoder self causal", 513, 513, True)):
    q = (torch.randn(B * lq, E, device=dev) * 0.7 * ops.QSCALE(dh)).to(bf)
    k = (torch.randn(B * lk, E, device=dev) * 0.7).to(bf)
    v = (torch.randn(B * lk, E, device=dev) * 0.7).to(bf)
    do = torch.randn(B * lq, E, device=dev).to(bf)
    cu_q, cu_k = engine.cu_from_lens([lq] * B, dev), engine.cu_from_lens([lk] * B, dev)
    lse = torch.empty(H * B * lq, device=dev)
    o = ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, lq, causal=causal, lse=lse, q_prescaled=True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    tf = timed(lambda: ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, lq, causal=causal, lse=lse, q_prescaled=True))
    tb = timed(lambda: ops.attn_varlen_bwd(q, k, v, o, do, lse, cu_q, cu_k, H, dh, lq, lk, causal, dq, dk, dv, q_prescaled=True))
    scores = B * H * lq * lk * (0.5 if causal else 1.0)
    print(f"{name:24s} fwd {tf*1e6:8.1f} us ({scores/tf/1e12:5.2f} T scores/s)   bwd {tb*1e6:8.1f} us ({scores/tb/1e12:5.2f} T scores/s)", flush=True)
