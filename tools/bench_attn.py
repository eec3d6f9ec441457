"""Varlen flash attention fwd / bwd timing on the path's shapes: python tools/bench_attn.py [iters]
(run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
only = os.environ.get("ACAI_BENCH_ATTN_ONLY")
for name, B, H, S, dh, dt in (("mae-decoder", 32, 16, 4096, 32, torch.bfloat16), ("mae-encoder", 32, 12, 1024, 64, torch.bfloat16),
                              ("tf-encoder", 16, 12, 4096, 64, torch.bfloat16),
                              ("omr-encoder-fp32", 8, 12, 4096, 64, torch.float32)):
    if only and name != only:
        continue
    E = H * dh
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.5).to(dev).to(dt)   # (the prescaled run reads the same numbers as an already scaled q)
    dout = torch.randn(B * S, E, generator=g).to(dev).to(dt)
    cu = engine.cu_from_lens([S] * B, dev)
    lse = torch.empty(H * B * S, device=dev)
    dqkv = torch.empty_like(qkv)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    for pre in (False, True):
        fwd = lambda: ops.attn_varlen(q, k, v, cu, cu, H, dh, S, lse=lse, q_prescaled=pre)
        o = fwd()
        bwd = lambda: ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu, cu, H, dh, S, S, False, dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:], q_prescaled=pre)
        bwd()
        res = {}
        for nm, fn in (("fwd", fwd), ("bwd", bwd)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            res[nm] = (time.perf_counter() - t0) / iters * 1e3
        scores = B * H * S * S
        print(f"{name}{' q-prescaled' if pre else ''}: fwd {res['fwd']:.3f} ms ({scores / res['fwd'] / 1e9:.2f} T scores/s, {4 * scores * dh / res['fwd'] / 1e9:.0f} TF)  "
              f"bwd {res['bwd']:.3f} ms ({10 * scores * dh / res['bwd'] / 1e9:.0f} TF algorithmic)", flush=True)
