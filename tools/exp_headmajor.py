"""Does a head-major K/V layout pay for the d_h = 32 attention?  Same work two ways: (a) the path's layout - q, k, v are column slices of one
[M, 3E] buffer (a head's row is 64 B at a 3072-B stride), (b) every (sequence, head) as its own contiguous [S, 32] matrix (H = 1, B x 16
sequences: rows 64 B apart).  python tools/exp_headmajor.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops
dev = torch.device("cuda", 0)
B, H, S, dh, dt = 32, 16, 4096, 32, torch.bfloat16
E = H * dh
g = torch.Generator().manual_seed(0)


def run(name, q, k, v, o_shape, cu, Hh, nseq_rows):
    lse = torch.empty(Hh * nseq_rows, device=dev)
    dout = torch.randn(o_shape, generator=g).to(dev).to(dt)
    fwd = lambda: ops.attn_varlen(q, k, v, cu, cu, Hh, dh, S, lse=lse)
    o = fwd()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    bwd = lambda: ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu, cu, Hh, dh, S, S, False, dq, dk, dv)
    bwd()
    for nm, fn in (("fwd", fwd), ("bwd", bwd)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print(f"{name:12s} {nm}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)


qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.5).to(dev).to(dt)
run("row-major", qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], (B * S, E), engine.cu_from_lens([S] * B, dev), H, B * S)
q = (torch.randn(B * H * S, dh, generator=g) * 0.5).to(dev).to(dt)
k = (torch.randn(B * H * S, dh, generator=g) * 0.5).to(dev).to(dt)
v = (torch.randn(B * H * S, dh, generator=g) * 0.5).to(dev).to(dt)
run("head-major", q, k, v, (B * H * S, dh), engine.cu_from_lens([S] * (B * H), dev), 1, B * H * S)
