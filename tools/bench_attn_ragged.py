"""The d_h = 32 attention backward on a RAGGED batch shaped like config 5's MAE decoder stream (images 256x1024 ... 768x3072 -> 1024 ... 9216
tokens): python tools/bench_attn_ragged.py   (ACAI_ATTN_BWD_1P=0: the two-kernel form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops

dev, bf, H, dh = torch.device("cuda", 0), torch.bfloat16, 16, 32
E = H * dh
g = torch.Generator().manual_seed(0)
sizes = [(256, 1024), (384, 1536), (512, 2048), (640, 2560), (768, 3072), (320, 1200), (448, 1808), (560, 2240)]
lens = [(h // 16) * (w // 16) for h, w in sizes] * 2
tot = sum(lens)
qkv = (torch.randn(tot, 3 * E, generator=g) * 0.5).to(dev).to(bf)
dout = torch.randn(tot, E, generator=g).to(dev).to(bf)
q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
cu = engine.cu_from_lens(lens, dev)
lse = torch.empty(H * tot, device=dev)
o = ops.attn_varlen(q, k, v, cu, cu, H, dh, max(lens), lse=lse, q_prescaled=True)
d = torch.empty_like(qkv)
bwd = lambda: ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu, cu, H, dh, max(lens), max(lens), False, d[:, :E], d[:, E:2 * E], d[:, 2 * E:], q_prescaled=True)
bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    bwd()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
scores = H * sum(l * l for l in lens)
print(f"ragged batch of {len(lens)} sequences ({min(lens)}..{max(lens)} tokens, {tot} in all): bwd {ms:.3f} ms ({scores / ms / 1e9:.2f} T scores/s)  "
      f"[ACAI_ATTN_BWD_1P={os.environ.get('ACAI_ATTN_BWD_1P', '1')}]")
