"""Attention forward / backward on a RAGGED batch shaped like config 5's streams (images 256x1024 ... 768x3072 -> 1024 ... 9216 tokens):
python tools/bench_attn_ragged.py [H dh]   (default 16 32: the MAE decoder; 12 64: the encoder.  ACAI_ATTN_BWD_1P=0: the two-kernel d_h = 32
backward; ACAI_XCD_ORDER=0 / 1: the XCD-aware block order of the one-dimensional grids forced off / on)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops

dev, bf = torch.device("cuda", 0), torch.bfloat16
H, dh = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 32)
E = H * dh
g = torch.Generator().manual_seed(0)
sizes = [(256, 1024), (384, 1536), (512, 2048), (640, 2560), (768, 3072), (320, 1200), (448, 1808), (560, 2240)]
lens = [(h // 16) * (w // 16) for h, w in sizes] * 2
if os.environ.get("ACAI_RAGGED_SORT"):   # 1: longest first, -1: shortest first (what the batch order is worth)
    lens = sorted(lens, reverse=os.environ["ACAI_RAGGED_SORT"] == "1")
tot = sum(lens)
qkv = (torch.randn(tot, 3 * E, generator=g) * 0.5).to(dev).to(bf)
dout = torch.randn(tot, E, generator=g).to(dev).to(bf)
q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
cu = engine.cu_from_lens(lens, dev)
lse = torch.empty(H * tot, device=dev)
o = ops.attn_varlen(q, k, v, cu, cu, H, dh, max(lens), lse=lse, q_prescaled=True)
d = torch.empty_like(qkv)
bwd = lambda: ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu, cu, H, dh, max(lens), max(lens), False, d[:, :E], d[:, E:2 * E], d[:, 2 * E:], q_prescaled=True)
fwd = lambda: ops.attn_varlen(q, k, v, cu, cu, H, dh, max(lens), lse=lse, q_prescaled=True)
res = {}
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t0) / 10 * 1e3
scores = H * sum(l * l for l in lens)
print(f"ragged batch of {len(lens)} sequences ({min(lens)}..{max(lens)} tokens, {tot} in all), {H} heads of {dh}: fwd {res['fwd']:.3f} ms  bwd {res['bwd']:.3f} ms "
      f"({scores / res['bwd'] / 1e9:.2f} T scores/s)  [ACAI_ATTN_BWD_1P={os.environ.get('ACAI_ATTN_BWD_1P', '1')} ACAI_XCD_ORDER={os.environ.get('ACAI_XCD_ORDER', 'auto')} ACAI_RAGGED_SORT={os.environ.get('ACAI_RAGGED_SORT', '0')}]")
