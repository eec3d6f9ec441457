"""Randomised comparison of the one-pass d_h = 32 attention backward with the two-kernel form on the same inputs (in one process: the second call
lends no workspace): python tools/stress_bwd1p.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops

cases, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = random.Random(seed)
dev, bf, dh = "cuda", torch.bfloat16, 32
worst = 0.0
for c in range(cases):
    B, H = rng.randint(1, 5), rng.randint(1, 4)
    cross = rng.random() < 0.4
    pick = lambda: rng.choice([rng.randint(1, 80), rng.randint(400, 700), 512, 513, 1024, rng.randint(900, 2100), 511, 64, 1536])
    lens_q = [pick() for _ in range(B)]
    lens_k = [pick() for _ in range(B)] if cross else list(lens_q)
    lens_q[rng.randrange(B)] = rng.choice([512, 600, 1200])     # the form needs max_q, max_k >= 512
    if not cross:
        lens_k = list(lens_q)
    else:
        lens_k[rng.randrange(B)] = rng.choice([512, 700, 1536, 2049])
    E = H * dh
    g = torch.Generator().manual_seed(seed * 1000 + c)
    q = (torch.randn(sum(lens_q), E, generator=g) * ops.QSCALE(dh)).to(dev).to(bf)
    kv = torch.randn(sum(lens_k), 2 * E, generator=g).to(dev).to(bf)
    k, v = kv[:, :E], kv[:, E:]
    do = torch.randn(sum(lens_q), E, generator=g).to(dev).to(bf)
    cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
    lse = torch.empty(H * sum(lens_q), device=dev)
    o = ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, max(lens_q), lse=lse, q_prescaled=True)
    outs = []
    for lend in (True, False):
        dq, dkv = torch.full_like(q, 7.0), torch.full_like(kv, 7.0)
        ops.attn_varlen_bwd(q, k, v, o, do, lse, cu_q, cu_k, H, dh, max(lens_q), max(lens_k), False, dq, dkv[:, :E], dkv[:, E:], q_prescaled=True, lend_workspace=lend)
        outs.append((dq.float().cpu(), dkv.float().cpu()))
    for name, a, b in (("dq", outs[1][0], outs[0][0]), ("dkv", outs[1][1], outs[0][1])):
        assert int(((b == 7.0) & (a != 7.0)).sum()) == 0, (c, name, "rows left at the 7.0 fill", lens_q, lens_k)
        rel = float((a - b).abs().max()) / max(1e-6, float(a.abs().max()))
        worst = max(worst, rel)
        assert rel <= 2.0 ** -7, (c, name, rel, B, H, lens_q, lens_k)
    print(f"case {c}: B {B} H {H} q {lens_q} k {lens_k} ok", flush=True)
print("all", cases, "cases agree; worst relative difference", worst)
