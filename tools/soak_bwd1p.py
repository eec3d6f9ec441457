"""Soak of the one-pass attention backward: the same call repeated, dK / dV must come out bit-identical every time and dQ within the rounding of an
fp32 sum taken in another order; equal-length and ragged batches alternate.  python tools/soak_bwd1p.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import engine, ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev, bf, H, dh = torch.device("cuda", 0), torch.bfloat16, 16, 32
E = H * dh
cases = []
for lens in ([4096] * 8, [1024, 2304, 4096, 6400, 9216, 1500, 3164, 4900], [513, 40, 1300, 700]):
    g = torch.Generator().manual_seed(len(lens))
    tot = sum(lens)
    qkv = (torch.randn(tot, 3 * E, generator=g) * 0.5).to(dev).to(bf)
    dout = torch.randn(tot, E, generator=g).to(dev).to(bf)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    cu = engine.cu_from_lens(lens, dev)
    lse = torch.empty(H * tot, device=dev)
    o = ops.attn_varlen(q, k, v, cu, cu, H, dh, max(lens), lse=lse, q_prescaled=True)
    cases.append((lens, q, k, v, o, dout, lse, cu))
ref = {}
worst = 0.0
for r in range(rounds):
    for ci, (lens, q, k, v, o, dout, lse, cu) in enumerate(cases):
        d = torch.empty(q.shape[0], 3 * E, dtype=bf, device=dev)
        ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu, cu, H, dh, max(lens), max(lens), False, d[:, :E], d[:, E:2 * E], d[:, 2 * E:], q_prescaled=True)
        if ci not in ref:
            ref[ci] = d.clone()
            assert bool(torch.isfinite(d.float()).all())
            continue
        assert torch.equal(d[:, E:], ref[ci][:, E:]), (r, ci, "dK / dV changed between runs")
        dd = float((d[:, :E].float() - ref[ci][:, :E].float()).abs().max()) / float(ref[ci][:, :E].float().abs().max())
        worst = max(worst, dd)
        assert dd <= 2.0 ** -7, (r, ci, dd)
    if r % 20 == 19:
        torch.cuda.synchronize()
        print(f"round {r + 1}: ok (worst dQ run-to-run difference {worst:.2e} of its maximum)", flush=True)
torch.cuda.synchronize()
print("soak ok:", rounds, "rounds x", len(cases), "batches")
