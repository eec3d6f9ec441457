"""Debug aid: dq / dk / dv of the d_h = 64 backward through the wide kernels (ACAI_ATTN64_BWD_WIDE as set) saved to a file, or two saved files compared.
python tools/dbg_bwd64w.py run out.pt [lq lk H] | python tools/dbg_bwd64w.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "run":
    from acai_omr_amd import engine, ops
    lq, lk, H = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (256, 400, 1)
    dev, bf, dh = "cuda", torch.bfloat16, int(sys.argv[6]) if len(sys.argv) > 6 else 64
    E = H * dh
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(lq, E, generator=g) * ops.QSCALE(dh)).to(dev).to(bf)
    k = torch.randn(lk, E, generator=g).to(dev).to(bf)
    v = torch.randn(lk, E, generator=g).to(dev).to(bf)
    do = torch.randn(lq, E, generator=g).to(dev).to(bf)
    cu_q, cu_k = engine.cu_from_lens([lq], dev), engine.cu_from_lens([lk], dev)
    lse = torch.empty(H * lq, device=dev)
    o = ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, lq, lse=lse, q_prescaled=True)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attn_varlen_bwd(q, k, v, o, do, lse, cu_q, cu_k, H, dh, lq, lk, False, dq, dk, dv, q_prescaled=True)
    torch.cuda.synchronize()
    torch.save({"dq": dq.float().cpu(), "dk": dk.float().cpu(), "dv": dv.float().cpu()}, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for n in ("dq", "dk", "dv"):
        x, y = a[n], b[n]
        d = (x - y).abs()
        print(n, "max|a|", float(x.abs().max()), "max|b|", float(y.abs().max()), "max diff", float(d.max()), "nan", int(torch.isnan(x).sum()), int(torch.isnan(y).sum()))
        rb = d.reshape(-1, 32, d.shape[1] // 32, 32).amax(dim=(1, 3)) if d.shape[0] % 32 == 0 else None
        if rb is not None:
            print("  per (32-row block, 32-col block) max diff:\n", rb[:16])
