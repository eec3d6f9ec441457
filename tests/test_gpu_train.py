"""Training-path parity on the GPU: forward + loss + gradients of the HIP autograd path against the golden vectors the
imported reference produced (tests/golden/{mae_*,tf_*}.pt), plus kernel-level checks of the backward kernels against
torch autograd on the CPU."""
import math

import pytest
import torch

from conftest import VOCAB, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from acai_omr_amd import _lib
    _lib.lib()
    return "cuda"


def md(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (128, 136, 64), (37, 19, 10), (513, 768, 256)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gemm_transposed_variants(dev, M, N, K, dtype):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(M * N + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    if dtype == "bf16":
        a, w = a.to(tdt).float(), w.to(tdt).float()
    ref = a.double() @ w.double().t()
    tol = 1e-4 if dtype == "fp32" else 1e-3
    ad, wd = a.to(dev).to(tdt), w.to(dev).to(tdt)
    for ta in (False, True):
        for tw in (False, True):
            A = ad.t().contiguous() if ta else ad
            W = wd.t().contiguous() if tw else wd
            y = ops.gemm(A, W, trans_a=ta, trans_w=tw)
            assert (y.cpu().double() - ref).abs().max() < tol, (ta, tw)


@pytest.mark.parametrize("H,dh,lens_q,lens_k,causal", [
    (2, 64, [8, 32, 200], None, False),
    (3, 32, [130, 1, 77], None, False),
    (2, 16, [65, 64], None, True),
    (1, 6, [5, 9, 3], None, False),
    (4, 12, [7, 12], [20, 13], False),
    (2, 64, [300], None, True),
    (2, 32, [333, 128], None, False),      # several full tiles in front of a ragged one
    (1, 64, [256], [400], False),
    # >= 512 queries and keys, d_h <= 32, bf16 prescaled, no mask: the two-blocks-per-wave backward kernels (ragged ends in both directions,
    # a workgroup whose last waves own no row, cross lengths)
    (2, 32, [513, 700], None, False),
    (2, 32, [600, 513], [1000, 577], False),
    (1, 24, [1025], [512], False),
    # d_h = 64, bf16 prescaled, no mask, >= 256 rows: the one-wave-per-SIMD kernels over the full 256-row blocks + the one-block kernels over the
    # rows past them (a one-row tail as in the decoder's 513 tokens, sequences shorter than a block, a ragged last key tile, cross lengths)
    (2, 64, [513, 700], None, False),
    (2, 64, [600, 256, 40], [1000, 577, 300], False),
    (1, 64, [1025], [512], False),
    (1, 64, [512], [330], False),
    (3, 64, [768], [129], False),
    # d_h = 32, bf16 prescaled, no mask, equal-length sequences with a multiple of 512 keys: the one-pass backward (attn_bwd1p.hip) against the
    # fp64 autograd reference - square, a query count that ends inside a 64-row tile, more keys than queries
    (2, 32, [1024, 1024], None, False),
    (3, 32, [600, 600], [1536, 1536], False),
    (2, 32, [700, 40, 1300], [1100, 300, 1025], False),      # ragged: partial key blocks, a short sequence, a one-key tail
])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("prescaled", [False, True])
def test_attn_backward(dev, H, dh, lens_q, lens_k, causal, dtype, prescaled):
    """prescaled: the kernels get q log2(e)/sqrt(dh) (the training path's in-projection epilogue) and still return the gradient with respect
    to the unscaled q."""
    from acai_omr_amd import engine, ops
    if prescaled and dh % (8 if dtype == "bf16" else 4):
        pytest.skip("the prescaled form exists for 16-byte-aligned heads only")
    lens_k = lens_k or lens_q
    g = torch.Generator().manual_seed(H * dh + sum(lens_q) + 7)
    E = H * dh
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    q = torch.randn(sum(lens_q), E, generator=g).to(tdt).float()
    k = torch.randn(sum(lens_k), E, generator=g).to(tdt).float()
    v = torch.randn(sum(lens_k), E, generator=g).to(tdt).float()
    dout = torch.randn(sum(lens_q), E, generator=g).to(tdt).float()
    qp = (q * ops.QSCALE(dh)).to(tdt)
    if prescaled:
        q = qp.float() / ops.QSCALE(dh)     # the reference differentiates with respect to the q the kernel effectively sees
    # reference: torch autograd in float64
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    out = torch.zeros(sum(lens_q), E, dtype=torch.float64)
    oq = ok = 0
    outs = []
    for lq, lk in zip(lens_q, lens_k):
        for h in range(H):
            sl = slice(h * dh, (h + 1) * dh)
            s = qr[oq:oq + lq, sl] @ kr[ok:ok + lk, sl].t() / math.sqrt(dh)
            if causal:
                s = s.masked_fill(~torch.ones(lq, lk, dtype=torch.bool).tril(), float("-inf"))
            outs.append((oq, lq, sl, torch.softmax(s, -1) @ vr[ok:ok + lk, sl]))
        oq += lq
        ok += lk
    loss = sum((o * dout[a:a + l, sl].double()).sum() for a, l, sl, o in outs)
    loss.backward()
    qd, kd, vd, dd = (t.to(dev).to(tdt) for t in (q, k, v, dout))
    if prescaled:
        qd = qp.to(dev)
    cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
    lse = torch.empty(H * sum(lens_q), device=dev)
    o = ops.attn_varlen(qd, kd, vd, cu_q, cu_k, H, dh, max(lens_q), causal=causal, lse=lse, q_prescaled=prescaled)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attn_varlen_bwd(qd, kd, vd, o, dd, lse, cu_q, cu_k, H, dh, max(lens_q), max(lens_k), causal, dq, dk, dv, q_prescaled=prescaled)
    tol = 3e-5 if dtype == "fp32" else 6e-2
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        err = (got.cpu().double() - ref).abs().max() / max(1.0, float(ref.abs().max()))
        assert err < tol, (name, float(err))


_BWD64W_SNIPPET = r"""
import sys, torch
from acai_omr_amd import engine, ops
dev, bf, dh, H = "cuda", torch.bfloat16, 64, 2
lens_q, lens_k = [600, 300, 256], [1000, 577, 256]
E = H * dh
g = torch.Generator().manual_seed(11)
q = (torch.randn(sum(lens_q), E, generator=g) * ops.QSCALE(dh)).to(dev).to(bf)
k = torch.randn(sum(lens_k), E, generator=g).to(dev).to(bf)
v = torch.randn(sum(lens_k), E, generator=g).to(dev).to(bf)
do = torch.randn(sum(lens_q), E, generator=g).to(dev).to(bf)
cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
lse = torch.empty(H * sum(lens_q), device=dev)
o = ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, max(lens_q), lse=lse, q_prescaled=True)
dq, dk, dv = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)
ops.attn_varlen_bwd(q, k, v, o, do, lse, cu_q, cu_k, H, dh, max(lens_q), max(lens_k), False, dq, dk, dv, q_prescaled=True)
torch.cuda.synchronize()
torch.save({"dq": dq.cpu(), "dk": dk.cpu(), "dv": dv.cpu()}, sys.argv[1])
"""


def test_attn_backward_wide_dh64_forms_equal_the_one_block_kernels(dev, tmp_path):
    """attn_bwd64w.hip (one wave per SIMD, two lane-owned blocks per wave): dQ is on by default, dK / dV is off (measured slower); both are
    the same arithmetic in the same summation order as attn_bwd.hip's one-block kernels, so every form must give the same gradients to the
    last place - ragged batch, tails past the last full 256-row block, a sequence of exactly one block, a ragged last streamed tile.  The form is chosen once
    per process (ACAI_ATTN64_BWD_WIDE), hence the child processes, one after the other."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("0", "1", "2", "3"):
        f = tmp_path / f"w{mode}.pt"
        env = dict(os.environ, ACAI_ATTN64_BWD_WIDE=mode, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", _BWD64W_SNIPPET, str(f)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = torch.load(f)
    for mode in ("1", "2", "3"):
        for n in ("dq", "dk", "dv"):
            a, b = outs["0"][n].float(), outs[mode][n].float()
            # full streamed tiles give the same bits (tools/dbg_bwd64w.py); in a ragged last tile the one-block kernels' general loop forms
            # 2^(c s - lse) from a zero-started accumulator where these start the accumulator at -lse: a last-place difference in a few elements
            d = (a - b).abs()
            assert float(d.max()) <= 2.0 ** -7 * float(a.abs().max()) and float((d > 0).float().mean()) < 2e-3, (mode, n, float(d.max()), float((d > 0).float().mean()))
    assert float(outs["0"]["dq"].float().abs().max()) < 6.0   # (every row was written: the 7.0 fill is gone)


_BWD1P_SNIPPET = r"""
import sys, torch
from acai_omr_amd import engine, ops
dev, bf, dh = "cuda", torch.bfloat16, 32
out = {}
for name, H, lens_q, lens_k in (("square", 3, [1024] * 3, [1024] * 3), ("ragged_last_tile", 2, [1000] * 2, [1536] * 2), ("smallest", 1, [512], [512]),
                               # ragged batches: keys past the last full 512-key block (the second launch), a sequence shorter than one block / one tile,
                               # a one-key tail, a sequence whose keys are all tail, fewer than 64 queries in the LAST sequence (the workspace's padding rows)
                               ("ragged", 2, [700, 40, 1300, 513], [1100, 300, 1025, 512]), ("ragged_short_last", 1, [600, 33], [640, 511])):
    E = H * dh
    B, lq, lk = len(lens_q), max(lens_q), max(lens_k)
    g = torch.Generator().manual_seed(sum(lens_q) + sum(lens_k))
    qkv = torch.randn(sum(lens_k), 3 * E, generator=g).to(dev).to(bf)     # k, v: strided views of one buffer, like the in-projection's output
    q = (torch.randn(sum(lens_q), E, generator=g) * ops.QSCALE(dh)).to(dev).to(bf)
    k, v = qkv[:, E:2 * E], qkv[:, 2 * E:]
    do = torch.randn(sum(lens_q), E, generator=g).to(dev).to(bf)
    cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
    lse = torch.empty(H * sum(lens_q), device=dev)
    o = ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, lq, lse=lse, q_prescaled=True)
    dq, dkv = torch.full_like(q, 7.0), torch.full_like(qkv, 7.0)
    for rep in range(2):   # (twice: the workspace is reused and must be re-zeroed by the call itself)
        ops.attn_varlen_bwd(q, k, v, o, do, lse, cu_q, cu_k, H, dh, lq, lk, False, dq, dkv[:, E:2 * E], dkv[:, 2 * E:], q_prescaled=True)
    torch.cuda.synchronize()
    out[name] = {"dq": dq.cpu(), "dk": dkv[:, E:2 * E].cpu(), "dv": dkv[:, 2 * E:].cpu(), "untouched": dkv[:, :E].cpu()}
torch.save(out, sys.argv[1])
"""


def test_attn_backward_one_pass_dh32_equals_the_two_kernel_form(dev, tmp_path):
    """attn_bwd1p.hip (bf16, d_h = 32, prescaled q, no mask, long sequences): P and dS are formed once per score and dQ is summed over the
    key blocks with fp32 atomics, so against the two-kernel form (ACAI_ATTN_BWD_1P=0) dK / dV agree to the last place and dQ to the rounding
    of an fp32 sum taken in another order.  Query counts that end inside a 64-row tile, strided k / v / dk / dv views, the smallest shape
    the form takes, ragged batches (partial key blocks, short sequences), and a second call over the same workspace.  The form is chosen
    once per process: child processes."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("0", "1"):
        f = tmp_path / f"p{mode}.pt"
        env = dict(os.environ, ACAI_ATTN_BWD_1P=mode, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", _BWD1P_SNIPPET, str(f)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = torch.load(f)
    for case in outs["0"]:
        two, one = outs["0"][case], outs["1"][case]
        assert float((one["untouched"].float() - 7.0).abs().max()) == 0.0, case
        for n in ("dq", "dk", "dv"):
            a, b = two[n].float(), one[n].float()
            assert float(b.abs().max()) < 6.0, (case, n)   # (every row was written: the 7.0 fill is gone)
            d = (a - b).abs()
            # one bf16 place of the larger magnitudes; dV comes from the same P packs in the same order
            assert float(d.max()) <= 2.0 ** -7 * float(a.abs().max()), (case, n, float(d.max()))
            if n == "dv":
                assert float((d > 0).float().mean()) < 1e-3, (case, float((d > 0).float().mean()))


def test_attn_backward_one_pass_random_ragged_batches(dev):
    """The one-pass d_h = 32 backward against the two-kernel form (the second call lends no workspace) on randomly drawn ragged self- and
    cross-attention batches: sequences shorter than a tile, lengths on and next to the 512-key block edges, up to five sequences and four
    heads (tools/stress_bwd1p.py draws more of the same)."""
    import random
    from acai_omr_amd import engine, ops
    rng = random.Random(7)
    bf, dh = torch.bfloat16, 32
    for c in range(14):
        B, H = rng.randint(1, 5), rng.randint(1, 4)
        pick = lambda: rng.choice([rng.randint(1, 80), rng.randint(400, 700), 512, 513, 1024, rng.randint(900, 2100), 511, 64, 1536])
        lens_q = [pick() for _ in range(B)]
        lens_q[rng.randrange(B)] = rng.choice([512, 600, 1200])
        lens_k = list(lens_q)
        if c % 3 == 2:
            lens_k = [pick() for _ in range(B)]
            lens_k[rng.randrange(B)] = rng.choice([512, 700, 1536, 2049])
        E = H * dh
        g = torch.Generator().manual_seed(100 + c)
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
            ops.attn_varlen_bwd(q, k, v, o, do, lse, cu_q, cu_k, H, dh, max(lens_q), max(lens_k), False, dq, dkv[:, :E], dkv[:, E:], q_prescaled=True,
                                lend_workspace=lend)
            outs.append((dq.float().cpu(), dkv.float().cpu()))
        for name, a, b in (("dq", outs[1][0], outs[0][0]), ("dkv", outs[1][1], outs[0][1])):
            assert int(((b == 7.0) & (a != 7.0)).sum()) == 0, (c, name, "rows left at the fill value", lens_q, lens_k)
            assert float((a - b).abs().max()) <= 2.0 ** -7 * max(1e-6, float(a.abs().max())), (c, name, B, H, lens_q, lens_k)


@pytest.mark.parametrize("H,dh,S,B", [(16, 32, 4096, 2), (12, 64, 4096, 1)])
def test_attn_prescaled_at_benchmark_size(dev, H, dh, S, B):
    """The attention kernels at the benchmarked sequence length (MAE decoder: 4096 tokens, 16 heads of 32; encoder: 12 heads of 64), bf16,
    through size-independent properties: rows of P sum to one (V = 1 gives 1), and the prescaled-q form (fast loops, accumulators started at
    -m / -lse) agrees with the plain form on the output, the log-sum-exp and all three gradients."""
    from acai_omr_amd import engine, ops
    g = torch.Generator().manual_seed(S + dh)
    E = H * dh
    qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.7).to(dev).to(torch.bfloat16)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    qp = (q.float() * ops.QSCALE(dh)).to(torch.bfloat16)
    q_eff = (qp.float() / ops.QSCALE(dh)).to(torch.bfloat16)   # what the prescaled kernels effectively see, for the plain kernels
    cu = engine.cu_from_lens([S] * B, dev)
    ones = torch.ones_like(v)
    for pre, qq in ((False, q_eff), (True, qp)):
        o1 = ops.attn_varlen(qq, k, ones, cu, cu, H, dh, S, q_prescaled=pre)
        assert float((o1.float() - 1.0).abs().max()) < 2e-2
    lse0, lse1 = torch.empty(H * B * S, device=dev), torch.empty(H * B * S, device=dev)
    o0 = ops.attn_varlen(q_eff, k, v, cu, cu, H, dh, S, lse=lse0)
    o1 = ops.attn_varlen(qp, k, v, cu, cu, H, dh, S, lse=lse1, q_prescaled=True)
    assert float((o0.float() - o1.float()).abs().max()) < 3e-2 and float((lse0 - lse1).abs().max()) < 2e-2
    dout = torch.randn(B * S, E, generator=g).to(dev).to(torch.bfloat16)
    d0, d1 = torch.empty_like(qkv), torch.empty_like(qkv)
    ops.attn_varlen_bwd(q_eff, k, v, o0, dout, lse0, cu, cu, H, dh, S, S, False, d0[:, :E], d0[:, E:2 * E], d0[:, 2 * E:])
    ops.attn_varlen_bwd(qp, k, v, o1, dout, lse1, cu, cu, H, dh, S, S, False, d1[:, :E], d1[:, E:2 * E], d1[:, 2 * E:], q_prescaled=True)
    for sl in (slice(0, E), slice(E, 2 * E), slice(2 * E, 3 * E)):
        a, b = d0[:, sl].float(), d1[:, sl].float()
        assert float((a - b).abs().max()) < 3e-2 * max(1.0, float(a.abs().max())), sl
        assert float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)) > 0.9995


def test_row_kernels_backward(dev):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(5)
    for rows, dim in [(37, 768), (5, 10), (130, 512)]:
        x = (torch.randn(rows, dim, generator=g) * 2 + 0.5).requires_grad_(True)
        w, b = torch.randn(dim, generator=g).requires_grad_(True), torch.randn(dim, generator=g).requires_grad_(True)
        dy = torch.randn(rows, dim, generator=g)
        torch.nn.functional.layer_norm(x, (dim,), w, b, 1e-5).backward(dy)
        dx, dw, db = ops.layernorm_bwd(x.detach().to(dev), w.detach().to(dev), dy.to(dev), 1e-5)
        assert md(dx, x.grad) < 1e-4 and md(dw, w.grad) < 1e-3 and md(db, b.grad) < 1e-3
    a = torch.randn(1000, generator=g).requires_grad_(True)
    dh = torch.randn(1000, generator=g)
    torch.nn.functional.gelu(a).backward(dh)
    assert md(ops.gelu_fwd(a.detach().to(dev)), torch.nn.functional.gelu(a)) < 1e-6
    assert md(ops.gelu_bwd(a.detach().to(dev), dh.to(dev)), a.grad) < 1e-5
    x = torch.randn(5000, 70, generator=g)
    assert md(ops.colsum(x.to(dev)), x.sum(0)) < 2e-3
    src, idx = torch.randn(300, 24, generator=g), torch.randint(0, 17, (300,), generator=g)
    dst = torch.zeros(17, 24)
    dst.index_add_(0, idx, src)
    assert md(ops.scatter_add_rows(src.to(dev), idx.to(torch.int32).to(dev), torch.zeros(17, 24, device=dev)), dst) < 1e-4
    # MAELoss KAT (reference tests/test_mae.py:169-180) through the HIP loss kernel
    from acai_omr_amd.models.models import MAELoss, OMRCELoss
    target = torch.cat([torch.tensor([[1, 1, 1], [2, 2, 2]], dtype=torch.float).unsqueeze(-1).repeat(1, 1, 6),
                        torch.tensor([[2, 2, 2], [3, 3, 3]], dtype=torch.float).unsqueeze(-1).repeat(1, 1, 6)], dim=-1)
    pred = torch.tensor([[2, 2, 2], [3, 3, 4]], dtype=torch.float).unsqueeze(-1).repeat(1, 1, 12)
    lm = torch.tensor([[1, 0, 0], [1, 0, 1]], dtype=torch.float)
    assert abs(float(MAELoss()(pred.to(dev), lm.to(dev), target.to(dev))) - 10.583329200744629) < 1e-5
    lg = torch.randn(6, 9, 227, generator=g).requires_grad_(True)
    tg = torch.randint(0, 227, (6, 9), generator=g)
    tg[0, 3:] = 1
    ref = torch.nn.functional.cross_entropy(lg.reshape(-1, 227), tg.reshape(-1), ignore_index=1)
    ref.backward()
    lgd = lg.detach().to(dev).requires_grad_(True)
    loss = OMRCELoss(1)(lgd, tg.to(dev))
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 and md(lgd.grad, lg.grad) < 1e-6


@pytest.mark.parametrize("name", ["mae_small", "mae_debug_ckpt"])
def test_mae_forward_loss_grads_vs_reference(dev, name):
    from acai_omr_amd.models.models import MAE, MAELoss
    fx = load_golden(name)
    cfg = fx["cfg"]
    mae = MAE(cfg["mask_ratio"], cfg["P"], cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"], decoder_hidden_dim=cfg["dec_dim"],
              encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"])
    mae.load_state_dict(fx["state_dict"])
    mae = mae.to(dev).train()
    batch = list(zip(fx["imgs"], fx["tgts"]))
    pred, loss_mask, target = mae(batch, noises=fx["noises"])
    assert torch.equal(loss_mask.cpu(), fx["loss_mask"])
    assert torch.equal(target.cpu(), fx["target"])
    valid = ~(torch.arange(fx["pred"].shape[1]).unsqueeze(0) >= torch.tensor([m.numel() for m in fx["noises"]]).unsqueeze(1))
    assert md(pred.cpu()[valid], fx["pred"][valid]) < 2e-4
    loss = MAELoss()(pred, loss_mask, target)
    assert abs(float(loss) - float(fx["loss"])) < 1e-4
    loss.backward()
    params = dict(mae.named_parameters())
    for n, gref in fx["grads"].items():
        assert md(params[n].grad, gref) < 2e-4 * max(1.0, float(gref.abs().max())), n
    # gradient confined to the used PE region (reference tests/test_mae.py:182-202)
    hp = max(t.shape[-2] // cfg["P"] for t in fx["imgs"])
    wp = max(t.shape[-1] // cfg["P"] for t in fx["imgs"])
    assert float(mae.encoder.pos_embedding.grad[hp:, :, :].abs().sum()) == 0 and float(mae.encoder.pos_embedding.grad[:, wp:, :].abs().sum()) == 0


@pytest.mark.parametrize("name", ["tf_small", "tf_dh64"])
def test_teacher_forced_train_step_vs_reference(dev, name):
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, TeacherForcedViTOMR
    fx = load_golden(name)
    cfg = fx["cfg"]
    enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                             num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"], transformer_dropout=0.0)
    dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"],
                     transformer_dropout=0.0)
    m = TeacherForcedViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"], transition_head_dropout=0.0)
    m.load_state_dict(fx["state_dict"])
    m = m.to(dev).train()
    pred, tgt = m(list(zip(fx["imgs"], fx["lmx"])))
    assert torch.equal(tgt.cpu(), fx["target"])
    valid = fx["target"] != 1
    assert md(pred.cpu()[valid], fx["pred"][valid]) < 1e-3
    loss = OMRCELoss(m.decoder.pad_idx)(pred, tgt)
    assert abs(float(loss) - float(fx["loss"])) < 1e-4
    loss.backward()
    params = dict(m.named_parameters())
    for n, gref in fx["grads"].items():
        assert md(params[n].grad, gref) < 3e-4 * max(1.0, float(gref.abs().max())), n


def test_scheduled_sampling_forward_train(dev):
    """ScheduledSamplingViTOMR.forward_train (models.py:798-838): with teacher_forcing_prob = 1 no position is sampled, the second decoder
    pass sees the gold embeddings and the step must equal the reference's teacher-forced golden step (pred, loss, gradients); with
    probability 0 every position takes the Gumbel-softmax expectation of the first pass, whose graph then carries gradient too."""
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    fx = load_golden("tf_small")
    cfg = fx["cfg"]

    def build():
        enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                                 num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"], transformer_dropout=0.0)
        dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"],
                         transformer_dropout=0.0)
        m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"], transition_head_dropout=0.0)
        m.load_state_dict(fx["state_dict"])
        return m.to(dev).train()

    batch = list(zip(fx["imgs"], fx["lmx"]))
    m = build()
    pred, tgt = m.forward_train(batch, 1.0, 0.5, False)
    valid = fx["target"] != 1
    assert torch.equal(tgt.cpu(), fx["target"]) and md(pred.cpu()[valid], fx["pred"][valid]) < 1e-3
    loss = OMRCELoss(m.decoder.pad_idx)(pred, tgt)
    assert abs(float(loss) - float(fx["loss"])) < 1e-4
    loss.backward()
    params = dict(m.named_parameters())
    for n, gref in fx["grads"].items():
        assert md(params[n].grad, gref) < 3e-4 * max(1.0, float(gref.abs().max())), n
    g_gold = {n: p.grad.clone() for n, p in params.items() if p.grad is not None}
    # every position sampled: finite, seeded, and a different gradient (the first pass contributes through the soft sample)
    m2 = build()
    torch.manual_seed(3)
    pred2, _ = m2.forward_train(batch, 0.0, 0.5, False)
    loss2 = OMRCELoss(m2.decoder.pad_idx)(pred2, tgt)
    loss2.backward()
    assert torch.isfinite(pred2).all() and abs(float(loss2) - float(loss)) > 1e-5
    g2 = {n: p.grad for n, p in m2.named_parameters() if p.grad is not None}
    assert all(torch.isfinite(v).all() for v in g2.values())
    assert md(g2["decoder.unembed.weight"], g_gold["decoder.unembed.weight"]) > 1e-6
    torch.manual_seed(3)
    pred3, _ = build().forward_train(batch, 0.0, 0.5, False)
    assert md(pred3, pred2) < 1e-5          # same seed, same sample
    assert torch.equal(m2.forward_eval(batch)[1].cpu(), fx["target"])


def test_twice_used_parameters_accumulate_in_place(dev):
    """Scheduled sampling with 0 < teacher_forcing_prob < 1: both decoder passes carry gradient, so every decoder parameter (and the encoder's,
    through the two passes' cross attention) gets two contributions in one backward.  The second contribution is accumulated by the
    weight-gradient GEMM / column-sum / LayerNorm-backward kernels into the first one's tensor (autograd_path._wgrad & co.) instead of by
    autograd's add: gradients must equal the plain form's (flag off) to summation order, and a later backward must not touch earlier tensors."""
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    from acai_omr_amd.train import autograd_path as AP
    fx = load_golden("tf_small")
    cfg = fx["cfg"]
    enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                             num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"], transformer_dropout=0.0)
    dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"],
                     transformer_dropout=0.0)
    m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"], transition_head_dropout=0.0)
    m.load_state_dict(fx["state_dict"])
    m = m.to(dev).train()
    batch = list(zip(fx["imgs"], fx["lmx"]))

    def grads(fuse):
        AP.PGRAD_FUSE = fuse
        try:
            m.zero_grad(set_to_none=True)
            torch.manual_seed(11)
            pred, tgt = m.forward_train(batch, 0.4, 0.5, False)
            OMRCELoss(m.decoder.pad_idx)(pred, tgt).backward()
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            AP.PGRAD_FUSE = True

    plain, fused = grads(False), grads(True)
    first = {n: g.clone() for n, g in fused.items()}
    again = grads(True)                      # a second backward pass of the same model: new graph task, nothing of the first is reused
    assert set(plain) == set(fused) == set(again)
    for n in plain:
        tol = 2e-5 * max(1.0, float(plain[n].abs().max()))
        assert md(fused[n], plain[n]) < tol, n
        assert md(again[n], plain[n]) < tol, n
        assert torch.equal(fused[n], first[n]), n
    assert any(n.startswith("decoder.decoder_blocks.layers.0.norm1") for n in plain)


def test_attn_backward_accumulate_dkv_flag(dev):
    """acai_attn_varlen_bwd, `causal` bit 1: dk, dv += instead of =.  Two backward calls with different incoming gradients, the second accumulating,
    equal the sum of the two plain calls (added in fp32, rounded once: at most one bf16 place from the sum of the two rounded results); dq is
    written, not accumulated; fp32 operands and stray flag bits are refused with an argument error."""
    from acai_omr_amd import _lib, engine, ops
    H, dh, lens_q, lens_k = 2, 64, [300, 513], [700, 260]
    E = H * dh
    g = torch.Generator().manual_seed(51)
    bf = torch.bfloat16
    q = (torch.randn(sum(lens_q), E, generator=g) * ops.QSCALE(dh)).to(dev).to(bf)
    k, v = (torch.randn(sum(lens_k), E, generator=g).to(dev).to(bf) for _ in range(2))
    d1, d2 = (torch.randn(sum(lens_q), E, generator=g).to(dev).to(bf) for _ in range(2))
    cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
    lse = torch.empty(H * sum(lens_q), device=dev)
    o = ops.attn_varlen(q, k, v, cu_q, cu_k, H, dh, max(lens_q), lse=lse, q_prescaled=True)

    def bwd(dout, dk, dv, acc):
        dq = torch.empty_like(q)
        ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu_q, cu_k, H, dh, max(lens_q), max(lens_k), False, dq, dk, dv, q_prescaled=True, accumulate_dkv=acc)
        return dq

    k1, v1, k2, v2 = (torch.empty_like(k) for _ in range(4))
    q1 = bwd(d1, k1, v1, False)
    q2 = bwd(d2, k2, v2, False)
    ka, va = k1.clone(), v1.clone()
    q2a = bwd(d2, ka, va, True)
    assert torch.equal(q2a, q2)
    for got, a, b in ((ka, k1, k2), (va, v1, v2)):
        ref = a.float() + b.float()
        assert float((got.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
        assert float((got.float() - ref).abs().mean()) <= 2.0 ** -9 * float(ref.abs().mean()) * 2
    # refused: fp32 operands, unknown flag bits
    qf, kf, vf, of, df = (t.float() for t in (q, k, v, o, d1))
    with pytest.raises(RuntimeError, match="accumulating dk / dv"):
        ops.attn_varlen_bwd(qf, kf, vf, of, df, lse, cu_q, cu_k, H, dh, max(lens_q), max(lens_k), False, torch.empty_like(qf), torch.empty_like(kf),
                            torch.empty_like(vf), accumulate_dkv=True)
    delta = torch.empty(H * sum(lens_q), device=dev)
    dq = torch.empty_like(q)
    rc = _lib.lib().acai_attn_varlen_bwd(q.data_ptr(), E, k.data_ptr(), E, v.data_ptr(), E, o.data_ptr(), E, d1.data_ptr(), E, dq.data_ptr(), E, k1.data_ptr(), E,
                                         v1.data_ptr(), E, lse.data_ptr(), delta.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(), 2, H, dh, max(lens_q), max(lens_k),
                                         sum(lens_q), 4, _lib.ACAI_BF16, 0.0, 0, 1, None)
    assert rc < 0 and b"bit 1" in _lib.lib().acai_last_error()


def test_shared_memory_gradient_accumulates_in_the_attention_backward(dev):
    """Scheduled sampling's two decoder passes attend to ONE projected memory per layer (autograd_path.decoder_forward shares it): under autocast the
    pass whose backward runs second adds its dK / dV inside the attention backward's epilogue (acai_attn_varlen_bwd, causal bit 1) to the tensor
    the first one returned, instead of autograd adding two [keys, 2E] tensors.  Same gradients as the plain form (flag off) up to one bf16 rounding
    of that sum - checked on the parameters the sum flows into (the cross attention's K / V rows, the transition head, the encoder)."""
    from torch.amp import autocast
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    from acai_omr_amd.train import autograd_path as AP
    torch.manual_seed(41)
    n_dec = 3
    enc = FineTuneOMREncoder(16, 8, 8, 2, num_layers=2, hidden_dim=64, num_heads=2, mlp_dim=128, transformer_dropout=0.0)
    dec = OMRDecoder(32, VOCAB, num_layers=n_dec, hidden_dim=64, num_heads=2, mlp_dim=128, transformer_dropout=0.0)    # d_h = 32: the aligned (vector) kernels
    m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dim=96, transition_head_dropout=0.0).to(dev).train()
    g = torch.Generator().manual_seed(42)
    batch = [(torch.rand(1, 32, 64, generator=g).to(dev), torch.cat([torch.tensor([0]), torch.randint(3, 227, (n,), generator=g), torch.tensor([2])]).to(dev))
             for n in (12, 7, 9)]
    calls = []
    orig = AP.ops.attn_varlen_bwd

    def spy(*a, **k):
        calls.append(bool(k.get("accumulate_dkv")))
        return orig(*a, **k)

    def grads(fuse):
        AP._KV_GRAD_FUSE = fuse
        AP.ops.attn_varlen_bwd = spy
        try:
            m.zero_grad(set_to_none=True)
            torch.manual_seed(11)
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                pred, tgt = m.forward_train(batch, 0.4, 0.5, False)
                loss = OMRCELoss(m.decoder.pad_idx)(pred, tgt)
            loss.backward()
            return {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            AP._KV_GRAD_FUSE = True
            AP.ops.attn_varlen_bwd = orig

    plain = grads(False)
    assert not any(calls)
    del calls[:]
    fused = grads(True)
    assert sum(calls) == n_dec, (sum(calls), len(calls))     # one accumulating backward per decoder layer's cross attention
    assert set(plain) == set(fused)
    for n in plain:
        a, b = plain[n], fused[n]
        assert float((a - b).abs().max()) <= 2e-2 * max(1e-3, float(a.abs().max())), n
        if a.numel() > 64:
            assert float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)) > 0.9999, n


def test_dropout_kernels(dev):
    """Counter-based dropout: keep rate / scaling, forward-backward mask consistency, attention-probability dropout checked against
    finite differences of the forward kernel itself (same seed -> same mask) and in expectation against the undropped output."""
    import math
    from acai_omr_amd import engine, ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(512, 384, generator=g).to(dev)
    res = torch.randn(512, 384, generator=g).to(dev)
    p = 0.25
    y = ops.dropout_add(x, res, p, 123)
    kept = (y != res)
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
    assert torch.allclose((y - res)[kept], x[kept] / (1 - p), atol=1e-5)
    dy = torch.randn(512, 384, generator=g).to(dev)
    dx = ops.dropout_add(dy, None, p, 123)
    assert torch.equal(dx != 0, kept | (dy == 0)) or float(((dx != 0) ^ kept).float().mean()) < 1e-4   # same mask in backward
    assert not torch.equal(ops.dropout_add(x, res, p, 124) != res, kept)                               # another seed, another mask
    # attention dropout
    H, dh, lens = 2, 16, [37, 20]
    E = H * dh
    q = torch.randn(sum(lens), 3 * E, generator=g).to(dev)
    cu = engine.cu_from_lens(lens, dev)
    pd, seed = 0.3, 777
    lse = torch.empty(H * sum(lens), device=dev)
    o = ops.attn_varlen(q[:, :E], q[:, E:2 * E], q[:, 2 * E:], cu, cu, H, dh, max(lens), lse=lse, dropout_p=pd, seed=seed)
    o0 = ops.attn_varlen(q[:, :E], q[:, E:2 * E], q[:, 2 * E:], cu, cu, H, dh, max(lens))
    # expectation: averaging over seeds approaches the undropped output
    acc = torch.zeros_like(o0)
    n = 200
    for s_ in range(n):
        acc += ops.attn_varlen(q[:, :E], q[:, E:2 * E], q[:, 2 * E:], cu, cu, H, dh, max(lens), dropout_p=pd, seed=1000 + s_)
    assert float((acc / n - o0).abs().mean()) < 0.05 * float(o0.abs().mean()) + 0.02
    # backward consistent with the forward (same mask): directional finite difference in fp32
    dout = torch.randn(o.shape, generator=g).to(dev)
    dq = torch.empty_like(q)
    ops.attn_varlen_bwd(q[:, :E], q[:, E:2 * E], q[:, 2 * E:], o, dout, lse, cu, cu, H, dh, max(lens), max(lens), False,
                        dq[:, :E], dq[:, E:2 * E], dq[:, 2 * E:], dropout_p=pd, seed=seed)
    d = torch.randn(q.shape, generator=g).to(dev)
    eps = 1e-2
    f = lambda t: float((ops.attn_varlen(t[:, :E], t[:, E:2 * E], t[:, 2 * E:], cu, cu, H, dh, max(lens), dropout_p=pd, seed=seed).double() * dout.double()).sum())
    num = (f(q + eps * d) - f(q - eps * d)) / (2 * eps)
    ana = float((dq.double() * d.double()).sum())
    assert abs(num - ana) < 2e-2 * max(1.0, abs(ana)), (num, ana)


def test_training_with_dropout_runs_and_is_seeded(dev):
    """The reference's real schedule trains with dropout 0.05 / 0.1 (omr_teacher_force_train.py:45-47): the HIP path applies it in train
    mode, is reproducible under torch.manual_seed, and reduces to the deterministic result in eval mode."""
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, TeacherForcedViTOMR
    fx = load_golden("tf_small")
    cfg = fx["cfg"]
    enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                             num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"], transformer_dropout=0.05)
    dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"],
                     transformer_dropout=0.1)
    m = TeacherForcedViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"], transition_head_dropout=0.05)
    m.load_state_dict(fx["state_dict"])
    m = m.to(dev).train()
    batch = list(zip(fx["imgs"], fx["lmx"]))
    losses = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        m.zero_grad()
        pred, tgt = m(batch)
        loss = OMRCELoss(1)(pred, tgt)
        loss.backward()
        losses.append(float(loss))
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    # same seed -> same masks (the loss reduction uses float atomics, so equality is to rounding); another seed -> another step
    assert abs(losses[0] - losses[1]) < 1e-5 and abs(losses[0] - losses[2]) > 1e-4
    assert abs(losses[0] - float(fx["loss"])) > 1e-4          # dropout really perturbs the step ...
    m.eval()
    with torch.no_grad():
        pred, tgt = m(batch)
    assert abs(float(OMRCELoss(1)(pred, tgt)) - float(fx["loss"])) < 1e-4   # ... and eval mode is the reference's deterministic value


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_decoder_checkpoint_grads_same_logits_and_gradients(dev, dropout):
    """OMRDecoder.forward(..., checkpoint_grads=True) (reference models.py:470-478: checkpoint_sequential, one segment per layer; the GRPO
    loop's setting): the per-layer recomputation must give the logits and every gradient of the plain forward - with dropout too (the masks
    are a hash of seeds drawn from torch's CPU generator, whose state the checkpoint restores)."""
    from acai_omr_amd.models.models import OMRDecoder
    torch.manual_seed(31)
    dec = OMRDecoder(64, VOCAB, num_layers=3, hidden_dim=128, num_heads=4, mlp_dim=256, transformer_dropout=dropout).to(dev).train()
    g = torch.Generator().manual_seed(32)
    B, T, S = 3, 20, 37
    seqs = torch.randint(3, 227, (B, T), generator=g).to(dev)
    lmx_mask = torch.zeros(B, T, dtype=torch.bool)
    lmx_mask[1, 15:] = True
    mem_mask = torch.zeros(B, S, dtype=torch.bool)
    mem_mask[2, 30:] = True
    mem0 = torch.randn(B, S, 128, generator=g)
    outs = []
    for ck in (False, True):
        torch.manual_seed(33)
        dec.zero_grad(set_to_none=True)
        mem = mem0.clone().to(dev).requires_grad_(True)
        logits = dec(seqs, mem, lmx_mask.to(dev), mem_mask.to(dev), checkpoint_grads=ck)
        w = torch.randn(logits.shape, generator=torch.Generator().manual_seed(34)).to(dev)
        (logits.float() * w * (~lmx_mask.to(dev)).unsqueeze(-1)).sum().backward()
        outs.append((logits.detach().float().clone(), mem.grad.clone(), {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None}))
    (l0, m0, g0), (l1, m1, g1) = outs
    assert torch.equal(l0, l1)
    assert set(g0) == set(g1) and len(g0) > 30
    assert float((m0 - m1).abs().max()) <= 1e-5 * max(1.0, float(m0.abs().max()))
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * max(1.0, float(g0[n].abs().max())), n     # (split-K float atomics: equal to rounding)


@pytest.mark.parametrize("dim", [256, 512, 768, 1024, 96])
def test_layernorm_bwd_fused_and_colsum_vec(dev, dim):
    """The one-pass LayerNorm backward (dim % 256 == 0) and the generic two-kernel form against torch autograd; bf16 side copy; 16-byte colsum."""
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(dim)
    rows = 1037
    x = (torch.randn(rows, dim, generator=g) * 2 + 0.5).to(dev)
    w = torch.randn(dim, generator=g).to(dev)
    b = torch.randn(dim, generator=g).to(dev)
    dy = torch.randn(rows, dim, generator=g).to(dev)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    torch.nn.functional.layer_norm(xr, (dim,), wr, br, 1e-5).backward(dy)
    if dim % 256 == 0:
        dx, dw, db, dxb, cs = ops.layernorm_bwd(x, w, dy, 1e-5, want_bf16=True, want_colsum=True)
        assert torch.equal(dxb, dx.to(torch.bfloat16))
        assert torch.allclose(cs, dxb.float().sum(0), atol=2e-3, rtol=1e-3)          # column sums of the bf16 copy (the consuming Linear's bias gradient)
        dx2, _, _, cs2 = ops.layernorm_bwd(x, w, dy, 1e-5, want_colsum=True)
        assert torch.equal(dx2, dx) and torch.allclose(cs2, dx.sum(0), atol=2e-3, rtol=1e-3)
    else:
        dx, dw, db = ops.layernorm_bwd(x, w, dy, 1e-5)
    assert torch.allclose(dx, xr.grad, atol=2e-5, rtol=1e-4)
    assert torch.allclose(dw, wr.grad, atol=2e-3, rtol=1e-4) and torch.allclose(db, br.grad, atol=2e-3, rtol=1e-4)
    for t in (dy, dy.to(torch.bfloat16), dy[:, : dim - 8]):
        assert torch.allclose(ops.colsum(t), t.float().sum(0), atol=2e-3, rtol=1e-3)


def test_fused_adamw_matches_torch(dev):
    """FusedAdamW (one HIP launch for every tensor) against torch.optim.AdamW over several steps: param groups with their own lr / weight decay
    (the layer-wise LR groups of models.py:761-781), an LR schedule, a frozen parameter, odd sizes and unaligned views, state_dict round trip."""
    from acai_omr_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(4)
    shapes = [(1024, 768), (768,), (3, 5), (17,), (40000,), (1,)]
    base = [torch.randn(s, generator=g) for s in shapes]

    def make():
        ps = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
        ps[3].requires_grad_(False)
        return ps

    pa, pb = make(), make()
    groups = lambda ps: [dict(params=ps[:2], lr=1.5e-4), dict(params=ps[2:4], lr=3e-3, weight_decay=0.0), dict(params=ps[4:], lr=1e-2, betas=(0.8, 0.9))]
    oa = torch.optim.AdamW(groups(pa), betas=(0.9, 0.95), weight_decay=0.05)
    ob = FusedAdamW(groups(pb), betas=(0.9, 0.95), weight_decay=0.05)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lambda e: 1.0 / (1 + e))
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda e: 1.0 / (1 + e))
    for it in range(6):
        for x, y in zip(pa, pb):
            if x.requires_grad:
                gr = torch.randn(x.shape, generator=g).to(dev) * (10.0 if it == 2 else 1.0)
                x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(), ob.step()
        sa.step(), sb.step()
        if it == 3:   # state_dict round trip into a fresh optimizer
            sd = ob.state_dict()
            ob = FusedAdamW(groups(pb), betas=(0.9, 0.95), weight_decay=0.05)
            ob.load_state_dict(sd)
            sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda e: 1.0 / (1 + e), last_epoch=it)
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-7), float((x - y).abs().max())
    for x, y in zip(pa, pb):
        if x.requires_grad:
            assert torch.allclose(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    assert set(ob.state_dict()["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    # grad_scale: the same step with gradients pre-multiplied
    pc, pd = make(), make()
    oc, od = FusedAdamW(pc, lr=1e-3), FusedAdamW(pd, lr=1e-3)
    for x, y in zip(pc, pd):
        if x.requires_grad:
            gr = torch.randn(x.shape, generator=g).to(dev)
            x.grad, y.grad = gr * 0.125, gr.clone()
    oc.step(), od.step(grad_scale=0.125)
    for x, y in zip(pc, pd):
        assert torch.equal(x, y)


def test_scatter_add_rows_shared_row(dev):
    """Row scatter-add with one shared index (the MAE mask token): equals index_add_ and the all-atomics form."""
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(9)
    rows, dim, table = 3000, 512, 901
    perm = torch.randperm(900, generator=g)
    idx = torch.full((rows,), 900, dtype=torch.int32)
    pos = torch.randperm(rows, generator=g)[:900]
    idx[pos] = perm.to(torch.int32)
    src = torch.randn(rows, dim, generator=g).to(dev)
    ref = torch.zeros(table, dim).index_add_(0, idx.long(), src.cpu())
    a = ops.scatter_add_rows(src, idx.to(dev), torch.zeros(table, dim, device=dev), shared_row=900)
    b = ops.scatter_add_rows(src, idx.to(dev), torch.zeros(table, dim, device=dev))
    assert torch.allclose(a.cpu(), ref, atol=2e-3, rtol=1e-4) and torch.allclose(b.cpu(), ref, atol=2e-3, rtol=1e-4)
    assert torch.equal(a[:900], b[:900])


def test_full_size_mae_step_vs_oracle(dev):
    """The FULL-SIZE MAE (ViT-B encoder 768/12/3072 x 12, decoder 512/16/3072 x 8; pre_train.py:156-159) on a small ragged batch: forward,
    MAELoss and the gradients of the fp32 training step against autograd through the CPU oracle on the same weights and injected noise -
    the golden fixtures use reduced widths, this runs the kernels at the real ones (d_h = 64 and 32, LayerNorm 768 / 512, every GEMM shape)."""
    import oracle.vitomr_oracle as O
    from acai_omr_amd.config import MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
    from acai_omr_amd.models.models import MAE, MAELoss
    torch.manual_seed(0)
    mae = MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH)
    g = torch.Generator().manual_seed(5)
    imgs = [torch.rand(1, 128, 256, generator=g), torch.rand(1, 256, 384, generator=g)]
    noises = [torch.rand((im.shape[-2] // PATCH_SIZE) * (im.shape[-1] // PATCH_SIZE), generator=g) for im in imgs]
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in mae.state_dict().items()}
    torch.set_num_threads(8)
    pred_o, lm_o, tgt_o, lens = O.mae_forward(list(zip(imgs, imgs)), noises, sd, PATCH_SIZE, MASK_RATIO, 12, 16, prec="fp32")
    loss_o = O.mae_loss(pred_o, lm_o, tgt_o)
    loss_o.backward()
    mae = mae.to(dev).train()
    pred, loss_mask, target = mae([(im.to(dev), im.to(dev)) for im in imgs], noises=noises)
    o = 0
    for b, n in enumerate(lens):
        assert md(pred[b, :n].cpu(), pred_o[o:o + n].detach()) < 1e-3
        assert torch.equal(loss_mask[b, :n].cpu(), lm_o[o:o + n]) and md(target[b, :n].cpu(), tgt_o[o:o + n]) < 1e-5
        o += n
    loss = MAELoss()(pred, loss_mask, target)
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 1e-4
    loss.backward()
    params = dict(mae.named_parameters())
    for name in ("mask_token", "decoder_pos_embedding", "decoder_unembed.weight", "decoder_embed.bias", "encoder.pos_embedding", "encoder.projection.weight",
                 "encoder.encoder_blocks.layers.0.self_attn.in_proj_weight", "encoder.encoder_blocks.layers.11.linear2.bias",
                 "decoder.decoder_blocks.layers.0.linear1.weight", "decoder.decoder_blocks.layers.7.norm2.weight"):
        gref = sd[name].grad
        assert gref is not None, name
        assert md(params[name].grad, gref) < 3e-4 * max(1.0, float(gref.abs().max())), name


def test_full_size_teacher_forced_step_vs_oracle(dev):
    """The FULL-SIZE TeacherForcedViTOMR (FineTuneOMREncoder 768 x 12, head 4096, OMRDecoder 1024/16/4096 x 12, V = 227;
    omr_teacher_force_train.py:265-284) on two small systems: logits, OMRCELoss and gradients of the fp32 step against autograd through the
    CPU oracle on the same weights."""
    import oracle.vitomr_oracle as O
    from acai_omr_amd.config import ENCODER_FINE_TUNE_DEPTH, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, TeacherForcedViTOMR
    torch.manual_seed(0)
    enc = FineTuneOMREncoder(PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH, ENCODER_FINE_TUNE_DEPTH, transformer_dropout=0.0)
    dec = OMRDecoder(MAX_LMX_SEQ_LEN, VOCAB, num_layers=NUM_DECODER_LAYERS, transformer_dropout=0.0)
    m = TeacherForcedViTOMR(enc, None, dec, transition_head_dropout=0.0)
    g = torch.Generator().manual_seed(6)
    imgs = [torch.rand(1, 64, 256, generator=g), torch.rand(1, 128, 192, generator=g)]
    lmx = [torch.cat([torch.tensor([0]), torch.randint(3, 227, (n,), generator=g), torch.tensor([2])]) for n in (17, 9)]
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    torch.set_num_threads(8)
    pred_o, tgt_o = O.teacher_forced_forward(list(zip(imgs, lmx)), sd, 12, 16, PATCH_SIZE, "fp32")
    loss_o = O.ce_loss(pred_o, tgt_o, 1)
    loss_o.backward()
    m = m.to(dev).train()
    pred, tgt = m([(im.to(dev), sq.to(dev)) for im, sq in zip(imgs, lmx)])
    assert torch.equal(tgt.cpu(), tgt_o)
    valid = tgt_o != 1
    assert md(pred.cpu()[valid], pred_o.detach()[valid]) < 1e-3
    loss = OMRCELoss(m.decoder.pad_idx)(pred, tgt)
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 1e-4
    loss.backward()
    params = dict(m.named_parameters())
    checked = 0
    for name in ("decoder.unembed.weight", "decoder.vocab_embedding.weight", "decoder.pos_embedding", "decoder.decoder_blocks.layers.0.multihead_attn.in_proj_weight",
                 "decoder.decoder_blocks.layers.11.linear1.bias", "decoder.decoder_blocks.layers.5.norm3.weight", "transition_head.0.weight", "transition_head.3.bias",
                 "encoder.fine_tune_blocks.layers.0.self_attn.out_proj.weight", "encoder.fine_tune_blocks.norm.bias"):
        if name not in params or not params[name].requires_grad:
            continue
        gref = sd[name].grad
        assert gref is not None, name
        assert md(params[name].grad, gref) < 3e-4 * max(1.0, float(gref.abs().max())), name
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize("ft_depth", [1, 2])
def test_fine_tune_freezing_and_llrd_like_reference(dev, ft_depth):
    """The behaviour the reference's tests/test_vitomr.py:203-338 pin (partial fine-tune, LLRD param groups), on its odd debug widths
    (hidden 10, one head of d_h = 10, mlp 1): after one optimizer step over `create_fine_tune_param_groups`, every decoder / transition-head /
    fine-tuned parameter has moved and every frozen one is bit-identical; with a partial fine-tune the patch projection and the positional
    embedding are frozen too, with a full one they train; every trainable parameter sits in exactly one group, decoder and head at the base
    LR, encoder groups at or below the fine-tune LR."""
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, TeacherForcedViTOMR
    kw = dict(num_layers=2, num_heads=1, hidden_dim=10, mlp_dim=1)
    torch.manual_seed(3)
    m = TeacherForcedViTOMR(FineTuneOMREncoder(16, 60, 200, ft_depth, **kw), None, OMRDecoder(1536, VOCAB, **kw)).to(dev).train()
    groups, _ = m.create_fine_tune_param_groups(100.0, 50.0, 0.99)
    groups = [{"params": list(g["params"]), "lr": g["lr"]} for g in groups]
    lr_of = {}
    for g in groups:
        for p in g["params"]:
            assert id(p) not in lr_of
            lr_of[id(p)] = g["lr"]
    for name, p in m.named_parameters():
        if p.requires_grad:
            assert id(p) in lr_of, name
            assert lr_of[id(p)] == 100.0 if ("decoder" in name or "transition" in name) else lr_of[id(p)] <= 50.0, name
    opt = torch.optim.SGD([g for g in groups if g["params"]])
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    g = torch.Generator().manual_seed(4)
    x = [(torch.rand(1, 64, 128, generator=g).to(dev), torch.randint(0, 227, (8,), generator=g).to(dev)),
         (torch.rand(1, 32, 32, generator=g).to(dev), torch.randint(0, 227, (6,), generator=g).to(dev))]
    pred, target = m(x)
    assert pred.shape == torch.Size([2, 7, 227])
    OMRCELoss(m.decoder.pad_idx)(pred, target).backward()
    opt.step()
    for name, p in m.named_parameters():
        moved = not torch.equal(p.detach(), before[name])
        if "frozen" in name or (ft_depth == 1 and ("pos_embedding" in name and "decoder" not in name or "projection" in name)):
            assert not p.requires_grad and not moved, name
        elif any(k in name for k in ("decoder", "fine_tune", "transition_head")) or ft_depth == 2:
            assert p.requires_grad and moved, name


def test_decoder_positional_gradient_confined_like_reference(dev):
    """tests/test_vitomr.py:92-124 on valid (suffix) masks: the learned positional table only receives gradient - and only moves under SGD -
    on the rows the batch used (5 input positions here), on the reference's debug widths (hidden 10, one head, mlp 1)."""
    from acai_omr_amd.models.models import OMRCELoss, OMRDecoder, batchify_and_split_lmx_seqs
    torch.manual_seed(5)
    dec = OMRDecoder(1536, VOCAB, hidden_dim=10, num_heads=1, num_layers=1, mlp_dim=1, transformer_dropout=0.0).to(dev).train()
    seqs = [torch.tensor([0, 2, 3, 226]), torch.tensor([0, 2, 2, 3, 4, 226])]
    inp, tgt, lmx_mask = batchify_and_split_lmx_seqs(seqs, dec.pad_idx, dev)
    latent = torch.ones(2, 10, 10, device=dev)
    latent_mask = torch.tensor([[False] * 7 + [True] * 3, [False] * 10], device=dev)
    before = dec.pos_embedding.detach().clone()
    opt = torch.optim.SGD(dec.parameters(), lr=0.01)
    pred = dec(inp, latent, lmx_mask, latent_mask)
    OMRCELoss(dec.pad_idx)(pred, tgt).backward()
    grad = dec.pos_embedding.grad.detach().clone()
    opt.step()
    assert float(grad[:5].abs().sum()) > 0 and float(grad[5:].abs().sum()) == 0
    assert not torch.equal(before[:5], dec.pos_embedding.detach()[:5]) and torch.equal(before[5:], dec.pos_embedding.detach()[5:])


@pytest.mark.gpu
def test_sample_and_mix_seqs_reference_vectors(dev):
    """tests/test_vitomr.py:340-364: the first position always keeps the gold <bos> embedding; with tf_prob = 0 and hard sampling every other
    position is the embedding of the arg-max token of the first pass (logit 100 against 5: the Gumbel noise cannot change it)."""
    import torch
    from conftest import VOCAB
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRDecoder, ScheduledSamplingViTOMR
    kw = dict(num_layers=2, num_heads=1, hidden_dim=10, mlp_dim=1)
    torch.manual_seed(0)
    m = ScheduledSamplingViTOMR(FineTuneOMREncoder(16, 60, 200, 1, **kw), None, OMRDecoder(1536, VOCAB, **kw)).to(dev)
    V = m.decoder.vocab_embedding.weight.shape[0]
    seqs = torch.full([1, 5], 10, dtype=torch.long, device=dev)
    seqs[:, 0] = m.decoder.bos_idx
    logits = torch.full([1, 5, V], 5.0, device=dev)
    logits[:, :, 2] = 100.0
    bos_emb = m.decoder.vocab_embedding(torch.tensor([m.decoder.bos_idx], device=dev))
    mixed = m.sample_and_mix_seqs(0.8, seqs, logits, 0.1, False, dev)
    assert mixed.shape == torch.Size([1, 5, m.encoder.hidden_dim]) and torch.equal(mixed[:, 0, :], bos_emb)
    mixed = m.sample_and_mix_seqs(0, seqs, logits, 0.1, True, dev)
    assert torch.equal(mixed[:, 0, :], bos_emb)
    assert torch.allclose(mixed[:, 1:, :], m.decoder.vocab_embedding(torch.tensor([2], device=dev)).expand(4, -1).unsqueeze(0), atol=1e-6)
    # and it is differentiable in the first pass' logits and the embedding matrix (the reference trains through it, M:801-817)
    lg = logits.clone().requires_grad_(True)
    m.sample_and_mix_seqs(0.0, seqs, lg, 1.0, False, dev).square().sum().backward()
    assert lg.grad is not None and float(lg.grad.abs().max()) > 0 and m.decoder.vocab_embedding.weight.grad is not None
    # the embedding lookup itself leaves the <pad> row without gradient (nn.Embedding(padding_idx), M:409): all-gold mix of a padded sequence
    m.zero_grad(set_to_none=True)
    seqs2 = seqs.clone()
    seqs2[:, -1] = m.decoder.pad_idx
    m.sample_and_mix_seqs(1.0, seqs2, logits, 1.0, False, dev).square().sum().backward()
    gw = m.decoder.vocab_embedding.weight.grad
    assert float(gw[m.decoder.pad_idx].abs().max()) == 0.0 and float(gw[10].abs().max()) > 0
