"""Kernel-level parity on the GPU: each C-ABI entry point against plain fp32 torch / the oracle on seeded inputs.
fp32 kernels must agree to ~1e-5 (exact-fp32 MFMA); bf16 kernels are compared against the same math on bf16-rounded
operands with fp32 accumulation (tolerance = a few bf16 ulps of the output)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from acai_omr_amd import _lib
    _lib.lib()  # fails loudly if the HIP library is not built
    return torch.device("cuda:0")


def rb(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("rows,dim", [(5, 768), (33, 1024), (7, 10), (1, 512), (130, 12)])
def test_layernorm(dev, rows, dim):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(rows * dim)
    x = torch.randn(rows, dim, generator=g) * 3 + 1
    w, b = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    y32, y16 = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-5, want_bf16=True)
    ref = torch.nn.functional.layer_norm(x, (dim,), w, b, 1e-5)
    assert (y32.cpu() - ref).abs().max() < 2e-5
    assert torch.equal(y16.cpu().float(), rb(y32.cpu()))


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (128, 128, 64), (37, 19, 10), (1, 227, 1024), (513, 768, 256), (260, 130, 72)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gemm_nt(dev, M, N, K, dtype):
    _check_gemm_nt(dev, M, N, K, dtype)


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 704])
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_gemm_nt_three_stage_tile(dev, K, dtype, variant):
    """The large-tile LDS-DMA kernels, each pinned in turn (256x128 two-stage, three-stage with counted vmcnt waits, persistent ring across
    tiles, 256x256, persistent 256x256 on the ring of half-stages, 7 = the ping-pong ring with the register epilogue - bf16 only, fp32 falls
    back to 6): every K-tile count modulo 3, one to many tiles, ragged M / N edges."""
    from acai_omr_amd import _lib
    if dtype == "fp32":
        K //= 2   # 32 floats per K-tile: same tile counts
    _lib.check(_lib.lib().acai_gemm_set_variant(variant), "acai_gemm_set_variant")
    try:
        _check_gemm_nt(dev, 8192 + 40, 2048 + 24, K, dtype)
    finally:
        _lib.lib().acai_gemm_set_variant(0)


def _check_gemm_nt(dev, M, N, K, dtype):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    if dtype == "bf16":
        a, w = rb(a), rb(w)
        ad, wd = a.to(dev).to(torch.bfloat16), w.to(dev).to(torch.bfloat16)
        tol = 3e-2
    else:
        ad, wd = a.to(dev), w.to(dev)
        tol = 2e-5 * max(1.0, math.sqrt(K) / 8)
    base = a.double() @ w.double().t() + bias.double()
    # plain
    y = ops.gemm_nt(ad, wd, bias.to(dev))
    assert (y.cpu().double() - base).abs().max() < (tol if dtype == "fp32" else 1e-4 * K ** 0.5 + 1e-4)
    # gelu + residual, fp32 out
    y = ops.gemm_nt(ad, wd, bias.to(dev), residual=res.to(dev), gelu=True)
    ref = torch.nn.functional.gelu(base.float()).double() + res.double()
    assert (y.cpu().double() - ref).abs().max() < (tol if dtype == "fp32" else 1e-3)
    # bf16 output with autocast-style rounding
    y = ops.gemm_nt(ad, wd, bias.to(dev), out_dtype=torch.bfloat16, round_bf16=True)
    err = (y.cpu().float().double() - base).abs() / (base.abs() + 1)
    assert err.max() < 1e-2
    # strided views (columns of a wider buffer), as the QKV split uses them
    wide = torch.zeros(M, K + 24, device=dev, dtype=ad.dtype)
    wide[:, 8:8 + K] = ad
    y2 = ops.gemm_nt(wide[:, 8:8 + K], wd, bias.to(dev))
    y1 = ops.gemm_nt(ad, wd, bias.to(dev))
    assert torch.equal(y1, y2)


def ref_attn(q, k, v, lens_q, lens_k, H, dh, causal):
    out = torch.zeros(q.shape[0], H * dh, dtype=torch.float64)
    oq = ok = 0
    for lq, lk in zip(lens_q, lens_k):
        for h in range(H):
            sl = slice(h * dh, (h + 1) * dh)
            s = q[oq:oq + lq, sl].double() @ k[ok:ok + lk, sl].double().t() / math.sqrt(dh)
            if causal:
                s = s.masked_fill(~torch.ones(lq, lk, dtype=torch.bool).tril(), float("-inf"))
            out[oq:oq + lq, sl] = torch.softmax(s, -1) @ v[ok:ok + lk, sl].double()
        oq += lq
        ok += lk
    return out


@pytest.mark.parametrize("H,dh,lens_q,lens_k,causal", [
    (2, 64, [8, 32, 200], None, False),
    (3, 32, [130, 1, 77], None, False),
    (2, 16, [65, 64], None, True),
    (1, 6, [5, 9, 3], None, False),
    (4, 12, [7, 12], [20, 13], False),     # cross attention: lq != lk
    (2, 64, [300], None, True),
    (12, 64, [1024], None, False),
    (1, 10, [70], None, True),
    # >= 512 queries, d_h <= 32, bf16 prescaled: the two-blocks-per-wave forward (ragged ends, a 1-row sequence, cross lengths, causal)
    (3, 32, [513, 640, 1], None, False),
    (2, 32, [600, 513], [1000, 577], False),
    (2, 32, [700, 130], None, True),
    (2, 24, [520], [64], False),
])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("prescaled", [False, True])
def test_attn_varlen(dev, H, dh, lens_q, lens_k, causal, dtype, prescaled):
    """prescaled: q carries log2(e)/sqrt(dh) (the training path's in-projection epilogue); the reference takes the same q, scaled back."""
    from acai_omr_amd import engine, ops
    if prescaled and dh % (8 if dtype == "bf16" else 4):
        pytest.skip("the prescaled form exists for 16-byte-aligned heads only")
    lens_k = lens_k or lens_q
    g = torch.Generator().manual_seed(H * dh + sum(lens_q))
    E = H * dh
    qkv = torch.randn(sum(lens_q), 3 * E, generator=g)
    kv = qkv if lens_k == lens_q else torch.randn(sum(lens_k), 3 * E, generator=g)
    qkv[:, :E] *= 2.0  # spread the scores
    if dtype == "bf16":
        qkv, kv = rb(qkv), rb(kv)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    qd, kd = qkv.to(dev).to(tdt), kv.to(dev).to(tdt)
    q_ref = qkv[:, :E]
    if prescaled:
        qp = (qkv[:, :E] * ops.QSCALE(dh)).to(tdt)
        qd = torch.cat([qp.to(dev), qd[:, E:]], 1)
        q_ref = qp.double() / ops.QSCALE(dh)
    cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
    out = ops.attn_varlen(qd[:, :E], kd[:, E:2 * E], kd[:, 2 * E:], cu_q, cu_k, H, dh, max(lens_q), causal=causal, q_prescaled=prescaled)
    ref = ref_attn(q_ref, kv[:, E:2 * E], kv[:, 2 * E:], lens_q, lens_k, H, dh, causal)
    err = (out.cpu().double() - ref).abs().max()
    # bf16: the output's own rounding is half an ulp = 2^-9 of its magnitude, P's is as much again
    assert err < (2e-5 if dtype == "fp32" else 1.2e-2 * max(1.0, float(ref.abs().max()))), err


def test_attn_varlen_prescaled_needs_aligned_heads(dev):
    """The prescaled-q form exists for 16-byte-aligned head rows only: anything else fails loudly instead of silently taking another path."""
    from acai_omr_amd import engine, ops
    H, dh, lens = 2, 12, [9, 5]
    qkv = torch.randn(sum(lens), 3 * H * dh).to(dev).to(torch.bfloat16)
    cu = engine.cu_from_lens(lens, dev)
    E = H * dh
    with pytest.raises(RuntimeError, match="q_prescaled"):
        ops.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, cu, H, dh, max(lens), q_prescaled=True)


@pytest.mark.parametrize("prescaled", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("causal,lens", [(False, [320, 200]), (True, [320, 200]), (False, [513, 200]), (True, [513, 130])])
def test_attn_varlen_reference_maximum_restart(dev, dtype, prescaled, causal, lens):
    """The tiles in front of the ragged end take their reference maximum from tile 0 alone.  Scores that later rise past it by more than
    2^80 (here: by ~130 in the log2 domain) overflow the row sum; the workgroup then starts over with a running maximum - same result.
    Causal and not, and with 513 queries (the fifth 128-query block holds one row: three of its waves are inactive and must keep the restarting
    waves' barrier count)."""
    from acai_omr_amd import engine, ops
    H, dh = 2, 32
    E = H * dh
    g = torch.Generator().manual_seed(11)
    q = torch.randn(sum(lens), E, generator=g) * 0.1 + 4.0
    k = torch.randn(sum(lens), E, generator=g) * 0.05
    v = torch.randn(sum(lens), E, generator=g)
    k[200] = 4.0          # sequence 0: one key 130 above everything before it, three tiles in
    k[250, :dh] = -4.0    # and a very negative one for head 0
    k[lens[0] + 100] = 3.0    # sequence 1
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    q, k, v = (t.to(tdt).float() for t in (q, k, v))
    qd = (q * ops.QSCALE(dh)).to(tdt) if prescaled else q.to(tdt)
    q_ref = qd.double() / ops.QSCALE(dh) if prescaled else q
    cu = engine.cu_from_lens(lens, dev)
    lse = torch.empty(H * sum(lens), device=dev)
    out = ops.attn_varlen(qd.to(dev), k.to(dev).to(tdt), v.to(dev).to(tdt), cu, cu, H, dh, max(lens), causal=causal, lse=lse, q_prescaled=prescaled)
    ref = ref_attn(q_ref, k, v, lens, lens, H, dh, causal)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(lse).all())
    err = (out.cpu().double() - ref).abs().max()
    assert err < (2e-5 if dtype == "fp32" else 1.2e-2 * max(1.0, float(ref.abs().max()))), err


@pytest.mark.parametrize("shift", [-4.0, 4.0])
def test_attn_varlen_two_block_forward_zero_reference_restart(dev, shift):
    """The two-blocks-per-wave forward (bf16, q prescaled, d_h = 32, >= 512 queries) takes probabilities as 2^score against a ZERO reference.
    Rows whose scores ALL sit ~130 below (or above) zero in the log2 domain underflow (overflow) every probability: the row sum leaves
    (2^-100, 2^100), the workgroup starts over with a running maximum, and the result is the plain softmax."""
    from acai_omr_amd import engine, ops
    H, dh, lens = 2, 32, [600, 520]
    E = H * dh
    g = torch.Generator().manual_seed(23)
    q = torch.randn(sum(lens), E, generator=g) * 0.1 + 4.0
    k = torch.randn(sum(lens), E, generator=g) * 0.05 + shift
    v = torch.randn(sum(lens), E, generator=g)
    bf = torch.bfloat16
    q, k, v = (t.to(bf).float() for t in (q, k, v))
    qd = (q * ops.QSCALE(dh)).to(bf)
    cu = engine.cu_from_lens(lens, dev)
    lse = torch.empty(H * sum(lens), device=dev)
    out = ops.attn_varlen(qd.to(dev), k.to(dev).to(bf), v.to(dev).to(bf), cu, cu, H, dh, max(lens), lse=lse, q_prescaled=True)
    ref = ref_attn(qd.double() / ops.QSCALE(dh), k, v, lens, lens, H, dh, False)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(lse).all())
    assert (out.cpu().double() - ref).abs().max() < 1.2e-2 * max(1.0, float(ref.abs().max()))
    assert float(lse.abs().min()) > 100.0        # the scores really are that far from zero


@pytest.mark.parametrize("H,lens_q,lens_k", [
    (2, [513, 640, 1], None),                 # ragged ends, a one-row sequence, a fifth 128-query block holding one row
    (3, [513, 130], [4096 + 17, 64]),         # the teacher-forced cross attention's shape; exactly one full tile
    (1, [40, 129, 192], [40, 65, 191]),       # one ragged tile; two tiles with one key in the second; three tiles
    (2, [300], [128]),                        # two full tiles: prologue + one steady tile + drain, nothing masked
    (1, [256], [1]),                          # a single key
    (2, [288, 289, 20], [700, 64, 1]),        # tails of exactly 32 rows (tail kernel: keys split over 8 waves, 11 tiles), of 33 (wide kernel) and a 20-row sequence
    (1, [513] * 3, [1100] * 3),               # equal lengths: the host launches the tail kernel because 513 % 256 = 1
])
def test_attn_fwd64_pipelined(dev, H, lens_q, lens_k):
    """attn_fwd64.hip (bf16, d_h = 64, q prescaled, no mask): the software-pipelined forward against the fp64 softmax, every tile-count class
    of its prologue / steady / masked / drain structure, with the log-sum-exp it hands to the backward pass."""
    from acai_omr_amd import engine, ops
    dh, bf = 64, torch.bfloat16
    lens_k = lens_k or lens_q
    E = H * dh
    g = torch.Generator().manual_seed(H + sum(lens_q) + sum(lens_k))
    q = rb(torch.randn(sum(lens_q), E, generator=g) * 2.0 * ops.QSCALE(dh))
    k, v = rb(torch.randn(sum(lens_k), E, generator=g)), rb(torch.randn(sum(lens_k), E, generator=g))
    cu_q, cu_k = engine.cu_from_lens(lens_q, dev), engine.cu_from_lens(lens_k, dev)
    lse = torch.full((H * sum(lens_q),), float("nan"), device=dev)
    out = ops.attn_varlen(q.to(dev).to(bf), k.to(dev).to(bf), v.to(dev).to(bf), cu_q, cu_k, H, dh, max(lens_q), lse=lse, q_prescaled=True)
    q_ref = q.double() / ops.QSCALE(dh)
    ref = ref_attn(q_ref, k, v, lens_q, lens_k, H, dh, False)
    assert (out.cpu().double() - ref).abs().max() < 1.2e-2 * max(1.0, float(ref.abs().max()))
    # lse[h][q] = log2 sum_k 2^(q' . k)
    oq = ok = 0
    lse = lse.cpu().double().view(H, -1)
    for lq, lk in zip(lens_q, lens_k):
        for h in range(H):
            sl = slice(h * dh, (h + 1) * dh)
            s = q[oq:oq + lq, sl].double() @ k[ok:ok + lk, sl].double().T
            want = torch.logsumexp(s * math.log(2.0), -1) / math.log(2.0)
            assert (lse[h, oq:oq + lq] - want).abs().max() < 2e-2
        oq += lq
        ok += lk


@pytest.mark.parametrize("shift", [-4.0, 4.0])
@pytest.mark.parametrize("lens", [[600, 520], [513, 70]])
def test_attn_fwd64_zero_reference_restart(dev, shift, lens):
    """attn_fwd64.hip takes probabilities as 2^score against a ZERO reference.  Rows whose scores all sit ~180 below (above) zero in the log2
    domain underflow (overflow) every probability: the row sum leaves (2^-100, 2^100), the workgroup takes the exact row maxima in a pre-pass
    and runs the same loop again with the scores starting at -m.  One sequence of the batch restarts, the other (ordinary scores) does not."""
    from acai_omr_amd import engine, ops
    H, dh, bf = 2, 64, torch.bfloat16
    E = H * dh
    g = torch.Generator().manual_seed(29)
    q = torch.randn(sum(lens), E, generator=g) * 0.1 + 4.0
    k = torch.randn(sum(lens), E, generator=g) * 0.05 + shift
    v = torch.randn(sum(lens), E, generator=g)
    q[lens[0]:] = torch.randn(lens[1], E, generator=g)          # second sequence: ordinary scores
    k[lens[0]:] = torch.randn(lens[1], E, generator=g)
    k[lens[0] + 5] *= 30.0                                      # ... except for one key that overflows some rows and underflows none
    q, k, v = (t.to(bf).float() for t in (q, k, v))
    qd = (q * ops.QSCALE(dh)).to(bf)
    cu = engine.cu_from_lens(lens, dev)
    lse = torch.empty(H * sum(lens), device=dev)
    out = ops.attn_varlen(qd.to(dev), k.to(dev).to(bf), v.to(dev).to(bf), cu, cu, H, dh, max(lens), lse=lse, q_prescaled=True)
    ref = ref_attn(qd.double() / ops.QSCALE(dh), k, v, lens, lens, H, dh, False)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(lse).all())
    assert (out.cpu().double() - ref).abs().max() < 1.2e-2 * max(1.0, float(ref.abs().max()))
    assert float(lse.view(H, -1)[:, :lens[0]].abs().min()) > 100.0        # the first sequence's scores really are that far from zero


@pytest.mark.parametrize("valid", [0, 8, 40 * 16, 40 * 16 + 8, 63 * 16 + 12, 1024])
def test_lds_dma_out_of_range_lanes_write_zeros(dev, valid):
    """The weight-gradient GEMMs fetch a ragged last token tile with `buffer_load_dwordx4 ... offen lds` through a resource whose num_records
    ends at the last token that exists, and rely on the lanes beyond it writing ZEROS into LDS (so that no second, register-staged launch is
    needed for the 16-token remainder of the 16 x 513 decoder stream).  That is observed gfx950 / ROCm 7.2 behaviour (per dword, only the
    VGPR offset is range-checked), not a documented guarantee - this test pins it."""
    from acai_omr_amd import _lib
    src = (torch.arange(256, dtype=torch.int32) + 0x1000).to(dev)
    out = torch.full((256,), -2, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().acai_debug_lds_dma_oob(src.data_ptr(), valid, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "acai_debug_lds_dma_oob")
    torch.cuda.synchronize()
    want = torch.arange(256, dtype=torch.int32) + 0x1000
    want[(valid + 3) // 4:] = 0      # every dword that does not lie wholly inside the resource reads as zero - never stale LDS bytes (0xFFFFFFFF)
    want[valid // 4:(valid + 3) // 4] = 0
    assert torch.equal(out.cpu(), want), (valid, out.cpu()[max(0, valid // 4 - 4):valid // 4 + 8])


def test_patchify_and_gather(dev):
    from acai_omr_amd import ops
    from oracle import vitomr_oracle as O
    g = torch.Generator().manual_seed(3)
    for (H, W, P) in [(32, 64, 16), (12, 20, 4), (48, 16, 16)]:
        img = torch.rand(1, H, W, generator=g)
        n = (H // P) * (W // P)
        out = torch.zeros(n + 3, P * P, device=dev)
        assert ops.patchify(img.to(dev), P, out, 2) == n
        assert torch.equal(out[2:2 + n].cpu(), O.patchify(img, P)[0])
        out16 = torch.zeros(n, P * P, device=dev, dtype=torch.bfloat16)
        ops.patchify(img.to(dev), P, out16, 0)
        assert torch.equal(out16.cpu().float(), rb(O.patchify(img, P)[0]))
    table = torch.randn(50, 24, generator=g)
    idx = torch.randint(0, 50, (33,), generator=g, dtype=torch.int32)
    add = torch.randn(33, 24, generator=g)
    assert torch.equal(ops.gather_rows(table.to(dev), idx.to(dev), add.to(dev)).cpu(), table[idx.long()] + add)
    x = torch.randn(1000 + 3, generator=g)
    assert torch.equal(ops.cast_bf16(x.to(dev)).cpu().float(), rb(x))


@pytest.mark.parametrize("B,N,K", [(8, 3072, 1024), (3, 227, 1024), (1, 48, 12), (8, 1024, 4096), (11, 100, 2052), (2, 5, 1)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_skinny_gemm(dev, B, N, K, dtype):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(B + N + K)
    x, w = torch.randn(B, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, res = torch.randn(N, generator=g), torch.randn(B, N, generator=g)
    if dtype == "bf16":
        wd = w.to(dev).to(torch.bfloat16)
        ref = (rb(x).double() @ rb(w).double().t() + bias.double())
        y = ops.skinny_gemm(x.to(dev), wd, bias.to(dev), round_bf16=True)
        # rounding to bf16 may flip on accumulation-order noise: allow one bf16 ulp
        assert ((y.cpu().double() - rb(ref.float()).double()).abs() <= ref.abs() * 2 ** -7 + 1e-6).all()
    else:
        ref = x.double() @ w.double().t() + bias.double()
        y = ops.skinny_gemm(x.to(dev), w.to(dev), bias.to(dev), residual=res.to(dev), gelu=True)
        ref2 = torch.nn.functional.gelu(ref.float()).double() + res.double()
        assert (y.cpu().double() - ref2).abs().max() < 1e-4


@pytest.mark.parametrize("H,dh,lens", [(16, 64, [1024, 77, 4096, 1]), (2, 6, [3, 1]), (4, 12, [20, 16, 13]), (1, 32, [700])])
@pytest.mark.parametrize("fused_merge", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_decode_attn(dev, H, dh, lens, dtype, fused_merge):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(H + dh + sum(lens))
    B = len(lens)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dhp = 8 if dtype == "bf16" else 4
    while dhp < dh:
        dhp *= 2
    q = torch.randn(B, H * dh, generator=g) * 2
    ks = [torch.randn(H, l, dh, generator=g) for l in lens]
    vs = [torch.randn(H, l, dh, generator=g) for l in lens]
    if dtype == "bf16":
        ks, vs = [rb(k) for k in ks], [rb(v) for v in vs]
    total = sum(lens) * H * dhp
    kc, vc = torch.zeros(total), torch.zeros(total)
    offs, o = [], 0
    for k, v, l in zip(ks, vs, lens):
        offs.append(o)
        kc[o:o + H * l * dhp].view(H, l, dhp)[..., :dh] = k
        vc[o:o + H * l * dhp].view(H, l, dhp)[..., :dh] = v
        o += H * l * dhp
    out = ops.decode_attn(q.to(dev), kc.to(dev).to(tdt), vc.to(dev).to(tdt), torch.tensor(offs, dtype=torch.int64, device=dev),
                          torch.tensor(lens, dtype=torch.int32, device=dev), H, dh, dhp, max(lens), fused_merge=fused_merge)
    for b in range(B):
        for h in range(H):
            s = (q[b, h * dh:(h + 1) * dh].double() @ ks[b][h].double().t()) / math.sqrt(dh)
            ref = torch.softmax(s, -1) @ vs[b][h].double()
            assert (out[b, h * dh:(h + 1) * dh].cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(300, 136, 64), (16384 + 8, 2048, 128)])
def test_gemm_nt_gelu_aux_modes(dev, dtype, shape):
    """acai_gemm_nt_ex: aux_mode 1 keeps the pre-activation next to its GELU; aux_mode 2 multiplies the product by gelu'(aux) - checked
    against the unfused kernels (gemm + gelu_fwd / gelu_bwd), which they replace in the training MLP."""
    from acai_omr_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    a = torch.randn(M, K, generator=g).to(dev).to(dt)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).to(dt)
    b = torch.randn(N, generator=g).to(dev)
    rnd = dtype == "bf16"
    pre = torch.empty(M, N, dtype=dt, device=dev)
    h = ops.gemm_nt(a, w, b, out_dtype=dt, gelu=True, round_bf16=rnd, pre_act=pre)
    pre_ref = ops.gemm_nt(a, w, b, out_dtype=dt, round_bf16=rnd)
    assert torch.equal(pre, pre_ref)
    assert torch.equal(h, ops.gelu_fwd(pre_ref))
    # derivative epilogue: (a . w^T) o gelu'(saved)
    saved = torch.randn(M, N, generator=g).to(dev).to(dt)
    fused = ops.gemm_nt(a, w, out_dtype=dt, round_bf16=rnd, gelu_grad_of=saved)
    ref = ops.gelu_bwd(saved, ops.gemm_nt(a, w, out_dtype=dt, round_bf16=rnd))
    if dtype == "bf16":   # two kernels, same formula: fp32 contraction may differ by an ulp, which can flip a bf16 rounding here and there
        d = (fused.float() - ref.float()).abs()
        assert bool((d <= 2.0 ** -7 * ref.float().abs() + 1e-6).all()) and float((d > 0).float().mean()) < 0.01, (float(d.max()), float((d > 0).float().mean()))
    else:
        assert torch.allclose(fused, ref, rtol=1e-5, atol=1e-6)   # fp32 contraction differences between the two kernels
    # round 4 (what MlpFn uses): aux_mode 3 keeps gelu'(pre-activation) next to the GELU, aux_mode 4 multiplies the product by the kept tensor
    dg = torch.empty(M, N, dtype=dt, device=dev)
    h3 = ops.gemm_nt(a, w, b, out_dtype=dt, gelu=True, round_bf16=rnd, gelu_grad_out=dg)
    assert torch.equal(h3, h)                                                    # the GELU itself is unchanged by what is kept beside it
    dg_ref = ops.gelu_bwd(pre_ref, torch.ones(M, N, dtype=dt, device=dev))       # gelu'(pre) * 1 through the stand-alone kernel
    dd = (dg.float() - dg_ref.float()).abs()
    assert float(dd.max()) <= (2.0 ** -7 if dtype == "bf16" else 2e-6), float(dd.max())
    prod = ops.gemm_nt(a, w, out_dtype=dt, round_bf16=rnd, times=saved)
    ref4 = (ops.gemm_nt(a, w, out_dtype=dt, round_bf16=rnd).float() * saved.float()).to(dt)
    if dtype == "bf16":
        assert torch.equal(prod, ref4)      # one multiply of two bf16 values, rounded once: exact in both
    else:
        assert torch.allclose(prod, ref4, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(300, 136, 64), (2048, 1536, 512), (1000, 2304, 768)])
def test_gemm_nt_col_scale(dev, dtype, shape):
    """acai_gemm_nt_ex scale_cols / col_scale: the first n columns of (a w^T + bias) are multiplied by s before the rounding - the
    in-projection hands q to the attention kernels as q log2(e)/sqrt(dh); every other column is bit-identical to the plain GEMM."""
    from acai_omr_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + 1)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    a = torch.randn(M, K, generator=g).to(dev).to(dt)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).to(dt)
    b = torch.randn(N, generator=g).to(dev)
    rnd = dtype == "bf16"
    n, sc = N // 3, ops.QSCALE(64)
    y = ops.gemm_nt(a, w, b, out_dtype=dt, round_bf16=rnd, col_scale=(n, sc))
    plain = ops.gemm_nt(a, w, b, out_dtype=dt, round_bf16=rnd)
    exact = ops.gemm_nt(a, w, b, out_dtype=torch.float32)   # unrounded fp32 accumulators + bias
    assert torch.equal(y[:, n:], plain[:, n:])
    want = (exact[:, :n] * sc).to(dt)
    if rnd:   # same fp32 number rounded once: the two kernels may differ in the last bit of the fp32 product
        d = (y[:, :n].float() - want.float()).abs()
        assert bool((d <= 2.0 ** -7 * want.float().abs() + 1e-6).all()) and float((d > 0).float().mean()) < 0.01
    else:
        assert torch.allclose(y[:, :n], want, rtol=1e-6, atol=1e-7)



@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(136, 264, 4096), (768, 2304, 8192 + 64), (512, 8, 640), (40, 1000, 192), (1024, 1024, 8208), (264, 136, 100)])
def test_gemm_weight_gradient_form(dev, dtype, shape):
    """dW = dY^T X (both operands contraction-major, split-K atomics into an fp32 gradient that may already hold a value): the LDS-DMA
    transposing-read kernel (bf16, token count % 64 == 0) and the register-staged form (fp32 / other shapes) against fp64."""
    from acai_omr_amd import ops
    M, N, K = shape      # dW is [M, N], K = tokens
    g = torch.Generator().manual_seed(M + N + K)
    dy, x = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    if dtype == "bf16":
        dy, x = rb(dy), rb(x)
    seed = torch.randn(M, N, generator=g)
    out = seed.clone().to(dev)
    ops.gemm(dy.to(dev).to(dt), x.to(dev).to(dt), trans_a=True, trans_w=True, out=out)
    ref = seed.double() + dy.double().t() @ x.double()
    tol = 2e-4 * math.sqrt(K)
    assert (out.cpu().double() - ref).abs().max() < tol, float((out.cpu().double() - ref).abs().max())
    # column-sliced operands (views into wider buffers), as the fused qkv gradient uses them
    wide = torch.zeros(K, M + 16, device=dev, dtype=dt)
    wide[:, 8:8 + M] = dy.to(dev).to(dt)
    out2 = seed.clone().to(dev)
    ops.gemm(wide[:, 8:8 + M], x.to(dev).to(dt), trans_a=True, trans_w=True, out=out2)
    assert (out2.cpu().double() - ref).abs().max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("K", [64, 192, 512])
@pytest.mark.parametrize("shape", [(8192 + 40, 2048 + 24), (16384 + 8, 512), (70000, 768)])
def test_gemm_nt_pingpong_epilogues(dev, K, shape):
    """gemm_nt_pp_kernel (variant 7: ping-pong wave groups, swapped 16x16x32 MFMAs, register epilogue with permlane16 exchanges and raw buffer
    stores) in EVERY epilogue form the training steps use, against the persistent 256x256 kernel it replaces (variant 6: LDS-transposed
    epilogue): several tiles per workgroup, ragged last tiles in both directions, strided outputs.  The two kernels sum the 64 products of a
    K-tile in different orders, so bf16 results may differ in the last bit here and there; fp32 results to contraction rounding."""
    from acai_omr_amd import _lib, ops
    M, N = shape
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev).to(bf)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).to(bf)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    saved = torch.randn(M, N, generator=g).to(dev).to(bf)
    wide = torch.zeros(M, N + 40, device=dev, dtype=bf)

    def run(variant):
        _lib.check(_lib.lib().acai_gemm_set_variant(variant), "acai_gemm_set_variant")
        try:
            out = {}
            out["plain_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, round_bf16=True)
            out["plain_f32"] = ops.gemm_nt(a, w, b)
            out["nobias_f32"] = ops.gemm_nt(a, w)
            out["res_f32"] = ops.gemm_nt(a, w, b, residual=res, round_bf16=True)
            out["gelu_res_f32"] = ops.gemm_nt(a, w, b, residual=res, gelu=True, round_bf16=True)
            pre = torch.empty(M, N, dtype=bf, device=dev)
            out["gelu_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, gelu=True, round_bf16=True, pre_act=pre)
            out["pre_bf16"] = pre
            out["dgelu_bf16"] = ops.gemm_nt(a, w, out_dtype=bf, round_bf16=True, gelu_grad_of=saved)
            dgo = torch.empty(M, N, dtype=bf, device=dev)      # round 4: the forward keeps gelu'(pre-activation), the backward multiplies by it
            out["gelu3_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, gelu=True, round_bf16=True, gelu_grad_out=dgo)
            out["dgelu_kept_bf16"] = dgo
            out["times_bf16"] = ops.gemm_nt(a, w, out_dtype=bf, round_bf16=True, times=saved)
            out["scale_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, round_bf16=True, col_scale=(N // 3 // 4 * 4, ops.QSCALE(64)))
            view = wide[:, 8:8 + N]
            ops.gemm_nt(a, w, b, out=view, round_bf16=True)
            out["strided_bf16"] = view.clone()
            torch.cuda.synchronize()
            return out
        finally:
            _lib.lib().acai_gemm_set_variant(0)

    new, old = run(7), run(6)
    base = (a.float().cpu().double() @ w.float().cpu().double().t() + b.cpu().double())
    assert (new["plain_f32"].cpu().double() - base).abs().max() < 1e-4 * K ** 0.5 + 1e-4
    assert float(wide[:, :8].abs().max()) == 0.0 and float(wide[:, 8 + N:].abs().max()) == 0.0    # nothing written outside the view
    for k in new:
        x, y = new[k].float(), old[k].float()
        d = (x - y).abs()
        if k.endswith("bf16"):
            assert bool((d <= 2.0 ** -7 * y.abs() + 1e-6).all()), (k, float(d.max()))
            assert float((d > 0).float().mean()) < 0.01, (k, float((d > 0).float().mean()))
        else:
            # a bf16-rounded linear output inside an fp32 sum can flip by one bf16 ulp of the linear term
            tol = 2.0 ** -7 * (y.abs() + 4.0) if ("res" in k) else 1e-5 * (y.abs() + 1.0) * K ** 0.5
            assert bool((d <= tol).all()), (k, float(d.max()))
            assert float((d > 1e-5 * (y.abs() + 1.0) * K ** 0.5).float().mean()) < 0.01, k


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8208, 1024, 1024), (8208, 4096, 1024), (8208, 1024, 4096), (16416, 1024, 1024), (4096 + 8, 2048, 1024)])
def test_gemm_nt_remainder_rows_split_off(dev, shape):
    """The auto rule launches the rows past a multiple of 256 separately where the full-tile part alone saves a round of tiles (the decoder stream of
    the teacher-forced step: 16 x 513 = 8208 rows; gemm.hip `split_rem`): every epilogue form the training steps use must come out as from ONE launch of
    a pinned kernel (variant 6) - the remainder rows above all: their A / C / residual / kept-tensor pointers are advanced by hand."""
    from acai_omr_amd import _lib, ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev).to(bf)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).to(bf)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    saved = torch.randn(M, N, generator=g).to(dev).to(bf)

    def run(variant):
        _lib.check(_lib.lib().acai_gemm_set_variant(variant), "acai_gemm_set_variant")
        try:
            out = {}
            out["plain_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, round_bf16=True)
            out["res_f32"] = ops.gemm_nt(a, w, b, residual=res, round_bf16=True)
            dgo = torch.full((M, N), 9.0, dtype=bf, device=dev)
            out["gelu3_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, gelu=True, round_bf16=True, gelu_grad_out=dgo)
            out["dgelu_kept_bf16"] = dgo
            out["times_bf16"] = ops.gemm_nt(a, w, out_dtype=bf, round_bf16=True, times=saved)
            out["scale_bf16"] = ops.gemm_nt(a, w, b, out_dtype=bf, round_bf16=True, col_scale=(N // 3 // 4 * 4, ops.QSCALE(64)))
            torch.cuda.synchronize()
            return out
        finally:
            _lib.lib().acai_gemm_set_variant(0)

    new, old = run(0), run(6)
    for k in new:
        for rows in (slice(0, M - M % 256), slice(M - M % 256, M)):      # the full-tile part, the remainder rows
            x, y = new[k][rows].float(), old[k][rows].float()
            d = (x - y).abs()
            if k.endswith("bf16"):
                assert bool((d <= 2.0 ** -7 * y.abs() + 1e-6).all()), (k, rows, float(d.max()))
                assert float((d > 0).float().mean()) < 0.01, (k, rows)
            else:
                assert bool((d <= 2.0 ** -7 * (y.abs() + 4.0)).all()), (k, rows, float(d.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16384 + 8, 512, 512), (70000, 768, 512), (8192 + 40, 3072 + 24, 768)])
def test_gemm_nt_pingpong_deferred_gelu(dev, shape):
    """Variant 8: the ping-pong ring with the GELU forms' deferred epilogue (PP_DEFER - the tile boundary stores the bf16 linear output, the
    GELU / gelu' step runs in 16-byte chunks beside the next tile's K-steps and the last tile's chunks after the loop).  Off by default (it
    measured slower), kept runnable: it rounds at the same places as the undeferred form, so the two agree bit for bit."""
    from acai_omr_amd import _lib, ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev).to(bf)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).to(bf)
    b = torch.randn(N, generator=g).to(dev)
    saved = torch.randn(M, N, generator=g).to(dev).to(bf)

    def run(variant):
        _lib.check(_lib.lib().acai_gemm_set_variant(variant), "acai_gemm_set_variant")
        try:
            pre = torch.empty(M, N, dtype=bf, device=dev)
            y = ops.gemm_nt(a, w, b, out_dtype=bf, gelu=True, round_bf16=True, pre_act=pre)
            dy = ops.gemm_nt(a, w, out_dtype=bf, round_bf16=True, gelu_grad_of=saved)
            torch.cuda.synchronize()
            return y, pre, dy
        finally:
            _lib.lib().acai_gemm_set_variant(0)

    for x, y in zip(run(8), run(7)):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3072, 512, 16384), (520, 264, 8192 + 40), (768, 2304, 4096), (512, 512, 32768),
                                   # 16 x 513 decoder tokens = 128 whole 64-token tiles + 16 rows: the ragged tile inside the ring (1024 x 1024) and the
                                   # ping-pong (4096 x 1024) kernel - rows beyond the end arrive as zeros through the buffer LDS-DMA
                                   (1024, 1024, 8208), (4096, 1024, 8208), (1024, 4096, 8208 + 63), (2048, 1024, 1024 + 1)])
def test_gemm_weight_gradient_pingpong(dev, shape):
    """dW = dY^T X on the ping-pong TN kernel (gemm_tn_pp_kernel: 256 x 256 tiles, transposing fragment reads out of the swizzled natural
    image, split-K atomics from the 16x16 accumulators), long token counts incl. a ragged last 64-token tile, ragged output edges."""
    from acai_omr_amd import ops
    M, N, K = shape     # dW is [M, N], K tokens
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(K, M, generator=g).to(torch.bfloat16)
    x = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    ref = dy.float().double().t() @ x.float().double()
    out = ops.gemm(dy.to(dev), x.to(dev), trans_a=True, trans_w=True, out_dtype=torch.float32)
    err = (out.cpu().double() - ref).abs().max()
    assert err < 2e-3 * float(ref.abs().max()) + 1e-4, float(err)
    # accumulation into an existing gradient (the arena hands out zeroed views; a second product adds on top)
    out2 = ops.gemm(dy.to(dev), x.to(dev), trans_a=True, trans_w=True, out=out)
    assert (out2.cpu().double() - 2 * ref).abs().max() < 4e-3 * float(ref.abs().max()) + 2e-4


@pytest.mark.gpu
def test_cast_weights_one_launch_equals_aten(dev):
    """acai_cast_weights: the bf16 copy, the transposed bf16 copy and the bf16-rounded fp32 copy of many fp32 tensors in one launch are bit for
    bit what ATen's casts / transposing copy give (ragged and odd shapes, vectors, NaN / inf / denormals kept)."""
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(5)
    shapes = [(3072, 512), (227, 1024), (512, 3072), (65, 130), (5, 3), (1, 7), (3072,), (227,), (1,), (64, 64), (130, 4)]
    srcs = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    srcs[3][0, :5] = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e-40, -0.0], device=dev)
    items = []
    for i, t in enumerate(srcs):
        two = t.dim() == 2
        items.append((t, two or i % 2 == 0, two, (not two) or i % 3 == 0))
    outs = ops.cast_weights(items)
    torch.cuda.synchronize()
    for (t, w16, w16t, w32), (d16, d16t, d32) in zip(items, outs):
        ref = t.to(torch.bfloat16)
        same = lambda a, b: torch.equal(a.view(torch.int16), b.view(torch.int16))
        assert (d16 is not None) == bool(w16) and (d16t is not None) == bool(w16t) and (d32 is not None) == bool(w32)
        if w16:
            assert d16.shape == t.shape and same(d16, ref)
        if w16t:
            assert d16t.shape == (t.shape[1], t.shape[0]) and d16t.is_contiguous() and same(d16t, ref.t().contiguous())
        if w32:
            assert d32.dtype == torch.float32 and torch.equal(d32.view(torch.int32), ref.float().view(torch.int32))


@pytest.mark.gpu
def test_weight_cache_refreshes_all_stale_copies_together(dev):
    """engine.WeightCache after an optimizer-style in-place update: the first request refreshes every stale copy in one launch; values equal
    fresh ATen casts, copies handed out before keep the old values (they may sit in an autograd graph), untouched parameters keep their copy."""
    from acai_omr_amd import engine
    torch.manual_seed(0)
    lin = [torch.nn.Linear(96, 200).to(dev), torch.nn.Linear(200, 72).to(dev), torch.nn.Linear(8, 8).to(dev)]
    wc = engine.WeightCache()
    before = [(wc.w(l.weight, "bf16"), wc.wt(l.weight, "bf16"), wc.b(l.bias, "bf16")) for l in lin]
    keep = [tuple(t.clone() for t in trip) for trip in before]
    with torch.no_grad():
        for l in lin[:2]:
            l.weight.mul_(1.5)
            l.bias.add_(0.25)
    launches = []
    orig = engine.ops.cast_weights
    engine.ops.cast_weights = lambda items: (launches.append(len(items)), orig(items))[1]
    try:
        after = [(wc.w(l.weight, "bf16"), wc.wt(l.weight, "bf16"), wc.b(l.bias, "bf16")) for l in lin]
    finally:
        engine.ops.cast_weights = orig
    assert launches == [4]                       # two weights (both kinds in one entry each) + two biases, one launch
    for l, (w, wt, b) in zip(lin, after):
        ref = l.weight.detach().to(torch.bfloat16)
        assert torch.equal(w, ref) and torch.equal(wt, ref.t().contiguous()) and torch.equal(b, l.bias.detach().to(torch.bfloat16).float())
    for old, kept in zip(before, keep):
        assert all(torch.equal(a, b) for a, b in zip(old, kept))
    assert all(a is b for a, b in zip(before[2], after[2]))      # the untouched layer's copies are the same tensors


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3072, 512, 16384), (1536, 512, 8192), (768, 768, 32768), (520, 264, 8192 + 40), (4096, 1024, 8208), (1024, 1024, 8208),
                                   (96, 40, 300)])
def test_gemm_dw_with_bias_gradient(dev, shape):
    """acai_gemm_dw: dW = dY^T X and db = column sums of dY from one launch - the ping-pong weight-gradient kernel sums its dY fragments with a
    ones operand (every tile-column count nbn = 1, 2, 4, 6; ragged token tiles and edges), other shapes take the separate column-sum pass -
    and accumulation into existing gradients."""
    from acai_omr_amd import ops
    M, N, K = shape     # dW is [M, N], K tokens
    g = torch.Generator().manual_seed(M + N + K + 1)
    dy = torch.randn(K, M, generator=g).to(torch.bfloat16)
    x = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    ref_w = dy.float().double().t() @ x.float().double()
    ref_b = dy.float().double().sum(0)
    dw, db = ops.gemm_dw(dy.to(dev), x.to(dev), want_bias=True)
    assert (dw.cpu().double() - ref_w).abs().max() < 2e-3 * float(ref_w.abs().max()) + 1e-4
    assert (db.cpu().double() - ref_b).abs().max() < 2e-3 * float(ref_b.abs().max()) + 1e-3
    dw2, db2 = ops.gemm_dw(dy.to(dev), x.to(dev), out=dw, bias_out=db)      # accumulate on top
    assert dw2 is dw and db2 is db
    assert (dw.cpu().double() - 2 * ref_w).abs().max() < 4e-3 * float(ref_w.abs().max()) + 2e-4
    assert (db.cpu().double() - 2 * ref_b).abs().max() < 4e-3 * float(ref_b.abs().max()) + 2e-3
    dw3, none = ops.gemm_dw(dy.to(dev), x.to(dev))                            # weight gradient alone
    assert none is None and (dw3.cpu().double() - ref_w).abs().max() < 2e-3 * float(ref_w.abs().max()) + 1e-4
