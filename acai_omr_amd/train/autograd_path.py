"""Differentiable (training) execution of the hot path: torch.autograd.Function wrappers around the HIP kernels.

Forward and backward arithmetic both run in the HIP library (GEMM NT/NN/TN, varlen attention fwd/bwd, LayerNorm fwd/bwd,
GELU, row gather / scatter-add, the two losses); torch.autograd only records the graph and accumulates `.grad`.
Everything works on packed token streams; the padded `(B, L, E)` views the reference API hands around are produced by a
differentiable row gather (`pad_rows`).

Step semantics reproduced: MAE pre-training runs fp32 outside autocast
(acai_omr/train/pre_train.py:54-62); the teacher-forced step runs its forward under autocast(bfloat16)
(acai_omr/train/omr_teacher_force_train.py:113-120) - precision follows `torch.is_autocast_enabled("cuda")` exactly as in
the inference path.  Dropout (train mode) is a counter-based hash mask regenerated in backward: residual dropouts through
`acai_dropout_add`, attention-probability dropout inside the flash kernels; seeds come from torch's CPU generator.
"""
import os

import weakref

import torch
from torch.autograd import Function

from .. import engine as EG
from .. import ops
from ..models.kv_caching import _wc


def _prec():
    return "bf16" if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 else "fp32"


def _cdt(prec):
    return torch.bfloat16 if prec == "bf16" else torch.float32


def _p_of(module, training):
    """Dropout probability of an nn.Dropout / nn.MultiheadAttention in train mode (0 in eval)."""
    if not training or module is None:
        return 0.0
    p = module.p if isinstance(module, torch.nn.Dropout) else getattr(module, "dropout", 0.0)
    return float(p)


def _next_seed():
    """Seeds of the counter-based dropout masks come from torch's CPU generator: torch.manual_seed() makes a step reproducible."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def _check_dropout(module):
    return None  # dropout is built: see DropoutAddFn and the attention kernels' dropout_p


# bf16 copy of a gradient emitted by the kernel that produced it (LayerNorm backward), for the GEMMs of the Function that receives that very
# tensor object next.  One entry, matched by object identity (a strong reference keeps the Python object, hence the identity, alive).
_SIDE = []
SIDE_HITS = [0, 0]   # [hits, misses]: test / profiling aid


def _side_put(t, tb, colsum=None):
    _SIDE[:] = [(t, tb, colsum)]


_FWD_SIDE = []   # forward twin: bf16 copy of a LayerNorm output, consumed by the cast in front of the next bf16 GEMM


def _fwd_put(t, tb):
    _FWD_SIDE[:] = [(t, tb)]


def _fwd_take(t):
    e = _FWD_SIDE.pop() if _FWD_SIDE else None
    if e is not None and e[0] is t:
        SIDE_HITS[0] += 1
        return e[1]
    SIDE_HITS[1] += 1
    return None


def _side_take2(t):
    """(bf16 copy or None, column sums or None) that the producer of gradient `t` (LayerNorm backward) left for its consumer."""
    e = _SIDE.pop() if _SIDE else None
    if e is not None and e[0] is t:
        SIDE_HITS[0] += 1
        return e[1], e[2]
    SIDE_HITS[1] += 1
    return None, None


def _side_take(t):
    return _side_take2(t)[0]


# ---- parameters used more than once in a graph ---------------------------------------------------------------------------
# Scheduled sampling runs the decoder twice, so every decoder parameter receives two gradient contributions in one backward pass, and autograd
# sums them with one `add` launch per parameter (~270 six-microsecond launches per teacher-forced step).  The weight-gradient GEMM, the column
# sums and the LayerNorm backward all ACCUMULATE into their (zeroed) outputs, so the second contribution goes straight into the first one's
# tensor - which autograd is still holding for the parameter's accumulation node - and reports "no gradient".  Keyed on the engine's graph-task
# id: a tensor of an earlier backward pass is never touched.  Weak references: nothing is kept alive.
_PGRAD = {}
PGRAD_FUSE = os.environ.get("ACAI_PGRAD_FUSE", "1") != "0"   # A/B and test aid
_ROWS_DW_FUSE = os.environ.get("ACAI_ROWS_DW_FUSE", "1") != "0"   # A/B aid


def _pgrad_prev(param):
    tid = torch._C._current_graph_task_id() if PGRAD_FUSE else -1
    if tid < 0 or param is None or not param.is_leaf:     # (slices of a parameter are fresh objects per call: their ids mean nothing)
        return None
    hit = _PGRAD.get(id(param))
    return hit[1]() if hit is not None and hit[0] == tid and hit[2]() is param else None


def _pgrad_note(param, g):
    tid = torch._C._current_graph_task_id()
    if tid >= 0 and g is not None and param is not None and param.is_leaf:
        _PGRAD[id(param)] = (tid, weakref.ref(g), weakref.ref(param))
    return g


# The same for a NON-LEAF tensor with two consumers in one graph (a projected memory attended to by two decoder passes): the gradient tensor the first
# consumer's backward returned, for the second one to add into.
_TGRAD = {}
_KV_GRAD_FUSE = os.environ.get("ACAI_KV_GRAD_FUSE", "1") != "0"   # A/B and test aid


def _tgrad_key(t):   # (the saved tensor may come back as another Python object: the memory it names is what identifies it inside one backward pass)
    return (t.data_ptr(), tuple(t.shape), t.stride(0))


def _tgrad_prev(t):
    tid = torch._C._current_graph_task_id()
    if tid < 0:
        return None
    hit = _TGRAD.get(_tgrad_key(t))
    return hit[1]() if hit is not None and hit[0] == tid else None


def _tgrad_note(t, g):
    tid = torch._C._current_graph_task_id()
    if tid >= 0:
        if len(_TGRAD) > 256:   # (entries of earlier passes)
            for k in [k for k, v in _TGRAD.items() if v[0] != tid or v[1]() is None]:
                del _TGRAD[k]
        _TGRAD[_tgrad_key(t)] = (tid, weakref.ref(g))


def _wgrad(W, dy, x):
    """dW = dy^T x (fp32, split-K atomics into a zeroed tensor)."""
    prev = _pgrad_prev(W)
    if prev is not None:
        ops.gemm(dy, x, trans_a=True, trans_w=True, out=prev)
        return None
    return _pgrad_note(W, ops.gemm(dy, x, trans_a=True, trans_w=True, out_dtype=torch.float32))


def _wbgrad(W, b, dy, x, cs, need_w, need_b):
    """(dW, db) of an nn.Linear.  When both are wanted and nobody formed the column sums yet (cs), ONE launch reads dy once for both
    (ops.gemm_dw: the weight-gradient kernel sums its dy fragments on the side) instead of a GEMM plus a column-sum pass over dy."""
    need_b = need_b and b is not None
    if not (need_w and need_b and cs is None and dy.dtype == torch.bfloat16):
        return (_wgrad(W, dy, x) if need_w else None), (_bgrad(b, dy, cs) if need_b else None)
    pw, pb = _pgrad_prev(W), _pgrad_prev(b)
    if (pw is None) != (pb is None):
        return _wgrad(W, dy, x), _bgrad(b, dy, cs)
    gw, gb = ops.gemm_dw(dy, x, out=pw, bias_out=pb, want_bias=True)
    if pw is None:
        return _pgrad_note(W, gw), _pgrad_note(b, gb)
    return None, None


def _wgrad_rows(W, r0, r1, dy, x):
    """Rows [r0, r1) of the gradient of a fused parameter (the q / k-v rows of nn.MultiheadAttention.in_proj_weight): the product lands in its
    rows of ONE full-size zeroed tensor per backward pass - autograd's slice backward made a zero-filled full tensor, a copy and an add per
    slice and pass (cross attention: ~100 of each per teacher-forced step)."""
    prev = _pgrad_prev(W)
    if prev is None:
        g = ops._ZEROS.take(W.shape[0], W.shape[1], W.device)
        _pgrad_note(W, g)
        ops.gemm(dy, x, trans_a=True, trans_w=True, out=g[r0:r1])
        return g
    ops.gemm(dy, x, trans_a=True, trans_w=True, out=prev[r0:r1])
    return None


def _bgrad_rows(b, r0, r1, dy, cs=None):
    prev = _pgrad_prev(b)
    g = prev if prev is not None else _pgrad_note(b, ops._ZEROS.take(1, b.shape[0], b.device).view(-1))
    if cs is not None:
        g[r0:r1].add_(cs)
    else:
        ops.colsum(dy, out=g[r0:r1])
    return None if prev is not None else g


def _bgrad(b, dy, cs=None):
    """db = column sums of dy (cs: already formed by the LayerNorm backward that produced dy)."""
    prev = _pgrad_prev(b)
    if prev is not None:
        if cs is not None:
            prev.add_(cs)
        else:
            ops.colsum(dy, out=prev)
        return None
    return _pgrad_note(b, cs if cs is not None else ops.colsum(dy))


# ---- autograd Functions ---------------------------------------------------------------------------------------------------
def _rows_param_grads(W, b, r0, r1, dyc, x, cs, need_w, need_b):
    """Gradient of rows r0:r1 of a fused parameter (the q or the k / v rows of nn.MultiheadAttention.in_proj_weight / _bias) as full-size tensors."""
    if _ROWS_DW_FUSE and need_w and need_b and cs is None and dyc.dtype == torch.bfloat16:
        # both gradients' rows from one pass over dy (ops.gemm_dw), into the full-size tensors of this backward pass
        pw, pb = _pgrad_prev(W), _pgrad_prev(b)
        gw = pw if pw is not None else _pgrad_note(W, ops._ZEROS.take(W.shape[0], W.shape[1], W.device))
        gb = pb if pb is not None else _pgrad_note(b, ops._ZEROS.take(1, b.shape[0], b.device).view(-1))
        ops.gemm_dw(dyc, x, out=gw[r0:r1], bias_out=gb[r0:r1])
        return (None if pw is not None else gw), (None if pb is not None else gb)
    return (_wgrad_rows(W, r0, r1, dyc, x) if need_w else None), (_bgrad_rows(b, r0, r1, dyc, cs) if need_b else None)


class LinearFn(Function):
    """y = x @ W^T + b (+ residual).  x in the compute dtype; y fp32 when a residual is added, else compute dtype (or fp32 on request)."""

    @staticmethod
    def forward(ctx, x, W, b, residual, prec, wc, out_fp32, col_scale=None):
        """col_scale = (n, s): the first n output columns leave the epilogue multiplied by s (a q projection for attention kernels that take a
        prescaled q; the attention backward returns the gradient w.r.t. the UNSCALED output, so nothing changes below)."""
        bf = prec == "bf16"
        Wc, bc = wc.w(W, prec), wc.b(b, prec)   # (a _SliceCache hands out its rows of the fused parameter W / b)
        out_dtype = torch.float32 if (residual is not None or out_fp32 or not bf) else torch.bfloat16
        y = ops.gemm_nt(x, Wc, bc, residual=residual, out_dtype=out_dtype, round_bf16=bf, col_scale=col_scale)
        ctx.save_for_backward(x, W, b)
        ctx.prec, ctx.wc, ctx.has_res, ctx.has_bias = prec, wc, residual is not None, b is not None
        ctx.rows = (wc.r0, wc.r1) if isinstance(wc, _SliceCache) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, b = ctx.saved_tensors
        prec, bf = ctx.prec, ctx.prec == "bf16"
        dy = dy.contiguous()
        dres = dy if ctx.has_res else None
        dyc, cs = _grad_copy_cs(dy, prec)
        dx = ops.gemm_nt(dyc, ctx.wc.wt(W, prec), out_dtype=x.dtype) if ctx.needs_input_grad[0] else None
        if ctx.rows is not None:
            dW, db = _rows_param_grads(W, b, ctx.rows[0], ctx.rows[1], dyc, x, cs, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        else:
            dW, db = _wbgrad(W, b, dyc, x, cs, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dW, db, dres, None, None, None, None


def _compute_copy(x32, prec):
    """The fp32 stream in the compute dtype: the bf16 copy its producing LayerNorm emitted, else one cast (autocast's input cast)."""
    if prec != "bf16" or x32.dtype == torch.bfloat16:
        return x32
    xb = _fwd_take(x32)
    return xb if xb is not None else ops.cast_bf16(x32)


def _grad_copy_cs(dy, prec):
    """Incoming gradient in the compute dtype (the bf16 copy LayerNorm backward emitted, else one cast) and, when the producer formed them,
    its column sums (= this Linear's bias gradient)."""
    bf = prec == "bf16"
    if bf and dy.dtype != torch.bfloat16:
        dyc, cs = _side_take2(dy)
        return (dyc, cs) if dyc is not None else (ops.cast_bf16(dy), None)
    if not bf and dy.dtype != torch.float32:
        return dy.float(), None
    if not bf:
        _, cs = _side_take2(dy)
        return dy, cs
    return dy, None


def _grad_copy(dy, prec):
    return _grad_copy_cs(dy, prec)[0]


class MlpFn(Function):
    """y = x + linear2(GELU(linear1(x))) on the fp32 stream x as ONE autograd node (no inner dropout).  The forward GEMM of linear1 writes the
    pre-activation and its GELU in one epilogue; the backward dX GEMM of linear2 multiplies by the GELU derivative in its epilogue; and the
    dX GEMM of linear1 adds the residual gradient in ITS epilogue - x has a single consumer, so autograd never runs its own accumulation add
    (nor the bf16 -> fp32 copy in front of it)."""

    @staticmethod
    def forward(ctx, x32, W1, b1, W2, b2, prec, wc, has_res=True):
        """has_res = False: y = linear2(GELU(linear1(x))) in the compute dtype, no residual (the transition head, models.py:669-670)."""
        bf = prec == "bf16"
        cdt = torch.bfloat16 if bf else torch.float32
        x = _compute_copy(x32, prec)
        # `a` keeps gelu'(pre-activation), not the pre-activation: the forward epilogue holds Phi(-|a|) for the GELU anyway, and the backward
        # epilogue of linear2's dX GEMM is then ONE multiply per element (it was VALU-bound on the derivative: tools/ablate_gelu_grad.py).
        # ACAI_GELU_KEEP_PRE=1 restores the kept pre-activation (A/B aid; bitwise the round-3 arithmetic)
        a = torch.empty(x.shape[0], W1.shape[0], dtype=cdt, device=x.device)
        if _GELU_KEEP_PRE:
            h = ops.gemm_nt(x, wc.w(W1, prec), wc.b(b1, prec), out_dtype=cdt, gelu=True, round_bf16=bf, pre_act=a)
        else:
            h = ops.gemm_nt(x, wc.w(W1, prec), wc.b(b1, prec), out_dtype=cdt, gelu=True, round_bf16=bf, gelu_grad_out=a)
        y = ops.gemm_nt(h, wc.w(W2, prec), wc.b(b2, prec), residual=x32 if has_res else None, out_dtype=torch.float32 if has_res else cdt, round_bf16=bf)
        ctx.save_for_backward(x, a, h, W1, W2, b1, b2)
        ctx.prec, ctx.wc, ctx.has_res = prec, wc, has_res
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, h, W1, W2, b1, b2 = ctx.saved_tensors
        prec, wc, bf = ctx.prec, ctx.wc, ctx.prec == "bf16"
        dy = dy.contiguous()
        dyc, cs = _grad_copy_cs(dy, prec)
        if _GELU_KEEP_PRE:
            da = ops.gemm_nt(dyc, wc.wt(W2, prec), out_dtype=a.dtype, round_bf16=bf, gelu_grad_of=a)     # (dY . W2) o gelu'(a)
        else:
            da = ops.gemm_nt(dyc, wc.wt(W2, prec), out_dtype=a.dtype, round_bf16=bf, times=a)            # (dY . W2) o [gelu'(a), kept by the forward]
        dW2, db2 = _wbgrad(W2, b2, dyc, h, cs, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        dx = None
        if ctx.needs_input_grad[0]:   # branch gradient (rounded to the compute dtype as the unfused path does) + residual gradient, fp32
            res = (dy.float() if dy.dtype != torch.float32 else dy) if ctx.has_res else None
            dx = ops.gemm_nt(da, wc.wt(W1, prec), residual=res, out_dtype=torch.float32, round_bf16=bf)
        dW1, db1 = _wbgrad(W1, b1, da, x, None, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dW1, db1, dW2, db2, None, None, None


class SelfAttnBlockFn(Function):
    """y = x + out_proj(SDPA(in_proj(x))) on the fp32 stream x as one autograd node (no dropout): as MlpFn, the residual gradient is added in
    the epilogue of the in-projection's dX GEMM."""

    @staticmethod
    def forward(ctx, x32, Wi, bi, Wo, bo, cu, H, max_len, causal, prec, wc):
        bf = prec == "bf16"
        cdt = torch.bfloat16 if bf else torch.float32
        E = x32.shape[1]
        dh = E // H
        x = _compute_copy(x32, prec)
        # the in-projection's epilogue hands q over as q * log2(e) / sqrt(dh) (one rounding, after the scale - torch's math SDPA scales q
        # before the product too): the attention kernels then spend no multiply per score.  The saved qkv holds that q'.
        pre = _q_prescale(prec, E, dh)
        qkv = ops.gemm_nt(x, wc.w(Wi, prec), wc.b(bi, prec), out_dtype=cdt, round_bf16=bf, col_scale=(E, ops.QSCALE(dh)) if pre else None)
        lse = torch.empty(H * x.shape[0], dtype=torch.float32, device=x.device)
        attn = ops.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, cu, H, dh, max_len, causal=causal, lse=lse, q_prescaled=pre)
        y = ops.gemm_nt(attn, wc.w(Wo, prec), wc.b(bo, prec), residual=x32, out_dtype=torch.float32, round_bf16=bf)
        ctx.save_for_backward(x, qkv, attn, lse, cu, Wi, Wo, bi, bo)
        ctx.cfg = (H, dh, max_len, causal, prec, wc, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, qkv, attn, lse, cu, Wi, Wo, bi, bo = ctx.saved_tensors
        H, dh, max_len, causal, prec, wc, pre = ctx.cfg
        bf = prec == "bf16"
        E = H * dh
        dy = dy.contiguous()
        dyc, cs = _grad_copy_cs(dy, prec)
        dattn = ops.gemm_nt(dyc, wc.wt(Wo, prec), out_dtype=attn.dtype, round_bf16=bf)
        dWo, dbo = _wbgrad(Wo, bo, dyc, attn, cs, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        dqkv = torch.empty_like(qkv)
        ops.attn_varlen_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], attn, dattn, lse, cu, cu, H, dh, max_len, max_len, causal,
                            dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:], q_prescaled=pre)   # dq: w.r.t. the UNSCALED in-projection output
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dqkv, wc.wt(Wi, prec), residual=dy.float() if dy.dtype != torch.float32 else dy, out_dtype=torch.float32, round_bf16=bf)
        dWi, dbi = _wbgrad(Wi, bi, dqkv, x, None, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dWi, dbi, dWo, dbo, None, None, None, None, None, None


_FUSED_MLP = os.environ.get("ACAI_FUSED_MLP", "1") != "0"   # A/B aid
_GELU_KEEP_PRE = os.environ.get("ACAI_GELU_KEEP_PRE", "0") == "1"   # A/B aid: MlpFn keeps the pre-activation (round 3) instead of its GELU derivative
_QPRESCALE = os.environ.get("ACAI_QPRESCALE", "1") != "0"   # A/B aid: q leaves the in-projection already scaled for the attention kernels
_LN_COLSUM = os.environ.get("ACAI_LN_COLSUM", "1") != "0"   # A/B aid: LayerNorm backward also forms the consuming Linear's bias gradient


def _mlp(x32, lin1, lin2, p_inner, p_out, prec, wc):
    """linear1 -> GELU -> (dropout) -> linear2 -> (dropout) + residual x32."""
    if _FUSED_MLP and p_inner <= 0.0 and p_out <= 0.0 and lin1.bias is not None and lin2.bias is not None:
        return MlpFn.apply(x32, lin1.weight, lin1.bias, lin2.weight, lin2.bias, prec, wc)
    xc = CastBf16Fn.apply(x32) if prec == "bf16" else x32
    a = LinearFn.apply(xc, lin1.weight, lin1.bias, None, prec, wc, False)
    h = GeluFn.apply(a)
    if p_inner > 0:
        h = DropoutAddFn.apply(h, None, p_inner, _next_seed())
    return _proj_residual(h, lin2.weight, lin2.bias, x32, p_out, prec, wc)


class SelfAttnFn(Function):
    """Packed self attention on a fused qkv tensor [M, 3E] -> [M, E]."""

    @staticmethod
    def forward(ctx, qkv, cu, H, dh, max_len, causal, dropout_p=0.0, pre=False):
        """pre: the q columns of qkv are prescaled (LinearFn col_scale); dqkv is w.r.t. the unscaled projection either way."""
        E = H * dh
        M = qkv.shape[0]
        lse = torch.empty(H * M, dtype=torch.float32, device=qkv.device)
        seed = _next_seed() if dropout_p > 0 else 0
        out = ops.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, cu, H, dh, max_len, causal=causal, lse=lse, dropout_p=dropout_p, seed=seed,
                              q_prescaled=pre)
        ctx.save_for_backward(qkv, out, lse, cu)
        ctx.cfg = (H, dh, max_len, causal, dropout_p, seed, pre)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        H, dh, max_len, causal, dropout_p, seed, pre = ctx.cfg
        E = H * dh
        dqkv = torch.empty_like(qkv)
        dout = dout.contiguous().to(qkv.dtype)
        ops.attn_varlen_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], out, dout, lse, cu, cu, H, dh, max_len, max_len, causal,
                            dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:], dropout_p=dropout_p, seed=seed, q_prescaled=pre)
        return dqkv, None, None, None, None, None, None, None


class CrossAttnFn(Function):
    """Packed cross attention: q [Mq, E], kv [Mk, 2E] -> [Mq, E]."""

    @staticmethod
    def forward(ctx, q, kv, cu_q, cu_k, H, dh, max_q, max_k, dropout_p=0.0, pre=False):
        E = H * dh
        lse = torch.empty(H * q.shape[0], dtype=torch.float32, device=q.device)
        seed = _next_seed() if dropout_p > 0 else 0
        out = ops.attn_varlen(q, kv[:, :E], kv[:, E:], cu_q, cu_k, H, dh, max_q, lse=lse, dropout_p=dropout_p, seed=seed, q_prescaled=pre)
        ctx.save_for_backward(q, kv, out, lse, cu_q, cu_k)
        ctx.cfg = (H, dh, max_q, max_k, dropout_p, seed, pre)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse, cu_q, cu_k = ctx.saved_tensors
        H, dh, max_q, max_k, dropout_p, seed, pre = ctx.cfg
        E = H * dh
        dq = torch.empty_like(q)
        # Two passes over ONE projected memory (scheduled sampling: decoder_forward shares `kv` between them): the pass whose backward runs second adds
        # its dK / dV inside the kernel's epilogue to the tensor the first one handed to autograd (which hands it on to the projection's backward only
        # after both nodes ran) and returns no gradient of its own - instead of autograd summing two [keys, 2E] tensors per layer (12 x ~150 us per
        # teacher-forced step).  Scoped to the current backward pass, as the in-place parameter gradients are.
        # (the kernel's accumulate form exists for bf16 with 16-byte aligned rows and head slices only - the condition of its vector path)
        fusable = _KV_GRAD_FUSE and kv.dtype == torch.bfloat16 and dh % 8 == 0 and kv.stride(0) % 8 == 0 and q.stride(0) % 8 == 0 and kv.stride(1) == 1
        prev = _tgrad_prev(kv) if fusable else None
        dkv = prev if prev is not None else torch.empty_like(kv)
        ops.attn_varlen_bwd(q, kv[:, :E], kv[:, E:], out, dout.contiguous().to(q.dtype), lse, cu_q, cu_k, H, dh, max_q, max_k, False,
                            dq, dkv[:, :E], dkv[:, E:], dropout_p=dropout_p, seed=seed, q_prescaled=pre, accumulate_dkv=prev is not None)
        if prev is not None:
            return dq, None, None, None, None, None, None, None, None, None
        if fusable:
            _tgrad_note(kv, dkv)
        return dq, dkv, None, None, None, None, None, None, None, None


class CrossAttnBlockFn(Function):
    """y = x + out_proj(SDPA(q_proj(x), kv)) on the fp32 stream x as one autograd node (no dropout), kv = the projected memory [Mk, 2E]: as
    SelfAttnBlockFn, x has a single consumer - the residual gradient is added in the epilogue of the q projection's dX GEMM, so autograd runs
    neither its accumulation add nor the bf16 -> fp32 copy in front of it (two launches per decoder layer and pass on the 8208-row stream)."""

    @staticmethod
    def forward(ctx, x32, kv, Win, bin_, Wo, bo, cu_q, cu_k, H, max_q, max_k, prec, wc, attn_mod):
        bf = prec == "bf16"
        cdt = torch.bfloat16 if bf else torch.float32
        E = x32.shape[1]
        dh = E // H
        x = _compute_copy(x32, prec)
        pre = _q_prescale(prec, E, dh)
        sl = _SliceCache(wc, attn_mod, 0, E)
        q = ops.gemm_nt(x, sl.w(Win, prec), sl.b(bin_, prec), out_dtype=cdt, round_bf16=bf, col_scale=(E, ops.QSCALE(dh)) if pre else None)
        lse = torch.empty(H * x.shape[0], dtype=torch.float32, device=x.device)
        attn = ops.attn_varlen(q, kv[:, :E], kv[:, E:], cu_q, cu_k, H, dh, max_q, lse=lse, q_prescaled=pre)
        y = ops.gemm_nt(attn, wc.w(Wo, prec), wc.b(bo, prec), residual=x32, out_dtype=torch.float32, round_bf16=bf)
        ctx.save_for_backward(x, q, kv, attn, lse, cu_q, cu_k, Win, bin_, Wo, bo)
        ctx.cfg = (H, dh, max_q, max_k, prec, wc, pre, attn_mod)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, q, kv, attn, lse, cu_q, cu_k, Win, bin_, Wo, bo = ctx.saved_tensors
        H, dh, max_q, max_k, prec, wc, pre, attn_mod = ctx.cfg
        bf = prec == "bf16"
        E = H * dh
        dy = dy.contiguous()
        dyc, cs = _grad_copy_cs(dy, prec)
        dattn = ops.gemm_nt(dyc, wc.wt(Wo, prec), out_dtype=attn.dtype, round_bf16=bf)
        dWo, dbo = _wbgrad(Wo, bo, dyc, attn, cs, ctx.needs_input_grad[4], ctx.needs_input_grad[5])
        dq = torch.empty_like(q)
        # (the second pass over a shared projected memory adds its dK / dV in the kernel's epilogue: see CrossAttnFn.backward)
        fusable = _KV_GRAD_FUSE and kv.dtype == torch.bfloat16 and dh % 8 == 0 and kv.stride(0) % 8 == 0 and q.stride(0) % 8 == 0 and kv.stride(1) == 1
        prev = _tgrad_prev(kv) if fusable else None
        dkv = prev if prev is not None else torch.empty_like(kv)
        ops.attn_varlen_bwd(q, kv[:, :E], kv[:, E:], attn, dattn, lse, cu_q, cu_k, H, dh, max_q, max_k, False, dq, dkv[:, :E], dkv[:, E:], q_prescaled=pre,
                            accumulate_dkv=prev is not None)
        if prev is None and fusable:
            _tgrad_note(kv, dkv)
        sl = _SliceCache(wc, attn_mod, 0, E)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dq, sl.wt(Win, prec), residual=dy.float() if dy.dtype != torch.float32 else dy, out_dtype=torch.float32, round_bf16=bf)
        dWin, dbin = _rows_param_grads(Win, bin_, 0, E, dq, x, None, ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        return dx, (None if prev is not None else dkv), dWin, dbin, dWo, dbo, None, None, None, None, None, None, None, None


class DropoutAddFn(Function):
    """out = residual + dropout_p(x) (residual may be None).  The mask is regenerated from the seed in backward (nothing is stored)."""

    @staticmethod
    def forward(ctx, x, residual, p, seed):
        ctx.cfg = (p, seed, x.dtype, residual is not None)
        return ops.dropout_add(x.contiguous(), residual, p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed, xdt, has_res = ctx.cfg
        dy = dy.contiguous()
        dx = ops.dropout_add(dy, None, p, seed, out_dtype=xdt)
        return dx, (dy if has_res else None), None, None


def _proj_residual(x, lin_w, lin_b, residual, p, prec, wc):
    """residual + dropout_p(Linear(x)): fused into the GEMM epilogue when p == 0."""
    if p <= 0.0:
        return LinearFn.apply(x, lin_w, lin_b, residual, prec, wc, True)
    y = LinearFn.apply(x, lin_w, lin_b, None, prec, wc, False)
    return DropoutAddFn.apply(y, residual, p, _next_seed())


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        bf = _prec() == "bf16"
        y, yb = ops.layernorm(x, w.detach(), b.detach(), eps, want_bf16=bf)
        if bf:
            _fwd_put(y, yb)   # the next bf16 GEMM reads this copy instead of casting y again
        ctx.save_for_backward(x, w, b)
        ctx.eps = eps
        ctx.autocast = bf
        ctx.bf = bf and x.shape[1] % 256 == 0 and x.shape[1] <= 1024   # the consumer of dx is a bf16 GEMM
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        fused = x.shape[1] % 256 == 0 and x.shape[1] <= 1024   # the one-pass kernel: can also leave dx's column sums for the consumer
        pw, pb = _pgrad_prev(w), _pgrad_prev(b)
        acc = (pw, pb) if (pw is not None and pb is not None) else None     # second use of this LayerNorm in the graph: add to the first's tensors
        if ctx.bf and not _LN_COLSUM:
            dx, dw, db, dxb = ops.layernorm_bwd(x, w.detach(), dy.contiguous().float(), ctx.eps, want_bf16=True, accum_into=acc)
            _side_put(dx, dxb)
        elif ctx.bf:
            dx, dw, db, dxb, cs = ops.layernorm_bwd(x, w.detach(), dy.contiguous().float(), ctx.eps, want_bf16=True, want_colsum=True, accum_into=acc)
            _side_put(dx, dxb, cs)
        elif fused and not ctx.autocast and _LN_COLSUM:
            dx, dw, db, cs = ops.layernorm_bwd(x, w.detach(), dy.contiguous().float(), ctx.eps, want_colsum=True, accum_into=acc)
            _side_put(dx, None, cs)
        else:
            dx, dw, db = ops.layernorm_bwd(x, w.detach(), dy.contiguous().float(), ctx.eps, accum_into=acc)
        _pgrad_note(w, dw), _pgrad_note(b, db)
        return dx, dw, db, None


class CastBf16Fn(Function):
    @staticmethod
    def forward(ctx, x):
        xb = _fwd_take(x)
        return xb if xb is not None else ops.cast_bf16(x)

    @staticmethod
    def backward(ctx, dy):
        return dy.float()


class GeluFn(Function):
    @staticmethod
    def forward(ctx, a):
        ctx.save_for_backward(a)
        return ops.gelu_fwd(a)

    @staticmethod
    def backward(ctx, dh):
        (a,) = ctx.saved_tensors
        return ops.gelu_bwd(a, dh.contiguous().to(a.dtype))


class MatmulKNFn(Function):
    """y[M,N] = a[M,K] @ W[K,N] with W stored [K][N] (an embedding matrix used as a linear map: the expected embedding of scheduled sampling,
    models.py:809).  Under autocast the operands are bf16 and so is y (what `@` does there); dA = dy W^T and dW = a^T dy on the same kernels."""

    @staticmethod
    def forward(ctx, a32, W, prec, wc):
        bf = prec == "bf16"
        ac = ops.cast_bf16(a32.contiguous()) if bf else a32.contiguous()
        Wc = wc.w(W, prec)
        ctx.save_for_backward(ac, Wc)
        ctx.bf = bf
        return ops.gemm(ac, Wc, trans_w=True, out_dtype=torch.bfloat16 if bf else torch.float32)

    @staticmethod
    def backward(ctx, dy):
        ac, Wc = ctx.saved_tensors
        dyc = dy.contiguous()
        dyc = (ops.cast_bf16(dyc.float()) if dyc.dtype != torch.bfloat16 else dyc) if ctx.bf else dyc.float()
        da = ops.gemm_nt(dyc, Wc, out_dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dW = ops.gemm(ac, dyc, trans_a=True, trans_w=True, out_dtype=torch.float32) if ctx.needs_input_grad[1] else None
        return da, dW, None, None


class GatherRowsFn(Function):
    """out[i] = table[idx[i]] (+ add[i]); backward scatter-adds into the table."""

    @staticmethod
    def forward(ctx, table, idx, add, shared_row=-1):
        """shared_row >= 0: the caller guarantees that no other table row is gathered twice (backward then needs no atomics for them)."""
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        ctx.has_add = add is not None
        ctx.shared_row = shared_row
        return ops.gather_rows(table.contiguous(), idx, add)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = dy.contiguous().float()
        dt = None
        if ctx.needs_input_grad[0]:
            dt = ops.scatter_add_rows(dy, idx, torch.zeros(ctx.shape, dtype=torch.float32, device=dy.device), shared_row=ctx.shared_row)
        return dt, None, (dy if ctx.has_add else None), None


class PeInterpFn(Function):
    """OMREncoder.interpolate_pe (models.py:291-302): (Hin, Win, E) table -> (h*w, E) bilinear resampling; backward = transposed stencil."""

    @staticmethod
    def forward(ctx, table, h, w):
        ctx.cfg = (int(h), int(w), tuple(table.shape))
        return ops.pe_interp(table.detach().contiguous(), h, w)

    @staticmethod
    def backward(ctx, dy):
        h, w, shp = ctx.cfg
        return ops.pe_interp_bwd(dy.contiguous().float(), h, w, shp), None, None


def pad_rows(packed, lens, fill_row=None):
    """Differentiable EG.pad_rows: (M,E) -> (B, Lmax, E) and mask; gradient of padded rows is dropped."""
    B, Lm, E, M = len(lens), max(lens), packed.shape[1], packed.shape[0]
    dev = packed.device
    fill = torch.zeros(1, E, device=dev) if fill_row is None else fill_row.detach().reshape(1, E).float()
    table = torch.cat([packed, fill], 0)
    idx = torch.full((B, Lm), M, dtype=torch.int32)
    mask = torch.ones(B, Lm, dtype=torch.bool)
    o = 0
    for b, l in enumerate(lens):
        idx[b, :l] = torch.arange(o, o + l, dtype=torch.int32)
        mask[b, :l] = False
        o += l
    out = GatherRowsFn.apply(table, ops.h2d(idx.reshape(-1), dev), None)
    return out.view(B, Lm, E), ops.h2d(mask, dev)


def unpad_rows(padded, mask):
    B, Lm, E = padded.shape
    lens = [Lm] * B if mask is None else (~mask).sum(dim=1).tolist()
    idx = torch.cat([torch.arange(b * Lm, b * Lm + l, dtype=torch.int32) for b, l in enumerate(lens)]).to(padded.device)
    return GatherRowsFn.apply(padded.reshape(B * Lm, E).float(), idx, None), lens


# ---- building blocks ---------------------------------------------------------------------------------------------------------
def _self_attn_block(x32, sa, cu, H, max_len, causal, p_attn, p_out, prec, wc):
    """in_proj -> SDPA -> out_proj -> (dropout) + residual x32 (torch's _sa_block + residual)."""
    if _FUSED_MLP and p_attn <= 0.0 and p_out <= 0.0 and sa.in_proj_bias is not None and sa.out_proj.bias is not None:
        return SelfAttnBlockFn.apply(x32, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, cu, H, max_len, causal, prec, wc)
    E = x32.shape[1]
    pre = _q_prescale(prec, E, E // H)
    qkv = _lin(x32, sa.in_proj_weight, sa.in_proj_bias, prec, wc, col_scale=(E, ops.QSCALE(E // H)) if pre else None)
    attn = SelfAttnFn.apply(qkv, cu, H, E // H, max_len, causal, p_attn, pre)
    return _proj_residual(attn, sa.out_proj.weight, sa.out_proj.bias, x32, p_out, prec, wc)


def _q_prescale(prec, E, dh):
    """Whether q leaves its projection already scaled for the attention kernels (bf16 path, 16-byte-aligned heads)."""
    return _QPRESCALE and prec == "bf16" and E % 8 == 0 and dh % 8 == 0


def _lin(x32, lin_w, lin_b, prec, wc, residual=None, out_fp32=False, col_scale=None):
    """nn.Linear on an fp32 activation: casts to the compute dtype (autocast's input cast) and applies LinearFn."""
    x = CastBf16Fn.apply(x32) if (prec == "bf16" and x32.dtype != torch.bfloat16) else x32
    return LinearFn.apply(x, lin_w, lin_b, residual, prec, wc, out_fp32, col_scale)


def encoder_stack(stack, x32, cu, max_len, H, prec, wc, training=False):
    """Post-LN blocks with the dropout sites of torch's TransformerEncoderLayer (attention probabilities, dropout1 after the
    out-projection, dropout after the activation, dropout2 after linear2) active when `training`."""
    E = x32.shape[1]
    dh = E // H
    for layer in stack.layers:
        sa = layer.self_attn
        y = _self_attn_block(x32, sa, cu, H, max_len, False, _p_of(sa, training), _p_of(layer.dropout1, training), prec, wc)
        x32 = LayerNormFn.apply(y, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
        y = _mlp(x32, layer.linear1, layer.linear2, _p_of(layer.dropout, training), _p_of(layer.dropout2, training), prec, wc)
        x32 = LayerNormFn.apply(y, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    if stack.norm is not None:
        x32 = LayerNormFn.apply(x32, stack.norm.weight, stack.norm.bias, stack.norm.eps)
    return x32


def _patches(enc, imgs, prec, select=None):
    """Patchify every image into one packed [M, P*P] tensor in the compute dtype (no gradient: inputs are data)."""
    P = enc.patch_size
    dims = [enc._grid(t) for t in imgs]
    lens = [h * w for h, w in dims]
    dev = enc._device()
    with torch.no_grad():
        patches = torch.empty(sum(lens), P * P, dtype=torch.float32, device=dev)
        r0 = 0
        for t in imgs:
            r0 += ops.patchify(t, P, patches, r0)
        if select is not None:
            o, rows = 0, []
            for l, s in zip(lens, select):
                rows.append(ops.h2d(s, dev, torch.int32) + o)
                o += l
            patches = ops.gather_rows(patches, torch.cat(rows).contiguous())
        if prec == "bf16":
            patches = ops.cast_bf16(patches)
    return patches, dims, lens


_GRID_ROWS = {}


def _grid_rows(h_p, w_p, Wm, dev):
    """Row indices of the (h_p, w_p) top-left block of a PE table with Wm columns (int32, on the device, cached)."""
    key = (h_p, w_p, Wm, str(dev))
    t = _GRID_ROWS.get(key)
    if t is None:
        t = ops.h2d((torch.arange(h_p, dtype=torch.int32).unsqueeze(1) * Wm + torch.arange(w_p, dtype=torch.int32).unsqueeze(0)).reshape(-1), dev)
        if len(_GRID_ROWS) < 256:
            _GRID_ROWS[key] = t
    return t


def _pe_rows(enc, table_param, dims, select=None):
    """Differentiable pos_embedding[:h_p,:w_p] rows (optionally a subset per image) of every image, packed.  Grids beyond the table are
    bilinearly interpolated (OMREncoder.batchify does so in every mode, models.py:315-318): their rows are appended to the gather table,
    so the gradient reaches pos_embedding through PeInterpFn."""
    dev, E = table_param.device, table_param.shape[-1]
    Hm, Wm = table_param.shape[0], table_param.shape[1]
    idx, extra = [], []
    base = Hm * Wm
    for i, (h_p, w_p) in enumerate(dims):
        if h_p > Hm or w_p > Wm:
            if not getattr(enc, "_allow_pe_interpolation", False):
                raise ValueError(f"{h_p} x {w_p} image is too large for max positional embedding grid of shape {Hm} x {Wm}")
            extra.append(PeInterpFn.apply(table_param, h_p, w_p))
            rows = torch.arange(base, base + h_p * w_p, dtype=torch.int32, device=dev)
            base += h_p * w_p
        else:
            rows = _grid_rows(h_p, w_p, Wm, dev)
        if select is not None:
            rows = rows[ops.h2d(select[i], dev)]   # device gather: no host round trip per image
        idx.append(rows)
    table = table_param.reshape(-1, E)
    if extra:
        table = torch.cat([table] + extra, 0)
    return GatherRowsFn.apply(table, torch.cat(idx), None)


def encoder_forward_packed(enc, x):
    """Training-path Encoder.forward_packed: returns (x32 packed, None, lens) with autograd history."""
    _check_dropout(enc)
    from ..models.models import _as_image_list
    prec, wc = _prec(), _wc(enc)
    from ..utils import PackedPatches
    if isinstance(x, PackedPatches):     # patch rows written by the resize kernel (utils.DynamicResize.to_patches)
        dims = list(x.dims)
        lens = [h * w for h, w in dims]
        patches = x.patches.to(enc._device())
        with torch.no_grad():
            patches = (patches if patches.dtype == torch.bfloat16 else ops.cast_bf16(patches.float().contiguous())) if prec == "bf16" else patches.float().contiguous()
    else:
        imgs = _as_image_list(x, enc._device())
        patches, dims, lens = _patches(enc, imgs, prec)
    pe = _pe_rows(enc, enc.pos_embedding, dims)
    x32 = LinearFn.apply(patches, enc.projection.weight, enc.projection.bias, pe, prec, wc, True)
    cu = EG.cu_from_lens(lens, x32.device)
    for st in enc._stacks():
        x32 = encoder_stack(st, x32, cu, max(lens), enc._num_heads(), prec, wc, training=enc.training)
    return x32, None, lens


def head_forward(head, x):
    _check_dropout(head)
    prec, wc = _prec(), _wc(head)
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]).float()
    if _FUSED_MLP and _p_of(head[2], head.training) <= 0.0 and head[0].bias is not None and head[3].bias is not None and x2.dtype == torch.float32:
        # no dropout between them: Linear -> GELU -> Linear as ONE node, the GELU (and its derivative, kept for the backward) in the first GEMM's epilogue
        y = MlpFn.apply(x2, head[0].weight, head[0].bias, head[3].weight, head[3].bias, prec, wc, False)
        return y.view(*shp[:-1], y.shape[-1])
    a = _lin(x2, head[0].weight, head[0].bias, prec, wc)
    h = GeluFn.apply(a)
    if _p_of(head[2], head.training) > 0:
        h = DropoutAddFn.apply(h, None, _p_of(head[2], head.training), _next_seed())
    y = LinearFn.apply(h, head[3].weight, head[3].bias, None, prec, wc, False)
    return y.view(*shp[:-1], y.shape[-1])


# ---- MAE (models.py:100-288) -------------------------------------------------------------------------------------------------
def _mae_prepare(mae, x, noises):
    from ..models.models import _as_image_list
    enc = mae.encoder
    imgs = _as_image_list(x, enc._device())
    dims = [enc._grid(t) for t in imgs]
    # mask_sequence (models.py:106-119) for every image; images with the same patch count share ONE batched argsort pair (the reference sorts
    # image by image: 2 x batch sort launches per step).  Everything stays on the device: no host round trip per image.
    dev = enc._device()
    B = len(dims)
    keep, restore, smask, kept = [None] * B, [None] * B, [None] * B, [None] * B
    groups = {}
    for i, (h, w) in enumerate(dims):
        groups.setdefault(h * w, []).append(i)
    for n, members in groups.items():
        k = int(n * (1 - enc.mask_ratio))
        if noises is None:
            noise = torch.rand(len(members), n, device=dev)
        else:
            noise = torch.stack([ops.h2d(noises[i], dev).reshape(n) for i in members])
        ids_shuffle = torch.argsort(noise, dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        seq_mask = (ids_restore >= k).to(torch.int)          # = ones with [:k] zeroed, gathered by ids_restore
        for j, i in enumerate(members):
            keep[i], restore[i], smask[i], kept[i] = ids_shuffle[j, :k], ids_restore[j], seq_mask[j], k
    return imgs, dims, keep, restore, smask, kept


def _mae_embed(enc, imgs, dims, keep, prec):
    """Kept patches -> projection + pos_embedding rows (models.py:153-161), packed.  A projection that is not an nn.Linear (the reference's
    tests swap in nn.Identity, tests/test_mae.py:61) is the caller's module and is simply called on the fp32 patch rows."""
    patches, _, _ = _patches(enc, imgs, prec if isinstance(enc.projection, torch.nn.Linear) else "fp32", select=keep)
    pe = _pe_rows(enc, enc.pos_embedding, dims, select=keep)
    if isinstance(enc.projection, torch.nn.Linear):
        return LinearFn.apply(patches, enc.projection.weight, enc.projection.bias, pe, prec, _wc(enc), True)
    return enc.projection(patches) + pe


def _projection_of_padding(enc):
    """What a zero-padded patch row becomes in the reference's batchify: projection(0) + 0 (models.py:155-161), i.e. the bias."""
    if isinstance(enc.projection, torch.nn.Linear):
        return None if enc.projection.bias is None else enc.projection.bias.detach()
    with torch.no_grad():
        return enc.projection(torch.zeros(1, enc.patch_size ** 2, device=enc._device())).reshape(-1)


def _mae_encode(mae, imgs, dims, keep, kept, prec):
    enc = mae.encoder
    x32 = _mae_embed(enc, imgs, dims, keep, prec)
    cu = EG.cu_from_lens(kept, x32.device)
    return encoder_stack(enc.encoder_blocks, x32, cu, max(kept), enc._num_heads(), prec, _wc(enc), training=enc.training)


class _EncShim:
    def __init__(self, enc):
        self.encoder = enc


def mae_mask_sequence(enc, t, h_p, w_p, noise=None):
    """MAEEncoder.mask_sequence (models.py:106-125): t is ONE unfolded image (1, C P^2, L).  Returns the reference's 6-tuple
    (t_masked (1, C P^2, L_keep), pos_embed_slice (L_keep, E), L, L_keep, seq_mask (L,) int32 in original order, ids_restore (L,))."""
    n = t.shape[-1]
    ids_keep, ids_restore, seq_mask, len_keep = enc.mask_ids(n, t.device, noise)
    t_masked = t.index_select(dim=-1, index=ids_keep)      # pure data movement of the caller's tensor, any dtype
    pe = _pe_rows(enc, enc.pos_embedding, [(h_p, w_p)], select=[ids_keep])
    return t_masked, pe, n, len_keep, seq_mask, ids_restore


def mae_encoder_batchify(enc, x, noises=None):
    """MAEEncoder.batchify (models.py:128-173): the reference's 8-tuple.  Padded rows of `embeddings` hold projection(0), as the reference's
    pad-then-project order leaves them."""
    imgs, dims, keep, restore, smask, kept = _mae_prepare(_EncShim(enc), x, noises)
    x32 = _mae_embed(enc, imgs, dims, keep, _prec())
    lens = [h * w for h, w in dims]
    emb, enc_mask = pad_rows(x32, kept, _projection_of_padding(enc))
    dec_mask = enc.create_attention_mask(lens, max(lens)).to(x32.device)
    seq_masks = torch.nested.as_nested_tensor(smask, layout=torch.jagged)
    ids_restore = torch.nested.as_nested_tensor(restore, layout=torch.jagged)
    return emb, enc_mask, dec_mask, kept, lens, seq_masks, ids_restore, dims


def mae_encoder_forward(enc, x, noises=None):
    """MAEEncoder.forward (models.py:176-180) with the reference's return tuple (padded latent, masks, lens, jagged tensors)."""
    _check_dropout(enc)
    imgs, dims, keep, restore, smask, kept = _mae_prepare(_EncShim(enc), x, noises)
    lat = _mae_encode(_EncShim(enc), imgs, dims, keep, kept, _prec())
    lens = [h * w for h, w in dims]
    padded, _ = pad_rows(lat, kept)
    dec_mask = enc.create_attention_mask(lens, max(lens)).to(lat.device)
    seq_masks = torch.nested.as_nested_tensor(smask, layout=torch.jagged)
    ids_restore = torch.nested.as_nested_tensor(restore, layout=torch.jagged)
    return padded, dec_mask, kept, lens, seq_masks, ids_restore, dims


def _restore_rows(mae, lat, kept, lens, restore, dims):
    """[kept tokens | mask tokens] unshuffled by ids_restore + decoder PE on the packed stream (models.py:219-241): one row gather whose table
    is the packed latent plus ONE mask-token row (the reference materialises N - keep copies of it per image)."""
    D = mae.decoder_hidden_dim
    table = torch.cat([lat, mae.mask_token.reshape(1, D)], 0)
    Mk = lat.shape[0]
    idx, o = [], 0
    for k, n, r in zip(kept, lens, restore):
        idx.append(torch.where(r < k, r + o, Mk))   # on the device: no host round trip per image
        o += k
    dpe = _pe_rows(mae.encoder, mae.decoder_pos_embedding, dims)
    return GatherRowsFn.apply(table, torch.cat(idx).to(torch.int32), dpe, Mk)   # ids_restore is a permutation: only the mask-token row repeats


def mae_prepare_for_decoder(mae, latent, kept_seq_lens, unmasked_seq_lens, batch_ids_restore, patchified_dims):
    """MAE.prepare_for_decoder (models.py:219-241): padded latent (B, L_keep_max, D) -> padded, positionally embedded decoder input
    (B, L_max, D); padded rows are zero."""
    dev = mae.decoder_pos_embedding.device
    B = latent.shape[0]
    kept = [int(k) for k in kept_seq_lens]
    lens = [int(n) for n in unmasked_seq_lens]
    latent = latent.to(device=dev, dtype=torch.float32)
    Lk = latent.shape[1]
    rows = ops.h2d(torch.cat([torch.arange(b * Lk, b * Lk + k, dtype=torch.int32) for b, k in enumerate(kept)]), dev)
    lat = GatherRowsFn.apply(latent.reshape(B * Lk, -1), rows, None)          # remove padding (models.py:225)
    restore = [ops.h2d(batch_ids_restore[i], dev) for i in range(B)]
    x32 = _restore_rows(mae, lat, kept, lens, restore, [tuple(d) for d in patchified_dims])
    return pad_rows(x32, lens)[0]


def mae_decoder_forward(dec, x, attention_mask):
    _check_dropout(dec)
    packed, lens = unpad_rows(x, attention_mask)
    cu = EG.cu_from_lens(lens, packed.device)
    H = dec.decoder_blocks.layers[0].self_attn.num_heads
    y = encoder_stack(dec.decoder_blocks, packed, cu, max(lens), H, _prec(), _wc(dec), training=dec.training)
    return pad_rows(y, lens)[0]


def mae_forward(mae, batch, noises=None, packed=False):
    """MAE.forward (models.py:249-269).  Returns (pred, loss_mask, target) padded as the reference does, or packed
    (pred [sum N, P^2], loss_mask [sum N] bool, target, lens) when packed=True."""
    _check_dropout(mae)
    xs, ys = zip(*batch)
    prec = _prec()
    imgs, dims, keep, restore, smask, kept = _mae_prepare(mae, list(xs), noises)
    lens = [h * w for h, w in dims]
    dev = mae.decoder_pos_embedding.device
    lat = _mae_encode(mae, imgs, dims, keep, kept, prec)
    wc = _wc(mae)
    lat = _lin(lat, mae.decoder_embed.weight, mae.decoder_embed.bias, prec, wc, out_fp32=True)
    x32 = _restore_rows(mae, lat, kept, lens, restore, dims)   # prepare_for_decoder on the packed stream
    cu = EG.cu_from_lens(lens, dev)
    H = mae.decoder.decoder_blocks.layers[0].self_attn.num_heads
    x32 = encoder_stack(mae.decoder.decoder_blocks, x32, cu, max(lens), H, prec, _wc(mae.decoder), training=mae.training)
    pred = _lin(x32, mae.decoder_unembed.weight, mae.decoder_unembed.bias, prec, wc, out_fp32=True)
    # targets and loss mask (no gradient)
    from ..models.models import _as_image_list
    with torch.no_grad():
        timgs = _as_image_list(list(ys), dev)
        target = _patches(mae.encoder, timgs, "fp32")[0]
        loss_mask = torch.cat([ops.h2d(m, dev).bool() for m in smask])
    if packed:
        return pred, loss_mask, target, lens
    ppred, pmask = pad_rows(pred, lens)
    ptarget = EG.pad_rows(target, lens)[0]
    plm = torch.zeros(len(lens), max(lens), dtype=torch.bool, device=dev)
    o = 0
    for b, l in enumerate(lens):
        plm[b, :l] = loss_mask[o:o + l]
        o += l
    return ppred, plm, ptarget


class MaeLossFn(Function):
    @staticmethod
    def forward(ctx, pred, mask_u8, target, count):
        loss, dpred = ops.mae_loss(pred, target, mask_u8, count, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None


def mae_loss(pred, loss_mask, target):
    """MAELoss.forward (models.py:273-288) on padded (N,L,D) or packed (M,D) tensors."""
    D = pred.shape[-1]
    p2 = pred.reshape(-1, D).float().contiguous()
    t2 = target.reshape(-1, D).float().contiguous()
    m = loss_mask.reshape(-1)
    count = float(m.sum().item())
    return MaeLossFn.apply(p2, m.to(torch.uint8).contiguous(), t2, count)


class CeLossFn(Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, count, label_smoothing=0.0):
        loss, dl = ops.ce_loss(logits, target, ignore_index, count, want_grad=True, label_smoothing=label_smoothing)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None


def ce_loss(pred, target_seqs, pad_idx, label_smoothing=0.0):
    """OMRCELoss.forward (models.py:784-796): mean CE over non-pad targets, nn.CrossEntropyLoss's label smoothing."""
    V = pred.shape[-1]
    lg = pred.reshape(-1, V).float().contiguous()
    tg = ops.h2d(target_seqs.reshape(-1), lg.device, torch.int64).contiguous()
    count = float((tg != pad_idx).sum().item())
    return CeLossFn.apply(lg, tg, pad_idx, count, float(label_smoothing))


# ---- teacher-forced decoder (models.py:445-483) ---------------------------------------------------------------------------------
_KV_SHARE = None   # set by shared_cross_kv(): {id(img_latent): {"mem": (mem32, lens_s, memc), "kv": {layer: kv}}}


class shared_cross_kv:
    """Context: decoder passes over the SAME image latent inside it share the packed memory and its per-layer cross K/V projections
    (ScheduledSamplingViTOMR.forward_train runs the decoder twice on one latent, models.py:820-838: the reference recomputes the 12
    memory projections - 65536 x 1024 x 2048 each at config 3 - in the second pass, forward and backward).  Gradients of both passes
    meet on the shared tensors, so the result is the reference's up to summation order."""

    def __enter__(self):
        global _KV_SHARE
        self._prev, _KV_SHARE = _KV_SHARE, {}
        return self

    def __exit__(self, *exc):
        global _KV_SHARE
        _KV_SHARE = self._prev
        return False


def decoder_forward(dec, input_seqs, img_latent, lmx_attention_mask, latent_attention_mask, token_idxs_input=True, checkpoint_grads=False):
    """checkpoint_grads (reference: models.py:470-478, `checkpoint_sequential` with one segment per decoder layer, used by the GRPO loop): each
    layer's activations are dropped after the forward and recomputed in the backward (torch.utils.checkpoint, non-reentrant; the dropout seeds
    come from torch's CPU generator, whose state the checkpoint restores for the recomputation).  Same logits and gradients; the cross K / V
    projection of a layer is part of its checkpointed segment unless it is shared between two passes (then it is computed once, outside)."""
    _check_dropout(dec)
    prec, wc = _prec(), _wc(dec)
    bf = prec == "bf16"
    dev, E, H = dec.pos_embedding.device, dec.hidden_dim, dec.num_heads
    dh = E // H
    B, T = input_seqs.shape[0], input_seqs.shape[1]
    lens_t = [T] * B if lmx_attention_mask is None else (~lmx_attention_mask).sum(dim=1).tolist()
    share = None if _KV_SHARE is None else _KV_SHARE.setdefault((id(img_latent), id(latent_attention_mask), prec), {"kv": {}})
    if share is not None and "mem" in share:
        mem32, lens_s, memc_shared = share["mem"]
    else:
        mem32, lens_s = unpad_rows(img_latent.to(dev), latent_attention_mask)
        memc_shared = None
    pos_idx = ops.h2d(torch.cat([torch.arange(t, dtype=torch.int32) for t in lens_t]), dev)
    pos = GatherRowsFn.apply(dec.pos_embedding, pos_idx, None)
    if token_idxs_input:
        tok = ops.h2d(torch.cat([input_seqs[b, :l] for b, l in enumerate(lens_t)]), dev, torch.int32).contiguous()
        x32 = GatherRowsFn.apply(dec.vocab_embedding.weight, tok, pos)
    else:
        idx = ops.h2d(torch.cat([torch.arange(b * T, b * T + l, dtype=torch.int32) for b, l in enumerate(lens_t)]), dev)
        x32 = GatherRowsFn.apply(input_seqs.reshape(B * T, E).float(), idx, pos)
    cu_t, cu_s = EG.cu_from_lens(lens_t, dev), EG.cu_from_lens(lens_s, dev)
    memc = memc_shared if memc_shared is not None else (CastBf16Fn.apply(mem32) if bf else mem32)
    if share is not None:
        share["mem"] = (mem32, lens_s, memc)
    mt, ms = max(lens_t), max(lens_s)
    tr = dec.training
    pre = _q_prescale(prec, E, dh)

    def layer(ly, x32, memc, kv_shared):
        sa, ca = ly.self_attn, ly.multihead_attn
        y = _self_attn_block(x32, sa, cu_t, H, mt, True, _p_of(sa, tr), _p_of(ly.dropout1, tr), prec, wc)
        x32 = LayerNormFn.apply(y, ly.norm1.weight, ly.norm1.bias, ly.norm1.eps)
        # (the fused parameters go in whole: LinearFn takes its rows through the _SliceCache and writes their gradient into the full-size tensor)
        kv = kv_shared if kv_shared is not None else LinearFn.apply(memc, ca.in_proj_weight, ca.in_proj_bias, None, prec, _SliceCache(wc, ca, E, 3 * E), False)
        if _FUSED_MLP and _p_of(ca, tr) <= 0.0 and _p_of(ly.dropout2, tr) <= 0.0 and ca.in_proj_bias is not None and ca.out_proj.bias is not None:
            y = CrossAttnBlockFn.apply(x32, kv, ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias, cu_t, cu_s, H, mt, ms, prec, wc, ca)
        else:
            xc = CastBf16Fn.apply(x32) if bf else x32
            q = LinearFn.apply(xc, ca.in_proj_weight, ca.in_proj_bias, None, prec, _SliceCache(wc, ca, 0, E), False, (E, ops.QSCALE(dh)) if pre else None)
            a = CrossAttnFn.apply(q, kv, cu_t, cu_s, H, dh, mt, ms, _p_of(ca, tr), pre)
            y = _proj_residual(a, ca.out_proj.weight, ca.out_proj.bias, x32, _p_of(ly.dropout2, tr), prec, wc)
        x32 = LayerNormFn.apply(y, ly.norm2.weight, ly.norm2.bias, ly.norm2.eps)
        y = _mlp(x32, ly.linear1, ly.linear2, _p_of(ly.dropout, tr), _p_of(ly.dropout3, tr), prec, wc)
        return LayerNormFn.apply(y, ly.norm3.weight, ly.norm3.bias, ly.norm3.eps)

    use_ckpt = checkpoint_grads and torch.is_grad_enabled()
    if use_ckpt:
        from torch.utils.checkpoint import checkpoint
    for li, ly in enumerate(dec.decoder_blocks.layers):
        kv = None
        if share is not None:   # two passes over one memory (scheduled sampling): the projection runs once, outside any checkpointed segment
            kv = share["kv"].get(li)
            if kv is None:
                ca = ly.multihead_attn
                kv = share["kv"][li] = LinearFn.apply(memc, ca.in_proj_weight, ca.in_proj_bias, None, prec, _SliceCache(wc, ca, E, 3 * E), False)
        x32 = checkpoint(layer, ly, x32, memc, kv, use_reentrant=False) if use_ckpt else layer(ly, x32, memc, kv)
    nrm = dec.decoder_blocks.norm
    x32 = LayerNormFn.apply(x32, nrm.weight, nrm.bias, nrm.eps)
    logits = _lin(x32, dec.unembed.weight, dec.unembed.bias, prec, wc, out_fp32=True)
    out = pad_rows(logits, lens_t)[0]
    return out.to(torch.bfloat16) if bf else out


class _SliceCache:
    """WeightCache view for a row slice of a fused in_proj parameter (autograd hands LinearFn the sliced tensor, the cache
    is keyed on the full parameter)."""

    def __init__(self, wc, attn, r0, r1):
        self.wc, self.attn, self.r0, self.r1 = wc, attn, r0, r1

    def w(self, _p, prec):
        return self.wc.w(self.attn.in_proj_weight, prec)[self.r0:self.r1]

    def wt(self, _p, prec):
        return self.wc.wt(self.attn.in_proj_weight, prec)[:, self.r0:self.r1]

    def b(self, _p, prec):
        return None if _p is None else self.wc.b(self.attn.in_proj_bias, prec)[self.r0:self.r1]
