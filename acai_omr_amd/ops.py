"""Thin tensor-level wrappers over the C ABI (include/acai_omr_hip.h).

PyTorch is plumbing here: it owns device memory and the current stream; every FLOP of the path runs in
the HIP library.  Every wrapper checks device / dtype / contiguity on the host before the launch (a
faulting kernel can take the whole node down) and raises RuntimeError on a non-zero return code.
"""
import functools

import math

import torch

from . import _lib
from ._lib import ACAI_BF16, ACAI_F32, GEMM_GELU, GEMM_ROUND_BF16  # noqa: F401


def _st(t=None):
    """Raw handle of torch's current stream on `t`'s device (the current device when t is None).  Every public wrapper below runs under
    `_on_operand_device`, which makes the operands' device current first, so both forms name the same stream."""
    return torch.cuda.current_stream(t.device if t is not None else None).cuda_stream


def _on_operand_device(fn):
    """Launch on the OPERANDS' device: a model on cuda:N while another device is current (set_up_omr_inference(device="cuda:1"), a DP rank
    that never called set_device) would otherwise get device 0's stream against device-N pointers - a memory fault without peer access.
    Mixed-device operands are refused."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        idx = -1
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if idx < 0:
                    idx = a.device.index
                elif a.device.index != idx:
                    raise RuntimeError(f"{fn.__name__}: operands on different GPUs (cuda:{idx} and {a.device})")
        if idx >= 0 and idx != torch.cuda.current_device():
            with torch.cuda.device(idx):
                return fn(*args, **kw)
        return fn(*args, **kw)
    return wrapped


def _dt(t):
    if t.dtype == torch.bfloat16:
        return ACAI_BF16
    if t.dtype == torch.float32:
        return ACAI_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _chk(t, name, dtype=None, rows_contig=True):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the product path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if rows_contig and t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: last dim must be contiguous")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    """Row stride of a 2-D tensor whose last dim is contiguous; a single-row tensor reports an arbitrary stride(0)."""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def layernorm(x, w, b, eps, want_f32=True, want_bf16=False, out_f32=None):
    _chk(x, "x", torch.float32)
    assert x.is_contiguous() and x.dim() == 2
    rows, dim = x.shape
    y32 = (out_f32 if out_f32 is not None else torch.empty_like(x)) if want_f32 else None
    y16 = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    assert w.numel() == dim and b.numel() == dim and w.dtype == torch.float32 and b.dtype == torch.float32
    _lib.check(_lib.lib().acai_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), _p(y32), _p(y16), rows, dim, _st()),
               "acai_layernorm_fwd")
    return y32, y16


def gemm_nt(a, w, bias=None, residual=None, out_dtype=torch.float32, gelu=False, round_bf16=False, out=None, pre_act=None, gelu_grad_of=None,
            col_scale=None, gelu_grad_out=None, times=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias) (+ residual).  a, w share a dtype (fp32 or bf16); 2-D, row stride free.
    pre_act (with gelu): [M,N] tensor of out's dtype that receives the pre-activation; gelu_grad_of: [M,N] saved pre-activation whose GELU
    derivative multiplies the (rounded) product; gelu_grad_out (with gelu): [M,N] tensor that receives gelu'(pre-activation) instead (the
    forward epilogue holds Phi(-|a|) anyway) and times: [M,N] saved tensor that multiplies the (rounded) product - together the MLP's backward
    epilogue is one multiply per element; col_scale = (n, s): columns [0, n) of (a @ w^T + bias) are multiplied by s before rounding
    (the in-projection hands q to the attention kernels as q * log2(e) / sqrt(dh), see attn_varlen(q_prescaled=True))."""
    _chk(a, "a"), _chk(w, "w")
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype and a.shape[1] == w.shape[1], (a.shape, w.shape, a.dtype, w.dtype)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == (M, N) and residual.stride(1) == 1
    flags = (GEMM_GELU if gelu else 0) | (GEMM_ROUND_BF16 if round_bf16 else 0)
    given = [(m, t) for m, t in ((1, pre_act), (2, gelu_grad_of), (3, gelu_grad_out), (4, times)) if t is not None]
    assert len(given) <= 1, "one auxiliary operand per launch"
    aux = given[0][1] if given else None
    if aux is not None or col_scale is not None:
        mode = 0
        if aux is not None:
            mode = given[0][0]
            assert aux.shape == (M, N) and aux.dtype == out.dtype and aux.stride(1) == 1
            assert gelu == (mode in (1, 3))
        ncol, cs = col_scale if col_scale is not None else (0, 1.0)
        _lib.check(_lib.lib().acai_gemm_nt_ex(a.data_ptr(), _ld(a), w.data_ptr(), _ld(w), _p(bias), _p(residual),
                                              _ld(residual) if residual is not None else 0, out.data_ptr(), _ld(out),
                                              aux.data_ptr() if aux is not None else None, _ld(aux) if aux is not None else 0,
                                              mode, M, N, K, _dt(a), _dt(out), flags, int(ncol), float(cs), _st()), "acai_gemm_nt_ex")
        return out
    _lib.check(_lib.lib().acai_gemm_nt(a.data_ptr(), _ld(a), w.data_ptr(), _ld(w), _p(bias), _p(residual),
                                       _ld(residual) if residual is not None else 0, out.data_ptr(), _ld(out),
                                       M, N, K, _dt(a), _dt(out), flags, _st()), "acai_gemm_nt")
    return out


def h2d(t, device, dtype=None):
    """Host tensor -> device WITHOUT stalling the host: staged in pinned memory (torch's caching host allocator) and copied asynchronously on
    the current stream.  `t.to(device)` from pageable memory synchronises the stream - inside a training step every such copy (cu_seqlens,
    index lists, optimizer tables) drained the GPU queue and left it idle until the host had enqueued the next kernels (MAE step: 6-9 ms of
    idle GPU per 120 ms step, tools/gpu_idle_from_trace.py).  Device tensors pass through."""
    if t.device.type == "cpu" and torch.device(device).type == "cuda":
        t = t.pin_memory().to(device, non_blocking=True)
    else:
        t = t.to(device)
    return t if dtype is None or t.dtype == dtype else t.to(dtype)


class _ZeroArena:
    """Zeroed fp32 scratch for the split-K weight-gradient GEMMs (they accumulate with float atomics into a zeroed output): slices of a few
    large chunks, each zeroed by ONE fill, instead of one `torch.zeros` launch per weight gradient (the MAE step issued ~380 five-microsecond
    fills, the teacher-forced step ~700).  A slice becomes `param.grad` (autograd keeps the tensor it is handed), so a chunk lives exactly as
    long as the gradients cut from it; a fresh chunk is taken from torch's caching allocator when the current one is used up."""
    CHUNK = 64 << 20   # elements (256 MB)
    SMALL_CHUNK, SMALL_MAX = 1 << 20, 1 << 16   # 4 MB chunks for slices of up to 256 KB: bias / LayerNorm / column-sum accumulators

    def __init__(self):
        self._buf, self._off = {}, {}

    def take(self, rows, cols, device):
        n = rows * cols
        if n >= self.CHUNK // 4:
            return torch.zeros(rows, cols, dtype=torch.float32, device=device)
        # one arena per (device, stream): a chunk is zero-filled on the stream that is current when it is created, and every slice of it is
        # consumed by a GEMM launched on that same stream (ops launch on torch's current stream), so fill and use are ordered.
        # Small slices (the 1-D accumulators that become bias / LayerNorm gradients) come from their own 4 MB chunks: a caller that keeps one
        # such gradient after training must not pin a 256 MB weight-gradient chunk with it.
        small = n <= self.SMALL_MAX
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, small)
        buf, off = self._buf.get(key), self._off.get(key, 0)
        if buf is None or off + n > buf.numel():
            buf, off = torch.zeros(self.SMALL_CHUNK if small else self.CHUNK, dtype=torch.float32, device=device), 0
            self._buf[key] = buf
        self._off[key] = off + ((n + 63) & ~63)   # 256-byte aligned slices
        return buf[off:off + n].view(rows, cols)


    def release(self):
        """Drop the cached chunks (gradients already cut from them keep their chunk alive until they are freed themselves)."""
        self._buf.clear()
        self._off.clear()


_ZEROS = _ZeroArena()


def release_scratch():
    """Free the zeroed weight-gradient arena's current chunks (up to 256 MB per device and stream stay cached after training ends).  Note for
    callers that serialise gradients: a weight gradient produced by `gemm(trans_a=True, trans_w=True)` is a VIEW of a 256 MB chunk -
    `torch.save(p.grad)` writes the whole storage; save `p.grad.clone()` instead (optimizer / model state dicts hold no gradients)."""
    _ZEROS.release()


def gemm(a, w, trans_a=False, trans_w=False, bias=None, residual=None, out_dtype=torch.float32, out=None):
    """out[M,N] = op(a) @ op(w)^T: logical a [M,K] (stored [K,M] if trans_a), logical w [N,K] (stored [K,N] if trans_w)."""
    _chk(a, "a"), _chk(w, "w")
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N, K2 = (w.shape[1], w.shape[0]) if trans_w else w.shape
    assert K == K2, (a.shape, w.shape, trans_a, trans_w)
    if out is None:
        # the weight-gradient form accumulates (split-K atomics) into an fp32 output
        out = _ZEROS.take(M, N, a.device) if (trans_a and trans_w) else torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == (M, N) and residual.stride(1) == 1
    _lib.check(_lib.lib().acai_gemm(a.data_ptr(), _ld(a), 1 if trans_a else 0, w.data_ptr(), _ld(w), 1 if trans_w else 0, _p(bias),
                                    _p(residual), _ld(residual) if residual is not None else 0, out.data_ptr(), _ld(out), M, N, K,
                                    _dt(a), _dt(out), 0, _st()), "acai_gemm")
    return out


def cast_bf16(x):
    _chk(x, "x", torch.float32)
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().acai_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _st()), "acai_cast_f32_bf16")
    return y


def patchify(img, P, out, row0):
    """img (1,H,W) fp32 -> rows row0.. of out [M, P*P]."""
    _chk(img, "img", torch.float32)
    assert img.dim() == 3 and img.shape[0] == 1 and img.is_contiguous()
    H, W = img.shape[1], img.shape[2]
    n = (H // P) * (W // P)
    assert out.dim() == 2 and out.shape[1] == P * P and row0 + n <= out.shape[0] and out.stride(1) == 1
    _lib.check(_lib.lib().acai_patchify(img.data_ptr(), H, W, P, out.data_ptr(), out.stride(0), row0, _dt(out), _st()), "acai_patchify")
    return n


def resize_bicubic_aa(img, size, clamp01=False):
    """img (C,H,W) fp32 on the GPU -> (C,OH,OW): bicubic, antialiased, align_corners=False (the float path of torchvision's resize that
    DynamicResize / PatchDivisibleResize call, acai_omr/utils/utils.py:325-330,351-356); clamp01 fuses DynamicResize's clamp (:367)."""
    _chk(img, "img", torch.float32)
    assert img.dim() == 3 and img.is_contiguous()
    C, H, W = img.shape
    OH, OW = int(size[0]), int(size[1])
    tmp = torch.empty(C, H, OW, dtype=torch.float32, device=img.device)
    out = torch.empty(C, OH, OW, dtype=torch.float32, device=img.device)
    _lib.check(_lib.lib().acai_resize_bicubic_aa(img.data_ptr(), C, H, W, tmp.data_ptr(), out.data_ptr(), OH, OW, int(bool(clamp01)), _st()),
               "acai_resize_bicubic_aa")
    return out


def resize_to_patches(img, size, P, out, row0, crop=None, clamp01=False):
    """One (1,H,W) or (H,W) image on the GPU - fp32 in [0,1], or uint8 (scaled by 1/255 on load) - resized to `size` = (OH, OW) (bicubic,
    antialiased, optional clamp) and written as the nn.Unfold(P, P) rows row0 .. of the packed patch stream `out` [rows, P*P] (fp32 or bf16):
    the resize's height pass stores patch rows directly - no resized image tensor, no patchify launch, no bf16 cast.
    crop = (top, left, h, w): the window of the resized image that is kept (DynamicResize's centre crop).  Returns the number of rows written."""
    assert img.is_cuda and img.dtype in (torch.float32, torch.uint8) and img.is_contiguous()
    if img.dim() == 3:
        assert img.shape[0] == 1, "one channel (NUM_CHANNELS = 1)"
        img = img[0]
    H, W = img.shape
    OH, OW = int(size[0]), int(size[1])
    top, left, ch, cw = (0, 0, OH, OW) if crop is None else [int(v) for v in crop]
    assert out.is_cuda and out.dim() == 2 and out.shape[1] == P * P and out.stride(1) == 1 and out.dtype in (torch.float32, torch.bfloat16)
    n = (ch // P) * (cw // P)
    assert row0 + n <= out.shape[0]
    tmp = torch.empty(H, OW, dtype=torch.float32, device=img.device)
    _lib.check(_lib.lib().acai_resize_to_patches(img.data_ptr(), 1 if img.dtype == torch.uint8 else 0, H, W, tmp.data_ptr(), out.data_ptr(), out.stride(0),
                                                 int(row0), OH, OW, top, left, ch, cw, int(P), _dt(out), int(bool(clamp01)), _st(img)), "acai_resize_to_patches")
    return n


def gather_rows(table, idx, add=None, out=None):
    _chk(table, "table", torch.float32), _chk(idx, "idx", torch.int32)
    assert table.dim() == 2 and table.is_contiguous() and idx.dim() == 1 and idx.is_contiguous()
    rows, dim = idx.numel(), table.shape[1]
    if out is None:
        out = torch.empty(rows, dim, dtype=torch.float32, device=table.device)
    assert out.is_contiguous() and out.shape == (rows, dim)
    if add is not None:
        assert add.shape == (rows, dim) and add.is_contiguous() and add.dtype == torch.float32
    _lib.check(_lib.lib().acai_gather_rows(table.data_ptr(), idx.data_ptr(), _p(add), out.data_ptr(), rows, dim, _st()), "acai_gather_rows")
    return out


def QSCALE(dh):
    """The factor a prescaled q carries: the softmax scale in the log2 domain."""
    return 1.4426950408889634 / math.sqrt(dh)


def attn_varlen(q, k, v, cu_q, cu_k, H, dh, max_q, causal=False, out=None, lse=None, dropout_p=0.0, seed=0, q_prescaled=False):
    """q [Mq, >=H*dh], k/v [Mk, >=H*dh] (2-D views with any row stride), cu_* int32 [B+1] on the GPU.
    q_prescaled: q already carries log2(e) / sqrt(dh) (QSCALE(dh); gemm_nt(col_scale=...) on the in-projection)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, n)
        assert t.dim() == 2 and t.dtype == q.dtype
    _chk(cu_q, "cu_q", torch.int32), _chk(cu_k, "cu_k", torch.int32)
    B = cu_q.numel() - 1
    assert cu_k.numel() == B + 1
    if out is None:
        out = torch.empty(q.shape[0], H * dh, dtype=q.dtype, device=q.device)
    assert out.shape[0] == q.shape[0] and out.stride(1) == 1 and out.dtype == q.dtype
    _lib.check(_lib.lib().acai_attn_varlen_fwd(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                               out.data_ptr(), out.stride(0), cu_q.data_ptr(), cu_k.data_ptr(), B, H, dh, int(max_q),
                                               1 if causal else 0, _dt(q), _p(lse), q.shape[0], float(dropout_p), int(seed) & 0xFFFFFFFF,
                                               1 if q_prescaled else 0, _st()),
               "acai_attn_varlen_fwd")
    return out


def cross_kv_prefill(mem, w_kv, b_kv, row_seq, row_pos, seq_off, seq_len, k_out, v_out, H, dh, dhp, round_bf16=False):
    _chk(mem, "mem"), _chk(w_kv, "w_kv")
    M, E = mem.shape
    assert w_kv.shape == (2 * E, E) and mem.dtype == w_kv.dtype == k_out.dtype == v_out.dtype
    assert row_seq.dtype == torch.int32 and row_pos.dtype == torch.int32 and seq_off.dtype == torch.int64 and seq_len.dtype == torch.int32
    assert row_seq.numel() == M and row_pos.numel() == M
    _lib.check(_lib.lib().acai_cross_kv_prefill(mem.data_ptr(), mem.stride(0), w_kv.data_ptr(), w_kv.stride(0), _p(b_kv), row_seq.data_ptr(),
                                                row_pos.data_ptr(), seq_off.data_ptr(), seq_len.data_ptr(), k_out.data_ptr(), v_out.data_ptr(),
                                                M, E, H, dh, dhp, _dt(mem), GEMM_ROUND_BF16 if round_bf16 else 0, _st()), "acai_cross_kv_prefill")


def skinny_gemm(x, w, bias=None, residual=None, gelu=False, round_bf16=False, out=None):
    """x [B,K] fp32, w [N,K] fp32 or bf16 -> y [B,N] fp32."""
    _chk(x, "x", torch.float32), _chk(w, "w")
    B, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    flags = (GEMM_GELU if gelu else 0) | (GEMM_ROUND_BF16 if round_bf16 else 0)
    _lib.check(_lib.lib().acai_skinny_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _p(bias), _p(residual),
                                           residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), B, N, K,
                                           _dt(w), flags, _st()), "acai_skinny_gemm")
    return out


def skinny_gemm_ex(x, w, bias=None, residual=None, gelu=False, round_bf16=False, out_dtype=torch.float32, ln=None, stats_out=None,
                   rln=None, rstats=None, ln_eps=1e-5):
    """Decode GEMV with its fusions: x [B,K] fp32 or bf16, w [N,K]; ln = (weight, bias) applied to x on load;
    rln = (weight, bias) + rstats [B,2] applied to the residual."""
    _chk(x, "x"), _chk(w, "w")
    B, K = x.shape
    N = w.shape[0]
    out = torch.empty(B, N, dtype=out_dtype, device=x.device)
    flags = (GEMM_GELU if gelu else 0) | (GEMM_ROUND_BF16 if round_bf16 else 0)
    _lib.check(_lib.lib().acai_skinny_gemm_ex(x.data_ptr(), x.stride(0), _dt(x), w.data_ptr(), w.stride(0), _p(bias), _p(residual),
                                              residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), _dt(out), B, N, K,
                                              _dt(w), flags, _p(ln[0]) if ln else None, _p(ln[1]) if ln else None, float(ln_eps), _p(stats_out),
                                              _p(rln[0]) if rln else None, _p(rln[1]) if rln else None, _p(rstats), _st()), "acai_skinny_gemm_ex")
    return out


def decode_attn(q, kc, vc, seq_off, seq_len, H, dh, dhp, max_len, round_out=False, chunk=256, fused_merge=False):
    """q [B, H*dh] fp32 (row stride free); kc/vc flat caches addressed by seq_off / seq_len (see header)."""
    _chk(q, "q", torch.float32)
    B = q.shape[0]
    nsplit = max(1, -(-int(max_len) // chunk))
    partial = torch.empty(B * H * nsplit * (dhp + 2), dtype=torch.float32, device=q.device)
    out = torch.empty(B, H * dh, dtype=torch.float32, device=q.device)
    tickets = torch.zeros(B * H, dtype=torch.int32, device=q.device) if fused_merge else None
    _lib.check(_lib.lib().acai_decode_attn(q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), seq_off.data_ptr(), seq_len.data_ptr(),
                                           partial.data_ptr(), out.data_ptr(), out.stride(0), B, H, dh, dhp, chunk, nsplit, _dt(kc),
                                           1 if round_out else 0, tickets.data_ptr() if tickets is not None else None, _st()), "acai_decode_attn")
    if tickets is not None:
        assert int(tickets.abs().sum().item()) == 0, "arrival counters must re-arm to zero"
    return out


_BWD_WS = {}   # (device, stream) -> the one-pass attention backward's fp32 query-gradient workspace (used only inside a call, in stream order)


def _bwd_workspace(device, nbytes):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _BWD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _BWD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def attn_varlen_bwd(q, k, v, o, dout, lse, cu_q, cu_k, H, dh, max_q, max_k, causal, dq, dk, dv, dropout_p=0.0, seed=0, q_prescaled=False, accumulate_dkv=False,
                    lend_workspace=True):
    """Gradients of attn_varlen w.r.t. q, k, v, written into the (strided) views dq, dk, dv.  q_prescaled: q is the forward's prescaled
    q; dq is still the gradient w.r.t. the unscaled in-projection output.  accumulate_dkv: dk, dv += (bf16, aligned operands only).
    lend_workspace=False: no workspace is offered, i.e. the two-kernel form whose dq is bit-reproducible (acai_attn_varlen_bwd_ws)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (dout, "dout"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk(t, n)
        assert t.dim() == 2 and t.dtype == q.dtype, n
    _chk(lse, "lse", torch.float32)
    B = cu_q.numel() - 1
    total_q = q.shape[0]
    assert lse.numel() == H * total_q and dout.shape == o.shape
    delta = torch.empty(H * total_q, dtype=torch.float32, device=q.device)
    flags = (1 if causal else 0) | (2 if accumulate_dkv else 0)
    total_k = k.shape[0]
    need = _lib.lib().acai_attn_varlen_bwd_workspace_bytes(B, H, dh, int(max_q), int(max_k), total_q, total_k, flags, _dt(q), float(dropout_p), 1 if q_prescaled else 0)
    ws = _bwd_workspace(q.device, need) if need and lend_workspace else None
    _lib.check(_lib.lib().acai_attn_varlen_bwd_ws(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), o.data_ptr(), o.stride(0),
                                                  dout.data_ptr(), dout.stride(0), dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(),
                                                  dv.stride(0), lse.data_ptr(), delta.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(), B, H, dh, int(max_q),
                                                  int(max_k), total_q, total_k, flags, _dt(q), float(dropout_p), int(seed) & 0xFFFFFFFF,
                                                  1 if q_prescaled else 0, _p(ws), need if ws is not None else 0, _st()),
               "acai_attn_varlen_bwd_ws")


def dropout_add(x, residual, p, seed, out_dtype=None):
    """out = residual + keep * x / (1 - p) (residual fp32 or None); the mask depends only on (seed, row, col)."""
    _chk(x, "x")
    assert x.dim() == 2 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=out_dtype or (torch.float32 if residual is not None else x.dtype), device=x.device)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == x.shape and residual.is_contiguous()
    _lib.check(_lib.lib().acai_dropout_add(x.data_ptr(), _p(residual), out.data_ptr(), x.shape[0], x.shape[1], float(p), int(seed) & 0xFFFFFFFF,
                                           _dt(x), _dt(out), _st()), "acai_dropout_add")
    return out


def layernorm_bwd(x, w, dy, eps, want_param_grads=True, want_bf16=False, want_colsum=False, accum_into=None):
    """Returns (dx, dw, db) or, with want_bf16 (dim % 256 == 0, dim <= 1024), (dx, dw, db, dx_bf16); want_colsum appends the column sums of
    dx (of its bf16 copy when one is written): the bias gradient of the nn.Linear whose output this LayerNorm normalised.
    accum_into = (dw, db): existing fp32 vectors the parameter gradients are ADDED to (returned dw / db are then None)."""
    _chk(x, "x", torch.float32), _chk(dy, "dy", torch.float32)
    assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape
    rows, dim = x.shape
    dx = torch.empty_like(x)
    dxb = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    n_acc = (2 if want_param_grads else 0) + (1 if want_colsum else 0)
    acc = _ZEROS.take(max(n_acc, 1), dim, x.device)   # zeroed accumulators: slices of the arena's chunk (one fill per 256 MB, not one per call)
    dw = acc[0] if want_param_grads else None
    db = acc[1] if want_param_grads else None
    dcs = acc[n_acc - 1] if want_colsum else None
    if accum_into is not None:
        assert want_param_grads
        dw, db = accum_into
        _chk(dw, "dw", torch.float32), _chk(db, "db", torch.float32)
        assert dw.is_contiguous() and db.is_contiguous() and dw.numel() == dim and db.numel() == dim
    _lib.check(_lib.lib().acai_layernorm_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), float(eps), dx.data_ptr(), _p(dxb), _p(dw), _p(db), _p(dcs),
                                             stats.data_ptr(), rows, dim, _st()), "acai_layernorm_bwd")
    if accum_into is not None:
        dw = db = None
    out = (dx, dw, db, dxb) if want_bf16 else (dx, dw, db)
    return out + (dcs,) if want_colsum else out


def gemm_dw(dy, x, out=None, bias_out=None, want_bias=False):
    """Both parameter gradients of an nn.Linear from one pass over dy: dW[M, N] (+)= dy[K, M]^T x[K, N] and, with want_bias or bias_out,
    db[M] (+)= column sums of dy.  out / bias_out: existing fp32 gradients to ACCUMULATE into (else zeroed arena slices).  Returns (dW, db)."""
    _chk(dy, "dy"), _chk(x, "x")
    assert dy.dim() == 2 and x.dim() == 2 and dy.dtype == x.dtype and dy.shape[0] == x.shape[0] and dy.stride(1) == 1 and x.stride(1) == 1
    K, M = dy.shape
    N = x.shape[1]
    if out is None:
        out = _ZEROS.take(M, N, dy.device)
    if bias_out is None and want_bias:
        bias_out = _ZEROS.take(1, M, dy.device).view(-1)
    assert out.shape == (M, N) and out.dtype == torch.float32 and out.stride(1) == 1
    if bias_out is not None:
        _chk(bias_out, "bias_out", torch.float32)
        assert bias_out.is_contiguous() and bias_out.numel() == M
    _lib.check(_lib.lib().acai_gemm_dw(dy.data_ptr(), _ld(dy), x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), _p(bias_out), M, N, K, _dt(dy), _st()),
               "acai_gemm_dw")
    return out, bias_out


def cast_weights(items):
    """One launch for the operand copies of many fp32 parameters (engine.WeightCache after an optimizer step).  items: (src, want16, want16t,
    want32r) with src a contiguous fp32 CUDA tensor of one or two dimensions; returns per item (bf16 copy | None, transposed bf16 copy
    [cols, rows] | None, bf16-rounded fp32 copy | None).  Replaces `t.to(bfloat16)`, `out.copy_(t.t())` and `t.to(bfloat16).to(float32)`."""
    if not items:
        return []
    dev = items[0][0].device
    arr = (_lib.AcaiCastEntry * len(items))()
    outs, tiles = [], 0
    for i, (src, w16, w16t, w32r) in enumerate(items):
        _chk(src, "src", torch.float32)
        assert src.is_contiguous() and src.dim() in (1, 2) and src.device == dev
        rows, cols = (1, src.shape[0]) if src.dim() == 1 else src.shape
        d16 = torch.empty(src.shape, dtype=torch.bfloat16, device=dev) if w16 else None
        d16t = torch.empty((cols, rows), dtype=torch.bfloat16, device=dev) if w16t else None
        d32 = torch.empty(src.shape, dtype=torch.float32, device=dev) if w32r else None
        arr[i].src, arr[i].dst16, arr[i].dst16t, arr[i].dst32r = src.data_ptr(), _p(d16), _p(d16t), _p(d32)
        arr[i].rows, arr[i].cols, arr[i].tile0 = rows, cols, tiles
        tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
        outs.append((d16, d16t, d32))
    table = h2d(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8), dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().acai_cast_weights(table.data_ptr(), len(items), tiles, _st()), "acai_cast_weights")
    return outs


def gelu_fwd(a):
    _chk(a, "a")
    assert a.is_contiguous()
    h = torch.empty_like(a)
    _lib.check(_lib.lib().acai_gelu_fwd(a.data_ptr(), h.data_ptr(), a.numel(), _dt(a), _st()), "acai_gelu_fwd")
    return h


def gelu_bwd(a, dh):
    _chk(a, "a"), _chk(dh, "dh")
    assert a.is_contiguous() and dh.is_contiguous() and a.dtype == dh.dtype and a.shape == dh.shape
    da = torch.empty_like(a)
    _lib.check(_lib.lib().acai_gelu_bwd(a.data_ptr(), dh.data_ptr(), da.data_ptr(), a.numel(), _dt(a), _st()), "acai_gelu_bwd")
    return da


def colsum(x, out=None):
    """Column sums of x (fp32).  out: an existing fp32 vector to ADD them to (the kernel accumulates with float atomics)."""
    _chk(x, "x")
    assert x.dim() == 2
    if out is None:
        out = _ZEROS.take(1, x.shape[1], x.device).view(-1)
    else:
        _chk(out, "out", torch.float32)
        assert out.is_contiguous() and out.numel() == x.shape[1]
    _lib.check(_lib.lib().acai_colsum(x.data_ptr(), x.stride(0), out.data_ptr(), x.shape[0], x.shape[1], _dt(x), _st()), "acai_colsum")
    return out


def scatter_add_rows(src, idx, dst, shared_row=-1):
    """dst[idx[r]] += src[r].  shared_row >= 0: the only index that may repeat (everything else is updated without atomics)."""
    _chk(src, "src", torch.float32), _chk(idx, "idx", torch.int32), _chk(dst, "dst", torch.float32)
    assert src.is_contiguous() and dst.is_contiguous() and src.shape[1] == dst.shape[1] and idx.numel() == src.shape[0]
    _lib.check(_lib.lib().acai_scatter_add_rows(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], int(shared_row), _st()),
               "acai_scatter_add_rows")
    return dst


def mae_loss(pred, target, mask, count, want_grad):
    """pred/target [R,D] fp32, mask [R] uint8; returns (loss scalar tensor, dpred or None) for loss = masked mean."""
    _chk(pred, "pred", torch.float32), _chk(target, "target", torch.float32), _chk(mask, "mask", torch.uint8)
    assert pred.is_contiguous() and target.is_contiguous() and pred.shape == target.shape and mask.numel() == pred.shape[0]
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    _lib.check(_lib.lib().acai_mae_loss(pred.data_ptr(), target.data_ptr(), mask.data_ptr(), 1.0 / float(count), loss.data_ptr(), _p(dpred),
                                        pred.shape[0], pred.shape[1], _st()), "acai_mae_loss")
    return loss[0], dpred


def pe_interp(table, h_out, w_out):
    """table (Hin, Win, E) fp32 -> (h_out * w_out, E): bilinear, align_corners=False (OMREncoder.interpolate_pe, models.py:291-302)."""
    _chk(table, "table", torch.float32)
    assert table.dim() == 3 and table.is_contiguous()
    Hin, Win, E = table.shape
    out = torch.empty(int(h_out) * int(w_out), E, dtype=torch.float32, device=table.device)
    _lib.check(_lib.lib().acai_pe_interp_fwd(table.data_ptr(), Hin, Win, E, out.data_ptr(), int(h_out), int(w_out), _st(table)), "acai_pe_interp_fwd")
    return out


def pe_interp_bwd(dout, h_out, w_out, table_shape):
    """Gradient of pe_interp w.r.t. the table: (Hin, Win, E) fp32."""
    _chk(dout, "dout", torch.float32)
    Hin, Win, E = table_shape
    assert dout.is_contiguous() and dout.shape == (int(h_out) * int(w_out), E)
    dt = torch.zeros(Hin, Win, E, dtype=torch.float32, device=dout.device)
    _lib.check(_lib.lib().acai_pe_interp_bwd(dout.data_ptr(), int(h_out), int(w_out), E, dt.data_ptr(), Hin, Win, _st(dout)), "acai_pe_interp_bwd")
    return dt


def ce_loss(logits, target, ignore_index, count, want_grad, label_smoothing=0.0):
    _chk(logits, "logits", torch.float32), _chk(target, "target", torch.int64)
    assert logits.dim() == 2 and target.numel() == logits.shape[0] and target.is_contiguous()
    loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty(logits.shape, dtype=torch.float32, device=logits.device) if want_grad else None
    _lib.check(_lib.lib().acai_ce_loss(logits.data_ptr(), logits.stride(0), target.data_ptr(), int(ignore_index), 1.0 / float(count), float(label_smoothing),
                                       loss.data_ptr(), _p(dl), logits.shape[0], logits.shape[1], _st(logits)), "acai_ce_loss")
    return loss[0], dl


for _name, _fn in list(globals().items()):
    if callable(_fn) and not _name.startswith("_") and getattr(_fn, "__module__", None) == __name__ and not isinstance(_fn, type):
        globals()[_name] = _on_operand_device(_fn)
del _name, _fn


class Graph:
    """hipGraph capture/replay of whatever is enqueued on the current stream between begin() and end()."""

    def __init__(self):
        self.exec = None

    def begin(self):
        _lib.check(_lib.lib().acai_graph_begin(_st()), "acai_graph_begin")

    def end(self):
        import ctypes
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().acai_graph_end(_st(), ctypes.byref(h)), "acai_graph_end")
        self.exec = h

    def launch(self):
        _lib.check(_lib.lib().acai_graph_launch(self.exec, _st()), "acai_graph_launch")

    def __del__(self):
        if self.exec is not None and _lib._lib is not None:
            _lib._lib.acai_graph_destroy(self.exec)
            self.exec = None
