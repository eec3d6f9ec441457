"""Fused multi-tensor AdamW for the training steps of the path.

Drop-in for `torch.optim.AdamW` as the reference uses it (`acai_omr/train/pre_train.py:105`: one group, lr 1.5e-4, betas (0.9, 0.95), weight
decay 0.05; `acai_omr/train/omr_teacher_force_train.py:207`: the layer-wise-LR param groups of `acai_omr/models/models.py:761-781`): same
constructor arguments, same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), `param_groups[i]["lr"]` is read every step so
torch LR schedulers (the reference's `LambdaLR`-style cosine/warm-up, `acai_omr/utils/utils.py:204-222`) drive it unchanged.  One HIP launch
(`acai_adamw_step`) updates all tensors; there is no CPU fallback."""
import ctypes
import math

import torch

from . import _lib, ops

CHUNK = 16384  # elements per workgroup


def _mark_modified(tensors):
    """The kernel writes the parameters behind autograd's back: bump their version counters, so that everything keyed on `_version` sees the
    write - `engine.WeightCache` (bf16 / transposed operand copies, refreshed once per optimizer step) and autograd's saved-tensor check."""
    tensors = list(tensors)
    bump = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if bump is not None:
        try:
            bump(tuple(tensors), tuple(t._version + 1 for t in tensors))
            return
        except TypeError:   # older signature
            pass
    torch._foreach_add_(tensors, 0)   # an in-place no-op moves the counters too




class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdamW: amsgrad / maximize are not used by the reference and not built")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._layout = None  # (key, chunk_tensor_dev, chunk_off_dev, n_chunks)

    # ---- tables -----------------------------------------------------------------------------------------------------------------------
    def _active(self):
        out = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedAdamW: dense fp32 parameters and gradients on the GPU only")
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("FusedAdamW: parameters and gradients must be contiguous")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                out.append((gi, p, st))
        return out

    def _build(self, active, dev, steps):
        """Tensor table (pointers + per-tensor bias corrections: torch keeps `step` per parameter, so a parameter that skipped steps - frozen
        for a while, no gradient - carries its own count) and the chunk table.  The chunk table is rebuilt only when the pointers change; the
        tensor table is re-uploaded every step (its bias corrections move, and gradient pointers may: ops._ZeroArena)."""
        key = tuple((gi, p.numel()) for gi, p, st in active)   # the chunk table depends on the tensor sizes only; pointers travel in the tensor table
        arr = (_lib.AcaiAdamWTensor * len(active))()
        for i, (gi, p, st) in enumerate(active):
            arr[i].p, arr[i].g, arr[i].m, arr[i].v = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            arr[i].n, arr[i].group = p.numel(), gi
            b1, b2 = self.param_groups[gi]["betas"]
            arr[i].bias_c1, arr[i].bias_c2_sqrt = 1.0 - b1 ** steps[i], math.sqrt(1.0 - b2 ** steps[i])
        raw = ops.h2d(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8), dev)
        if self._layout is None or self._layout[0] != key:
            ct, co = [], []
            for i, (gi, p, st) in enumerate(active):
                for off in range(0, p.numel(), CHUNK):
                    ct.append(i)
                    co.append(off)
            self._layout = (key, ops.h2d(torch.tensor(ct, dtype=torch.int32), dev), ops.h2d(torch.tensor(co, dtype=torch.int64), dev), len(ct))
        return raw, self._layout[1], self._layout[2], self._layout[3]

    # ---- step ---------------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        active = self._active()
        if not active:
            return loss
        dev = active[0][1].device
        steps = []
        for gi, p, st in active:
            st["step"] += 1
            steps.append(float(st["step"]))
        garr = (_lib.AcaiAdamWGroup * len(self.param_groups))()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            garr[gi].lr, garr[gi].beta1, garr[gi].beta2, garr[gi].eps = float(group["lr"]), b1, b2, group["eps"]
            garr[gi].weight_decay = group["weight_decay"]
        gdev = ops.h2d(torch.frombuffer(bytearray(bytes(garr)), dtype=torch.uint8), dev)
        tdev, ctd, cod, n_chunks = self._build(active, dev, steps)
        with torch.cuda.device(dev):
            st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.lib().acai_adamw_step(tdev.data_ptr(), gdev.data_ptr(), ctd.data_ptr(), cod.data_ptr(), n_chunks, CHUNK, float(grad_scale), st_),
                       "acai_adamw_step")
        _mark_modified(p for _, p, _ in active)
        return loss
