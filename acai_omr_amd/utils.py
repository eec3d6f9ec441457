"""Host-side helpers the callers on either side of the hot path use: the detokenise edge of inference and
the learning-rate schedules of the two training loops.  Pure Python / stock torch schedulers; they drive `optim.FusedAdamW` exactly as they
drive `torch.optim.AdamW` (only `param_groups[i]["lr"]` changes); and the image-side resize transforms (8f-2), whose arithmetic runs in a HIP
kernel."""
import math

import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

from .config import LMX_EOS_TOKEN


def stringify_lmx_seq(lmx_seq, idxs_to_tokens):
    """(T,) tensor of LMX token indices starting with <bos> -> one LMX string without <bos> / a trailing <eos>
    (acai_omr/utils/utils.py:194-202; consumer acai_omr/ui/routes.py:68-86)."""
    toks = [idxs_to_tokens[idx.item()] for idx in lmx_seq]
    if toks[-1] == LMX_EOS_TOKEN:
        toks.pop(-1)
    return " ".join(toks[1:])


def stepwise_cosine_anneal_with_warmup(optimizer, warmup_steps, total_epochs, final_lr, num_steps_per_epoch):
    """Linear warm-up from 0.5 % of the base LR, then cosine annealing to final_lr, stepped per minibatch (utils.py:204-208)."""
    warmup = LinearLR(optimizer, start_factor=5e-3, end_factor=1.0, total_iters=warmup_steps)
    anneal = CosineAnnealingLR(optimizer, T_max=total_epochs * num_steps_per_epoch - warmup_steps, eta_min=final_lr)
    return SequentialLR(optimizer, schedulers=[warmup, anneal], milestones=[warmup_steps])


def cosine_anneal_with_warmup(optimizer, warmup_epochs, total_epochs, final_lr, num_train_batches=None):
    """The schedule of pre_train.py:107 (per epoch) and omr_teacher_force_train.py:210 (per minibatch when num_train_batches is given)
    (utils.py:212-222)."""
    if not num_train_batches:
        warmup = LinearLR(optimizer, start_factor=5e-3, end_factor=1.0, total_iters=warmup_epochs)
        anneal = CosineAnnealingLR(optimizer, T_max=total_epochs - warmup_epochs, eta_min=final_lr)
        return SequentialLR(optimizer, schedulers=[warmup, anneal], milestones=[warmup_epochs])
    warm = warmup_epochs * num_train_batches
    warmup = LinearLR(optimizer, start_factor=5e-3, end_factor=1.0, total_iters=warm)
    anneal = CosineAnnealingLR(optimizer, T_max=(total_epochs - warmup_epochs) * num_train_batches, eta_min=final_lr)
    return SequentialLR(optimizer, schedulers=[warmup, anneal], milestones=[warm])


def ragged_collate_fn(batch):
    """DataLoader collate for ragged (image, target) examples: the model layer packs them itself (utils.py:225-229)."""
    return list(batch)


# ---- image-side transforms ------------------------------------------------------------------------------------------------
def dynamic_resize_target(height, width, patch_size, max_seq_len):
    """Target (height, width) of `DynamicResize.forward` (acai_omr/utils/utils.py:343-349): integer aspect ratio (floor division), the short
    side = patch_size * floor(sqrt(max_seq_len / aspect)), the long side = short * aspect."""
    if width > height:
        aspect_ratio = width // height
        target_height = patch_size * math.floor(math.sqrt(max_seq_len / aspect_ratio))
        target_width = target_height * aspect_ratio
    else:
        aspect_ratio = height // width
        target_width = patch_size * math.floor(math.sqrt(max_seq_len / aspect_ratio))
        target_height = target_width * aspect_ratio
    return target_height, target_width


def _center_crop(img, out_h, out_w):
    """torchvision `center_crop` for an image at least as large as the crop (the only case DynamicResize reaches, utils.py:360-364):
    top = round((H - out_h) / 2), left = round((W - out_w) / 2) (Python banker's rounding, as torchvision computes them)."""
    h, w = img.shape[-2], img.shape[-1]
    if out_h > h or out_w > w:
        raise ValueError("center crop larger than the image")
    top, left = int(round((h - out_h) / 2.0)), int(round((w - out_w) / 2.0))
    return img[..., top:top + out_h, left:left + out_w]


def _device_image(img, device=None):
    """GPU tensors stay on THEIR device; CPU tensors are uploaded to `device` (default: the current GPU)."""
    if not torch.is_tensor(img) or img.dim() != 3:
        raise TypeError("expected a C x H x W tensor (decode PIL images with ToImage / ToDtype first, as the reference pipelines do)")
    if not torch.cuda.is_available():
        raise RuntimeError("acai_omr_amd transforms run on the GPU (HIP resize kernel); there is no CPU fallback")
    if img.is_cuda:
        return img.to(dtype=torch.float32).contiguous()
    return img.to(device=device if device is not None else torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32).contiguous()


class PackedPatches:
    """A batch of images as the encoder's projection GEMM wants it: the nn.Unfold(P, P) rows of every image in ONE packed [sum N, P*P] tensor
    (fp32, or bf16 for an autocast encoder) plus the patch grid (h_p, w_p) of each image.  `DynamicResize.to_patches` / `resize_batch_to_patches`
    produce it straight from the resize kernel; `Encoder` / `OMREncoder` / `FineTuneOMREncoder` (`forward`, `forward_packed`) and the
    `inference()` entry point accept it wherever they accept a list of image tensors."""

    def __init__(self, patches, dims, patch_size):
        self.patches, self.dims, self.patch_size = patches, [tuple(d) for d in dims], patch_size

    def __len__(self):
        return len(self.dims)


class PatchDivisibleResize(torch.nn.Module):
    """`PatchDivisibleResize` (acai_omr/utils/utils.py:309-330): resize to the nearest lower patch-divisible size, bicubic + antialias, on the GPU.
    Takes a C x H x W tensor (CPU tensors are uploaded once); returns a GPU tensor."""

    def __init__(self, patch_size, device=None):
        super().__init__()
        self.patch_size = patch_size
        self.device = device   # extension: target GPU for CPU inputs (default: the current device)

    def forward(self, img):
        from . import ops
        img = _device_image(img, self.device)
        _, h, w = img.shape
        new_w = max(w // self.patch_size * self.patch_size, self.patch_size)
        new_h = max(h // self.patch_size * self.patch_size, self.patch_size)
        return ops.resize_bicubic_aa(img, (new_h, new_w))


class DynamicResize(torch.nn.Module):
    """`DynamicResize` (acai_omr/utils/utils.py:334-367): same constructor and the same arithmetic -- target size from the integer aspect ratio and
    the sequence budget, bicubic antialiased resize, optional centre crop to the positional-embedding grid, clamp to [0, 1] -- with the resize and
    the clamp in one HIP launch pair on the GPU.  Takes the float C x H x W tensor the reference's `ToImage -> ToDtype(float32, scale=True)`
    produce (a CPU tensor is uploaded once; this replaces the per-example `.to(device)` of `pre_train.py:56`) and returns a GPU tensor."""

    def __init__(self, patch_size, max_seq_len, pe_max_height, pe_max_width, crop_imgs, device=None):
        super().__init__()
        self.device = device   # extension: target GPU for CPU inputs (default: the current device)
        self.patch_size = patch_size
        self.max_seq_len = max_seq_len
        self.pe_max_height = pe_max_height
        self.pe_max_width = pe_max_width
        self.crop_imgs = crop_imgs

    def forward(self, img):
        from . import ops
        img = _device_image(img, self.device)
        target_height, target_width = dynamic_resize_target(img.shape[-2], img.shape[-1], self.patch_size, self.max_seq_len)
        img = ops.resize_bicubic_aa(img, (target_height, target_width), clamp01=True)
        if self.crop_imgs:
            if target_height / self.patch_size > self.pe_max_height:
                img = _center_crop(img, self.pe_max_height * self.patch_size, img.shape[-1])
            if target_width / self.patch_size > self.pe_max_width:
                img = _center_crop(img, img.shape[-2], self.pe_max_width * self.patch_size)
            img = img.contiguous()
        return img


    # ---- extension (second half): resize straight into the packed patch stream -----------------------------------------------
    def _plan(self, h, w):
        """Target size of `forward` for an h x w input and the centre-crop window (top, left, height, width) inside it."""
        th, tw = dynamic_resize_target(h, w, self.patch_size, self.max_seq_len)
        ch, cw = th, tw
        if self.crop_imgs:
            if th / self.patch_size > self.pe_max_height:
                ch = self.pe_max_height * self.patch_size
            if tw / self.patch_size > self.pe_max_width:
                cw = self.pe_max_width * self.patch_size
        top, left = int(round((th - ch) / 2.0)), int(round((tw - cw) / 2.0))
        return (th, tw), (top, left, ch, cw)

    def to_patches(self, imgs, dtype=torch.float32):
        """`forward` + the encoder's Unfold in one step for a LIST of images: every image - a (1,H,W) / (H,W) float tensor in [0,1] or a uint8
        tensor (what `v2.ToImage` yields; `ToDtype(float32, scale=True)`'s 1/255 is applied on load) - is resized, clamped, cropped and written
        as patch rows of ONE packed tensor by the resize kernel's height pass.  Returns a `PackedPatches` the encoders accept directly;
        patchify(forward(img)) gives the same rows bit for bit."""
        from . import ops
        if torch.is_tensor(imgs):
            imgs = [imgs]
        dev_imgs, plans = [], []
        for img in imgs:
            if not torch.is_tensor(img) or img.dim() not in (2, 3):
                raise TypeError("expected (1,H,W) or (H,W) tensors")
            if not torch.cuda.is_available():
                raise RuntimeError("acai_omr_amd transforms run on the GPU (HIP resize kernel); there is no CPU fallback")
            d = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
            t = img if img.is_cuda else img.to(d)
            t = (t if t.dtype == torch.uint8 else t.to(torch.float32)).contiguous()
            dev_imgs.append(t)
            plans.append(self._plan(t.shape[-2], t.shape[-1]))
        P = self.patch_size
        dims = [(crop[2] // P, crop[3] // P) for _, crop in plans]
        out = torch.empty(sum(h * w for h, w in dims), P * P, dtype=dtype, device=dev_imgs[0].device)
        r0 = 0
        for t, (size, crop) in zip(dev_imgs, plans):
            r0 += ops.resize_to_patches(t, size, P, out, r0, crop=crop, clamp01=True)
        return PackedPatches(out, dims, P)
