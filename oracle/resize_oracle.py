"""CPU oracle for the image resize of SURVEY 8f-2.  TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

The reference's `DynamicResize` / `PatchDivisibleResize` (acai_omr/utils/utils.py:309-367) call torchvision's resize on a float32 C x H x W tensor
with `InterpolationMode.BICUBIC, antialias=True`.  torchvision (pinned 0.22.1 in the reference's poetry.lock; NOT installed here, so the
reference's own classes cannot be imported: an ordinary ModuleNotFoundError) forwards that case to
`torch.nn.functional.interpolate(img[None], size, mode="bicubic", align_corners=False, antialias=True)`, i.e. aten's `_upsample_bicubic2d_aa`.
This file restates aten's published algorithm (aten/src/ATen/native/cpu/UpSampleKernel.cpp, `_compute_indices_min_size_weights_aa` and the
separable width-then-height passes) in numpy, keeping its float / double promotions.

Parity is PINNED against that third-party implementation itself: `tests/test_resize.py` compares every function here with
`torch.nn.functional.interpolate(..., antialias=True)` of the installed torch on the CPU (no file of /root/reference is involved), and the
target-size arithmetic of DynamicResize against hand-computed cases read off utils.py:343-349."""
import math

import numpy as np


def _cubic_aa(x):
    a = np.float32(-0.5)
    x = np.abs(np.float32(x))
    if x < 1:
        return np.float32(((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + np.float32(1))
    if x < 2:
        return np.float32((((x - np.float32(5)) * x + np.float32(8)) * x - np.float32(4)) * a)
    return np.float32(0)


def axis_weights(in_size, out_size):
    """Per output index: (xmin, normalised float32 weights) as `_compute_indices_min_size_weights_aa` builds them (bicubic: interp_size 4)."""
    scale = np.float32(in_size) / np.float32(out_size)
    support = np.float32(2.0) * scale if scale >= 1.0 else np.float32(2.0)
    invscale = np.float32(1.0) / scale if scale >= 1.0 else np.float32(1.0)
    max_taps = int(math.ceil(float(support))) * 2 + 1
    out = []
    for i in range(out_size):
        center = np.float32(float(scale) * (i + 0.5))
        xmin = max(int(float(np.float32(center - support)) + 0.5), 0)
        xsize = min(int(float(np.float32(center + support)) + 0.5), in_size) - xmin
        xsize = min(max(xsize, 0), max_taps)
        w = np.zeros(xsize, dtype=np.float32)
        total = np.float32(0)
        for j in range(xsize):
            w[j] = _cubic_aa(np.float32((float(np.float32(j + xmin) - center) + 0.5) * float(invscale)))
            total = np.float32(total + w[j])
        inv = np.float32(1.0) / total if total != 0 else np.float32(0)
        out.append((xmin, (w * inv).astype(np.float32)))
    return out


def resize_bicubic_aa(img, size, clamp01=False):
    """img: float32 array (C, H, W) -> (C, OH, OW); width pass first, then height, sequential float32 accumulation per output sample."""
    img = np.asarray(img, dtype=np.float32)
    C, H, W = img.shape
    OH, OW = size
    wx, wy = axis_weights(W, OW), axis_weights(H, OH)
    tmp = np.zeros((C, H, OW), dtype=np.float32)
    for ox, (xmin, w) in enumerate(wx):
        acc = np.zeros((C, H), dtype=np.float32)
        for j in range(len(w)):
            acc = (acc + img[:, :, xmin + j] * w[j]).astype(np.float32) if j else (img[:, :, xmin] * w[0]).astype(np.float32)
        tmp[:, :, ox] = acc
    out = np.zeros((C, OH, OW), dtype=np.float32)
    for oy, (ymin, w) in enumerate(wy):
        acc = np.zeros((C, OW), dtype=np.float32)
        for j in range(len(w)):
            acc = (acc + tmp[:, ymin + j, :] * w[j]).astype(np.float32) if j else (tmp[:, ymin, :] * w[0]).astype(np.float32)
        out[:, oy, :] = acc
    return np.clip(out, 0.0, 1.0) if clamp01 else out


def dynamic_resize(img, patch_size, max_seq_len, pe_max_height, pe_max_width, crop_imgs):
    """`DynamicResize.forward` (utils.py:343-367) on a float32 (C, H, W) array."""
    height, width = img.shape[-2], img.shape[-1]
    if width > height:
        aspect_ratio = width // height
        target_height = patch_size * math.floor(math.sqrt(max_seq_len / aspect_ratio))
        target_width = target_height * aspect_ratio
    else:
        aspect_ratio = height // width
        target_width = patch_size * math.floor(math.sqrt(max_seq_len / aspect_ratio))
        target_height = target_width * aspect_ratio
    out = resize_bicubic_aa(img, (target_height, target_width))
    if crop_imgs:
        if target_height / patch_size > pe_max_height:
            ch = pe_max_height * patch_size
            top = int(round((out.shape[-2] - ch) / 2.0))
            out = out[:, top:top + ch, :]
        if target_width / patch_size > pe_max_width:
            cw = pe_max_width * patch_size
            left = int(round((out.shape[-1] - cw) / 2.0))
            out = out[:, :, left:left + cw]
    return np.clip(out, 0.0, 1.0)
