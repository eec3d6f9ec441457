"""the resize transforms.  CPU: the numpy oracle against aten's own `_upsample_bicubic2d_aa` (what torchvision's float resize
calls) and DynamicResize's size arithmetic; GPU: the HIP kernel and the `DynamicResize` / `PatchDivisibleResize` mirrors against both."""
import math

import numpy as np
import pytest
import torch

from oracle import resize_oracle

SIZES = [((1, 37, 53), (16, 32)), ((1, 64, 200), (64, 256)), ((3, 50, 41), (75, 90)), ((1, 90, 300), (32, 96)), ((1, 33, 33), (33, 33)),
         ((1, 7, 500), (16, 16)), ((1, 120, 17), (16, 64)), ((1, 1, 1), (4, 4)), ((1, 2, 3), (7, 5)), ((2, 5, 1), (3, 9)), ((1, 40, 40), (1, 1))]


def _aten(img, size):
    return torch.nn.functional.interpolate(img[None], size=size, mode="bicubic", align_corners=False, antialias=True)[0]


@pytest.mark.parametrize("shape,size", SIZES)
def test_oracle_matches_aten(shape, size):
    g = torch.Generator().manual_seed(sum(shape) + sum(size))
    img = torch.rand(*shape, generator=g)
    ref = _aten(img, size).numpy()
    got = resize_oracle.resize_bicubic_aa(img.numpy(), size)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-6


def test_dynamic_resize_target_sizes():
    from acai_omr_amd.utils import dynamic_resize_target
    # utils.py:343-349 by hand: 4:1 system -> aspect 4, 16 * floor(sqrt(1024 / 4)) = 256 high, 1024 wide
    assert dynamic_resize_target(500, 2000, 16, 1024) == (256, 1024)
    assert dynamic_resize_target(500, 2400, 16, 1024) == (256, 1024)          # 2400 // 500 = 4: the ratio is floored
    assert dynamic_resize_target(700, 700, 16, 1024) == (512, 512)            # square goes through the else branch
    assert dynamic_resize_target(3000, 1000, 16, 1024) == (16 * 18 * 3, 16 * 18)
    assert dynamic_resize_target(600, 1000, 16, 4096) == (1024, 1024)         # 1000 // 600 = 1
    for (h, w) in [(480, 1999), (1200, 333), (64, 64)]:
        th, tw = dynamic_resize_target(h, w, 16, 1024)
        assert th % 16 == 0 and tw % 16 == 0 and (th // 16) * (tw // 16) <= 1024


def test_dynamic_resize_reference_cases_sizes():
    """The three cases of the reference's tests/test_datasets.py:127-141 (patch 2, budget 10, grid 4 x 8), on the size arithmetic."""
    from acai_omr_amd.utils import dynamic_resize_target
    for hw in [(6, 10), (10, 6)]:
        th, tw = dynamic_resize_target(*hw, 2, 10)
        assert (tw / 2) * (th / 2) <= 10
    th, tw = dynamic_resize_target(100, 200, 2, 10)
    assert tw / 2 < 8 and th / 2 < 4


def test_transforms_refuse_without_gpu():
    from acai_omr_amd.utils import DynamicResize
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DynamicResize(16, 1024, 60, 200, False)(torch.rand(1, 40, 160))
    with pytest.raises(TypeError):
        DynamicResize(16, 1024, 60, 200, False)("not a tensor")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from acai_omr_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", SIZES + [((1, 1400, 5000), (512, 2048)), ((1, 300, 1100), (512, 2048))])
def test_hip_resize_matches_aten_and_oracle(dev, shape, size):
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + sum(size))
    img = torch.rand(*shape, generator=g)
    got = ops.resize_bicubic_aa(img.to(dev), size).cpu()
    ref = _aten(img, size)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-6          # fp32; the two differ only in fused multiply-adds
    if shape[1] * shape[2] <= 64 * 200:
        assert np.abs(got.numpy() - resize_oracle.resize_bicubic_aa(img.numpy(), size)).max() < 2e-6
    clamped = ops.resize_bicubic_aa(img.to(dev), size, clamp01=True).cpu()
    assert torch.equal(clamped, got.clamp(0.0, 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("hw,crop", [((300, 1250), False), ((1250, 300), False), ((90, 4000), True), ((4000, 90), True), ((333, 333), True)])
def test_dynamic_resize_mirror(dev, hw, crop):
    """End to end against the oracle's restatement of utils.py:343-367 and against aten for the resize itself; bicubic overshoot is clamped; the
    wide / tall cases exceed the 60 x 200 positional grid and are centre-cropped."""
    from acai_omr_amd.utils import DynamicResize
    g = torch.Generator().manual_seed(hw[0] + hw[1])
    img = (torch.rand(1, *hw, generator=g) > 0.5).float()     # black / white: overshoots on both sides
    t = DynamicResize(16, 1024 if not crop else 4096, 60, 200, crop)
    out = t(img)                                              # CPU tensor in, GPU tensor out
    assert out.is_cuda and out.is_contiguous()
    ref = resize_oracle.dynamic_resize(img.numpy(), 16, t.max_seq_len, 60, 200, crop)
    assert tuple(out.shape) == ref.shape
    assert out.shape[-2] % 16 == 0 and out.shape[-1] % 16 == 0
    if crop:
        assert out.shape[-2] // 16 <= 60 and out.shape[-1] // 16 <= 200
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert torch.equal(t(img.to(dev)), out)                   # device tensor in: same result


@pytest.mark.gpu
def test_dynamic_resize_reference_cases(dev):
    """tests/test_datasets.py:127-141 through the transform itself."""
    from acai_omr_amd.utils import DynamicResize
    resize = DynamicResize(2, 10, 4, 8, False)
    for shape in [(1, 6, 10), (1, 10, 6)]:
        img = resize(torch.rand(*shape))
        assert (img.shape[-1] / 2) * (img.shape[-2] / 2) <= 10
    img = resize(torch.rand(1, 100, 200))
    assert img.shape[-1] / 2 < 8 and img.shape[-2] / 2 < 4


@pytest.mark.gpu
def test_patch_divisible_resize_mirror(dev):
    from acai_omr_amd.utils import PatchDivisibleResize
    img = torch.rand(1, 123, 457, generator=torch.Generator().manual_seed(5))
    out = PatchDivisibleResize(16)(img)
    assert tuple(out.shape) == (1, 112, 448)
    assert (out.cpu() - _aten(img, (112, 448))).abs().max() < 2e-6
    assert tuple(PatchDivisibleResize(16)(torch.rand(1, 9, 40)).shape) == (1, 16, 32)   # floor would be 0: minimum one patch


@pytest.mark.gpu
def test_resized_image_feeds_the_encoder(dev):
    """The transform's output is what the encoder's batchify takes: patch-divisible, inside the positional grid, on the device."""
    from acai_omr_amd.utils import DynamicResize
    img = torch.rand(1, 500, 2100, generator=torch.Generator().manual_seed(9))
    out = DynamicResize(16, 1024, 60, 200, False)(img)
    assert tuple(out.shape) == (1, 256, 1024)


@pytest.mark.gpu
def test_resize_properties_at_full_size(dev):
    """Size-independent properties on a scan-sized input (1400 x 5000 -> 512 x 2048): a constant image stays that constant (every window's
    weights sum to one), the map is linear, and the output stays inside the input's range up to the bicubic overshoot bound."""
    from acai_omr_amd import ops
    g = torch.Generator().manual_seed(31)
    x, y = torch.rand(1, 1400, 5000, generator=g).to(dev), torch.rand(1, 1400, 5000, generator=g).to(dev)
    size = (512, 2048)
    c = ops.resize_bicubic_aa(torch.full((1, 1400, 5000), 0.625, device=dev), size)
    assert (c - 0.625).abs().max() < 1e-6
    rx, ry = ops.resize_bicubic_aa(x, size), ops.resize_bicubic_aa(y, size)
    mix = ops.resize_bicubic_aa((0.25 * x + 3.0 * y).contiguous(), size)
    assert (mix - (0.25 * rx + 3.0 * ry)).abs().max() < 2e-5
    assert float(rx.min()) > -0.2 and float(rx.max()) < 1.2


@pytest.mark.gpu
def test_resize_into_patch_stream(dev):
    """SURVEY 8f-2, second half: `DynamicResize.to_patches` = the resize kernel's height pass writing nn.Unfold(P, P) rows of the packed stream
    (fp32 or bf16, uint8 inputs scaled by 1/255 on load, centre crop applied) - the same rows, bit for bit, as patchify(DynamicResize(img)),
    and an encoder fed the `PackedPatches` returns what it returns for the image tensors."""
    from acai_omr_amd import ops
    from acai_omr_amd.models.models import OMREncoder
    from acai_omr_amd.utils import DynamicResize
    g = torch.Generator().manual_seed(5)
    P = 16
    tr = DynamicResize(P, 512, 60, 200, True)
    imgs = [torch.rand(1, 300, 1100, generator=g), torch.rand(1, 97, 1900, generator=g), torch.rand(1, 700, 333, generator=g)]
    pk = tr.to_patches(imgs)
    rows = []
    for im in imgs:
        r = tr(im)
        out = torch.empty((r.shape[-2] // P) * (r.shape[-1] // P), P * P, device=dev)
        ops.patchify(r, P, out, 0)
        rows.append(out)
    want = torch.cat(rows)
    assert pk.patches.shape == want.shape and [h * w for h, w in pk.dims] == [t.shape[0] for t in rows]
    assert torch.equal(pk.patches, want)
    assert torch.equal(tr.to_patches(imgs, dtype=torch.bfloat16).patches, want.to(torch.bfloat16))
    # uint8 images: ToImage's output, ToDtype(float32, scale=True) folded into the load
    u8 = [(im * 255).round().to(torch.uint8) for im in imgs]
    pk8 = tr.to_patches(u8)
    want8 = tr.to_patches([t.float() / 255.0 for t in u8])
    assert torch.equal(pk8.patches, want8.patches)
    # a crop case: a 1:1 aspect at a large budget exceeds a small positional grid in both directions
    tr2 = DynamicResize(P, 4096, 20, 30, True)
    big = torch.rand(1, 900, 1000, generator=g)
    r2 = tr2(big)
    assert r2.shape[-2] == 20 * P and r2.shape[-1] == 30 * P
    o2 = torch.empty(20 * 30, P * P, device=dev)
    ops.patchify(r2, P, o2, 0)
    assert torch.equal(tr2.to_patches([big]).patches, o2)
    # the encoder takes the packed patches wherever it takes images
    enc = OMREncoder(P, 60, 200, num_layers=2, hidden_dim=64, num_heads=2, mlp_dim=128).to(dev).eval()
    with torch.no_grad():
        a, ma = enc([tr(im) for im in imgs])
        b, mb = enc(pk)
    assert torch.equal(ma, mb) and torch.equal(a, b)
