"""GPU parity of codetr_mask_pyramid (csrc/mask_pyramid.hip) against the ATen formulation it replaces:
F.interpolate(nearest) -> bool, cumsum along y / x, get_valid_ratio's sums, flatten + cat.  Integer work: bit-exact."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(img_masks, shapes):
    m4 = img_masks.float().unsqueeze(1)
    masks = [F.interpolate(m4, size=tuple(hw)).to(torch.bool).squeeze(1) for hw in shapes]
    flat = torch.cat([m.flatten(1) for m in masks], 1)
    ycum = [(~m).cumsum(1, dtype=torch.float32) for m in masks]
    xcum = [(~m).cumsum(2, dtype=torch.float32) for m in masks]
    counts = torch.stack([torch.stack((torch.sum(~m[:, 0, :], 1), torch.sum(~m[:, :, 0], 1)), -1) for m in masks], 1)
    return flat, ycum, xcum, counts.float()


def _pyramid_shapes(H, W, strides=(8, 16, 32, 64, 128)):
    return [(-(-H // s), -(-W // s)) for s in strides]


@pytest.mark.parametrize("B,H,W,kind", [
    (1, 1280, 1920, "zeros"),          # the benchmark input: no padding
    (2, 1280, 1920, "padded"),         # batch of two images padded to a common size
    (3, 608, 608, "padded"),
    (2, 333, 517, "random"),           # sizes no stride divides + arbitrary masks: the index rule itself
    (1, 97, 61, "ones"),
])
def test_mask_pyramid_matches_aten(B, H, W, kind):
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(3)
    if kind == "zeros":
        img = torch.zeros(B, H, W, device=DEV)
    elif kind == "ones":
        img = torch.ones(B, H, W, device=DEV)
    elif kind == "random":
        img = (torch.rand(B, H, W, device=DEV, generator=g) < 0.4).float()
    else:
        img = torch.ones(B, H, W, device=DEV)
        for b in range(B):
            h = int(H * (0.55 + 0.45 * b / max(B - 1, 1)))
            w = int(W * (1.0 - 0.37 * b / max(B - 1, 1)))
            img[b, :h, :w] = 0
    shapes = _pyramid_shapes(H, W)
    before = _cabi.CALLS["mask_pyramid"]
    flat, ycum, xcum, counts = hip_ops.mask_pyramid(img, shapes)
    assert _cabi.CALLS["mask_pyramid"] == before + 1
    rflat, rycum, rxcum, rcounts = _reference(img, shapes)
    assert flat.dtype == torch.bool and torch.equal(flat, rflat)
    assert torch.equal(counts, rcounts)
    start = 0
    for lvl, hw in enumerate(shapes):
        y, x = hip_ops.level_cums(ycum, xcum, B, start, hw)
        assert torch.equal(y, rycum[lvl]) and torch.equal(x, rxcum[lvl]), f"level {lvl}"
        start += hw[0] * hw[1]
    # bool / uint8 inputs are taken as they are
    flat2 = hip_ops.mask_pyramid(img.bool(), shapes)[0]
    assert torch.equal(flat2, rflat)


def test_mask_pyramid_rejects_bad_arguments():
    from codetr import _cabi

    img = torch.zeros(1, 64, 64, device=DEV, dtype=torch.bool)
    with pytest.raises(RuntimeError):
        _cabi.mask_pyramid(img, [(8, 8)] * 9)  # more than 8 levels
    with pytest.raises(RuntimeError):
        _cabi.mask_pyramid(img, [(0, 8)])
