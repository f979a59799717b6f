"""codetr_encoder_projections_posgen_*: the encoder's two projections with the positional operand GENERATED in the kernel
from the running sums of the padding mask, against the same launch READING the tensor codetr_sine_pos_tokens_* writes
(reference positional_encoding.py:58-93 + transformer.py:508-519).  Same fp32 operations in the same order -> the same
bits; the tensor-reading route is itself pinned to the two GEMMs and to fp32 in tests/test_linear_gpu.py, and the encoding
to the reference's SinePositionalEncoding in tests/test_sine_pos_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pyramid(H, W, L=5):
    out, h, w = [], H, W
    for _ in range(L):
        out.append((h, w))
        h, w = -(-h // 2), -(-w // 2)
    return out


def _setup(B, shapes, dtype, padded, seed):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(seed)
    H0, W0 = shapes[0][0] * 8, shapes[0][1] * 8
    masks = torch.zeros(B, H0, W0, dtype=torch.bool, device=DEV)
    if padded:
        for b in range(B):   # bottom / right padding, different per image
            masks[b, H0 - (37 * (b + 1)) % (H0 // 3):, :] = True
            masks[b, :, W0 - (53 * (b + 2)) % (W0 // 3):] = True
    mask_flat, ycum, xcum, _ = hip_ops.mask_pyramid(masks, shapes)
    S = sum(h * w for h, w in shapes)
    K, L = 256, len(shapes)
    level_embed = torch.randn(L, K, device=DEV, generator=g).to(dtype)
    pos = torch.empty(B, S, K, dtype=dtype, device=DEV)
    st, cums = 0, []
    consts = dict(temperature=20.0, scale=2 * 3.141592653589793, eps=1e-6, offset=0.0, normalize=True)
    for l, hw in enumerate(shapes):
        c = hip_ops.level_cums(ycum, xcum, B, st, hw)
        cums.append(c)
        hip_ops.sine_pos_tokens_into(None, pos, st, level_embed[l], K // 2, consts["temperature"], consts["scale"], consts["eps"],
                                     consts["offset"], consts["normalize"], cums=c)
        st += hw[0] * hw[1]
    pos._codetr_posgen = {"B": B, "S": S, "cums": cums, "shapes": [tuple(s) for s in shapes], "level_embed": level_embed, **consts}
    x = torch.randn(B, S, K, device=DEV, generator=g).to(dtype)
    Nv, Np = 256, 512
    wc = (torch.randn(Nv + Np, K, device=DEV, generator=g) / 16).to(dtype)
    bc = torch.randn(Nv + Np, device=DEV, generator=g).to(dtype)
    mask = mask_flat if padded else None
    return x, pos, wc, bc, mask, Nv


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("B,H,W,padded", [(1, 160, 240, False), (2, 100, 152, True), (3, 83, 131, True)],
                         ids=["one_image", "padded", "odd_pyramid"])
def test_generated_operand_is_bit_identical(B, H, W, padded, dtype):
    from codetr import _cabi, hip_ops

    shapes = _pyramid(H, W)
    x, pos, wc, bc, mask, Nv = _setup(B, shapes, dtype, padded, seed=H + W + B)
    outs = []
    with torch.no_grad():
        for gen in (True, False):
            hip_ops.ENC_POSGEN = gen
            before = dict(_cabi.CALLS)
            try:
                both = hip_ops.encoder_projections(x, pos, wc, bc, mask, Nv, 32)
            finally:
                hip_ops.ENC_POSGEN = True
            assert both is not None
            assert _cabi.CALLS["encoder_projections_posgen"] - before["encoder_projections_posgen"] == (1 if gen else 0)
            outs.append(both)
    (v1, p1), (v0, p0) = outs
    assert torch.equal(v1, v0)
    assert torch.equal(p1.view(torch.int16), p0.view(torch.int16))


def test_turn_downs():
    """a recipe for another batch / token count is ignored (the tensor is read), too few rows decline as the plain entry does"""
    from codetr import _cabi, hip_ops

    shapes = _pyramid(160, 240)
    x, pos, wc, bc, mask, Nv = _setup(1, shapes, torch.float16, False, seed=3)
    pos._codetr_posgen["B"] = 2
    with torch.no_grad():
        before = _cabi.CALLS["encoder_projections_posgen"]
        assert hip_ops.encoder_projections(x, pos, wc, bc, mask, Nv, 32) is not None
        assert _cabi.CALLS["encoder_projections_posgen"] == before
        assert hip_ops.encoder_projections(x[:, :100], pos[:, :100], wc, bc, None, Nv, 32) is None
