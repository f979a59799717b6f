"""BASELINE full size (Co-DINO Swin-L, 1920x1280, fp16) on the GPU: no CPU oracle run is feasible at this size (the
fp32 oracle needs ~70 s per 608x608 image), so the checks are size-independent properties of the path (③):
  * the output contract of CoDETR.forward (shapes, dtypes, finite, boxes inside the image, scores descending,
    labels in range),
  * determinism (the same input twice -> identical tensors: no atomics on the inference path),
  * image independence (what makes image sharding exact): a batch of two images gives each image the
    detections it gets alone, up to fp16 noise -- the batch changes which GEMM kernel serves a layer (tile shapes are
    chosen by M), not the arithmetic,
  * every hand-written kernel family served the run (call counters)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def swin_l():
    import bench

    return bench.build_model(torch.device(DEV), torch.float16)


def test_full_size_contract_determinism_and_image_independence(swin_l):
    from codetr import _cabi

    H, W = 1280, 1920
    g = torch.Generator(device=DEV).manual_seed(7)
    img = torch.randn(2, 3, H, W, device=DEV, generator=g).half()
    mask = torch.zeros(2, H, W, device=DEV, dtype=torch.float16)
    mask[1, 1100:, :] = 1      # second image padded at the bottom and the right
    mask[1, :, 1700:] = 1
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        b2, s2, l2 = swin_l(img, mask)
        b2r, s2r, l2r = swin_l(img, mask)
        singles = [swin_l(img[i:i + 1], mask[i:i + 1]) for i in range(2)]
    # (the decoder's MSDA / sine embedding / FFN / LayerNorms run inside codetr_decoder_layer_f16)
    for k in ("linear", "layernorm", "window_attention", "decoder_layer", "ffn_fused", "groupnorm_tokens",
              "sine_pos_tokens", "mask_pyramid", "encoder_geometry", "patch_merge_layernorm",
              "msda_encoder", "patch_im2col", "mha_attention", "topk"):
        assert _cabi.CALLS[k] > before[k], f"{k} kernels did not run"
    # contract
    assert b2.shape == (2, 300, 4) and s2.shape == (2, 300) and l2.shape == (2, 300)
    assert b2.dtype == torch.float16 and l2.dtype == torch.int64
    valid = ~torch.isnan(s2)   # padded proposals can carry the reference's NaN semantics (tests/helpers_model.py)
    assert valid.float().mean() > 0.9
    assert torch.isfinite(b2[valid]).all()
    assert (b2[valid] >= 0).all() and (b2[..., 0::2][valid] <= W).all() and (b2[..., 1::2][valid] <= H).all()
    assert ((l2 >= 0) & (l2 < 80)).all()
    sv = torch.where(valid, s2.float(), torch.full_like(s2.float(), -1.0))
    assert (sv[:, :-1] >= sv[:, 1:] - 1e-6).all(), "scores must come out in descending order"
    # determinism
    assert torch.equal(torch.nan_to_num(b2, 0), torch.nan_to_num(b2r, 0)) and torch.equal(l2, l2r)
    assert torch.equal(torch.nan_to_num(s2, 0), torch.nan_to_num(s2r, 0))
    # image independence: the sorted score profile of each image is the one it gets alone (fp16 noise; with random
    # weights scores are nearly tied, so individual detections may swap ranks -- the profile may not move)
    for i, (bs, ss, ls) in enumerate(singles):
        a, b = sv[i, :100], torch.nan_to_num(ss[0, :100].float(), nan=-1.0)
        # (fp16 noise in the encoder also moves a few of the 900 near-tied two-stage proposals in and out of the top-k,
        # so the bound is on the profile, 10 % of the top score, not on individual detections)
        assert (a - b).abs().max() <= 0.1 * a.abs().max() + 2e-3, f"image {i}: batch changed its scores"
