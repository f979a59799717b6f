"""Row-wise top-k kernel (csrc/topk.hip, codetr_topk_*) against torch: the two selections of the detection head
(reference transformer.py:560-561: top 900 of S per-token scores; co_dino_head.py:183-186: top 300 of 900 x 80 sigmoid
scores).  Values must equal torch.topk's exactly (it is a selection, not arithmetic); indices must be the stable
descending order (ties by ascending index), which pins them completely; NaN first as in torch."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _check(x, k):
    from codetr import _cabi, hip_ops

    before = _cabi.CALLS["topk"]
    with torch.no_grad():
        v, i = hip_ops.topk(x, k)
    torch.cuda.synchronize()
    assert _cabi.CALLS["topk"] == before + 1
    # stable descending order of the whole row fixes the indices; every NaN counts as the largest value (torch's
    # documented rule and the CUDA implementation's; torch-ROCm's topk leaves sign-bit NaNs at the bottom)
    xf = x.float()
    nan = torch.isnan(xf)
    key = torch.where(nan, torch.full_like(xf, float("inf")), xf.clamp(min=-3e38, max=3e38))   # +-inf strictly inside NaN
    order = torch.sort(key, dim=-1, descending=True, stable=True)[1][..., :k]
    assert torch.equal(i, order), "indices are not the stable descending order"
    if not nan.any():
        tv, _ = torch.topk(xf, k, dim=-1)
        assert torch.equal(v.float(), tv), "values differ from torch.topk"
    assert torch.equal(torch.gather(x, -1, i).view(torch.int16), v.view(torch.int16))


@pytest.mark.parametrize("rows,n,k", [(1, 204600, 900), (8, 72000, 300), (2, 30785, 900), (1, 30785, 900), (1, 8193, 300), (1, 77, 50), (3, 1000, 1000), (1, 1024, 1024),
                                      (4, 5000, 1), (2, 77, 50), (1, 1, 1)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_matches_torch_topk(rows, n, k, dtype):
    g = torch.Generator(device=DEV).manual_seed(rows * 1000 + n + k)
    _check(torch.randn(rows, n, device=DEV, generator=g).to(dtype), k)


def test_heavy_ties_like_sigmoid_scores():
    """fp16 sigmoid scores cluster in a few hundred distinct values: thousands of exact ties around the threshold"""
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.sigmoid(torch.randn(4, 72000, device=DEV, generator=g) * 0.05).half()
    assert x.unique().numel() < 400
    _check(x, 300)
    _check(torch.zeros(2, 5000, device=DEV, dtype=torch.float16), 300)           # all equal: indices 0..299
    _check(torch.cat((torch.zeros(1, 100, device=DEV), -torch.zeros(1, 100, device=DEV)), 1).half(), 150)


def test_nan_inf_and_negative_values():
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(2, 4000, device=DEV, generator=g).half()
    x[0, 5] = float("nan")
    x[0, 3000] = float("nan")
    x[0, 17] = float("inf")
    x[1, 9] = float("-inf")
    x[1, 100:200] = -65504.0
    _check(x, 900)
    _check(-x.abs(), 64)


def test_unsupported_falls_back_to_torch():
    from codetr import _cabi, hip_ops

    x = torch.randn(2, 5000, device=DEV)
    before = _cabi.CALLS["topk"]
    v, i = hip_ops.topk(x, 10)                      # fp32: library path
    assert _cabi.CALLS["topk"] == before
    tv, ti = torch.topk(x, 10, dim=-1)
    assert torch.equal(v, tv) and torch.equal(i, ti)
    with torch.no_grad():
        v, i = hip_ops.topk(x.half(), 2000)         # k > 1024: library path
    assert _cabi.CALLS["topk"] == before and v.shape == (2, 2000)
