"""Dense multi-head attention kernel (csrc/mha_attention.hip, codetr_mha_attention_*) against the fp32 formula
softmax(q k^T / sqrt(32)) v -- the core of nn.MultiheadAttention in the decoder's self-attention (reference
transformer_mmcv.py:394-428).  Cases: the decoder shape (900 queries, 8 heads), key counts around the 128-key chunk and
32-row padding boundaries, strided q / k views of a fused projection, large logits, bf16."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ref(q, k, v, H):
    B, Nq, C = q.shape
    sp = lambda t: t.float().reshape(B, -1, H, 32).transpose(1, 2)  # noqa: E731
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / 32 ** 0.5, -1)
    return (a @ sp(v)).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("B,N,H", [(2, 900, 8), (1, 1, 1), (1, 17, 2), (3, 127, 4), (1, 128, 8), (1, 129, 8),
                                   (1, 1000, 8), (1, 1024, 2)])
def test_matches_fp32_softmax_attention(B, N, H):
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(N + H)
    C = H * 32
    q, k, v = (torch.randn(B, N, C, device=DEV, generator=g).half() for _ in range(3))
    before = _cabi.CALLS["mha_attention"]
    out = hip_ops.mha_self_attention(q, k, v, H)
    torch.cuda.synchronize()
    assert _cabi.CALLS["mha_attention"] == before + 1
    torch.testing.assert_close(out.float(), _ref(q, k, v, H), rtol=5e-3, atol=3e-3)


def test_strided_views_of_a_fused_projection_and_cross_lengths():
    """q | k as column halves of one [B, N, 2C] tensor (what MultiheadAttention.forward_bf passes); Nq != Nk"""
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(1)
    H, C = 8, 256
    qk = torch.randn(2, 300, 2 * C, device=DEV, generator=g).half()
    v = torch.randn(2, 300, C, device=DEV, generator=g).half()
    q, k = qk[..., :C], qk[..., C:]
    assert _cabi.mha_attention_supported(q, k, v, H)
    out = torch.empty(2, 300, C, dtype=torch.float16, device=DEV)
    _cabi.mha_attention(q, k, v, H, out)
    torch.testing.assert_close(out.float(), _ref(q, k, v, H), rtol=5e-3, atol=3e-3)
    q2 = torch.randn(2, 70, C, device=DEV, generator=g).half()          # fewer queries than keys
    out2 = torch.empty(2, 70, C, dtype=torch.float16, device=DEV)
    _cabi.mha_attention(q2, k, v, H, out2)
    torch.testing.assert_close(out2.float(), _ref(q2, k, v, H), rtol=5e-3, atol=3e-3)


def test_large_logits_stay_finite():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(2)
    q, k, v = (torch.randn(1, 400, 64, device=DEV, generator=g).half() for _ in range(3))
    q = q * 30            # logits of several hundred: the running max must keep exp2 in range
    out = hip_ops.mha_self_attention(q, k, v, 2)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out.float(), _ref(q, k, v, 2), rtol=5e-3, atol=5e-3)


def test_bf16_and_fallback():
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(3)
    q, k, v = (torch.randn(2, 333, 128, device=DEV, generator=g).bfloat16() for _ in range(3))
    out = hip_ops.mha_self_attention(q, k, v, 4)
    torch.testing.assert_close(out.float(), _ref(q, k, v, 4), rtol=2e-2, atol=2e-2)
    before = _cabi.CALLS["mha_attention"]
    q, k, v = (torch.randn(1, 1500, 64, device=DEV, generator=g).half() for _ in range(3))   # > 1024 keys: library path
    out = hip_ops.mha_self_attention(q, k, v, 2)
    assert _cabi.CALLS["mha_attention"] == before
    torch.testing.assert_close(out.float(), _ref(q, k, v, 2), rtol=5e-3, atol=3e-3)
