"""GPU parity of the native positional-encoding kernel against the reference's captured output (golden) and
against the module's own PyTorch formulation in fp32.  Tolerance: fp16 rounding of values in [-1, 1] + level
embedding: 1e-3 abs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_sine_pos_tokens_vs_reference_golden():
    from codetr import hip_ops

    g = np.load(os.path.join(GOLDEN, "model_posenc.npz"))
    mask = torch.as_tensor(g["mask"]).to(DEV)
    B, H, W = mask.shape
    dest = torch.zeros(B, H * W + 9, 256, device=DEV, dtype=torch.float16)
    hip_ops.sine_pos_tokens_into(mask, dest, 4, None, 128, 20, 2 * np.pi, 1e-6, 0.0, True)
    torch.cuda.synchronize()
    ref = torch.as_tensor(g["out"]).permute(0, 2, 3, 1).reshape(B, H * W, 256)  # reference: [B,256,H,W]
    torch.testing.assert_close(dest[:, 4:4 + H * W].float().cpu(), ref, rtol=0, atol=1e-3)
    assert (dest[:, :4] == 0).all() and (dest[:, 4 + H * W:] == 0).all()


@pytest.mark.parametrize("H,W,normalize", [(1, 1, True), (20, 30, True), (33, 17, False), (320, 480, True)])
def test_sine_pos_tokens_vs_module_fp32(H, W, normalize):
    from codetr import hip_ops
    from codetr.positional_encoding import SinePositionalEncoding

    g = torch.Generator(device=DEV).manual_seed(H)
    mask = torch.zeros(2, H, W, dtype=torch.bool, device=DEV)
    mask[1, :, int(W * 0.7):] = True
    mask[1, int(H * 0.8):, :] = True
    lvl = (torch.randn(256, device=DEV, generator=g) * 0.5).half()
    pe = SinePositionalEncoding(num_feats=128, temperature=20, normalize=normalize)
    ref = pe.forward_tokens(mask, dtype=torch.float32) + lvl.float()
    dest = torch.empty(2, H * W, 256, device=DEV, dtype=torch.float16)
    hip_ops.sine_pos_tokens_into(mask, dest, 0, lvl, 128, 20, pe.scale, pe.eps, pe.offset, normalize)
    tol = 2e-3 if normalize else 2e-2  # un-normalised: angles up to H, sin/cos of large arguments in fp32 vs fp32
    torch.testing.assert_close(dest.float(), ref, rtol=0, atol=tol)


def test_sine_pos_tokens_bf16_matches_f16_kernel():
    """bf16 storage instantiation: same fp32 arithmetic as the f16 kernel, rounded to bf16 instead"""
    from codetr import hip_ops

    B, H, W, nf = 2, 9, 13, 128
    mask = torch.zeros(B, H, W, dtype=torch.bool, device=DEV)
    mask[1, 7:, :] = True
    mask[1, :, 10:] = True
    le = torch.randn(2 * nf, device=DEV)
    outs = {}
    for dt in (torch.float16, torch.bfloat16):
        dest = torch.zeros(B, H * W + 5, 2 * nf, dtype=dt, device=DEV)
        hip_ops.sine_pos_tokens_into(mask, dest, 5, le.to(dt), nf, 20, 2 * 3.141592653589793, 1e-6, 0.0, True)
        outs[dt] = dest[:, 5:].float()
    torch.cuda.synchronize()
    torch.testing.assert_close(outs[torch.bfloat16], outs[torch.float16], rtol=1e-2, atol=2e-2)
