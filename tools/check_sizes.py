"""Full model (Co-DINO Swin-L, fp16, random-init) at image sizes whose pyramids the 16x8 regions / 4x4 patches do not
divide: the LDS-staged encoder MSDA kernel must leave the detections bit-identical to the general fused kernel, and the
stem / neck gathers must agree with the ATen formulations to fp16 noise.   python tools/check_sizes.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
import bench  # noqa: E402
from codetr import _cabi, hip_ops  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, torch.float16)
ok = True
for (H, W) in [(608, 608), (800, 1333), (750, 1000), (1280, 1920)]:
    g = torch.Generator(device=dev).manual_seed(H + W)
    img = torch.randn(2, 3, H, W, device=dev, generator=g).half()
    mask = torch.zeros(2, H, W, device=dev, dtype=torch.float16)
    mask[1, int(H * 0.9):, :] = 1
    mask[1, :, int(W * 0.85):] = 1
    outs = {}
    for enc in (True, False):
        hip_ops.MSDA_ENCODER = enc
        before = _cabi.CALLS["msda_encoder"]
        with torch.no_grad():
            outs[enc] = model(img, mask)
        used = _cabi.CALLS["msda_encoder"] - before
        assert (used > 0) == enc, (enc, used)
    hip_ops.MSDA_ENCODER = True
    same = all(torch.equal(torch.nan_to_num(a.float(), 0), torch.nan_to_num(b.float(), 0)) for a, b in zip(outs[True], outs[False]))
    saved = hip_ops.patch_embed_supported
    hip_ops.patch_embed_supported = lambda *a, **k: False    # the stem through the library convolution
    try:
        with torch.no_grad():
            ref = model(img, mask)
    finally:
        hip_ops.patch_embed_supported = saved
    s_new, s_ref = torch.nan_to_num(outs[True][1].float(), 0), torch.nan_to_num(ref[1].float(), 0)
    prof = (s_new[:, :100] - s_ref[:, :100]).abs().max().item() / max(s_ref.abs().max().item(), 1e-6)
    print(f"{W}x{H}: encoder kernel == general kernel: {same};  gather+GEMM stem / neck vs ATen: top-100 score profile "
          f"differs by {prof:.3%} of the top score")
    # (random-init weights: near-tied scores, fp16 noise in the stem moves a few of the 900 proposals in and out of the
    # two-stage top-k -- the same 5-10 % profile wobble as between batch sizes, tests/test_full_size_gpu.py)
    ok &= same and prof < 0.2
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
