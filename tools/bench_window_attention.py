"""Micro-benchmark of the fused window-attention kernel at the four Swin-L stages of a 1920x1280 image (GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_linear import timeit  # noqa: E402
from codetr import _cabi, hip_ops  # noqa: E402

if os.environ.get("CODETR_LIB"):   # timing experiments: a diagnostic build of the library (tools/micro/build_variant.sh)
    _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name, (H, W), heads in (("stage0", (320, 480), 6), ("stage1", (160, 240), 12), ("stage2", (80, 120), 24),
                            ("stage3", (40, 60), 48)):
    C = heads * 32
    qkv = torch.randn(B, H * W, 3 * C, device="cuda").half()
    bias = torch.randn(3 * C, device="cuda").half()
    rel = torch.randn(heads, 144, 144, device="cuda").half()
    for shift in (0, 6):
        t = timeit(lambda: hip_ops.swin_window_attention(qkv, bias, rel, (H, W), heads, 12, shift))
        nwin = -(-H // 12) * -(-W // 12)
        waves = B * nwin * heads
        mb = (qkv.numel() + qkv.numel() // 3) * 2 / 1e6
        print(f"{name} B={B} shift={shift}: {t * 1e6:8.1f} us   {waves} problems -> {t * 1e6 / (waves / 2048):6.2f} us per "
              f"problem at 2048 resident; qkv+out {mb:.0f} MB -> {mb / t / 1e6:.2f} TB/s")
