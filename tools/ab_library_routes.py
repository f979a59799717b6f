"""A/B timing of the three ops that have a PyTorch-ROCm library route next to their native kernel -- decoder
self-attention (SDPA), the Swin stem (MIOpen convolution), row-wise top-k (torch.topk / rocPRIM).  The product always
takes the native kernel where it applies (rounds 1-2 measured: level or faster); this tool re-measures by patching
codetr.hip_ops / codetr._cabi from the OUTSIDE -- the shipped host has no switch for it.
    python tools/ab_library_routes.py [--batch 8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from codetr import _cabi, hip_ops

    dev, B = "cuda:0", a.batch
    g = torch.Generator(device=dev).manual_seed(0)
    # decoder self-attention: 900 queries, 8 heads x 32
    q, k, v = (torch.randn(B, 900, 256, device=dev, generator=g).half() for _ in range(3))
    t_nat = timed(lambda: hip_ops.mha_self_attention(q, k, v, 8))
    saved = _cabi.mha_attention_supported
    _cabi.mha_attention_supported = lambda *x: False
    t_lib = timed(lambda: hip_ops.mha_self_attention(q, k, v, 8))
    _cabi.mha_attention_supported = saved
    print(f"decoder self-attention  native {t_nat:8.1f} us   SDPA {t_lib:8.1f} us")
    # top-k of the two-stage scores
    x = torch.randn(B, 204600, device=dev, generator=g).half()
    with torch.no_grad():
        t_nat = timed(lambda: hip_ops.topk(x, 900, want_values=False))
    t_lib = timed(lambda: torch.topk(x, 900, dim=-1))
    print(f"top-900 of 204 600      native {t_nat:8.1f} us   torch.topk {t_lib:8.1f} us")
    # Swin stem
    img = torch.randn(B, 3, 1280, 1920, device=dev, generator=g).half()
    w = torch.randn(192, 3, 4, 4, device=dev, generator=g).half()
    b = torch.randn(192, device=dev, generator=g).half()
    with torch.no_grad():
        t_nat = timed(lambda: hip_ops.patch_embed(img, w, b))
        t_lib = timed(lambda: torch.nn.functional.conv2d(img, w, b, stride=4).flatten(2).transpose(1, 2).contiguous())
    print(f"Swin stem (4x4 / 4)     native {t_nat:8.1f} us   conv2d + transpose {t_lib:8.1f} us")


if __name__ == "__main__":
    main()
