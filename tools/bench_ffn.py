"""Micro-benchmark: fused FFN kernel vs two native GEMMs at the encoder shape (GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_linear import timeit  # noqa: E402
from codetr import _cabi, hip_ops  # noqa: E402

if os.environ.get("CODETR_LIB"):   # timing experiments: a diagnostic build of the library (e.g. -DCODETR_FFN_NO8)
    _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None   # (import codetr loaded the product build)

for M in (818400, 204600, 73656, 30785):   # 4 images, 1 image, one 1152x768 image, one 608x608 image
    x = torch.randn(M, 256, device="cuda").half()
    w1 = (torch.randn(2048, 256, device="cuda") / 16).half()
    b1 = torch.randn(2048, device="cuda").half()
    w2 = (torch.randn(256, 2048, device="cuda") / 45).half()
    b2 = torch.randn(256, device="cuda").half()
    tf = timeit(lambda: hip_ops.ffn_fused(x, w1, b1, w2, b2))
    gam, bet, pos = torch.ones(256, device="cuda").half(), torch.zeros(256, device="cuda").half(), torch.randn_like(x)
    tl = timeit(lambda: hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(gam, bet, 1e-5), pos=pos))

    tli = timeit(lambda: hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(gam, bet, 1e-5), ln_in=(gam, bet, 1e-5)))
    print(f"M={M}: LN -> FFN -> LN one kernel (the encoder layer's form) {tli * 1e6:.1f} us")

    def three():
        y = hip_ops.layer_norm(hip_ops.ffn_fused(x, w1, b1, w2, b2), gam, bet, 1e-5)
        return y, y + pos

    t3 = timeit(three)

    def two():
        h = hip_ops.linear(x, w1, b1, act="relu")
        return hip_ops.linear(h, w2, b2, residual=x)

    t2 = timeit(two)
    fl = 2 * 2.0 * M * 256 * 2048
    print(f"M={M}: FFN+LN+pos one kernel {tl * 1e6:.1f} us, as three kernels {t3 * 1e6:.1f} us")
    print(f"M={M}: fused FFN {tf * 1e6:.1f} us ({fl / tf / 1e12:.0f} TF/s)   two native GEMMs {t2 * 1e6:.1f} us ({fl / t2 / 1e12:.0f} TF/s)")
    if "--fp8" in sys.argv:
        lnp = (gam, bet, 1e-5)
        t8 = timeit(lambda: hip_ops.ffn_fp8(x, w1, b1, w2, b2, 0.012, 0.02, ln=lnp, pos=pos, ln_in=lnp))
        t16 = timeit(lambda: hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=lnp, pos=pos, ln_in=lnp))
        print(f"M={M}: LN+FFN+LN+pos  fp8 {t8 * 1e6:.1f} us ({fl / t8 / 1e12:.0f} TF/s)   fp16 {t16 * 1e6:.1f} us ({fl / t16 / 1e12:.0f} TF/s)")
