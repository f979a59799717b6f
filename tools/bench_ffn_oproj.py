"""Fused FFN with the attention output projection folded in (codetr_ffn_oproj_relu_ln2_*) against the two launches it
replaces (codetr_linear_* with the residual epilogue, then the fused FFN with ln_in), encoder shape.  GPU box only.
    python tools/bench_ffn_oproj.py [M ...]      CODETR_LIB=<diagnostic build> for the -DCODETR_FFN_ABL experiments"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_linear import timeit  # noqa: E402
from codetr import _cabi, hip_ops  # noqa: E402

if os.environ.get("CODETR_LIB"):
    _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None

with torch.no_grad():
    for M in [int(a) for a in sys.argv[1:]] or [818400, 204600]:
        r = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).half()   # noqa: E731
        attn, ident, pos = r(M, 256), r(M, 256), r(M, 256)
        wo, bo, w1, b1, w2, b2 = r(256, 256, k=1 / 16), r(256), r(2048, 256, k=1 / 16), r(2048), r(256, 2048, k=1 / 45), r(256)
        gam, bet = torch.ones(256, device="cuda").half(), torch.zeros(256, device="cuda").half()
        ln = (gam, bet, 1e-5)
        t_proj = timeit(lambda: hip_ops.linear(attn, wo, bo, residual=ident))
        x0 = hip_ops.linear(attn, wo, bo, residual=ident)
        t_ffn = timeit(lambda: hip_ops.ffn_fused(x0, w1, b1, w2, b2, ln=ln, pos=pos, ln_in=ln))
        t_one = timeit(lambda: hip_ops.ffn_oproj_fused(attn, wo, bo, ident, w1, b1, w2, b2, ln, pos=pos, ln_in=ln))
        print(f"M={M}: output_proj + identity {t_proj * 1e6:.1f} us, LN-FFN-LN+pos {t_ffn * 1e6:.1f} us, sum {(t_proj + t_ffn) * 1e6:.1f} us"
              f" | one launch {t_one * 1e6:.1f} us  {os.environ.get('CODETR_LIB', '')}")
