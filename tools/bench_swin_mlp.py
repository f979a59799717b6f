#!/usr/bin/env python3
"""Swin stage-0 / stage-1 MLP (norm2 -> fc1 -> GELU -> fc2 -> + identity, reference codetr/swin.py:331-352) at the bench's
launch shape: the fused kernel (csrc/swin_mlp.hip) against the three launches it replaces.
    python tools/bench_swin_mlp.py [--images 4]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    a = ap.parse_args()
    from codetr import _cabi, hip_ops

    if os.environ.get("CODETR_LIB"):   # timing experiments: a diagnostic build of the library (tools/micro/build_variant.sh)
        _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None

    for C, tokens in ((192, 153600), (384, 38400)):
        M = tokens * a.images
        g = torch.Generator(device="cuda").manual_seed(C)
        r = lambda *s, k=1.0: (torch.randn(*s, device="cuda", generator=g) * k).half()  # noqa: E731
        x, gam, bet = r(M, C), (1 + 0.1 * torch.randn(C, device="cuda", generator=g)).half(), r(C, k=0.1)
        w1, b1, w2, b2 = r(4 * C, C, k=C ** -0.5), r(4 * C, k=0.2), r(C, 4 * C, k=(4 * C) ** -0.5), r(C, k=0.2)

        def separate():
            if hip_ops.linear_ln_supported(x, gam, w1):
                h = hip_ops.linear_ln(x, gam, bet, 1e-5, w1, b1, act="gelu")
            else:
                h = hip_ops.linear(hip_ops.layer_norm(x, gam, bet, 1e-5), w1, b1, act="gelu")
            return hip_ops.linear(h, w2, b2, residual=x)

        fused = lambda: hip_ops.swin_mlp(x, gam, bet, 1e-5, w1, b1, w2, b2)  # noqa: E731
        y1, y2 = separate(), fused()
        rel = float((y1.float() - y2.float()).norm() / y1.float().norm())
        ts, tf = timeit(separate), timeit(fused)
        fl = 2.0 * M * C * 4 * C * 2
        print(f"C {C} M {M}: separate launches {ts:8.1f} us ({fl / ts / 1e6:6.0f} TF/s)   fused {tf:8.1f} us ({fl / tf / 1e6:6.0f} TF/s)   "
              f"rel L2 between them {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
