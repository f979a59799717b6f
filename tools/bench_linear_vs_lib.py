"""Round-5 yardstick: the 16 Swin-L linears of one forward (4 stages x qkv / proj /
fc1 + GELU / fc2 + residual) at the bench's launch shape (--images 4) on ONE box in ONE run:
  native   hip_ops.linear -- the hand-written MFMA kernels with bias / GELU / residual fused in the epilogue
  lib      torch.nn.functional.linear (hipBLASLt: GEMM + bias only -- NOT the same op for fc1 / proj / fc2)
  lib+epi  the same op through the library: F.linear, then F.gelu or the residual add as separate ATen kernels
    python tools/bench_linear_vs_lib.py [--images 4] > profiles/r05_linear_vs_hipblaslt.txt"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

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
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--res", default="1920x1280")
    a = ap.parse_args()
    from codetr import _cabi, hip_ops

    W_, H_ = (int(v) for v in a.res.split("x"))
    shapes = []
    C, h, w = 192, H_ // 4, W_ // 4
    for s in range(4):
        hp, wp = -(-h // 12) * 12, -(-w // 12) * 12        # window-padded token count feeds qkv / proj
        tp, t = a.images * hp * wp, a.images * h * w
        shapes += [(f"swin{s}.qkv", tp, 3 * C, C, None, False), (f"swin{s}.proj", tp, C, C, None, True),
                   (f"swin{s}.fc1", t, 4 * C, C, "gelu", False), (f"swin{s}.fc2", t, C, 4 * C, None, True)]
        C, h, w = 2 * C, -(-h // 2), -(-w // 2)
    print(f"# tools/bench_linear_vs_lib.py --images {a.images} --res {a.res} on {torch.cuda.get_device_name(0)}: us per launch "
          f"(TF/s); ratio = native / lib+epi (< 1: the native kernel is faster than the library route for the SAME op)")
    wins = 0
    for name, M, N, K, act, res in shapes:
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(M, K, device="cuda", generator=g).half()
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
        b = torch.randn(N, device="cuda", generator=g).half()
        r = torch.randn(M, N, device="cuda", generator=g).half() if res else None
        before = dict(_cabi.CALLS)
        tn = timeit(lambda: hip_ops.linear(x, w, b, act=act, residual=r))
        kern = [k for k in ("linear_pp", "linear_sk", "linear_tile256", "linear_tile128", "linear_xs") if _cabi.CALLS[k] > before[k]]
        tl = timeit(lambda: F.linear(x, w, b))

        def lib_epi():
            y = F.linear(x, w, b)
            if act == "gelu":
                y = F.gelu(y)
            if r is not None:
                y = y + r
            return y

        te = timeit(lib_epi) if (act or res) else tl
        fl = 2.0 * M * N * K
        wins += tn <= te
        print(f"{name:11s} M={M:7d} N={N:5d} K={K:5d} {'gelu' if act else '    '} {'+res' if res else '    '} | native {tn * 1e6:7.1f} "
              f"({fl / tn / 1e12:6.0f}) [{','.join(kern)}] | lib gemm+bias {tl * 1e6:7.1f} ({fl / tl / 1e12:6.0f}) | lib+epi "
              f"{te * 1e6:7.1f} ({fl / te / 1e12:6.0f}) | ratio {tn / te:.2f}  vs gemm+bias only {tn / tl:.2f}")
    print(f"# native <= library route (same op) on {wins} of {len(shapes)} shapes")


if __name__ == "__main__":
    main()
