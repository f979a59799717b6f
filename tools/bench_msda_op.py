#!/usr/bin/env python3
"""The PUBLIC op torch.ops.codetr.multi_scale_deformable_attention alone (reference codetr/csrc/ms_deform_attn.cu:211-261,
762-779), timed with HIP events on its launch stream:
  encoder shape   Nq = S (1920x1280: 204 600 queries), fp16, batch 1: the windowed kernel (csrc/msda_op4.hip) + the general
                  kernel's skipped launch behind it; query i samples around pixel i with --spread pixels of normal spread
  decoder shape   Nq = 900, the general kernel (106.1 MB per image)
Algorithmic bytes: value + locations + weights + output, each touched once.
    python tools/bench_msda_op.py [--spread 0 1 2 3 4 8] [--pmc]      (--pmc: three launches of each shape, for rocprofv3)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
HBM = 8000.0
M, D, L, P = 8, 32, 5, 4


def pyramid(H, W):
    out, h, w = [], -(-H // 4), -(-W // 4)      # strides 4 .. 64 (Co-DINO Swin-L 5-scale: 320 x 480 ... 20 x 30 at 1920x1280)
    for _ in range(L):
        out.append((h, w))
        h, w = -(-h // 2), -(-w // 2)
    return out


def inputs(B, shapes, Nq, spread, dev, seed=0):
    ss = torch.tensor(shapes, dtype=torch.int64, device=dev)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    S = int(ss.prod(1).sum())
    g = torch.Generator(device=dev).manual_seed(seed)
    value = torch.randn(B, S, M, D, device=dev, generator=g).half()
    norm = torch.stack((ss[:, 1], ss[:, 0]), -1).float()[None, None, None, :, None, :]
    if Nq == S:
        refs = []
        for (h, w) in shapes:
            ys, xs = torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w, indexing="ij")
            refs.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
        ref = torch.cat(refs)[None, :, None, None, None, :]
    else:
        ref = torch.rand(B, Nq, 1, 1, 1, 2, device=dev, generator=g) * 0.8 + 0.1
    loc = (ref + torch.randn(B, Nq, M, L, P, 2, device=dev, generator=g) * spread / norm).half().contiguous()
    w = torch.softmax(torch.randn(B, Nq, M, L * P, device=dev, generator=g), -1).view(B, Nq, M, L, P).half().contiguous()
    return value, ss, ls, loc, w, S


def alg_bytes(B, S, Nq):
    return 2 * B * (S * M * D + 3 * Nq * M * L * P + Nq * M * D)


def time_op(args, iters):
    op = torch.ops.codetr.multi_scale_deformable_attention
    import time
    t0 = time.time()
    while iters > 3 and time.time() - t0 < 0.25:   # warm-up by time (clock state: tools/micro/op_after_burn.py); not under --pmc
        for _ in range(50):
            op(*args, 64)
        torch.cuda.synchronize()
    for _ in range(3):
        op(*args, 64)
    st = torch.cuda.current_stream()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(st)
        op(*args, 64)
        b.record(st)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    ts = ts[len(ts) // 10: len(ts) - len(ts) // 10]   # (a host-side hiccup between two launches is not the kernel)
    return sum(ts) / len(ts) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="1920x1280")
    ap.add_argument("--spread", type=float, nargs="*", default=[0.0, 1.0, 2.0, 3.0, 4.0, 8.0])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--pmc", action="store_true")
    a = ap.parse_args()
    import codetr  # noqa: F401
    from codetr import _cabi

    W_, H_ = (int(v) for v in a.res.split("x"))
    shapes = pyramid(H_, W_)
    dev = "cuda:0"
    rows = []
    for sp in (a.spread if not a.pmc else a.spread[:1]):
        value, ss, ls, loc, w, S = inputs(a.batch, shapes, sum(h * w_ for h, w_ in shapes), sp, dev)
        served = bool(_cabi.load().codetr_msda_op4_supported(2, a.batch, S, M, D, L, S, P))
        t = time_op((value, ss, ls, loc, w), 3 if a.pmc else a.iters)
        nb = alg_bytes(a.batch, S, S)
        rows.append({"shape": "encoder", "spread_px": sp, "us": round(t * 1e6, 1), "GB/s": round(nb / t / 1e9, 1),
                     "frac": round(nb / t / 1e9 / HBM, 4), "bytes": nb, "windowed_kernel": served})
        print(json.dumps(rows[-1]), flush=True)
        del value, loc, w
    for B in ((1,) if a.pmc else (1, 4)):
        value, ss, ls, loc, w, S = inputs(B, shapes, 900, 0.05 * 200, dev, seed=1)
        t = time_op((value, ss, ls, loc, w), 3 if a.pmc else a.iters)
        nb = alg_bytes(B, S, 900)
        rows.append({"shape": "decoder", "batch": B, "us": round(t * 1e6, 1), "GB/s": round(nb / t / 1e9, 1),
                     "frac": round(nb / t / 1e9 / HBM, 4), "bytes": nb})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
