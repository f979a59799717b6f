"""The X-stationary GEMM (csrc/gemm_f16.hip linear_xs_kernel) at its launch shapes of one 4-image 1920x1280 forward: the
encoder's value projection (head-major output) and (offsets | logits) projection (query + query_pos folded into the operand
load), Swin stage 0's norm1 -> qkv and norm2 -> fc1 + GELU.  us per launch, algorithmic TB/s (X (+X2) read once, Y written
once) and TF/s -- these launches sit between the two roofs.
    python tools/bench_linear_xs.py [--images 4] [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def timeit(fn, iters):
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    from codetr import _cabi, hip_ops

    if os.environ.get("CODETR_LIB"):   # timing experiments: a diagnostic build of the library (-DCODETR_XS_ABL=mask)
        _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None
    S = 204600                                   # tokens of one 1920x1280 image over the five levels
    T0 = 480 * 320                               # Swin stage-0 tokens of one image
    B = a.images
    g = torch.Generator(device="cuda").manual_seed(1)

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, device="cuda", generator=g) * scale).half()

    rows = []
    # encoder value projection: [B, S, 256] -> head-major [B, 8, S, 32]
    x = rnd(B, S, 256)
    w, b = rnd(256, 256, scale=1 / 16), rnd(256)
    rows.append(("enc.value (head-major)", B * S, 256, 256, 1, lambda: hip_ops.linear(x, w, b, head_major=32)))
    rows.append(("enc.value (row-major)", B * S, 256, 256, 1, lambda: hip_ops.linear(x, w, b)))
    # encoder (offsets | logits): (query + pos) @ W^T, N = 512 (lane-major packed)
    p = rnd(B, S, 256)
    w5, b5 = rnd(512, 256, scale=1 / 16), rnd(512)
    rows.append(("enc.packed (x + pos)", B * S, 512, 256, 2, lambda: hip_ops.linear_xadd(x, p, w5, b5)))
    w3, b3 = rnd(384, 256, scale=1 / 16), rnd(384)
    rows.append(("enc.offsets|logits 384", B * S, 384, 256, 2, lambda: hip_ops.linear_xadd(x, p, w3, b3)))
    w7, b7 = rnd(768, 256, scale=1 / 16), rnd(768)
    rows.append(("enc.value+packed traffic", B * S, 768, 256, 2, lambda: hip_ops.linear_xadd(x, p, w7, b7)))
    wc, bc = torch.cat((w, w5), 0).contiguous(), torch.cat((b, b5), 0).contiguous()
    rows.append(("enc.value+packed ONE launch", B * S, 768, 256, 2, lambda: hip_ops.encoder_projections(x, p, wc, bc, None, 256, 32)))
    # Swin stage 0: norm1 -> qkv, norm2 -> fc1 + GELU
    x0 = rnd(B * T0, 192)
    gm, bt = rnd(192), rnd(192)
    wq, bq = rnd(576, 192, scale=1 / 14), rnd(576)
    wf, bf = rnd(768, 192, scale=1 / 14), rnd(768)
    rows.append(("swin0.norm1+qkv", B * T0, 576, 192, 1, lambda: hip_ops.linear_ln(x0, gm, bt, 1e-5, wq, bq)))
    rows.append(("swin0.norm2+fc1+gelu", B * T0, 768, 192, 1, lambda: hip_ops.linear_ln(x0, gm, bt, 1e-5, wf, bf, act="gelu")))
    rows.append(("swin0.qkv (no norm)", B * T0, 576, 192, 1, lambda: hip_ops.linear(x0, wq, bq)))
    rows.append(("swin0.fc1+gelu (no norm)", B * T0, 768, 192, 1, lambda: hip_ops.linear(x0, wf, bf, act="gelu")))
    rows.append(("layer_norm 192", B * T0, 192, 0, 1, lambda: hip_ops.layer_norm(x0, gm, bt, 1e-5)))
    rows.append(("add 256", B * S, 256, 0, 2, lambda: hip_ops.add(x, p) if hasattr(hip_ops, "add") else x + p))
    rows.append(("enc.packed (no add)", B * S, 512, 256, 1, lambda: hip_ops.linear(x, w5, b5)))
    print(f"# tools/bench_linear_xs.py --images {B} {a.tag} on {torch.cuda.get_device_name(0)}")
    for name, M, N, K, nx, fn in rows:
        before = dict(_cabi.CALLS)
        t = timeit(fn, a.iters)
        kern = [k for k in ("encoder_projections", "linear_xs", "linear_xadd", "linear_ln", "linear_sk", "linear_tile256", "linear_tile128")
                if _cabi.CALLS.get(k, 0) > before.get(k, 0)]
        byt = 2.0 * M * (nx * K + N)
        print(f"{name:26s} M={M:7d} N={N:4d} K={K:4d} | {t * 1e6:7.1f} us  {byt / t / 1e12:5.2f} TB/s  "
              f"{2.0 * M * N * K / t / 1e12:5.0f} TF/s  [{','.join(kern)}]")


if __name__ == "__main__":
    with torch.no_grad():
        main()
