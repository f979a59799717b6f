"""Micro-benchmark of the MSDA HIP kernel at the model shapes (GPU box only).

    python tools/bench_msda.py [--res 1920x1280] [--dtype f16] [--batch 1] [--iters 50]

Times the C-ABI call with HIP events on the launch stream and reports algorithmic GB/s
(BASELINE.md section 3 formula) for the encoder- and decoder-shaped calls.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def pyramid(h, w):
    c = lambda a, b: -(-a // b)  # noqa: E731
    lv = [(c(h, 4), c(w, 4)), (c(h, 8), c(w, 8)), (c(h, 16), c(w, 16)), (c(h, 32), c(w, 32))]
    lv.append((c(lv[-1][0], 2), c(lv[-1][1], 2)))
    return lv


def algorithmic_bytes(B, S, Nq, M=8, D=32, L=5, P=4, e=2):
    return e * (B * S * M * D + 3 * B * Nq * M * L * P + B * Nq * M * D) + 24 * L


def make_inputs(B, shapes, Nq, dtype, dev, seed=0, realistic=True):
    M, D, L, P = 8, 32, len(shapes), 4
    ss = torch.tensor(shapes, dtype=torch.int64, device=dev)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    S = int(ss.prod(1).sum())
    g = torch.Generator(device=dev).manual_seed(seed)
    value = torch.randn(B, S, M, D, device=dev, generator=g).to(dtype)
    if realistic and Nq == S:
        # encoder-like: reference point = own pixel centre, offsets of a few pixels (what the
        # trained model produces: init bias grid is +-1..4 px, ms_deform_attn init_weights)
        refs = []
        for (h, w) in shapes:
            ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, device=dev) / h,
                                    torch.linspace(0.5, w - 0.5, w, device=dev) / w, indexing="ij")
            refs.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
        ref = torch.cat(refs)[None, :, None, None, None, :]  # [1,S,1,1,1,2]
        off_px = torch.randn(B, Nq, M, L, P, 2, device=dev, generator=g) * 3.0
        norm = torch.stack((ss[:, 1], ss[:, 0]), -1).float()[None, None, None, :, None, :]
        loc = (ref + off_px / norm).to(dtype)
    else:
        loc = torch.rand(B, Nq, M, L, P, 2, device=dev, generator=g).to(dtype)
    w = torch.rand(B, Nq, M, L, P, device=dev, generator=g)
    w = (w / w.sum((-1, -2), keepdim=True)).to(dtype)
    return value, ss, ls, loc.contiguous(), w.contiguous(), S


def time_op(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    st = torch.cuda.current_stream()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(st)
        fn()
        b.record(st)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="1920x1280")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--uniform", action="store_true", help="uniform random locations instead of encoder-like")
    ap.add_argument("--fused", action="store_true", help="also time the fused-prologue variant on equivalent inputs")
    a = ap.parse_args()
    import codetr  # noqa: F401

    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
    e = 4 if a.dtype == "f32" else 2
    W, H = (int(x) for x in a.res.split("x"))
    shapes = pyramid(H, W)
    dev = "cuda:0"
    op = torch.ops.codetr.multi_scale_deformable_attention
    for tag, nq in (("enc", None), ("dec", 900)):
        value, ss, ls, loc, w, S = make_inputs(a.batch, shapes, nq or sum(h * w_ for h, w_ in shapes), dt, dev,
                                               realistic=not a.uniform)
        Nq = loc.shape[1]
        med, best = time_op(lambda: op(value, ss, ls, loc, w, 64), a.iters)
        nbytes = algorithmic_bytes(a.batch, S, Nq, e=e)
        if a.fused and a.dtype != "f32":
            from codetr import hip_ops
            M, L, P = 8, len(shapes), 4
            # equivalent fused inputs: reference = the sampling location of point 0, offsets relative to it, logits = log w
            ref = loc[:, :, 0, :, 0, :].contiguous()  # [B,Nq,L,2]
            norm = torch.stack((ss[:, 1], ss[:, 0]), -1).to(dt)[None, None, None, :, None, :]
            off = ((loc - ref[:, :, None, :, None, :]) * norm).reshape(a.batch, Nq, -1)
            logits = torch.log(w.float().clamp_min(1e-6)).to(dt).reshape(a.batch, Nq, -1)
            proj = torch.cat((off, logits), -1).contiguous()
            fm, fb = time_op(lambda: hip_ops.msda_fused(value, ss, ls, proj, 0, M * L * P * 2, ref, L, P), a.iters)
            print(f"{a.res} {a.dtype} B={a.batch} {tag}: FUSED median {fm * 1e6:9.1f} us  best {fb * 1e6:9.1f} us")
            vhm = value.permute(0, 2, 1, 3).contiguous()
            hm, hb = time_op(lambda: hip_ops.msda_fused(vhm, ss, ls, proj, 0, M * L * P * 2, ref, L, P, head_major=True), a.iters)
            print(f"{a.res} {a.dtype} B={a.batch} {tag}: FUSED head-major median {hm * 1e6:9.1f} us  best {hb * 1e6:9.1f} us")
        print(f"{a.res} {a.dtype} B={a.batch} {tag}: S={S} Nq={Nq} median {med * 1e6:9.1f} us  best {best * 1e6:9.1f} us  "
              f"algorithmic {nbytes / 1e6:8.1f} MB -> {nbytes / med / 1e9:8.1f} GB/s  "
              f"(gather volume {a.batch * Nq * 8 * 20 * 4 * 32 * e / 1e9:.2f} GB -> {a.batch * Nq * 8 * 20 * 4 * 32 * e / med / 1e12:.2f} TB/s)")


if __name__ == "__main__":
    main()
