"""Encoder MSDA at BASELINE's pyramid: LDS-staged kernel (codetr_msda_encoder_forward_f16) against the general fused
kernel on the same inputs.  Offsets follow the reference's initialisation (multi_scale_deformable_attention.py:90-115:
head m points along angle 2 pi m / M, point p at distance p + 1 pixels) plus Gaussian noise of `--noise` pixels.
    python tools/bench_msda_encoder.py [--batch 1] [--noise 0.5] [--halo 4]"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--noise", type=float, default=0.5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--windows", type=int, default=1, help="1: windows from the bias grid, 0: symmetric halo")
    ap.add_argument("--passes", type=int, default=3, help="3: three-pass kernel, 1: single pass")
    ap.add_argument("--counts", type=int, default=0, help="1: reference points computed in fp32 from valid counts")
    ap.add_argument("--v4", type=int, default=0, help="1: the round-5 packed kernel (codetr_msda_encoder_forward_packed_f16)")
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--region", default="16x16")
    ap.add_argument("--budget", type=int, default=0, help="LDS bytes per workgroup (default 40 KiB x 256 / threads ... see code)")
    ap.add_argument("--cap", type=float, default=40.0, help="largest window margin in pixels")
    ap.add_argument("--hm", type=int, default=1, help="1: head-major value map [B, M, S, D]")
    a = ap.parse_args()
    from codetr import _cabi, hip_ops

    if os.environ.get("CODETR_LIB"):   # timing experiments: a diagnostic build of the library (e.g. -DMSDA_ENC_ABLATE)
        _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None   # (import codetr loaded the product build)

    dev = "cuda:0"
    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    B, M, D, L, P = a.batch, 8, 32, 5, 4
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(B, S, M, D, device=dev, generator=g).half()
    th = torch.arange(M, device=dev) * (2 * math.pi / M)
    grid = torch.stack((th.cos(), th.sin()), -1)
    grid = grid / grid.abs().max(-1, keepdim=True)[0]
    off = grid[:, None, None, :] * (torch.arange(P, device=dev) + 1)[None, None, :, None]      # [M,1,P,2]
    off = off.expand(M, L, P, 2)[None, None] + a.noise * torch.randn(B, S, M, L, P, 2, device=dev, generator=g)
    logits = torch.randn(B, S, M * L * P, device=dev, generator=g)
    proj = torch.cat((off.reshape(B, S, -1), logits), -1).half().contiguous()
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid(torch.arange(h, device=dev) + 0.5, torch.arange(w, device=dev) + 0.5, indexing="ij")
        refs.append(torch.stack((xs.reshape(-1) / w, ys.reshape(-1) / h), -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(B, S, L, 2).half().contiguous()
    ss = torch.tensor(shapes, dtype=torch.int64, device=dev)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    bias = off_bias = (grid[:, None, None, :] * (torch.arange(P, device=dev) + 1)[None, None, :, None]).expand(M, L, P, 2)
    counts = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev)[None].expand(B, L, 2).contiguous()
    # (the round-3/4 kernels -- --v4 0, --passes, --windows, --counts -- left the library in round 6:
    # tools/micro/experiments/msda_encoder_v3.hip)
    hip_ops.MSDA_V4_THREADS = a.threads
    hip_ops.MSDA_V4_REGION = tuple(int(v) for v in a.region.split("x"))
    hip_ops.MSDA_V4_LDS_BUDGET = a.budget if a.budget else (40 * 1024 if a.threads == 256 else 64 * 1024)
    hip_ops.MSDA_V4_MARGIN_CAP = a.cap
    idx = torch.tensor(_cabi.msda_pack_projection_index(M, L, P), device=dev)
    packed = proj[..., idx.clamp_min(0)].clone()
    packed[..., idx < 0] = 0
    packed = packed.contiguous()
    win = hip_ops.msda_encoder_windows_packed(bias.reshape(-1), shapes, M, L, P)
    vhm = value.permute(0, 2, 1, 3).contiguous() if a.hm else None
    enc = lambda: hip_ops.msda_encoder_packed(vhm if a.hm else value, shapes, packed, P, win, counts, bool(a.hm))  # noqa: E731
    print("v4 windows head 0/1:", win[0], win[1], "lds",
          _cabi.msda_encoder_packed_lds_bytes(shapes, M, P, win, hip_ops.MSDA_V4_REGION, a.threads))
    gen = lambda: hip_ops.msda_fused(value, ss, ls, proj, 0, M * L * P * 2, ref, L, P)  # noqa: E731
    o1, o2 = enc(), gen()
    assert o1 is not None
    same = torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    rel = ((o1.double() - o2.double()).norm() / o2.double().norm()).item()
    t_enc, t_gen = timed(enc), timed(gen)
    alg = 2 * (B * S * M * D + 3 * B * S * M * L * P + B * S * M * D)
    print(f"batch {B} noise {a.noise}: encoder kernel {t_enc:8.1f} us "
          f"({alg / t_enc / 1e6:6.2f} TB/s algorithmic)   general fused {t_gen:8.1f} us   identical: {same} "
          f"rel L2 vs general {rel:.2e}"
          f"  v4 threads {a.threads} region {a.region} cap {a.cap} hm {a.hm} budget {hip_ops.MSDA_V4_LDS_BUDGET}")


if __name__ == "__main__":
    main()
