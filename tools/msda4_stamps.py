"""In-kernel timeline of the encoder MSDA kernel (csrc/msda_encoder4.hip built with -DMSDA4_STAMPS; diagnostic, never shipped):
    tools/micro/build_variant.sh co-detr-tensorrt_amd/csrc/msda_encoder4.hip stamps -DMSDA4_STAMPS
    CODETR_LIB=tools/micro/_bin/libcodetr_stamps.so python tools/msda4_stamps.py [--batch 4] [--noise 2.0]
Wave 0 of every workgroup stamps, per pass: the barrier in front, the DMA issue, the wait for the staged data, the barrier behind
it, the gather (+ fix-up); plus the prologue and the workgroup's life.  Printed: means over the workgroups, in cycles and as a
share of the workgroup's life.  --threads 1024 reads the stamps of the persistent experiment instead
(tools/micro/experiments/msda_encoder4_v5_persistent.hip copied over csrc/msda_encoder4.hip for the variant build)."""
import argparse
import ctypes
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--noise", type=float, default=2.0)
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--region", default="16x16")
    ap.add_argument("--budget", type=int, default=64 * 1024)
    a = ap.parse_args()
    from codetr import _cabi, hip_ops

    _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None
    dev = "cuda:0"
    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    B, M, D, L, P = a.batch, 8, 32, 5, 4
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(B, M, S, D, device=dev, generator=g).half()
    th = torch.arange(M, device=dev) * (2 * math.pi / M)
    grid = torch.stack((th.cos(), th.sin()), -1)
    grid = grid / grid.abs().max(-1, keepdim=True)[0]
    bias = (grid[:, None, None, :] * (torch.arange(P, device=dev) + 1)[None, None, :, None]).expand(M, L, P, 2)
    off = bias[None, None] + a.noise * torch.randn(B, S, M, L, P, 2, device=dev, generator=g)
    logits = torch.randn(B, S, M * L * P, device=dev, generator=g)
    proj = torch.cat((off.reshape(B, S, -1), logits), -1).half().contiguous()
    del off, logits
    counts = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev)[None].expand(B, L, 2).contiguous()
    idx = torch.tensor(_cabi.msda_pack_projection_index(M, L, P), device=dev)
    packed = proj[..., idx.clamp_min(0)].clone()
    packed[..., idx < 0] = 0
    packed = packed.contiguous()
    del proj
    hip_ops.MSDA_V4_THREADS = a.threads
    hip_ops.MSDA_V4_REGION = tuple(int(v) for v in a.region.split("x"))
    hip_ops.MSDA_V4_LDS_BUDGET = a.budget
    win = hip_ops.msda_encoder_windows_packed(bias.reshape(-1), shapes, M, L, P)
    enc = lambda: hip_ops.msda_encoder_packed(value, shapes, packed, P, win, counts, True)  # noqa: E731
    for _ in range(3):
        enc()
    torch.cuda.synchronize()
    rw, rh = hip_ops.MSDA_V4_REGION
    tiles = B * M * (-(-shapes[0][1] // rw)) * (-(-shapes[0][0] // rh))
    st = torch.zeros(tiles * 20, dtype=torch.int64, device=dev)
    lib = ctypes.CDLL(os.environ["CODETR_LIB"])
    assert lib.codetr_msda4_set_stamps(ctypes.c_void_p(st.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    enc()
    e1.record()
    torch.cuda.synchronize()
    s = st.view(tiles, 20).double().cpu()
    if a.threads == 1024:   # the persistent form: sums over a workgroup's tiles
        s = s[s[:, 12] > 0]
        life, nt = s[:, 12].mean().item(), s[:, 11].mean().item()
        names5 = ["geometry + softmax", "landed 0", "landed 1", "landed 2", "issue pass 1", "issue pass 2", "next tile (geometry, rows, pass-0 issue)",
                  "gather 0", "gather 1", "gather 2", "output"]
        print(f"batch {B} noise {a.noise} px: {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), {s.shape[0]} workgroups x {nt:.1f} tiles, "
              f"life {life:.0f} cycles = {life / nt:.0f} per tile")
        for i, n in enumerate(names5):
            v = s[:, i].mean().item()
            print(f"  {n:42s} {v / nt:8.0f} per tile  ({v / life * 100:4.1f} %)")
        return
    s = s[s[:, 16] > 0]
    life = s[:, 16].mean().item()
    print(f"batch {B} noise {a.noise} px: {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), {s.shape[0]} workgroups, "
          f"mean life {life:.0f} cycles")
    names = ["barrier in front", "DMA issue", "wait for data", "barrier", "gather + fix-up"]
    print(f"  prologue {s[:, 15].mean().item():8.0f}  ({s[:, 15].mean().item() / life * 100:4.1f} %)")
    tot = [0.0] * 5
    for p in range(3):
        row = []
        for k in range(5):
            v = s[:, p * 5 + k].mean().item()
            tot[k] += v
            row.append(f"{names[k]} {v:7.0f}")
        print(f"  pass {p}: " + "  ".join(row))
    print("  sums  : " + "  ".join(f"{names[k]} {tot[k]:7.0f} ({tot[k] / life * 100:4.1f} %)" for k in range(5)))
    print(f"  p10 / p50 / p90 of 'wait for data' per pass: " + "  ".join(
        "/".join(f"{torch.quantile(s[:, p * 5 + 2], q).item():.0f}" for q in (0.1, 0.5, 0.9)) for p in range(3)))


if __name__ == "__main__":
    main()
