"""Micro-benchmark: native fused linear (codetr_linear_f16) vs ATen/hipBLASLt F.linear (+ the separate
epilogue kernels) on the Linear shapes of Co-DINO Swin-L at 1920x1280 (GPU box only)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))

SHAPES = [  # (name, M, N, K, act, residual)
    ("swin0.qkv", 155520, 576, 192, None, False), ("swin0.proj", 155520, 192, 192, None, False),
    ("swin0.fc1", 153600, 768, 192, "gelu", False), ("swin0.fc2", 153600, 192, 768, None, True),
    ("swin1.qkv", 40320, 1152, 384, None, False), ("swin1.fc1", 38400, 1536, 384, "gelu", False),
    ("swin1.fc2", 38400, 384, 1536, None, True),
    ("swin2.qkv", 10080, 2304, 768, None, False), ("swin2.proj", 10080, 768, 768, None, False),
    ("swin2.fc1", 9600, 3072, 768, "gelu", False), ("swin2.fc2", 9600, 768, 3072, None, True),
    ("swin3.qkv", 2880, 4608, 1536, None, False), ("swin3.fc1", 2400, 6144, 1536, "gelu", False),
    ("swin3.fc2", 2400, 1536, 6144, None, True),
    ("enc.value_proj", 204600, 256, 256, None, False), ("enc.offsets", 204600, 320, 256, None, False),
    ("enc.attw", 204600, 160, 256, None, False), ("enc.off|attw", 204600, 480, 256, None, False), ("enc.out_proj", 204600, 256, 256, None, True),
    ("enc.ffn1", 204600, 2048, 256, "relu", False), ("enc.ffn2", 204600, 256, 2048, None, True),
    ("head.cls", 204600, 80, 256, None, False), ("dec.q", 900, 256, 256, None, False),
]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


def main():
    from codetr import hip_ops

    dev = "cuda:0"
    tot_n = tot_t = 0.0
    print(f"{'layer':16s} {'M':>7s} {'N':>5s} {'K':>5s}   native us   TF/s   GB/s |   aten us   TF/s | speedup")
    for name, M, N, K, act, res in SHAPES:
        x = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        b = torch.randn(N, device=dev).half()
        r = torch.randn(M, N, device=dev).half() if res else None

        def aten():
            y = F.linear(x, w, b)
            if act == "relu":
                y = F.relu(y, inplace=True)
            elif act == "gelu":
                y = F.gelu(y)
            if r is not None:
                y = y + r
            return y

        tn = timeit(lambda: hip_ops.linear(x, w, b, act=act, residual=r))
        ta = timeit(aten)
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K + M * N * (2 if res else 1))
        tot_n += tn
        tot_t += ta
        print(f"{name:16s} {M:7d} {N:5d} {K:5d}  {tn * 1e6:9.1f} {fl / tn / 1e12:6.1f} {by / tn / 1e9:6.0f} | {ta * 1e6:9.1f} {fl / ta / 1e12:6.1f} | {ta / tn:5.2f}x")
    print(f"sum native {tot_n * 1e3:.2f} ms, aten {tot_t * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
