import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "co-detr-tensorrt_amd")
import torch, torch.nn.functional as F
from bench_linear import timeit
from codetr import hip_ops
S = [("swin2.qkv", 80640, 2304, 768, None, False), ("swin2.proj", 80640, 768, 768, None, True),
     ("swin2.fc1", 76800, 3072, 768, "gelu", False), ("swin2.fc2", 76800, 768, 3072, None, True),
     ("swin3.qkv", 23040, 4608, 1536, None, False), ("swin3.fc1", 19200, 6144, 1536, "gelu", False),
     ("swin3.fc2", 19200, 1536, 6144, None, True), ("swin1.qkv", 322560, 1152, 384, None, False),
     ("swin1.fc1", 307200, 1536, 384, "gelu", False), ("swin1.fc2", 307200, 384, 1536, None, True),
     ("swin0.fc2", 1228800, 192, 768, None, True)]
for name, M, N, K, act, res in S:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda").half(); r = torch.randn(M, N, device="cuda").half() if res else None
    tn = timeit(lambda: hip_ops.linear(x, w, b, act=act, residual=r))
    tg = timeit(lambda: F.linear(x, w, b))   # GEMM + bias only (no act / residual kernels)
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M:8d} N={N:5d} K={K:5d}  native {tn*1e6:8.1f} us {fl/tn/1e12:7.1f} TF/s | hipBLASLt gemm+bias only {tg*1e6:8.1f} us {fl/tg/1e12:7.1f} TF/s")
if "--fp8" in sys.argv:
    FP8 = torch.float8_e4m3fn
    from codetr import _cabi
    for name, M, N, K, act, res in S:
        if K % 128:
            continue
        x8 = (torch.randn(M, K, device="cuda") * 40).to(FP8); w8 = (torch.randn(N, K, device="cuda") * 60).to(FP8)
        ws = torch.rand(N, device="cuda") * 1e-3; b = torch.randn(N, device="cuda").half()
        r = torch.randn(M, N, device="cuda").half() if res else None
        out8 = act == "gelu"
        out = torch.empty(M, N, dtype=FP8 if out8 else torch.float16, device="cuda")
        tn = timeit(lambda: _cabi.linear_fp8(x8, w8, ws, 0.01, b, r, act, out, 0.05 if out8 else 0.0))
        fl = 2.0 * M * N * K
        # the same GEMM with MX block scales on the activation (and, for fc1, on the e4m3 output)
        sx = torch.full((_cabi.mx_scale_bytes(M, K),), 0x7F, dtype=torch.uint8, device="cuda")
        sy = torch.empty(_cabi.mx_scale_bytes(M, N), dtype=torch.uint8, device="cuda") if out8 else None
        tm = timeit(lambda: _cabi.linear_fp8mx(x8, sx, w8, ws, b, r, act, out, sy))
        print(f"fp8 {name:12s} M={M:8d} N={N:5d} K={K:5d}  static {tn*1e6:8.1f} us {fl/tn/1e12:7.1f} TF/s | MX {tm*1e6:8.1f} us {fl/tm/1e12:7.1f} TF/s  (out {'e4m3' if out8 else 'f16'})")
