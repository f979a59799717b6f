import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "co-detr-tensorrt_amd")
import torch
from bench_linear import timeit
from codetr import hip_ops
# the X-stationary kernel's shapes at 8 images (M = 8 x 204 600 transformer rows, 8 x 153 600 Swin stage-0 rows)
S = [("enc.offs|logits", 1636800, 480, 256, None), ("dec.vproj x6", 1636800, 1536, 256, None), ("swin0.qkv", 1228800, 576, 192, None),
     ("swin0.fc1", 1228800, 768, 192, "gelu"), ("enc.offs|logits b4", 818400, 480, 256, None)]
for name, M, N, K, act in S:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda").half()
    t = timeit(lambda: hip_ops.linear(x, w, b, act=act))
    by = 2.0 * (M * K + M * N)
    print(f"{name:20s} M={M:8d} N={N:5d} K={K:4d}  {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s  {by/t/1e12:5.2f} TB/s (X in + Y out)")
