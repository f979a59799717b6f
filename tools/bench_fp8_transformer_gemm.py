"""e4m3 (MX block scales) against fp16 on the encoder's K = 256 GEMMs (value_proj / output_proj shape) -- BASELINE config 5
names the transformer GEMMs; this is the measurement behind leaving them in fp16."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_linear import timeit
from codetr import hip_ops
hip_ops.FP8_MIN_TILES = 0
for B in (4,):
    M = 204600 * B
    x = torch.randn(M, 256, device="cuda").half()
    w = (torch.randn(256, 256, device="cuda") / 16).half()
    b = torch.randn(256, device="cuda").half()
    r = torch.randn(M, 256, device="cuda").half()
    t16 = timeit(lambda: hip_ops.linear(x, w, b))
    t16r = timeit(lambda: hip_ops.linear(x, w, b, residual=r))
    x8, sx = hip_ops.cast_fp8mx(x)
    tc = timeit(lambda: hip_ops.cast_fp8mx(x))
    t8 = timeit(lambda: hip_ops.linear_fp8mx(x8, sx, w, b))
    t8r = timeit(lambda: hip_ops.linear_fp8mx(x8, sx, w, b, residual=r))
    y16 = hip_ops.linear(x, w, b).float(); y8 = hip_ops.linear_fp8mx(x8, sx, w, b).float()
    rel = float((y8 - y16).norm() / y16.norm())
    print(f"M={M} N=256 K=256: fp16 {t16*1e6:.1f} us (+residual {t16r*1e6:.1f}); e4m3 GEMM {t8*1e6:.1f} us (+residual {t8r*1e6:.1f}), "
          f"cast of the fp16 activation {tc*1e6:.1f} us; rel. L2 of the e4m3 result vs fp16 {rel:.2e}")
