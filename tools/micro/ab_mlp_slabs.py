import sys, torch
sys.path.insert(0, "co-detr-tensorrt_amd")
from codetr import hip_ops
def timeit(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, C) in ((614400, 192), (153600, 384)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, C, device="cuda", generator=g).half()
    w1 = (torch.randn(4 * C, C, device="cuda", generator=g) / C ** 0.5).half(); b1 = torch.randn(4 * C, device="cuda", generator=g).half()
    w2 = (torch.randn(C, 4 * C, device="cuda", generator=g) / (4 * C) ** 0.5).half(); b2 = torch.randn(C, device="cuda", generator=g).half()
    big = torch.randn(256 << 20, device="cuda").half()   # 512 MB: flush the infinity cache between measurements
    def whole():
        h = hip_ops.linear(x, w1, b1, act="gelu")
        return hip_ops.linear(h, w2, b2, residual=x)
    def slabs(n):
        def run():
            outs = []
            step = -(-M // n // 256) * 256
            for s in range(0, M, step):
                xs = x[s:s + step]
                h = hip_ops.linear(xs, w1, b1, act="gelu")
                outs.append(hip_ops.linear(h, w2, b2, residual=xs))
            return outs
        return run
    print(f"M {M} C {C}: whole {timeit(whole):.1f} us", end="")
    for n in (2, 4, 8, 16, 32):
        print(f" | {n} slabs {timeit(slabs(n)):.1f}", end="")
    print(flush=True)
