import sys, torch
sys.path.insert(0, "co-detr-tensorrt_amd")
from codetr import hip_ops
for n in (30785, 73656, 204600):
    x = torch.randn(1, n, device="cuda").half()
    for _ in range(3): hip_ops.topk(x, 900)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): hip_ops.topk(x, 900)
    e1.record(); torch.cuda.synchronize()
    print(n, round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us")
