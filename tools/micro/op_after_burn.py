"""Does the public op's launch time depend on what the GPU did just before?  The bench line times it behind ~60 s of timed
forwards and reads 351-355 us where tools/bench_msda_op.py alone reads 326 (3 px): time it cold, after a GEMM burn, after the
burn plus a pause, and with its tensors allocated behind a fragmented pool."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_msda_op as t  # noqa: E402

import codetr  # noqa: E402,F401


def burn(seconds):
    a = torch.randn(8192, 8192, device="cuda").half()
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(50):
            a @ a
        torch.cuda.synchronize()


shapes = t.pyramid(1280, 1920)
args = t.inputs(1, shapes, sum(h * w for h, w in shapes), 3.0, "cuda")[:5]
print("cold            %.1f us" % (t.time_op(args, 30) * 1e6))
burn(20)
print("after 20 s burn %.1f us" % (t.time_op(args, 30) * 1e6))
time.sleep(5)
print("burn + 5 s idle %.1f us" % (t.time_op(args, 30) * 1e6))
# fragmented pool: many small live allocations, then the op's tensors again
junk = [torch.empty(3 * 1024 * 1024 + 4096 * (i % 7), dtype=torch.uint8, device="cuda") for i in range(3000)]
del junk[::2]
args2 = t.inputs(1, shapes, sum(h * w for h, w in shapes), 3.0, "cuda", seed=1)[:5]
print("fragmented pool %.1f us" % (t.time_op(args2, 30) * 1e6))
