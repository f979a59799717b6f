"""encoder projections at the 4-image launch shape: positional operand read (codetr_encoder_projections_*) vs generated in the
kernel (codetr_encoder_projections_posgen_*), us per launch"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from codetr import hip_ops  # noqa: E402
import test_encoder_projections_gpu as t  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = t._pyramid(320, 480)
x, pos, wc, bc, mask, Nv = t._setup(B, shapes, torch.float16, False, seed=1)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


with torch.no_grad():
    for rep in range(2):
        for gen in (False, True):
            hip_ops.ENC_POSGEN = gen
            print("generated" if gen else "read     ", "%.1f us" % timeit(lambda: hip_ops.encoder_projections(x, pos, wc, bc, mask, Nv, 32)))
