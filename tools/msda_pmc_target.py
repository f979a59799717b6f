"""Target for rocprofv3 --pmc runs: a handful of MSDA launches at the 1920x1280 encoder shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda import make_inputs, pyramid  # noqa: E402

import codetr  # noqa: E402,F401

shapes = pyramid(1280, 1920)
S = sum(h * w for h, w in shapes)
value, ss, ls, loc, w, S = make_inputs(1, shapes, S, torch.float16, "cuda:0", realistic=("--uniform" not in sys.argv))
op = torch.ops.codetr.multi_scale_deformable_attention
for _ in range(5):
    out = op(value, ss, ls, loc, w, 64)
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))
