#!/usr/bin/env python3
"""How many host threads serve the fp32 CPU oracle best on this box?  (bench.py's cpu_baseline: 128 torch threads on the
GPU box's 256 logical CPUs took 95 s for a 608x608 image that 8 threads of the build container finish in 15 s.)
    python tools/probe_cpu_oracle_threads.py [HxW]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import codetr  # noqa: E402
import codetr_fp32 as M  # noqa: E402

H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "384x384").split("x"))
torch.manual_seed(0)
model = codetr.build_CoDETR(bench.CFG, None, "cpu")
model.init_weights()
sd = {k: v.detach().float() for k, v in model.state_dict().items()}
img = torch.randn(1, 3, H, W)
mask = torch.zeros(1, H, W)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "default threads", torch.get_num_threads(),
      "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
for n in (8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 8):
        continue
    torch.set_num_threads(n)
    os.environ["OMP_NUM_THREADS"] = str(n)
    t0 = time.perf_counter()
    with torch.no_grad():
        M.codetr_forward(sd, img, mask, backbone="swin", num_heads=(6, 12, 24, 48), window_size=12)
    print(f"threads {n:4d}: {time.perf_counter() - t0:7.2f} s", flush=True)
