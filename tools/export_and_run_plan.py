#!/usr/bin/env python3
"""Export the launch plan of the full Co-DINO Swin-L model at WxH / batch B (fp16, seeded random-init weights as
bench.py) and replay it with the C++ runner:   python tools/export_and_run_plan.py [--res 1920x1280] [--batch 1]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "co-detr-tensorrt_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from codetr.export import export_plan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="1920x1280")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--plan", default="/tmp/codetr_swinl.plan")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
W, H = (int(v) for v in a.res.split("x"))
dev = torch.device("cuda", 0)
model = bench.build_model(dev, torch.float16)
g = torch.Generator(device=dev).manual_seed(42)
img = torch.randn(a.batch, 3, H, W, device=dev, generator=g).half()
mask = torch.zeros(a.batch, H, W, device=dev, dtype=torch.float16)
t0 = time.time()
info = export_plan(model, img, mask, a.plan)
info["export_s"] = round(time.time() - t0, 1)
info["plan_bytes"] = os.path.getsize(a.plan)
with torch.no_grad():
    boxes, scores, labels = model(img, mask)
out_dir = a.plan + ".out"
os.makedirs(out_dir, exist_ok=True)
p = subprocess.run([os.path.join(ROOT, "runner", "codetr_runner"), "--plan", a.plan, "--lib",
                    os.path.join(ROOT, "co-detr-tensorrt_amd", "codetr", "libcodetr_hip.so"), "--iters", str(a.iters),
                    "--out-dir", out_dir], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if p.returncode != 0:
    sys.exit("runner failed: " + p.stderr[-2000:])
rep = json.loads(p.stdout.strip().splitlines()[-1])
import numpy as np  # noqa: E402

same = (np.array_equal(np.fromfile(out_dir + "/boxes.bin", np.uint16), boxes.cpu().numpy().reshape(-1).view(np.uint16))
        and np.array_equal(np.fromfile(out_dir + "/labels.bin", np.int64), labels.cpu().numpy().reshape(-1)))
print(json.dumps({"res": a.res, "batch": a.batch, "export": info, "runner": rep, "identical_to_python_host": bool(same)}))
