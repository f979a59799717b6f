"""Print every stage's error of the headline case (tests/test_headline_gpu.py) without asserting: batch of 4 vs the
oracle rows, alone vs oracle, batch vs alone.   python tools/diag_headline.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import fullsize_cases as F  # noqa: E402
from helpers_model import row_error_stats, valid_topk  # noqa: E402

NAME, DEV = "swinl_1920x1280", "cuda:0"
fx = F.load_fixture(NAME)
model, full, img, mask = F.build_case(NAME)
model = model.to(DEV).half().eval()
picks0 = torch.from_numpy(fx["topk_indices"]).to(DEV)
imgs, masks = [img], [mask]
for i, pad in enumerate((None, (0.8, 0.85), None)):
    a, b = F.case_input(NAME, image_seed=101 + i, pad=pad)
    imgs.append(a)
    masks.append(b)
x4, m4 = torch.cat(imgs).to(DEV).half(), torch.cat(masks).to(DEV).half()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    return np.linalg.norm((a - b)[fin]) / max(np.linalg.norm(b[fin]), 1e-30)


with torch.no_grad():
    cap = {}
    model(x4, m4, capture=cap)
    picks = valid_topk(cap["enc_outputs_class"].float(), cap["enc_outputs_coord_unact"].float(), 900, bound=50.0)
    picks[0] = picks0[0]
    cap4 = {}
    model(x4, m4, forced_topk_indices=picks, capture=cap4)
    cap1 = {}
    model(x4[:1], m4[:1], forced_topk_indices=picks0, capture=cap1)
got4, got1 = F.sample_capture(NAME, cap4, image=0, images=4), F.sample_capture(NAME, cap1)
import codetr.hip_ops as _ho
print(f"hip_ops.MSDA_ENCODER={_ho.MSDA_ENCODER}")
for k in got4:
    r4, r1, r41 = rel(got4[k], fx[k]), rel(got1[k], fx[k]), rel(got4[k], got1[k])
    rr = row_error_stats(got4[k], fx[k], 1e-2) if got4[k].ndim == 2 else (0, 0, 0)
    print(f"{k:20s} b4-vs-oracle {r4:.3e}  alone-vs-oracle {r1:.3e}  b4-vs-alone {r41:.3e}   worst row ratio (of 1e-2) {rr[0]:.2f} elem frac {rr[1]:.3f}")
