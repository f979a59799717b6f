"""Diagnostics of tests/test_decoder_layer_oracle_gpu.py: per-phase error statistics of one decoder layer against the
oracle fixture, the worst rows, and what those rows look like (reference boxes, attention sharpness)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "co-detr-tensorrt_amd"), ROOT):
    sys.path.insert(0, p)

import test_decoder_layer_oracle_gpu as T  # noqa: E402
from helpers_model import row_error_stats  # noqa: E402

fx, got = T._run_layer()
for name in ("qpos", "x_out", "ref_out_unact", "qpos_next"):
    g = got[name].float().cpu().numpy()[0].astype(np.float64)
    r = fx[name][0].astype(np.float64)
    rel = np.linalg.norm(g - r) / np.linalg.norm(r)
    ratio, frac, worst = row_error_stats(g, r, 3e-3)
    rows = np.linalg.norm(g - r, axis=1) / np.maximum(np.linalg.norm(r, axis=1), 1e-30)
    order = np.argsort(-rows)[:8]
    print(f"{name:14s} rel L2 {rel:.3e}  worst row ratio {ratio:.2f} (row {worst})  worst elem frac {frac:.3f}  "
          f"row rel-err p50 {np.median(rows):.2e} p99 {np.quantile(rows, 0.99):.2e} max {rows.max():.2e}")
    print("   worst rows:", [(int(i), f"{rows[i]:.2e}") for i in order])
    if name == "x_out":
        ref_in = torch.from_numpy(fx["ref_in_unact"][0]).float().sigmoid()
        for i in order[:5]:
            print(f"   row {int(i)}: box (cx, cy, w, h) = {[round(float(v), 3) for v in ref_in[i]]}  |x_in| {np.linalg.norm(fx['x_in'][0][i].astype(np.float64)):.2f}")
