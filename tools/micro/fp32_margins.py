"""Print the error margins of tests/test_model_gpu.py::test_full_codetr_fp32_vs_oracle[swin] (rel-L2 / worst element
against each bound).   python tools/micro/fp32_margins.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
import numpy as np  # noqa: E402

import helpers_model  # noqa: E402

orig = helpers_model.assert_close_lowp


def spy(actual, ref, rel_l2=1e-2, max_abs=None, what=""):
    a, r = np.asarray(actual, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    ok = np.isfinite(r)
    err = np.linalg.norm(a[ok] - r[ok]) / max(np.linalg.norm(r[ok]), 1e-30)
    worst = np.abs(a[ok] - r[ok]).max() / max(np.abs(r[ok]).max(), 1e-30)
    print(f"MARGIN {what:24s} relL2 {err:.2e} / {rel_l2:.0e}   worst {worst:.2e} / {max_abs}")
    return orig(actual, ref, rel_l2, max_abs, what)


import test_model_gpu  # noqa: E402

test_model_gpu.assert_close_lowp = spy
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    test_model_gpu.test_full_codetr_fp32_vs_oracle("swin", (76, 100))
