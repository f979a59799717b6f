"""Generate tests/golden/fullsize_<case>.npz: sampled rows of the fp32 CPU ORACLE's intermediates for the
full-architecture cases of tests/fullsize_cases.py (BASELINE configs 1-3: R50 608x608, Swin-L 608x608, Swin-L 1152x768).

Run in the build container (CPU, ~1-2 minutes per case); the GPU tests only read the .npz:

    python tests/golden/make_fullsize_rows.py [case ...]

These fixtures come from oracle/ (pinned against the imported reference by tests/test_oracle_model.py and the
model_*.npz fixtures), not from the reference itself: they exist because the oracle cannot run at these sizes inside
a GPU test's time budget and its full intermediates are too large to commit."""
import os
import sys
import time
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import codetr_fp32 as M  # noqa: E402
import fullsize_cases as F  # noqa: E402
from helpers_model import valid_topk  # noqa: E402


def make(name):
    c = F.CASES[name]
    t0 = time.time()
    _, sd, img, mask = F.build_case(name)
    kw = dict(backbone="swin", num_heads=(6, 12, 24, 48), window_size=12) if c["backbone"] == "swin" else dict(backbone="r50")
    cap = {}
    with torch.no_grad():
        boxes, scores, labels = M.codetr_forward(sd, img, mask, forced_topk=partial(valid_topk, bound=50.0), capture=cap, **kw)
    out = F.sample_capture(name, cap)
    out.update(topk_indices=cap["topk_indices"].numpy(), spec_digest=np.array(F.spec_digest(sd)),
               scores=scores.numpy(), labels=labels.numpy(), boxes=boxes.numpy(),
               level_shapes=cap["spatial_shapes"].numpy())
    np.savez_compressed(F.fixture_path(name), **out)
    print(f"{name}: oracle {time.time() - t0:.1f} s -> {F.fixture_path(name)} "
          f"({os.path.getsize(F.fixture_path(name)) / 1024:.0f} KiB); top scores {scores[0, :3].tolist()}")


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(F.CASES)):
        make(n)
