"""Python-free deployment (reference codetr_inference.cpp:322-438): codetr/export.py records one fp16
forward into a launch plan, runner/codetr_runner (C++ over the C ABI, no Python / PyTorch) replays it -- eagerly and
from a hipGraph -- and must return the same detections as the Python host, bit for bit, also for new inputs."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers_model import seeded_params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RUNNER = os.path.join(ROOT, "runner", "codetr_runner")
LIB = os.path.join(ROOT, "co-detr-tensorrt_amd", "codetr", "libcodetr_hip.so")


def _model():
    import codetr
    from test_model_gpu import _tiny_codetr_cfg

    torch.manual_seed(0)
    cfg = _tiny_codetr_cfg("swin")
    cfg["backbone"].update(embed_dims=64, num_heads=[2, 4, 8, 16], window_size=12)
    cfg["neck"]["in_channels"] = [64, 128, 256, 512]
    model = codetr.CoDETR(**cfg)
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 5, scale=1.0))
    model.load_state_dict(full)
    return model.to(DEV).half().eval()


def _run(args):
    p = subprocess.run([RUNNER] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    return p


def _outputs(d):
    return (np.fromfile(os.path.join(d, "boxes.bin"), dtype=np.float16), np.fromfile(os.path.join(d, "scores.bin"), dtype=np.float16),
            np.fromfile(os.path.join(d, "labels.bin"), dtype=np.int64))


def test_plan_replay_matches_python_host(tmp_path):
    from codetr.export import export_plan

    model = _model()
    g = torch.Generator(device=DEV).manual_seed(3)
    img = torch.randn(2, 3, 160, 192, device=DEV, generator=g).half()
    mask = torch.zeros(2, 160, 192, device=DEV, dtype=torch.float16)
    mask[1, :, 170:] = 1
    plan = str(tmp_path / "tiny.plan")
    info = export_plan(model, img, mask, plan)
    assert info["launches"] > 100 and os.path.getsize(plan) > 1 << 20
    with torch.no_grad():
        boxes, scores, labels = model(img, mask)
    for extra in ([], ["--no-graph"]):
        out = tmp_path / ("out" + "_".join(extra))
        out.mkdir()
        p = _run(["--plan", plan, "--lib", LIB, "--iters", "3", "--out-dir", str(out)] + extra)
        assert p.returncode == 0, p.stderr[-2000:]
        rep = json.loads(p.stdout.strip().splitlines()[-1])
        assert rep["launches"] == info["launches"] and rep["hipgraph"] == (not extra)
        b, s, l = _outputs(str(out))
        assert np.array_equal(b.view(np.uint16), boxes.cpu().numpy().reshape(-1).view(np.uint16))
        assert np.array_equal(s.view(np.uint16), scores.cpu().numpy().reshape(-1).view(np.uint16))
        assert np.array_equal(l, labels.cpu().numpy().reshape(-1))
    # a different image through the same plan
    img2 = torch.randn(2, 3, 160, 192, device=DEV, generator=g).half()
    (tmp_path / "in.bin").write_bytes(img2.cpu().numpy().tobytes())
    out = tmp_path / "out2"
    out.mkdir()
    p = _run(["--plan", plan, "--lib", LIB, "--iters", "2", "--out-dir", str(out), "--input", f"batch_inputs={tmp_path / 'in.bin'}"])
    assert p.returncode == 0, p.stderr[-2000:]
    with torch.no_grad():
        boxes2, scores2, labels2 = model(img2, mask)
    b, s, l = _outputs(str(out))
    assert np.array_equal(b.view(np.uint16), boxes2.cpu().numpy().reshape(-1).view(np.uint16))
    assert np.array_equal(l, labels2.cpu().numpy().reshape(-1))
    assert not np.array_equal(l, labels.cpu().numpy().reshape(-1)) or not torch.equal(boxes, boxes2)


def test_runner_fails_loudly(tmp_path):
    p = _run(["--plan", str(tmp_path / "missing.plan"), "--lib", LIB])
    assert p.returncode != 0 and "cannot open" in p.stderr
    (tmp_path / "bad.plan").write_bytes(b"not a plan at all, just bytes" * 4)
    p = _run(["--plan", str(tmp_path / "bad.plan"), "--lib", LIB])
    assert p.returncode != 0 and "not a plan" in p.stderr
    p = _run(["--plan", str(tmp_path / "bad.plan"), "--lib", str(tmp_path / "nolib.so")])
    assert p.returncode != 0 and "cannot load" in p.stderr


def test_export_refuses_a_forward_with_foreign_kernels(tmp_path):
    """an odd-sized Swin map takes PatchMerging's F.pad (ATen): a plan could not replay it, so the export must refuse"""
    from codetr.export import export_plan

    model = _model()
    img = torch.randn(1, 3, 152, 200, device=DEV).half()       # 38 x 50 -> 19 x 25 (odd) -> pad
    mask = torch.zeros(1, 152, 200, device=DEV, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="not exportable"):
        export_plan(model, img, mask, str(tmp_path / "odd.plan"))


def test_plan_at_a_published_size_covers_the_large_shape_entries():
    """tools/export_and_run_plan.py at 1152x768 (BASELINE config 3: 73 656 encoder tokens): the entry points that only serve
    large launches -- the one-launch encoder projections with the positional operand generated in the kernel, the fused
    FFN with the output projection inside, the ping-pong GEMM -- are recorded with their host arrays and replayed
    bit-identically by the C++ runner (a wrapper that hands the exporter a cast pointer instead of the ctypes array fails
    here, not in the small model above)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "export_and_run_plan.py"), "--res", "1152x768", "--batch", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["identical_to_python_host"] is True, rep
    assert rep["runner"]["launches"] == rep["export"]["launches"] > 200
