"""CPU: the oracle (oracle/msda_ref.c + numpy restatement) against the golden vectors captured from
the reference's own Python (tests/golden/make_golden.py), and live against the imported reference
when /root/reference is present.  Tolerances are the reference tests' own
(tests/test_multi_scale_deformable_attention.py:282-283, 319-320, 363-364, 62)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import msda_oracle as O
from conftest import GOLDEN

CASES = ["msda_g1", "msda_g2", "msda_g3_dec", "msda_g3_enc", "msda_g4"]


def _args(g):
    return g["value"], g["spatial_shapes"], g["level_start_index"], g["sampling_loc"], g["attn_weight"]


@pytest.mark.parametrize("case", CASES)
def test_c_oracle_matches_reference_golden(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    out64 = O.msda_forward_c(*_args(g), dtype=np.float64)
    out32 = O.msda_forward_c(*_args(g), dtype=np.float32)
    if "out_f64" in g.files:
        ref = g["out_f64"].astype(np.float64)
        stored_f64 = g["out_f64"].dtype == np.float64
        err = np.abs(out64 - ref)
        if stored_f64:  # reference double test: abs < 1e-18, rel < 1e-15 on the seed-3 case
            assert err.max() < 1e-15 * max(1.0, np.abs(ref).max())
            if case == "msda_g1":
                assert err.max() < 1e-18
                assert (err / np.abs(ref)).max() < 1e-15
        else:
            assert err.max() < 2e-7
    ref32 = g["out_f32"].astype(np.float64)
    # reference float test: abs < 1e-9 / rel < 1e-6 on g1; generic: rtol 1e-5, atol 1e-6 (:492)
    np.testing.assert_allclose(out32, ref32, rtol=1e-5, atol=1e-6)
    if case == "msda_g1":
        assert np.abs(out32 - ref32).max() < 1e-9
        assert (np.abs(out32 - ref32) / np.abs(ref32)).max() < 1e-6
    if "out_f16" in g.files:  # the reference's fp16 path vs our fp32 arithmetic: its own fp16 tolerance
        np.testing.assert_allclose(out32, g["out_f16"].astype(np.float64), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("case", CASES)
def test_numpy_restatement_agrees_with_c(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    a = O.msda_forward_c(*_args(g), dtype=np.float64)
    b = O.msda_forward_numpy(*_args(g), dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)


def test_im2col_step_contract():
    g = np.load(os.path.join(GOLDEN, "msda_g2.npz"))  # batch 2
    O.msda_forward_c(*_args(g), im2col_step=2)
    O.msda_forward_c(*_args(g), im2col_step=64)  # min(B, step) = 2
    O.msda_forward_c(*_args(g), im2col_step=1)
    g3 = {k: g[k] for k in g.files}
    three = [np.concatenate([g3[k], g3[k][:1]]) for k in ("value", "sampling_loc", "attn_weight")]
    with pytest.raises(ValueError, match="must divide im2col_step"):
        O.msda_forward_c(three[0], g["spatial_shapes"], g["level_start_index"], three[1], three[2], im2col_step=2)


def test_linearity_and_weight_scaling():
    """size-independent properties the op must have: linear in value and in attn_weight."""
    rng = np.random.default_rng(0)
    ss = np.array([[7, 5], [3, 4]], dtype=np.int64)
    ls = O.level_start_index_from_shapes(ss)
    S = int((ss[:, 0] * ss[:, 1]).sum())
    v1, v2 = rng.random((2, S, 3, 4)), rng.random((2, S, 3, 4))
    loc = rng.random((2, 6, 3, 2, 3, 2)) * 1.4 - 0.2
    w = rng.random((2, 6, 3, 2, 3))
    f = lambda v, ww: O.msda_forward_c(v, ss, ls, loc, ww, dtype=np.float64)  # noqa: E731
    np.testing.assert_allclose(f(v1 + 2 * v2, w), f(v1, w) + 2 * f(v2, w), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(f(v1, 3 * w), 3 * f(v1, w), rtol=1e-12, atol=1e-13)
    # constant value map + in-image points + weights summing to 1 -> the constant
    loc_in = rng.random((2, 6, 3, 2, 3, 2)) * 0.5 + 0.25
    wn = w / w.sum((-1, -2), keepdims=True)
    out = O.msda_forward_c(np.full_like(v1, 0.75), ss, ls, loc_in, wn, dtype=np.float64)
    np.testing.assert_allclose(out, 0.75, rtol=1e-12)


def test_out_of_range_points_contribute_zero():
    ss = np.array([[4, 4]], dtype=np.int64)
    ls = np.array([0], dtype=np.int64)
    v = np.ones((1, 16, 1, 2))
    loc = np.array([-0.2, 0.5, 1.3, 0.5, 0.5, -0.3, 0.5, 1.26]).reshape(1, 1, 1, 1, 4, 2)
    w = np.ones((1, 1, 1, 1, 4))
    assert np.all(O.msda_forward_c(v, ss, ls, loc, w, dtype=np.float64) == 0)
    # exactly on the border (x = 0 -> w_im = -0.5): half of the border pixel
    loc = np.array([0.0, 0.5]).reshape(1, 1, 1, 1, 1, 2)
    out = O.msda_forward_c(v, ss, ls, loc, np.ones((1, 1, 1, 1, 1)), dtype=np.float64)
    np.testing.assert_allclose(out, 0.5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/codetr"), reason="reference tree not present")
def test_live_against_imported_reference():
    """Fresh random cases through the reference's Python (separate process: it owns the module
    name `codetr`) and through the oracle."""
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import _ref_import as R, msda_oracle as O
f = R.ref('ops').multi_scale_deformable_attention_pytorch
torch.manual_seed(7)
worst = 0.0
for (B, Nq, M, D, shapes, P) in [(2, 5, 3, 6, [(5, 7), (2, 3)], 3), (1, 9, 8, 32, [(6, 9), (3, 5), (2, 3), (1, 2), (1, 1)], 4)]:
    ss = torch.tensor(shapes); S = int(ss.prod(1).sum()); L = len(shapes)
    v = torch.randn(B, S, M, D, dtype=torch.float64)
    loc = torch.rand(B, Nq, M, L, P, 2, dtype=torch.float64) * 1.3 - 0.15
    w = torch.rand(B, Nq, M, L, P, dtype=torch.float64)
    ref = f(v, ss, loc, w).numpy()
    got = O.msda_forward_c(v.numpy(), ss.numpy(), O.level_start_index_from_shapes(ss.numpy()), loc.numpy(), w.numpy(), dtype=np.float64)
    worst = max(worst, float(np.abs(ref - got).max()))
print('WORST', worst)
assert worst < 1e-13
"""
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code, os.path.join(here, "golden"), os.path.join(here, "..", "oracle")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
