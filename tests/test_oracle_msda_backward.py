"""CPU: the backward oracle (oracle/msda_backward_oracle.py, closed-form col2im gradients) against gradients produced
by torch.autograd through the IMPORTED reference's own differentiable formulation (tests/golden/msda_grad.npz, made by
tests/golden/make_golden.py grad): the reference's gradient-test geometry for every channel count it checks
(tests/test_multi_scale_deformable_attention.py:367-414) and a model-shaped case with samples outside the maps."""
import os

import numpy as np
import pytest

import msda_backward_oracle as BO
import msda_oracle as O
from conftest import GOLDEN

CASES = ["c4", "c30", "c32", "c64", "c71", "c1025", "model"]


def load_case(tag):
    g = np.load(os.path.join(GOLDEN, "msda_grad.npz"))
    d = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + ".")}
    shapes = d["shapes"]
    d["level_start"] = np.concatenate(([0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1])).astype(np.int64)
    return d


@pytest.mark.parametrize("tag", CASES)
def test_backward_oracle_matches_reference_autograd(tag):
    d = load_case(tag)
    gv, gl, gw = BO.msda_backward(d["value"], d["shapes"], d["level_start"], d["loc"], d["w"], d["go"])
    for name, got, ref in (("value", gv, d["grad_value"]), ("loc", gl, d["grad_loc"]), ("w", gw, d["grad_w"])):
        scale = np.abs(ref).max() + 1e-300
        assert np.abs(got - ref).max() <= 1e-12 * scale, (tag, name, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("tag", ["c4", "c71", "model"])
def test_forward_oracle_matches_the_same_fixture(tag):
    """the forward outputs stored next to the gradients pin the forward C oracle on these geometries too"""
    d = load_case(tag)
    out = O.msda_forward_c(d["value"], d["shapes"], d["level_start"], d["loc"], d["w"], dtype=np.float64)
    np.testing.assert_allclose(out, d["out"], rtol=1e-12, atol=1e-15)
