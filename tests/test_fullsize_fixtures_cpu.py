"""The committed full-size oracle-row fixtures (tests/golden/fullsize_*.npz) still describe the current model layout
and the current oracle (CPU only; the GPU comparisons are in tests/test_timed_route_gpu.py)."""
from functools import partial

import numpy as np
import pytest
import torch

import codetr_fp32 as M
import fullsize_cases as F
from helpers_model import valid_topk


@pytest.mark.parametrize("name", sorted(F.CASES))
def test_fixture_matches_parameter_layout(name):
    import codetr
    import os

    fx = F.load_fixture(name)
    model = codetr.build_CoDETR(os.path.join(F.CFG_DIR, F.CASES[name]["cfg"]), None, "cpu")
    assert str(fx["spec_digest"]) == F.spec_digest(model.state_dict())
    for k in ("memory", "enc_outputs_class", "final_state", "outputs_classes", "outputs_coords", "topk_indices", "neck4"):
        assert k in fx and np.isfinite(fx[k]).all(), k
    assert fx["topk_indices"].shape == (F.CASES[name]["B"], 900)


def test_oracle_reproduces_the_r50_fixture():
    """the cheapest case end to end (about 5 s): fixture == what oracle/codetr_fp32.py computes today"""
    name = "r50_608"
    fx = F.load_fixture(name)
    _, sd, img, mask = F.build_case(name)
    cap = {}
    with torch.no_grad():
        M.codetr_forward(sd, img, mask, backbone="r50", forced_topk=partial(valid_topk, bound=50.0), capture=cap)
    assert np.array_equal(cap["topk_indices"].numpy(), fx["topk_indices"])
    got = F.sample_capture(name, cap)
    for k, v in got.items():
        np.testing.assert_allclose(v, fx[k], rtol=1e-4, atol=1e-4 * float(np.abs(fx[k]).max()), err_msg=k)


def test_oracle_reproduces_the_single_decoder_layer_fixture():
    """tests/golden/decoder_layer_1920x1280.npz == what oracle/codetr_fp32.decoder computes today for the layer under test
    on the stored (fp16-rounded) layer inputs (about 5 s): the GPU test's expectation is the oracle's, not an edited file"""
    import decoder_layer_case as D

    fx = D.load_fixture()
    dec, reg = D.build_decoder()
    sd = D.state_dict(dec, reg)
    lid = int(fx["layer"])
    one = {k: v for k, v in sd.items() if ".layers." not in k and not k.startswith("reg.")}
    one.update({k.replace(f"dec.layers.{lid}.", "dec.layers.0."): v for k, v in sd.items() if k.startswith(f"dec.layers.{lid}.")})
    one.update({k.replace(f"reg.{lid}.", "reg.0."): v for k, v in sd.items() if k.startswith(f"reg.{lid}.")})
    memory, pad, vr, ss, start = D.memory_and_masks()
    cap = []
    with torch.no_grad():
        M.decoder(one, "dec", torch.from_numpy(fx["x_in"]).float(), memory, pad, torch.from_numpy(fx["ref_in_unact"]).float(),
                  vr, ss, start, "reg", layer_capture=cap)
    for k in ("qpos", "x_out", "ref_out_unact"):
        np.testing.assert_allclose(cap[0][k].numpy(), fx[k], rtol=1e-4, atol=1e-4 * float(np.abs(fx[k]).max()), err_msg=k)
