"""The configuration BASELINE.json's metric is quoted on -- Co-DINO Swin-L 1920x1280 fp16 -- against the fp32 oracle AT
ITS OWN SIZE and at the launch shapes bench.py replays (4-image graphs, default dispatch thresholds):

  * tests/golden/fullsize_swinl_1920x1280.npz holds oracle rows of one padded 1920x1280 image (made in the build
    container by tests/golden/make_fullsize_rows.py: 70 s of the CPU oracle), including the first / last row of every
    pyramid level and rows either side of multiples of 128 / 256 -- the tile edges of the GEMM / FFN kernels;
  * the product runs a batch of FOUR images (image 0 = the fixture's image, three others, one of them padded
    differently), nothing lowered or forced except the proposal selection (the reference disables its own value asserts
    for that instability, tests/test_export.py:638-655);
  * the call counters prove which kernels served it: (x + pos) folded into the offsets GEMM (>= 400 k rows), the
    LDS-staged encoder MSDA kernel, the persistent fused FFN, X-stationary / 256-tile / split-K GEMMs;
  * image 0's rows are compared with the oracle's (tensor rel-L2 AND the per-row bounds of helpers_model.assert_rows_
    close), and with the rows the same image gets alone (B = 1: other GEMM kernels serve it; fp16 noise only).

Tolerances as tests/test_timed_route_gpu.py: rel-L2 <= 1e-2 (decoder-side tensors 2.5e-2); per row <= 5 x that; in no row
more than 1 % of the elements beyond 4 x the bound (decoder side: 5 %).  Batch of 4 vs alone: 5e-3 up to the encoder;
the decoder side gets the oracle bound (2.5e-2): on these random weights the six-layer decoder with its layer-to-layer
box refinement amplifies a 1e-3 difference of the memory to ~2e-2 of its output whatever its source (measured: batch vs
alone 1.9e-2 from a 1.1e-3 memory difference; vs the oracle 2.2e-2 from 1.6e-3) -- individual queries' refinement
trajectories diverge, which is also why the element criterion is wider there.

Measured (gpurun_out/parity_report_headline.json): memory 1.6e-3 since the encoder's MSDA kernel computes its reference
points in fp32 (1.14e-2 with the fp16 reference-point tensor: a quarter pixel of resolution on the 480-wide level)."""
import json
import os

import numpy as np
import pytest
import torch

import fullsize_cases as F
from conftest import ROOT
from helpers_model import assert_rows_close, valid_topk

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAME = "swinl_1920x1280"
DEEP = {"final_state": 2.5, "outputs_classes": 2.5, "outputs_coords": 2.5}


@pytest.fixture(scope="module")
def case():
    fx = F.load_fixture(NAME)
    model, full, img, mask = F.build_case(NAME)
    assert str(fx["spec_digest"]) == F.spec_digest(full), "fixture was made for another parameter layout: regenerate"
    return fx, model.to(DEV).half().eval(), img, mask


def _batch4(img, mask):
    imgs, masks = [img], [mask]
    for i, pad in enumerate((None, (0.8, 0.85), None)):
        a, b = F.case_input(NAME, image_seed=101 + i, pad=pad)
        imgs.append(a)
        masks.append(b)
    return torch.cat(imgs).to(DEV).half(), torch.cat(masks).to(DEV).half()


def _write_report(tag, errs):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        path = os.path.join(d, "parity_report_headline.json")
        rep = json.load(open(path)) if os.path.isfile(path) else {}
        rep[tag] = errs
        with open(path, "w") as f:
            json.dump(rep, f, indent=1, sort_keys=True)


def _compare(got, want, rel, tag, deep_rel=None):
    errs = {}
    for k, v in got.items():
        bound = (deep_rel if deep_rel is not None else rel * DEEP[k]) if k in DEEP else rel
        e, row, frac = assert_rows_close(v, want[k], bound, f"{tag}: {k}", elem_frac=0.05 if k in DEEP else 0.01)
        errs[k] = {"rel_l2": float(f"{e:.3e}"), "worst_row_over_limit": round(row, 3), "worst_row_outlier_frac": round(frac, 4)}
    return errs


def test_headline_batch4_vs_oracle_rows_and_alone(case):
    from codetr import _cabi

    fx, model, img, mask = case
    x4, m4 = _batch4(img, mask)
    picks0 = torch.from_numpy(fx["topk_indices"]).to(DEV)
    with torch.no_grad():
        # proposals of images 1-3: the reference's rule on the product's own encoder outputs (finite boxes only)
        cap = {}
        model(x4, m4, capture=cap)
        picks = valid_topk(cap["enc_outputs_class"].float(), cap["enc_outputs_coord_unact"].float(), 900, bound=50.0)
        picks[0] = picks0[0]
        del cap
        cap4 = {}
        out4 = model(x4, m4, forced_topk_indices=picks, capture=cap4)
        before = dict(_cabi.CALLS)
        out4b = model(x4, m4, forced_topk_indices=picks)      # no hook: the launches bench.py replays
        torch.cuda.synchronize()
        calls = {k: _cabi.CALLS[k] - before[k] for k in before}
    assert cap4["route"] == "tokens"
    # per encoder layer: one projection launch (value + packed offsets | logits), the packed MSDA kernel, one launch from the
    # attention output to the layer output (output_proj + identity, norm1, FFN, norm2, + query_pos)
    assert calls["encoder_projections"] == 6 and calls["msda_encoder_packed"] == 6 and calls["ffn_oproj_fused"] == 6, calls
    assert calls["encoder_projections_posgen"] == 6, calls   # the positional operand is generated in the kernel, not read
    assert calls["linear_xadd"] == 0 and calls["ffn_fused"] == 6, calls
    # (decoder: the head-only launch + one per layer, the self-attention cores between them)
    assert calls["decoder_layer"] == 7 and calls["mha_attention"] == 6 and calls["window_attention"] == 24, calls
    assert calls["linear_xs"] > 0 and calls["linear_splitk"] > 0, calls
    # round 6: every linear of Swin stages 2 and 3 (20 blocks x 4) and stage 1's fc2 run on the ping-pong persistent GEMM
    assert calls["linear_pp"] >= 80 and calls["linear_sk"] > 0, calls
    # round 6: the MLPs of Swin stages 0 and 1 (2 + 2 blocks) are ONE launch each (norm2, fc1, GELU, fc2, identity); norm1 of
    # stage 0 stays folded into its qkv GEMM (2 launches)
    assert calls["swin_mlp"] == 4 and calls["linear_ln"] == 2, calls
    assert calls["linear_tile128"] > 0 and calls["topk"] == 1, calls
    for a, b in zip(out4, out4b):   # the capture hook changes nothing
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    got = F.sample_capture(NAME, cap4, image=0, images=4)
    errs = _compare(got, fx, 1e-2, "batch of 4 vs oracle")
    np.testing.assert_allclose(out4[1][0].float().cpu().numpy(), fx["scores"][0], rtol=2e-2, atol=2e-3)
    _write_report("batch4_image0_vs_oracle_fp16", errs)
    # the same image alone
    with torch.no_grad():
        cap1 = {}
        model(x4[:1], m4[:1], forced_topk_indices=picks0, capture=cap1)
    alone = F.sample_capture(NAME, cap1)
    _write_report("batch4_image0_vs_alone_fp16", _compare(got, alone, 5e-3, "batch of 4 vs alone", deep_rel=2.5e-2))
    _write_report("alone_vs_oracle_fp16", _compare(alone, fx, 1e-2, "alone vs oracle"))


@pytest.mark.parametrize("mode", ["mx", "static"])
def test_headline_fp8_batch4_vs_oracle_rows(case, mode):
    """BASELINE config 5 at the headline size and the bench's launch shapes: the same four images through fp8.enable
    (Swin stage 1-3 linears on e4m3 -- MX block scales, or static scales calibrated on images 2, 3 -- and the encoder's
    fused FFN on e4m3 with static scales calibrated on images 2, 3), image 0 against the oracle's fixture rows.  The
    bounds are the measured errors of e4m3 operands (3 mantissa bits compounding over 22 Swin blocks and 6 FFNs; the
    same with either scaling) with ~1.5x headroom -- they pin the path against breakage, they do not
    certify 0.1 AP; the row criterion (no row above 5x the tensor bound) holds here as for fp16."""
    from codetr import _cabi, fp8
    from helpers_model import row_error_stats

    fx, model, img, mask = case
    x4, m4 = _batch4(img, mask)
    picks0 = torch.from_numpy(fx["topk_indices"]).to(DEV)
    try:
        with torch.no_grad():
            assert fp8.calibrate(model, x4[2:], m4[2:]) == 24
            sat = fp8.saturation(model, x4[:2], m4[:2])      # the evaluated images against scales calibrated on the others
            assert sat["tensors"] >= 4 * 22 and sat["saturating"] == 0 and sat["worst_ratio"] <= 1.0, sat
            fp8.enable(model, True, mode)
            cap = {}
            model(x4, m4, capture=cap)
            picks = valid_topk(cap["enc_outputs_class"].float(), cap["enc_outputs_coord_unact"].float(), 900, bound=50.0)
            picks[0] = picks0[0]
            del cap
            before = dict(_cabi.CALLS)
            cap4 = {}
            model(x4, m4, forced_topk_indices=picks, capture=cap4)
            torch.cuda.synchronize()
            calls = {k: _cabi.CALLS[k] - before[k] for k in before}
        n8 = calls["linear_fp8"]
        assert n8 == 4 * 22 and calls["ffn_fp8"] == 6 and calls["ffn_fused"] == 0, calls      # stages 1-3: 2 + 18 + 2 blocks
        assert calls["msda_encoder_packed"] == 6 and calls["encoder_projections"] == 6 and calls["ffn_oproj_fused"] == 0, calls
        got = F.sample_capture(NAME, cap4, image=0, images=4)
        bounds = {"backbone0": 5e-3, "backbone1": 7e-2, "backbone2": 1.2e-1, "backbone3": 1.4e-1, "memory": 1.2e-1,
                  "enc_outputs_class": 1.2e-1, "neck0": 5e-3}   # measured: 7.9e-4, 4.5e-2, 8.1e-2, 8.9e-2, 9.1e-2, 8.5e-2, 9.2e-4
        errs = {}
        for k, v in got.items():
            a, r = np.asarray(v, np.float64), np.asarray(fx[k], np.float64)
            fin = np.isfinite(r)
            assert (np.isfinite(a) == fin).all(), k
            e = float(np.linalg.norm((a - r)[fin]) / max(np.linalg.norm(r[fin]), 1e-30))
            b = bounds.get(k, 1.4e-1)       # neck levels 1-4 and the decoder side: measured 4.4e-2 ... 9.7e-2
            ratio, frac, worst = row_error_stats(a, r, b)
            errs[k] = {"rel_l2": float(f"{e:.3e}"), "bound": b, "worst_row_over_limit": round(ratio / 5.0, 3),
                       "worst_row_outlier_frac": round(frac, 4)}
        _write_report(f"batch4_image0_vs_oracle_fp8_{mode}", errs)
        print("headline fp8", mode, {k: v["rel_l2"] for k, v in errs.items()})
        for k, v in errs.items():
            assert v["rel_l2"] <= v["bound"], (k, v)
            assert v["worst_row_over_limit"] <= 1.0, (k, v)
    finally:
        fp8.enable(model, False)
