"""GPU parity of the small geometry kernels of the detection transformer against the ATen formulations they replace
(codetr.transformer's own helper functions, which mirror the reference line by line):
  codetr_encoder_geometry_f16  get_reference_points / per-level scaling / proposals / keep-drop state
  codetr_row_max_f16           enc_outputs_class.max(-1)[0]
  codetr_query_sine_embed_f16  sigmoid x valid ratios + gen_sineembed_for_position of the decoder boxes
  codetr_linear_* row state 2  `memory * keep` folded into enc_output
Reference points, per-level points and keep/drop states are fp16-exact restatements: bit-exact.  Proposals and sine
embeddings are computed in fp32 and rounded once (the ATen fp16 path rounds after every op): compared against the
fp32 evaluation of the same formula at 1 fp16 ulp."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _masks(B, shapes, pad):
    g = torch.Generator(device=DEV).manual_seed(1)
    ms = []
    for h, w in shapes:
        m = torch.zeros(B, h, w, dtype=torch.bool, device=DEV)
        if pad:
            for b in range(B):
                m[b, int(h * (0.6 + 0.4 * torch.rand((), device=DEV, generator=g))):, :] = True
                m[b, :, int(w * (0.5 + 0.5 * torch.rand((), device=DEV, generator=g))):] = True
        ms.append(m)
    return ms


@pytest.mark.parametrize("B,shapes,pad", [
    (1, [(160, 240), (80, 120), (40, 60), (20, 30), (10, 15)], False),
    (2, [(76, 76), (38, 38), (19, 19), (10, 10), (5, 5)], True),
    (3, [(42, 65), (21, 33), (11, 17), (6, 9), (3, 5)], True),
])
def test_encoder_geometry_matches_aten(B, shapes, pad):
    from codetr import _cabi, hip_ops
    from codetr import transformer as T

    masks = _masks(B, shapes, pad)
    mask_flat = torch.cat([m.flatten(1) for m in masks], 1)
    vr = torch.stack([T.get_valid_ratio(m, dtype=torch.float16) for m in masks], 1)
    before = _cabi.CALLS["encoder_geometry"]
    ref, ref_lvl, prop, state = hip_ops.encoder_geometry(vr, mask_flat, shapes)
    assert _cabi.CALLS["encoder_geometry"] == before + 1
    ref0 = T.get_reference_points([tuple(s) for s in shapes], vr, device=DEV)
    assert torch.equal(ref, ref0)
    assert torch.equal(ref_lvl, ref0[:, :, None] * vr[:, None])
    # proposals: fp32 evaluation of the same formula on the same fp16 inputs
    lvl = T.get_lvl_repeated(masks, dtype=torch.float16)
    wh = (0.05 * (2.0 ** lvl)).expand(B, -1).reshape(B, -1, 1)
    p = torch.cat((ref0, wh, wh), -1).float()
    logit = torch.log(p / (1 - p))
    logit_h = logit.half()
    inside = ((logit_h > -4.6) & (logit_h < 4.6)).all(-1)
    keep = inside & ~mask_flat
    # a logit within an ulp of the +-4.6 bound may land on either side; everything else must agree
    near = ((logit.abs() - 4.6).abs() < 4e-3).any(-1)
    assert torch.equal((state == 0)[~near], keep[~near])
    assert set(state.unique().tolist()) <= {0, 2}
    k = state == 0
    assert ((prop[k].float() - logit[k]).abs() <= 2.0 ** -10 * logit[k].abs() + 1e-6).all()
    d = prop[~k]
    fin = torch.isfinite(logit_h[~k])
    assert (d[fin] == 65504).all() and torch.isnan(d[~fin]).all()


def test_row_max_matches_torch_and_propagates_nan():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(0)
    for rows, C in [(204600, 80), (1000, 81), (7, 8), (33, 300)]:
        x = torch.randn(rows, C, device=DEV, generator=g).half()
        x[3, C // 2] = float("nan")
        x[5] = float("-inf")
        y = hip_ops.row_max(x)
        ref = x.max(-1)[0]
        assert torch.isnan(y[3]) and torch.isnan(ref[3])
        ok = torch.ones(rows, dtype=torch.bool, device=DEV)
        ok[3] = False
        assert torch.equal(y[ok], ref[ok])
    x3 = torch.randn(2, 50, 80, device=DEV, generator=g).half()
    assert torch.equal(hip_ops.row_max(x3), x3.max(-1)[0])


@pytest.mark.parametrize("B,Nq,d,L", [(1, 900, 4, 5), (2, 77, 4, 4), (2, 50, 2, 5)])
def test_query_sine_embed_matches_formula(B, Nq, d, L):
    from codetr import _cabi, hip_ops
    from codetr.transformer import DinoTransformerDecoder

    g = torch.Generator(device=DEV).manual_seed(4)
    ref = (torch.randn(B, Nq, d, device=DEV, generator=g) * 2).half()
    vr = (0.5 + 0.5 * torch.rand(B, L, 2, device=DEV, generator=g)).half()
    assert hip_ops.query_sine_embed_supported(ref, vr, 128)
    before = _cabi.CALLS["query_sine_embed"]
    ref_in, emb = hip_ops.query_sine_embed(ref, vr, 128)
    assert _cabi.CALLS["query_sine_embed"] == before + 1
    vrd = torch.cat((vr, vr), -1) if d == 4 else vr
    ref_in0 = ref[:, :, None].sigmoid() * vrd[:, None]
    # sigmoid: one rounding on both sides, exp implementations may differ in the last fp32 bit -> 1 fp16 ulp
    assert ref_in.shape == ref_in0.shape
    assert ((ref_in.float() - ref_in0.float()).abs() <= 2.0 ** -10 * ref_in0.float().abs() + 1e-7).all()
    emb0 = DinoTransformerDecoder.gen_sineembed_for_position(ref_in[:, :, 0, :].float(), 128)  # fp32 formula
    assert emb.shape == (B, Nq, d * 128) and emb.dtype == torch.float16
    # fp32 argument up to 2 pi: sin/cos error of the fast path + the final fp16 rounding
    assert (emb.float() - emb0).abs().max() <= 1.5e-3
    assert math.isfinite(float(emb.float().abs().sum()))


def test_linear_row_state_2_is_a_zero_input_row():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(2, 300, 256, device=DEV, generator=g).half()
    for N, act in [(256, None), (256, "relu"), (5, None)]:   # row-store path, activation, ragged-N path
        w = (torch.randn(N, 256, device=DEV, generator=g) / 16).half()
        b = torch.randn(N, device=DEV, generator=g).half()
        state = torch.zeros(2, 300, dtype=torch.uint8, device=DEV)
        state[0, 17:90] = 2
        state[1, ::7] = 2
        y = hip_ops.linear(x, w, b, act=act, row_mask=state)
        keep = (state == 0).unsqueeze(-1).half()
        y0 = hip_ops.linear(x * keep, w, b, act=act)
        assert torch.equal(y, y0)
