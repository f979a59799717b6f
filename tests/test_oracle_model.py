"""CPU: the functional fp32 oracle (oracle/codetr_fp32.py) against module-level outputs captured
from the reference's own Python classes (tests/golden/model_*.npz, made by make_golden.py).
Tolerance: fp32 re-association noise only (different but equivalent op order), 2e-4 abs on O(1)
activations after up to 4 transformer layers; tighter where the chain is short."""
import os

import numpy as np
import torch

import codetr_fp32 as M
from conftest import GOLDEN
from helpers_model import seeded_params, unpack_param_spec


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_sine_positional_encoding():
    g = _load("model_posenc")
    out = M.sine_positional_encoding(_t(g["mask"]), torch.float32)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-6)


def test_msda_module_2d_and_4d_reference_points():
    g = _load("model_msda_module")
    sd = {"m." + k: v for k, v in seeded_params(unpack_param_spec(g), int(g["seed"])).items()}
    ss, ls = _t(g["spatial_shapes"]), _t(g["level_start_index"])
    kpm = _t(g["key_padding_mask"])
    value = _t(g["value"]).permute(1, 0, 2)  # reference modules are sequence-first
    qpos = _t(g["query_pos"]).permute(1, 0, 2)
    out2 = M.msda_module(sd, "m", value, None, qpos, kpm, _t(g["ref2"]), ss, ls)
    np.testing.assert_allclose(out2.permute(1, 0, 2).numpy(), g["out2"], rtol=1e-4, atol=2e-5)
    q4 = _t(g["query4"]).permute(1, 0, 2)
    qp4 = _t(g["query_pos4"]).permute(1, 0, 2)
    out4 = M.msda_module(sd, "m", q4, value, qp4, kpm, _t(g["ref4"]), ss, ls)
    np.testing.assert_allclose(out4.permute(1, 0, 2).numpy(), g["out4"], rtol=1e-4, atol=2e-5)


def transformer_fixture():
    g = _load("model_transformer")
    sd = {}
    for tag in ("t", "c", "r"):
        sub = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + ".spec.")}
        sd.update(seeded_params(unpack_param_spec(sub), int(g[tag + ".seed"])))
    gen = torch.Generator().manual_seed(int(g["feat_seed"]))
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    feats = [torch.randn(2, 256, h, w, generator=gen) for h, w in shapes]
    return g, sd, feats


def test_transformer_encoder_decoder_with_padding():
    g, sd, feats = transformer_fixture()
    img_mask = _t(g["img_mask"])
    masks = [torch.nn.functional.interpolate(img_mask[:, None], size=f.shape[-2:]).to(torch.bool).squeeze(1) for f in feats]
    pos = [M.sine_positional_encoding(m, torch.float32) for m in masks]
    cap = {}
    state, refs = M.transformer(sd, feats, masks, pos, num_query=40, capture=cap)
    np.testing.assert_allclose(cap["memory"].numpy(), g["memory"], rtol=1e-4, atol=1e-4)
    # the reference's own top-k must be reproduced here (fp32 noise ~1e-6 on logits that are well separated);
    # if it is, the decoder outputs agree too
    np.testing.assert_allclose(state.numpy(), g["final_state"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(refs.numpy(), g["final_refs_unact"], rtol=2e-4, atol=2e-4)


def head_fixture():
    g = _load("model_head")
    sd = seeded_params(unpack_param_spec(g), int(g["seed"]))
    gen = torch.Generator().manual_seed(int(g["feat_seed"]))
    feats = [torch.randn(2, 256, h, w, generator=gen) for h, w in [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]]
    return g, sd, feats


def test_head_forward_vs_reference_codinohead():
    """the oracle's head (mask pyramid -> encodings -> transformer -> branches -> top-k -> decode) against the
    REFERENCE's own CoDINOHead.forward (tests/golden/model_head.npz; only mmdet's DINOHead constructor chain and
    bbox_cxcywh_to_xyxy were stand-ins when it was captured)"""
    g, sd, feats = head_fixture()
    cap = {}
    boxes, scores, labels = M.head(sd, feats, _t(g["img_mask"]), num_query=40, max_per_img=25, capture=cap)
    assert np.array_equal(cap["topk_indices"].numpy(), g["proposal_topk"])   # the reference's own proposal selection
    assert np.array_equal(labels.numpy(), g["labels"])
    np.testing.assert_allclose(scores.numpy(), g["scores"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(boxes.numpy(), g["boxes"], rtol=1e-4, atol=2e-3)   # pixels of a 64 x 48 image


def test_swin_tiny_padding_shift_merging():
    g = _load("model_swin_tiny")
    sd = seeded_params(unpack_param_spec(g), int(g["seed"]), scale=2.0)
    outs = M.swin_forward(sd, _t(g["img"]), num_heads=(2, 4), window_size=4, out_indices=(0, 1))
    np.testing.assert_allclose(outs[0].numpy(), g["out0"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(outs[1].numpy(), g["out1"], rtol=1e-4, atol=2e-5)


# ---- known-answer tests for the third-party pieces the reference tree cannot pin (mmdet absent) ----


def test_channel_mapper_known_answer():
    """1x1 conv + GN(32) per level; extra level = 3x3 stride-2 conv on the RAW last input."""
    torch.manual_seed(0)
    feats = [torch.randn(1, 4, 6, 6), torch.randn(1, 8, 3, 3)]
    sd = {}
    for i, c in enumerate((4, 8)):
        sd[f"neck.convs.{i}.conv.weight"] = torch.randn(64, c, 1, 1)
        sd[f"neck.convs.{i}.gn.weight"] = torch.rand(64) + 0.5
        sd[f"neck.convs.{i}.gn.bias"] = torch.randn(64)
    sd["neck.extra_convs.0.conv.weight"] = torch.randn(64, 8, 3, 3)
    sd["neck.extra_convs.0.gn.weight"] = torch.rand(64) + 0.5
    sd["neck.extra_convs.0.gn.bias"] = torch.randn(64)
    outs = M.channel_mapper(sd, feats)
    assert [tuple(o.shape) for o in outs] == [(1, 64, 6, 6), (1, 64, 3, 3), (1, 64, 2, 2)]
    # hand computation of one group of level 0
    y = torch.einsum("oc,chw->ohw", sd["neck.convs.0.conv.weight"][:, :, 0, 0], feats[0][0])
    grp = y[:2]
    ref = (grp - grp.mean()) / torch.sqrt(grp.var(unbiased=False) + 1e-5)
    ref = ref * sd["neck.convs.0.gn.weight"][:2, None, None] + sd["neck.convs.0.gn.bias"][:2, None, None]
    torch.testing.assert_close(outs[0][0, :2], ref, rtol=1e-4, atol=1e-5)
    # extra conv consumes feats[-1] (raw), output 2x2 = ceil(3/2)
    y2 = torch.nn.functional.conv2d(feats[1], sd["neck.extra_convs.0.conv.weight"], stride=2, padding=1)
    assert y2.shape[-2:] == (2, 2)


def test_head_decode_known_answer():
    """top-k over (query, class), label = idx % C, box = idx // C, cxcywh -> xyxy, scale, clamp."""
    B, Nq, C = 1, 3, 4
    cls = torch.full((B, Nq, C), -10.0)
    cls[0, 2, 1] = 3.0   # best
    cls[0, 0, 3] = 1.0   # second
    coords = torch.tensor([[[0.5, 0.5, 0.2, 0.4], [0.1, 0.1, 0.1, 0.1], [0.9, 0.95, 0.4, 0.2]]])
    scores, idx = torch.topk(cls.sigmoid().view(B, -1), 2, dim=-1)
    assert idx.tolist() == [[2 * C + 1, 0 * C + 3]]
    labels, q = idx % C, idx // C
    assert labels.tolist() == [[1, 3]] and q.tolist() == [[2, 0]]
    box = torch.gather(coords, 1, q[..., None].expand(-1, -1, 4))
    cx, cy, w, h = box.unbind(-1)
    xyxy = torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), -1) * torch.tensor([100.0, 50, 100, 50])
    xyxy = torch.minimum(xyxy.clamp(min=0), torch.tensor([100.0, 50, 100, 50]))
    torch.testing.assert_close(xyxy, torch.tensor([[[70.0, 42.5, 100.0, 50.0], [40.0, 15.0, 60.0, 35.0]]]))
