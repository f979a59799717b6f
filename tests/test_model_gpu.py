"""GPU parity of the host package's modules (computing through hip_ops / the HIP MSDA kernel)
against (a) outputs captured from the reference's own modules (tests/golden/model_*.npz) and
(b) the functional fp32 CPU oracle on identical seeded weights and inputs.

Tolerances: fp32 GPU vs fp32 reference 2e-4 abs on O(1) activations (accumulation-order noise over
up to 4 layers; rocBLAS / MIOpen vs CPU kernels); fp16 GPU vs fp32 oracle is checked on
intermediate tensors with forced-equal top-k (final boxes are top-k-unstable on
random weights) as relative L2 error <= 1e-2 of each tensor plus a cap on the worst element (fp16
has 2^-11 relative rounding per op; after 4-12 residual layers on O(1..10) activations single
elements drift by a few 1e-2 while the tensor as a whole stays within 1 %)."""
import os

import numpy as np
import pytest
import torch

import codetr_fp32 as M
from conftest import GOLDEN, ROOT
from helpers_model import assert_close_lowp, seeded_params, unmatched_detections, unpack_param_spec, valid_topk

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a, dtype=None):
    t = torch.from_numpy(np.asarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def test_posenc_vs_reference():
    from codetr.positional_encoding import SinePositionalEncoding

    g = _load("model_posenc")
    pe = SinePositionalEncoding(num_feats=128, temperature=20, normalize=True)
    out = pe(_t(g["mask"]), dtype=torch.float32)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], atol=5e-6)


def test_msda_module_vs_reference():
    from codetr.multi_scale_deformable_attention import MultiScaleDeformableAttention

    g = _load("model_msda_module")
    m = MultiScaleDeformableAttention(embed_dims=256, num_levels=5, dropout=0.0).to(DEV).eval()
    m.load_state_dict({k: v.to(DEV) for k, v in seeded_params(unpack_param_spec(g), int(g["seed"])).items()})
    ss, ls = _t(g["spatial_shapes"]), _t(g["level_start_index"])
    with torch.no_grad():
        out2 = m(_t(g["value"]), value=None, query_pos=_t(g["query_pos"]), key_padding_mask=_t(g["key_padding_mask"]),
                 reference_points=_t(g["ref2"]), spatial_shapes=ss, level_start_index=ls)
        out4 = m(_t(g["query4"]), value=_t(g["value"]), query_pos=_t(g["query_pos4"]),
                 key_padding_mask=_t(g["key_padding_mask"]), reference_points=_t(g["ref4"]), spatial_shapes=ss,
                 level_start_index=ls)
    np.testing.assert_allclose(out2.cpu().numpy(), g["out2"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(out4.cpu().numpy(), g["out4"], rtol=1e-4, atol=5e-5)
    # fp16 module vs the fp32 reference output
    mh = m.half()
    with torch.no_grad():
        o = mh(_t(g["value"]).half(), value=None, query_pos=_t(g["query_pos"]).half(),
               key_padding_mask=_t(g["key_padding_mask"]), reference_points=_t(g["ref2"]).half(), spatial_shapes=ss,
               level_start_index=ls)
    assert_close_lowp(o.float().cpu().numpy(), g["out2"], rel_l2=5e-3, max_abs=5e-2, what="msda module fp16")


def _transformer_fixture():
    from test_oracle_model import transformer_fixture

    return transformer_fixture()


def _build_transformer(sd, dtype):
    from codetr.transformer import CoDinoTransformer
    import torch.nn as nn

    cfg = dict(with_coord_feat=False, num_co_heads=2, num_feature_levels=5, as_two_stage=True, two_stage_num_proposals=40,
               encoder=dict(type="DetrTransformerEncoder", num_layers=2, with_cp=4, transformerlayers=dict(
                   type="BaseTransformerLayer",
                   attn_cfgs=dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=5, dropout=0.0),
                   feedforward_channels=64, ffn_dropout=0.0, operation_order=("self_attn", "norm", "ffn", "norm"))),
               decoder=dict(type="DinoTransformerDecoder", num_layers=2, return_intermediate=True, transformerlayers=dict(
                   type="DetrTransformerDecoderLayer",
                   attn_cfgs=[dict(type="MultiheadAttention", embed_dims=256, num_heads=8, dropout=0.0),
                              dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=5, dropout=0.0)],
                   feedforward_channels=64, ffn_dropout=0.0,
                   operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))))
    t = CoDinoTransformer(**cfg)
    cls_b = nn.ModuleList(nn.Linear(256, 80) for _ in range(3))
    reg_b = nn.ModuleList(nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 4))
                          for _ in range(3))
    pre = "query_head.transformer."
    t.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    cls_b.load_state_dict({k[len("query_head.cls_branches."):]: v for k, v in sd.items() if "cls_branches" in k})
    reg_b.load_state_dict({k[len("query_head.reg_branches."):]: v for k, v in sd.items() if "reg_branches" in k})
    return t.to(DEV, dtype).eval(), cls_b.to(DEV, dtype), reg_b.to(DEV, dtype)


def test_transformer_fp32_vs_reference():
    from codetr.positional_encoding import SinePositionalEncoding

    g, sd, feats = _transformer_fixture()
    t, cls_b, reg_b = _build_transformer(sd, torch.float32)
    img_mask = _t(g["img_mask"])
    feats = [f.to(DEV) for f in feats]
    masks = [torch.nn.functional.interpolate(img_mask[:, None], size=f.shape[-2:]).to(torch.bool).squeeze(1) for f in feats]
    pe = SinePositionalEncoding(num_feats=128, temperature=20, normalize=True)
    pos = [pe(m, dtype=torch.float32) for m in masks]
    cap = {}
    with torch.no_grad():
        state, refs = t(feats, masks, pos, reg_branches=reg_b, cls_branches=cls_b, capture=cap)
    np.testing.assert_allclose(cap["memory"].cpu().numpy(), g["memory"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(state.cpu().numpy(), g["final_state"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(refs.cpu().numpy(), g["final_refs_unact"], rtol=5e-4, atol=5e-4)


def test_transformer_fp16_vs_oracle_forced_topk():
    from codetr.positional_encoding import SinePositionalEncoding

    g, sd, feats = _transformer_fixture()
    img_mask = torch.from_numpy(g["img_mask"])
    masks_c = [torch.nn.functional.interpolate(img_mask[:, None], size=f.shape[-2:]).to(torch.bool).squeeze(1) for f in feats]
    pos_c = [M.sine_positional_encoding(m, torch.float32) for m in masks_c]
    cap_o = {}
    M.transformer(sd, feats, masks_c, pos_c, num_query=40, capture=cap_o)
    picks = valid_topk(cap_o["enc_outputs_class"], cap_o["enc_outputs_coord_unact"], 40)
    state_o, refs_o = M.transformer(sd, feats, masks_c, pos_c, num_query=40, forced_topk=picks, capture=cap_o)
    t, cls_b, reg_b = _build_transformer(sd, torch.float16)
    pe = SinePositionalEncoding(num_feats=128, temperature=20, normalize=True)
    masks = [m.to(DEV) for m in masks_c]
    pos = [pe(m, dtype=torch.float16) for m in masks]
    cap = {}
    with torch.no_grad():
        state, refs = t([f.to(DEV).half() for f in feats], masks, pos, reg_branches=reg_b, cls_branches=cls_b,
                        forced_topk_indices=cap_o["topk_indices"].to(DEV), capture=cap)
    assert_close_lowp(cap["memory"].float().cpu().numpy(), cap_o["memory"].numpy(), 1e-2, 0.1, "encoder memory fp16")
    assert_close_lowp(state.float().cpu().numpy(), state_o.numpy(), 1e-2, 0.1, "decoder state fp16")
    assert_close_lowp(refs.float().cpu().numpy(), refs_o.numpy(), 1e-2, 0.1, "decoder refs fp16")


def test_swin_tiny_vs_reference_fp32_and_fp16():
    from codetr.swin import SwinTransformer

    g = _load("model_swin_tiny")
    sd = seeded_params(unpack_param_spec(g), int(g["seed"]), scale=2.0)
    s = SwinTransformer(pretrain_img_size=64, embed_dims=32, depths=(2, 2), num_heads=(2, 4), window_size=4,
                        strides=(4, 2), out_indices=(0, 1), drop_path_rate=0.0, patch_norm=True)
    res = s.load_state_dict({k[len("backbone."):]: v for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all("relative_position_index" in k for k in res.missing_keys)
    s = s.to(DEV).eval()
    with torch.no_grad():
        outs = s(_t(g["img"]))
    np.testing.assert_allclose(outs[0].cpu().numpy(), g["out0"], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(outs[1].cpu().numpy(), g["out1"], rtol=2e-4, atol=5e-5)
    with torch.no_grad():
        outs_h = s.half()(_t(g["img"]).half())
    assert_close_lowp(outs_h[0].float().cpu().numpy(), g["out0"], 1e-2, 0.1, "swin stage-0 out fp16")
    assert_close_lowp(outs_h[1].float().cpu().numpy(), g["out1"], 1e-2, 0.1, "swin stage-1 out fp16")


def _tiny_codetr_cfg(backbone):
    """A structurally complete CoDETR (all module kinds) small enough for the CPU oracle."""
    from codetr.config import Config

    cfg = Config.fromfile(os.path.join(ROOT, "co-detr-tensorrt_amd", "configs",
                                       "co_dino_5scale_swin_l_16xb1_16e_o365tococo.py" if backbone == "swin"
                                       else "co_dino_5scale_r50_8xb2_1x_coco.py"))
    m = {k: v for k, v in cfg.model.items()}
    m.pop("type")
    if backbone == "swin":
        m["backbone"].update(embed_dims=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=4)
        m["neck"]["in_channels"] = [32, 64, 128, 256]
    m["query_head"]["num_query"] = 50
    m["query_head"]["transformer"]["encoder"]["num_layers"] = 2
    m["query_head"]["transformer"]["decoder"]["num_layers"] = 2
    m["test_cfg"] = [dict(max_per_img=20)]
    return m


@pytest.mark.parametrize("backbone,hw", [("swin", (76, 100)), ("r50", (96, 128))])
def test_full_codetr_fp32_vs_oracle(backbone, hw):
    """End to end: backbone -> neck -> head, padded second image, product fp32 on GPU vs CPU oracle.

    (Round 1 wrapped this test in a retry because about one fresh-box run in twenty failed its last assertion.  The
    cause was the assertion, not a kernel: detections were compared as sets of tuples ROUNDED to 6 / 2 decimals, and a
    GPU sigmoid one ulp away from the CPU's next to a rounding boundary put the same detection into different cells --
    tools/diag_fp32_flake.py reproduces it and shows every stage bit-stable run to run.  Membership is now by
    tolerance, helpers_model.unmatched_detections, and there is no retry.)"""
    _full_codetr_fp32_vs_oracle(backbone, hw)


def _full_codetr_fp32_vs_oracle(backbone, hw):
    import codetr

    torch.manual_seed(0)
    model = codetr.CoDETR(**_tiny_codetr_cfg(backbone))
    model.init_weights()
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    sd = seeded_params(spec, 77, scale=1.5)
    full = dict(model.state_dict())
    full.update(sd)
    for k in full:
        if k.endswith("running_var"):
            full[k] = torch.rand(full[k].shape) + 0.5
        elif k.endswith("running_mean"):
            full[k] = torch.randn(full[k].shape) * 0.1
    model.load_state_dict(full)
    model = model.to(DEV).eval()
    H, W = hw
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, H, W, generator=g)
    mask = torch.zeros(2, H, W)
    mask[1, :, int(W * 0.8):] = 1
    mask[1, int(H * 0.9):, :] = 1
    cap_o = {}
    kw = dict(num_heads=(1, 2, 4, 8), window_size=4) if backbone == "swin" else {}
    M.codetr_forward(full, img, mask, backbone=backbone, num_query=50, max_per_img=20, capture=cap_o, **kw)
    picks = valid_topk(cap_o["enc_outputs_class"], cap_o["enc_outputs_coord_unact"], 50)
    boxes_o, scores_o, labels_o = M.codetr_forward(full, img, mask, backbone=backbone, num_query=50, max_per_img=20,
                                                   forced_topk=picks, capture=cap_o, **kw)
    cap = {}
    with torch.no_grad():
        boxes, scores, labels = model(img.to(DEV), mask.to(DEV), forced_topk_indices=cap_o["topk_indices"].to(DEV),
                                      capture=cap)
    # fp32 vs fp32: rel-L2 <= 1e-5..1e-4 and the worst element within 1e-3 of the tensor's max
    # (random-weight activations reach 1e3 in the un-normalised ResNet; absolute bounds are meaningless)
    for i, (a, b) in enumerate(zip(cap["backbone_feats"], cap_o["backbone_feats"])):
        assert_close_lowp(a.cpu().numpy(), b.numpy(), 1e-4, 1e-3, f"backbone level {i}")
    for i, (a, b) in enumerate(zip(cap["neck_feats"], cap_o["neck_feats"])):
        assert_close_lowp(a.cpu().numpy(), b.numpy(), 1e-4, 1e-3, f"neck level {i}")
    assert_close_lowp(cap["memory"].cpu().numpy(), cap_o["memory"].numpy(), 2e-4, 2e-3, "encoder memory")
    assert_close_lowp(cap["outputs_classes"].cpu().numpy(), cap_o["outputs_classes"].numpy(), 5e-4, 5e-3, "class logits")
    assert_close_lowp(cap["outputs_coords"].cpu().numpy(), cap_o["outputs_coords"].numpy(), 5e-4, 5e-3, "box coords")
    # the two-stage scores that drive the proposal top-k agree (the selection itself is forced equal:
    # it is unstable under 1e-6 noise on random weights, reference tests/test_export.py:638-655)
    assert_close_lowp(cap["enc_outputs_class"].cpu().numpy(), cap_o["enc_outputs_class"].numpy(), 5e-4, 5e-3, "enc cls")
    # final detections.  With random weights many sigmoid scores tie (saturate), so WHICH candidates win
    # is not comparable across implementations; what is: (1) the score multiset, (2) the decode
    # arithmetic -- re-deriving (boxes, scores, labels) from the product's own logits / coords with the
    # oracle's decode must reproduce the product's outputs, modulo the order inside exact ties.
    assert boxes.shape == (2, 20, 4) and labels.dtype == torch.int64 and scores.shape == (2, 20)
    np.testing.assert_allclose(scores.cpu().numpy(), scores_o.numpy(), rtol=5e-3, atol=1e-4)
    bx, sc, lb = M.decode_detections(cap["outputs_classes"].cpu(), cap["outputs_coords"].cpu(), H, W, 20, 80)
    torch.testing.assert_close(scores.cpu(), sc, rtol=1e-6, atol=1e-7, equal_nan=True)  # NaN: padded proposals, see helpers_model
    for bi in range(2):
        missing = unmatched_detections((boxes[bi], scores[bi], labels[bi]), (bx[bi], sc[bi], lb[bi]))
        assert not missing, f"image {bi}: detections of the oracle decode missing from the product's output: {missing}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_forward_is_bit_stable_with_poisoned_allocator(dtype):
    """The same inputs 8 times with every torch.empty handing out NaNs (helpers_model.poison_allocator): every captured
    stage and the detections are bit-identical run to run -- no race, no atomics, no read of unwritten workspace on the
    inference path (fp32: ATen GEMMs + the native mask-pyramid / MSDA f32 kernels; fp16: the hand-written kernels)."""
    import codetr
    from helpers_model import poison_allocator

    if dtype == torch.float32:
        # the fp32 parity route runs the neck's convolutions through MIOpen, whose algorithm search changes the
        # result of the 3x3 / stride-2 extra level between the first call and later ones (tools/diag_fp32_flake.py:
        # everything up to the backbone output is bit-stable, the neck's last level is the first tensor to move);
        # pin the library to its deterministic choice so that this test is about OUR kernels
        saved = (torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark)
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    try:
        _bit_stable(dtype)
    finally:
        if dtype == torch.float32:
            torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = saved


def _bit_stable(dtype):
    import codetr
    from helpers_model import poison_allocator

    torch.manual_seed(0)
    model = codetr.CoDETR(**_tiny_codetr_cfg("swin"))
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 77, scale=1.5))
    model.load_state_dict(full)
    model = model.to(DEV, dtype).eval()
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 76, 100, generator=g).to(DEV, dtype)
    mask = torch.zeros(2, 76, 100)
    mask[1, :, 80:] = 1
    mask[1, 68:, :] = 1
    mask = mask.to(DEV, dtype)
    stages = ("memory", "enc_outputs_class", "topk_coords_unact", "final_state", "outputs_classes", "outputs_coords")
    first = None
    for it in range(8):
        poison_allocator(DEV)
        cap = {}
        with torch.no_grad():
            out = model(img, mask, capture=cap)
        torch.cuda.synchronize()
        snap = [t.clone() for t in cap["backbone_feats"]] + [t.clone() for t in cap["neck_feats"]] \
            + [cap[k].clone() for k in stages] + [o.clone() for o in out]
        if first is None:
            first = snap
            continue
        for i, (a, b) in enumerate(zip(snap, first)):
            assert torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0)), \
                f"run {it}: tensor {i} differs from the first run"


def test_token_major_path_equals_nchw_path():
    """CoDETR.forward's token-major route (Swin tokens -> linear + native GroupNorm into [B,S,256] -> head) against
    the generic NCHW route of the same model (`route="nchw"`), fp16, padded second image."""
    import codetr
    from codetr import _cabi

    torch.manual_seed(0)
    cfg = _tiny_codetr_cfg("swin")
    cfg["backbone"].update(embed_dims=64, num_heads=[2, 4, 8, 16], window_size=12)
    cfg["neck"]["in_channels"] = [64, 128, 256, 512]
    model = codetr.CoDETR(**cfg)
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 5, scale=1.0))
    model.load_state_dict(full)
    model = model.to(DEV).half().eval()
    g = torch.Generator().manual_seed(2)
    img = torch.randn(2, 3, 150, 200, generator=g).to(DEV).half()
    mask = torch.zeros(2, 150, 200, device=DEV, dtype=torch.float16)
    mask[1, :, 170:] = 1
    cap = {}
    with torch.no_grad():
        model(img, mask, capture=cap, route="nchw")  # generic path, to obtain a NaN-free proposal selection
        picks = valid_topk(cap["enc_outputs_class"].float().cpu(), cap["enc_outputs_coord_unact"].float().cpu(), 50).to(DEV)
        cap = {}
        b0, s0, l0 = model(img, mask, forced_topk_indices=picks, capture=cap, route="nchw")   # NCHW route
        assert cap["route"] == "nchw"
        before = dict(_cabi.CALLS)
        b1, s1, l1 = model(img, mask, forced_topk_indices=picks)                     # token-major route
    assert _cabi.CALLS["groupnorm_tokens"] - before["groupnorm_tokens"] == 5  # 4 mapped levels + the stride-2 extra level
    assert _cabi.CALLS["mask_pyramid"] - before["mask_pyramid"] == 1
    torch.testing.assert_close(s1.float(), s0.float(), rtol=2e-2, atol=2e-3)
    # near-tied scores may swap ranks between the two routes (fp16 noise): match detections as sets --
    # same label, score within 2 %, box within 1 px (200 px wide image)
    matched = 0
    for bi in range(2):
        for k in range(20):
            ok = (l0[bi] == l1[bi, k]) & ((s0[bi] - s1[bi, k]).abs() <= 2e-2 * s1[bi, k].abs() + 2e-3) \
                & ((b0[bi].float() - b1[bi, k].float()).abs().max(-1)[0] <= 1.0)
            matched += int(ok.any())
    assert matched >= 36, matched


def test_decoder_projects_all_layer_values_in_one_gemm():
    """DinoTransformerDecoder._project_values: value_proj(memory) of the six cross-attentions as one N = 1536 GEMM
    (memory read once) == the six per-layer projections, bit for bit, padding mask included."""
    from codetr import _cabi, hip_ops
    from codetr.transformer import DinoTransformerDecoder

    torch.manual_seed(0)
    cfg = dict(type="DetrTransformerDecoderLayer",
               attn_cfgs=[dict(type="MultiheadAttention", embed_dims=256, num_heads=8, dropout=0.0),
                          dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=5, dropout=0.0)],
               feedforward_channels=2048, ffn_dropout=0.0,
               operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))
    dec = DinoTransformerDecoder(return_intermediate=True, transformerlayers=cfg, num_layers=6).to(DEV).half().eval()
    for layer in dec.layers:  # distinct, non-trivial projections
        torch.nn.init.normal_(layer.attentions[1].value_proj.weight, std=0.05)
        torch.nn.init.normal_(layer.attentions[1].value_proj.bias, std=0.5)
    B, S = 2, 20000
    g = torch.Generator(device=DEV).manual_seed(3)
    mem = torch.randn(B, S, 256, device=DEV, generator=g).half()
    mask = torch.rand(B, S, device=DEV, generator=g) < 0.2
    before = _cabi.CALLS["linear"]
    with torch.no_grad():
        vs = dec._project_values(mem, mask)
    assert vs is not None and len(vs) == 6 and _cabi.CALLS["linear"] == before + 1
    for lid, layer in enumerate(dec.layers):
        vp = layer.attentions[1].value_proj
        ref = hip_ops.linear(mem, vp.weight, vp.bias, row_mask=mask)
        assert vs[lid].shape == ref.shape and vs[lid].is_contiguous() and torch.equal(vs[lid], ref)
    with torch.no_grad():
        assert dec._project_values(mem[:, :100], mask[:, :100]) is None  # short memories: per-layer projections
