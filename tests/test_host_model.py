"""CPU: host logic of the product package -- config loading, model construction, checkpoint-key
parity with the reference, loader behaviour.  No compute (the product computes on the GPU only)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from helpers_model import unpack_param_spec

CFG_DIR = os.path.join(ROOT, "co-detr-tensorrt_amd", "configs")
SWIN_CFG = os.path.join(CFG_DIR, "co_dino_5scale_swin_l_16xb1_16e_o365tococo.py")
R50_CFG = os.path.join(CFG_DIR, "co_dino_5scale_r50_8xb2_1x_coco.py")


def test_config_base_chain_delete_and_base_refs():
    from codetr.config import Config

    cfg = Config.fromfile(SWIN_CFG)
    m = cfg.model
    assert m.type == "CoDETR"
    assert m.backbone.type == "SwinTransformer" and "depth" not in m.backbone          # _delete_=True
    assert m.backbone.depths == [2, 2, 18, 2] and m.backbone.num_heads == [6, 12, 24, 48]
    assert m.neck.in_channels == [192, 384, 768, 1536] and m.neck.num_outs == 5       # deep merge
    assert m.query_head.transformer.encoder.with_cp == 6                               # 3-level deep merge
    assert m.query_head.transformer.encoder.num_layers == 6
    assert m.query_head.positional_encoding.temperature == 20
    assert m.use_lsj is False                                                          # from the r50 child
    assert cfg.dataset_type == "CocoDataset"                                           # from the mmdet:: base
    r50 = Config.fromfile(R50_CFG)
    assert r50.model.backbone.type == "ResNet"
    assert r50.test_pipeline[0]["backend_args"] is None                                # `_base_.backend_args`


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree not present")
def test_reference_config_files_load_and_agree():
    """The reference's own config files go through the same loader; their inference-relevant model
    section equals ours."""
    from codetr.config import Config

    ref = Config.fromfile("/root/reference/configs/co_dino_5scale_swin_l_16xb1_16e_o365tococo.py")
    ours = Config.fromfile(SWIN_CFG)
    rb = dict(ref.model.backbone)
    rb.pop("init_cfg")
    assert rb == dict(ours.model.backbone)
    assert ref.model.neck == ours.model.neck
    for k in ("num_query", "num_classes", "as_two_stage", "transformer", "positional_encoding"):
        assert ref.model.query_head[k] == ours.model.query_head[k], k
    assert ref.model.test_cfg[0] == ours.model.test_cfg[0]
    assert ref.model.data_preprocessor.mean == ours.model.data_preprocessor.mean


@pytest.fixture(scope="module")
def swin_l_model():
    import codetr

    return codetr.build_CoDETR(SWIN_CFG, None, "cpu")


def test_state_dict_matches_reference_keys_and_shapes(swin_l_model):
    g = np.load(os.path.join(GOLDEN, "state_dict_keys.npz"))
    ref = dict(unpack_param_spec(g))
    own = {k: tuple(v.shape) for k, v in swin_l_model.state_dict().items()}
    for k, shp in ref.items():
        assert k in own, f"reference key {k} missing"
        assert own[k] == shp, (k, own[k], shp)
    # backbone, transformer AND the head's own parameters (class / box branches, `downsample`) are pinned by the
    # reference-built key list: nothing of ours under those prefixes may be missing from it either
    extra = [k for k in own if k.startswith(("backbone.", "query_head.")) and k not in ref]
    assert not extra, extra
    assert sum(1 for k in ref if k.startswith("query_head.cls_branches.")) == 14        # 7 x (weight, bias)
    assert sum(1 for k in ref if k.startswith("query_head.reg_branches.")) == 42        # 7 x 3 Linears
    assert ref["query_head.downsample.0.weight"] == (256, 256, 3, 3)
    blk = swin_l_model.backbone.stages[0].blocks[0].attn.w_msa
    assert np.array_equal(blk.relative_position_index.numpy(), g["rel_index"])
    # the neck's keys follow mmdet's ConvModule names (mmdet is absent: no reference-built list) -- spot-check the
    # contract of SURVEY 8(f)-2
    for k in ("neck.convs.0.conv.weight", "neck.convs.3.gn.bias", "neck.extra_convs.0.conv.weight",
              "neck.extra_convs.0.gn.weight", "query_head.cls_branches.6.weight", "query_head.reg_branches.6.4.bias",
              "query_head.reg_branches.0.0.weight", "query_head.downsample.0.weight", "query_head.downsample.1.bias"):
        assert k in own, k
    assert "neck.convs.0.conv.bias" not in own  # conv followed by GN has no bias
    assert own["query_head.cls_branches.0.weight"] == (80, 256)


def test_model_sizes(swin_l_model):
    nb = sum(p.numel() for p in swin_l_model.backbone.parameters())
    nt = sum(p.numel() for p in swin_l_model.query_head.transformer.parameters())
    assert abs(nb / 1e6 - 195.2) < 0.05 and abs(nt / 1e6 - 18.0) < 0.05  # SURVEY section 8 [probe]


def test_build_returns_model_only_without_weights_and_pair_with(tmp_path, swin_l_model):
    import codetr

    assert isinstance(swin_l_model, codetr.CoDETR) and not swin_l_model.training
    sd = {k: v for k, v in swin_l_model.state_dict().items()}
    sd["rpn_head.rpn_conv.weight"] = torch.zeros(1)          # training-only keys are ignored
    sd["query_head.label_embedding.weight"] = torch.zeros(1)
    path = tmp_path / "ckpt.pth"
    torch.save({"state_dict": sd, "meta": {"dataset_meta": {"CLASSES": ("a", "b")}}}, path)
    with pytest.warns(UserWarning, match="unexpected keys"):
        model, meta = codetr.build_CoDETR(SWIN_CFG, str(path), "cpu")
    assert meta["classes"] == ("a", "b") and meta["palette"] == "coco"
    assert torch.equal(model.state_dict()["backbone.norm3.weight"], sd["backbone.norm3.weight"])


def test_r50_config_builds():
    import codetr

    m = codetr.build_CoDETR(R50_CFG, None, "cpu")
    assert type(m.backbone).__name__ == "ResNet"
    assert m.neck.convs[3].conv.weight.shape == (256, 2048, 1, 1)
    assert "backbone.layer3.5.conv2.weight" in m.state_dict()


def test_msda_module_contract():
    from codetr.multi_scale_deformable_attention import MultiScaleDeformableAttention

    with pytest.raises(ValueError, match="divisible"):
        MultiScaleDeformableAttention(embed_dims=100, num_heads=8)
    m = MultiScaleDeformableAttention(embed_dims=256, num_levels=5)
    assert m.sampling_offsets.weight.shape == (8 * 5 * 4 * 2, 256) and m.attention_weights.weight.shape == (160, 256)
    # grid-pattern bias: head 0 points along +x with magnitudes 1..4
    b = m.sampling_offsets.bias.view(8, 5, 4, 2)
    torch.testing.assert_close(b[0, :, :, 0], torch.arange(1.0, 5.0).expand(5, 4))
    assert float(b[0, :, :, 1].abs().max()) < 1e-6
    # the product never computes on the CPU
    q = torch.zeros(3, 1, 256)
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(q, reference_points=torch.zeros(1, 3, 5, 2), spatial_shapes=torch.ones(5, 2, dtype=torch.long),
          level_start_index=torch.zeros(5, dtype=torch.long))


def test_swin_converter_known_answer():
    from codetr.swin import swin_converter

    C = 2
    red = torch.arange(3 * 4 * C, dtype=torch.float32).view(3, 4 * C)  # official order: [x0 | x1 | x2 | x3] blocks of C
    norm = torch.arange(4 * C, dtype=torch.float32)
    out = swin_converter({"layers.0.downsample.reduction.weight": red, "layers.0.downsample.norm.weight": norm,
                          "layers.1.blocks.0.attn.qkv.weight": torch.zeros(1), "layers.1.blocks.0.mlp.fc1.bias": torch.zeros(1),
                          "patch_embed.proj.weight": torch.zeros(1), "head.weight": torch.zeros(1)})
    assert set(out) == {"backbone.stages.0.downsample.reduction.weight", "backbone.stages.0.downsample.norm.weight",
                        "backbone.stages.1.blocks.0.attn.w_msa.qkv.weight", "backbone.stages.1.blocks.0.ffn.layers.0.0.bias",
                        "backbone.patch_embed.projection.weight"}
    # unfold order is channel-major: new[c*4 + j] = old[perm[j]*C + c] with perm = (0, 2, 1, 3)
    exp = torch.tensor([norm[p * C + c] for c in range(C) for p in (0, 2, 1, 3)])
    torch.testing.assert_close(out["backbone.stages.0.downsample.norm.weight"], exp)
    torch.testing.assert_close(out["backbone.stages.0.downsample.reduction.weight"][1], red[1][exp.long()])
