"""Import the reference's Python (read-only tree at /root/reference) in THIS build container.

Test infrastructure only.  Used by ``make_golden.py`` (fixture generation) and by the
``-m "not gpu"`` tests that cross-check the oracle against the live reference when the
reference tree is present.  Nothing here travels to the GPU box: the reference does not
exist there, and every caller skips when ``REFERENCE_ROOT`` is missing.

The reference cannot be imported as a normal package: ``codetr/__init__.py`` (reference
codetr/__init__.py:8-19) demands two CUDA ``.so`` files, ``codetr/ops.py`` imports
tensorrt / torch_tensorrt (ops.py:4-10) and the model modules import mmengine / mmcv /
mmdet, none of which exist in this image.  The stand-ins below are trivial wrappers over
``torch.nn`` that give those third-party names their documented behaviour (mmcv 2.x /
mmengine 0.x / mmdet v3.3.0 semantics); they contain no reference code.
"""
import copy
import importlib
import math
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("CODETR_REFERENCE_ROOT", "/root/reference")

_LIB = None  # keeps the torch.library.Library objects alive


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "codetr", "ops.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _ConfigDict(dict):
    """dict with attribute access; missing attributes raise AttributeError (deepcopy-safe)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)
        self._is_init = False

    def init_weights(self):
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()


class _ModuleList(nn.ModuleList, _BaseModule):
    def __init__(self, modules=None, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class _Sequential(nn.Sequential, _BaseModule):
    def __init__(self, *args, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _xavier_init(module, gain=1, bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _trunc_normal_init(module, mean=0, std=1, a=-2, b=2, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.trunc_normal_(module.weight, mean, std, a, b)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cfg.pop("requires_grad", None)
    if t == "LN":
        return "ln" + str(postfix), nn.LayerNorm(num_features, **cfg)
    if t == "GN":
        return "gn" + str(postfix), nn.GroupNorm(num_channels=num_features, **cfg)
    raise NotImplementedError(t)


def _build_activation_layer(cfg):
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t == "ReLU":
        return nn.ReLU(**cfg)
    if t == "GELU":
        return nn.GELU()
    raise NotImplementedError(t)


def _build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get("type", "Conv2d") in ("Conv2d", "Conv")
    return nn.Conv2d(*args, **kwargs)


def _build_dropout(cfg):
    # eval-mode only: DropPath and Dropout are both the identity at inference
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t == "DropPath":
        return nn.Identity()
    if t == "Dropout":
        return nn.Dropout(cfg.get("drop_prob", 0.0))
    raise NotImplementedError(t)


class _LayerScale(nn.Module):
    def __init__(self, dim, inplace=False, data_format="channels_last", scale=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim) * scale)

    def forward(self, x):
        return x * self.weight


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _Logger:
    @classmethod
    def get_current_instance(cls):
        return cls()

    def warning(self, *a, **k):
        pass

    info = warning


class _DINOHeadStandIn(nn.Module):
    """Stand-in for mmdet v3.3.0 ``DINOHead`` (-> DeformableDETRHead -> DETRHead constructor chain), reduced to what
    the reference's ``CoDINOHead`` reads at inference: ``num_classes``, ``cls_out_channels`` (= num_classes when the
    classification loss uses sigmoid, else + 1: DETRHead.__init__), ``num_reg_fcs`` (2), ``as_two_stage``,
    ``test_cfg``, ``loss_cls`` (only ``use_sigmoid`` is read), then ``self._init_layers()`` -- which is the REFERENCE's
    own method (co_dino_head.py:74-118).  Contains no reference code."""

    def __init__(self, num_classes=80, embed_dims=256, num_reg_fcs=2, sync_cls_avg_factor=False, as_two_stage=False,
                 loss_cls=None, loss_bbox=None, loss_iou=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.embed_dims = embed_dims
        self.num_reg_fcs = num_reg_fcs
        self.as_two_stage = as_two_stage
        self.train_cfg = train_cfg
        self.test_cfg = _ConfigDict(test_cfg if test_cfg is not None else dict(max_per_img=100))
        self.loss_cls = _ConfigDict(loss_cls if loss_cls is not None else dict(type="CrossEntropyLoss", use_sigmoid=False))
        self.cls_out_channels = num_classes if self.loss_cls.get("use_sigmoid", False) else num_classes + 1
        self._init_layers()


def _bbox_cxcywh_to_xyxy(bbox):
    """mmdet.structures.bbox.bbox_cxcywh_to_xyxy (published formula): (cx, cy, w, h) -> (x1, y1, x2, y2)"""
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(cx - 0.5 * w), (cy - 0.5 * h), (cx + 0.5 * w), (cy + 0.5 * h)], dim=-1)


def install_shims():
    """Put the third-party stand-ins and the op schemas in place (idempotent)."""
    global _LIB
    if "codetr" in sys.modules and getattr(sys.modules["codetr"], "_is_reference_shim", False):
        return
    if "codetr" in sys.modules:
        raise RuntimeError(
            "a different 'codetr' package is already imported in this process; the reference "
            "must be imported in its own process (see tests/golden/make_golden.py)"
        )
    # --- tensorrt / torch_tensorrt (reference ops.py:4-10, decorator at :189) ---
    _mod("tensorrt", ITensor=object)
    tt = _mod("torch_tensorrt")
    tt_d = _mod("torch_tensorrt.dynamo")
    tt_c = _mod(
        "torch_tensorrt.dynamo.conversion",
        ConversionContext=object,
        dynamo_tensorrt_converter=lambda *a, **k: (lambda f: f),
    )
    tt_u = _mod("torch_tensorrt.dynamo.conversion.converter_utils", get_trt_tensor=None)
    tt.dynamo = tt_d
    tt_d.conversion = tt_c
    tt_c.converter_utils = tt_u

    # --- op schemas, exactly the strings of deformable_attention_torch.cpp:17-23 ---
    _LIB = torch.library.Library("codetr", "DEF")
    _LIB.define(
        "multi_scale_deformable_attention(Tensor value, Tensor spatial_shapes, "
        "Tensor level_start_index, Tensor sampling_loc, Tensor attn_weight, "
        "int im2col_step) -> Tensor"
    )
    _LIB.define(
        "multi_scale_deformable_attention_backward(Tensor value, Tensor "
        "spatial_shapes, Tensor level_start_index, Tensor sampling_loc, Tensor "
        "attn_weight, Tensor grad_output, Tensor(a!) grad_value, Tensor(b!) "
        "grad_sampling_loc, Tensor(c!) grad_attn_weight, int im2col_step) -> ()"
    )

    # --- mmengine ---
    me = _mod("mmengine", ConfigDict=_ConfigDict)
    me.config = _mod("mmengine.config", ConfigDict=_ConfigDict, Config=object)
    me.model = _mod(
        "mmengine.model",
        BaseModule=_BaseModule,
        ModuleList=_ModuleList,
        Sequential=_Sequential,
        constant_init=_constant_init,
        xavier_init=_xavier_init,
    )
    me.model.weight_init = _mod(
        "mmengine.model.weight_init",
        xavier_init=_xavier_init,
        constant_init=_constant_init,
        trunc_normal_=nn.init.trunc_normal_,
        trunc_normal_init=_trunc_normal_init,
    )
    me.utils = _mod("mmengine.utils", to_2tuple=_to_2tuple)
    me.logging = _mod("mmengine.logging", MMLogger=_Logger)
    me.runner = _mod("mmengine.runner")
    me.runner.checkpoint = _mod(
        "mmengine.runner.checkpoint", CheckpointLoader=object, _load_checkpoint=None, _load_checkpoint_to_model=None
    )

    # --- mmcv ---
    mc = _mod("mmcv")
    mc.cnn = _mod(
        "mmcv.cnn",
        Linear=nn.Linear,
        build_norm_layer=_build_norm_layer,
        build_activation_layer=_build_activation_layer,
        build_conv_layer=_build_conv_layer,
    )
    mc.cnn.bricks = _mod("mmcv.cnn.bricks")
    mc.cnn.bricks.drop = _mod("mmcv.cnn.bricks.drop", build_dropout=_build_dropout)
    mc.cnn.bricks.transformer = _mod("mmcv.cnn.bricks.transformer", build_dropout=_build_dropout)
    mc.cnn.bricks.scale = _mod("mmcv.cnn.bricks.scale", LayerScale=_LayerScale)

    # --- mmdet (only the names the importable modules touch) ---
    md = _mod("mmdet")
    md.utils = _mod("mmdet.utils", OptMultiConfig=object)
    md.models = _mod("mmdet.models", DINOHead=_DINOHeadStandIn)
    md.models.layers = _mod("mmdet.models.layers")
    md.structures = _mod("mmdet.structures")
    md.structures.bbox = _mod("mmdet.structures.bbox", bbox_cxcywh_to_xyxy=_bbox_cxcywh_to_xyxy)

    # --- the reference package itself, bypassing its __init__ ---
    pkg = types.ModuleType("codetr")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "codetr")]
    pkg._is_reference_shim = True
    sys.modules["codetr"] = pkg
    # swin.py:13 imports mmdet's PatchEmbed/PatchMerging; the reference carries equivalent
    # source in transformer_mmcv.py:100-316, bind those.
    tm = importlib.import_module("codetr.transformer_mmcv")
    md.models.layers.PatchEmbed = tm.PatchEmbed
    md.models.layers.PatchMerging = tm.PatchMerging


def ref(module: str):
    """``ref('ops')`` -> the reference's ``codetr.ops`` module object."""
    install_shims()
    return importlib.import_module("codetr." + module)


# ----------------------------------------------------------------------------------------
# model configs used for the golden captures (values from reference configs, lsj:58-106,
# swin:10-27; written out here as plain dicts because mmengine's Config is not available)
# ----------------------------------------------------------------------------------------
def transformer_cfg(num_levels=5, num_layers=(6, 6), num_query=900, ffn=2048):
    enc_layers, dec_layers = num_layers
    return dict(
        with_coord_feat=False,
        num_co_heads=2,
        num_feature_levels=num_levels,
        as_two_stage=True,
        two_stage_num_proposals=num_query,
        encoder=dict(
            type="DetrTransformerEncoder",
            num_layers=enc_layers,
            with_cp=4,
            transformerlayers=dict(
                type="BaseTransformerLayer",
                attn_cfgs=dict(
                    type="MultiScaleDeformableAttention", embed_dims=256, num_levels=num_levels, dropout=0.0
                ),
                feedforward_channels=ffn,
                ffn_dropout=0.0,
                operation_order=("self_attn", "norm", "ffn", "norm"),
            ),
        ),
        decoder=dict(
            type="DinoTransformerDecoder",
            num_layers=dec_layers,
            return_intermediate=True,
            transformerlayers=dict(
                type="DetrTransformerDecoderLayer",
                attn_cfgs=[
                    dict(type="MultiheadAttention", embed_dims=256, num_heads=8, dropout=0.0),
                    dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=num_levels, dropout=0.0),
                ],
                feedforward_channels=ffn,
                ffn_dropout=0.0,
                operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"),
            ),
        ),
    )


def make_branches(num_pred=7, embed=256, num_classes=80):
    """cls/reg branches as CoDINOHead._init_layers builds them (co_dino_head.py:95-113)."""
    cls = nn.Linear(embed, num_classes)
    reg = nn.Sequential(nn.Linear(embed, embed), nn.ReLU(), nn.Linear(embed, embed), nn.ReLU(), nn.Linear(embed, 4))
    cls_branches = nn.ModuleList([copy.deepcopy(cls) for _ in range(num_pred)])
    reg_branches = nn.ModuleList([copy.deepcopy(reg) for _ in range(num_pred)])
    return cls_branches, reg_branches


def randomize_(module: nn.Module, seed: int, scale: float = 1.0, prefix: str = ""):
    """Seeded, non-degenerate parameters so every code path (bias, LN affine, rel-pos table)
    contributes to the captured outputs (default mmdet init leaves many of them at 0/1).
    Returns the ordered [(prefixed name, shape)] spec so a fixture can store it instead of the
    weights; tests rebuild the same tensors with tests/helpers_model.seeded_params."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from helpers_model import seeded_params

    spec = [(name, tuple(p.shape)) for name, p in module.named_parameters()]
    vals = seeded_params(spec, seed, scale)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(vals[name])
    return [(prefix + n, s) for n, s in spec]
