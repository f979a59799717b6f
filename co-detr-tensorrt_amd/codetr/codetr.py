"""``CoDETR`` and ``build_CoDETR`` -- host-side mirror of reference codetr/codetr.py:15-170.

Same call contract: ``build_CoDETR(model_file, weights_file=None, device="cuda")`` returns the
model alone when no weights are given and ``(model, dataset_meta)`` otherwise (reference :161-170);
``CoDETR.forward(batch_inputs[bs,3,H,W], img_masks[bs,H,W]) -> (boxes[bs,300,4], scores[bs,300],
labels[bs,300])``.

Differences by design: config files are read by ``codetr.config`` (no mmengine), the neck is
this package's ``ChannelMapper`` (no mmdet registry), checkpoints by ``codetr.checkpoint``; a
``ResNet`` backbone is accepted in addition to ``SwinTransformer`` (BASELINE config 1 names an R50
model that the reference's constructor assertion, :51, cannot build).
"""
import warnings
from typing import Optional, Tuple

import torch.nn as nn
from torch import Tensor

from .co_dino_head import CoDINOHead
from .config import Config
from .neck import ChannelMapper
from .resnet import ResNet
from .swin import SwinTransformer

_BACKBONES = {"SwinTransformer": SwinTransformer, "ResNet": ResNet}

COCO_CLASSES = (
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant",
    "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard",
    "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle",
    "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange", "broccoli",
    "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet",
    "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink",
    "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush",
)


class CoDETR(nn.Module):
    def __init__(self, backbone, neck=None, query_head=None, train_cfg=[None, None], test_cfg=[None, None], **kwargs):
        """`kwargs` swallows the training-only entries of the config's model dict (aux heads,
        data_preprocessor, eval_module ...), exactly like the reference (:41)."""
        super().__init__()
        backbone = dict(backbone)
        kind = backbone.pop("type")
        if kind not in _BACKBONES:
            raise AssertionError(f"backbone type must be one of {sorted(_BACKBONES)}, got {kind}")
        self.backbone = _BACKBONES[kind](**backbone)
        if neck is not None:
            neck = dict(neck)
            if neck.pop("type") != "ChannelMapper":
                raise AssertionError("neck type must be ChannelMapper")
            self.neck = ChannelMapper(**neck)
        if query_head is None:
            raise AssertionError("query_head is required")
        query_head = dict(query_head)
        head_idx = 0
        query_head.update(
            train_cfg=train_cfg[head_idx] if (train_cfg is not None and train_cfg[head_idx] is not None) else None)
        query_head.update(test_cfg=test_cfg[head_idx])
        if query_head.pop("type") != "CoDINOHead":
            raise AssertionError("query_head type must be CoDINOHead")
        self.query_head = CoDINOHead(**query_head)
        self.query_head.init_weights()

    def init_weights(self):
        """Seeded default init of every sub-module (no checkpoint): what ``build_CoDETR(cfg, None)``
        gives in the reference plus backbone/neck init, so random-weight benchmarks are well-conditioned."""
        self.backbone.init_weights()
        if hasattr(self, "neck"):
            self.neck.init_weights()
        self.query_head.init_weights()

    def forward(self, batch_inputs: Tensor, img_masks: Tensor, forced_topk_indices=None,
                capture=None, route=None) -> Tuple[Tensor, Tensor, Tensor]:
        """Reference signature (:66-90) plus three test hooks that do not exist there: `forced_topk_indices` (see
        CoDinoTransformer.forward), `capture` (a dict that receives intermediates; it never changes which kernels
        run) and `route` ("nchw" forces the generic NCHW backbone -> neck -> head route that ResNet / fp32 models
        take; default: token-major wherever the backbone and neck support it)."""
        if route not in (None, "tokens", "nchw"):
            raise ValueError(f"route must be None, 'tokens' or 'nchw', got {route!r}")
        if (route != "nchw" and hasattr(self.backbone, "forward_tokens") and hasattr(self, "neck")
                and batch_inputs.is_cuda):
            # token-major path: backbone stage outputs stay [B, HW, C], the neck's 1x1 convs are linears over
            # tokens, GroupNorm writes straight into the encoder's [B, S, 256] input -- no NCHW round trip
            tokens = self.backbone.forward_tokens(batch_inputs)
            if self.neck.tokens_supported(tokens):
                flat, shapes = self.neck.forward_tokens(tokens)
                if capture is not None:
                    # NCHW *views* of the token-major maps (no copies, nothing recomputed): same keys as the NCHW route
                    capture["route"] = "tokens"
                    capture["backbone_feats"] = [t.view(-1, *hw, t.shape[-1]).permute(0, 3, 1, 2) for t, hw in tokens]
                    starts = [0]
                    for h, w in shapes:
                        starts.append(starts[-1] + h * w)
                    capture["neck_feats"] = [flat[:, a:b].view(-1, h, w, flat.shape[-1]).permute(0, 3, 1, 2)
                                             for (h, w), a, b in zip(shapes, starts[:-1], starts[1:])]
                return self.query_head.forward_flat(flat, shapes, img_masks, forced_topk_indices=forced_topk_indices,
                                                    capture=capture)
            elif route == "tokens":
                raise RuntimeError("route='tokens': the neck does not take token-major maps for this model / dtype")
            feats = [t.view(-1, *hw, t.shape[-1]).permute(0, 3, 1, 2).contiguous() for t, hw in tokens]
        else:
            if route == "tokens":
                raise RuntimeError("route='tokens' needs a token-major backbone, a neck and GPU tensors")
            feats = self.backbone(batch_inputs)
        if capture is not None:
            capture["route"] = "nchw"
            capture["backbone_feats"] = feats
        feats = self.neck(feats)
        if capture is not None:
            capture["neck_feats"] = feats
        return self.query_head(feats, img_masks, forced_topk_indices=forced_topk_indices, capture=capture)


def get_dataset_meta(checkpoint):
    """class names from the checkpoint's meta, COCO by default (reference :93-126)."""
    meta = checkpoint.get("meta", {}) if isinstance(checkpoint, dict) else {}
    if "dataset_meta" in meta:
        dataset_meta = {k.lower(): v for k, v in meta["dataset_meta"].items()}
    elif "CLASSES" in meta:
        dataset_meta = {"classes": meta["CLASSES"]}
    else:
        warnings.warn("dataset_meta or class names are not saved in the checkpoint's meta data, "
                      "use COCO classes by default.")
        dataset_meta = {"classes": COCO_CLASSES}
    dataset_meta["palette"] = "coco"
    return dataset_meta


def build_CoDETR(model_file: str, weights_file: Optional[str] = None, device: str = "cuda"):
    cfg = Config.fromfile(model_file)
    if cfg.model.get("pretrained") is not None:
        del cfg.model["pretrained"]
    if isinstance(cfg.model.get("backbone"), dict):
        cfg.model.backbone.pop("init_cfg", None)  # never fetch backbone weights from the network
    model_cfg = {k: v for k, v in cfg.model.items()}
    if model_cfg.pop("type") != "CoDETR":
        raise AssertionError("model type must be CoDETR")
    model = CoDETR(**model_cfg)
    model.cfg = cfg
    if weights_file is None:
        model.to(device)
        model.eval()
        return model
    from .checkpoint import load_checkpoint, load_checkpoint_to_model

    checkpoint = load_checkpoint(weights_file, map_location="cpu")
    load_checkpoint_to_model(model, checkpoint)
    model.to(device)
    model.eval()
    return model, get_dataset_meta(checkpoint)
