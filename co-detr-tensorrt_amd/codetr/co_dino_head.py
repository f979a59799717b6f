"""``CoDINOHead`` -- host-side mirror of reference codetr/co_dino_head.py:17-210.

The reference derives from mmdet's ``DINOHead`` (third party); the parts of that constructor
chain that matter at inference are restated here: ``num_classes``, ``cls_out_channels``
(= num_classes for sigmoid losses), ``num_reg_fcs = 2``, ``as_two_stage``, ``test_cfg`` and the
bias initialisation of the prediction branches.  Training-only arguments (losses' weights,
``dn_cfg``, assigners ...) are accepted and ignored, as the reference ignores them at inference.
"""
import copy
import math
from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops
from .positional_encoding import SinePositionalEncoding
from .transformer import CoDinoTransformer, run_mlp


def bbox_cxcywh_to_xyxy(bbox):
    """(cx, cy, w, h) -> (x1, y1, x2, y2)  (mmdet.structures.bbox.bbox_cxcywh_to_xyxy)."""
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


class CoDINOHead(nn.Module):
    def __init__(self, *args, num_query=900, transformer=None, in_channels=2048, max_pos_coords=300, dn_cfg=None,
                 use_zero_padding=False,
                 positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                 num_classes=80, embed_dims=256, num_reg_fcs=2, as_two_stage=False, sync_cls_avg_factor=False,
                 loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False), loss_bbox=None, loss_iou=None,
                 train_cfg=None, test_cfg=dict(max_per_img=100), init_cfg=None, **kwargs):
        super().__init__()
        transformer = copy.deepcopy(dict(transformer))
        if "two_stage_num_proposals" in transformer:
            if transformer["two_stage_num_proposals"] != num_query:
                raise AssertionError("two_stage_num_proposals must be equal to num_query for DINO")
        else:
            transformer["two_stage_num_proposals"] = num_query
        transformer["as_two_stage"] = True
        self.num_query = num_query
        self.num_classes = num_classes
        self.num_reg_fcs = num_reg_fcs
        self.as_two_stage = as_two_stage
        self.train_cfg = train_cfg
        self.test_cfg = dict(test_cfg) if test_cfg is not None else {}
        self.use_sigmoid = bool(dict(loss_cls).get("use_sigmoid", False))
        self.cls_out_channels = num_classes if self.use_sigmoid else num_classes + 1
        if transformer.pop("type") != "CoDinoTransformer":
            raise AssertionError("transformer must be CoDinoTransformer")
        self.transformer = CoDinoTransformer(**transformer)
        self.embed_dims = self.transformer.embed_dims
        pe = dict(positional_encoding)
        if pe.pop("type") != "SinePositionalEncoding":
            raise AssertionError("positional_encoding must be SinePositionalEncoding")
        self.positional_encoding = SinePositionalEncoding(**pe)
        if self.positional_encoding.num_feats * 2 != self.embed_dims:
            raise AssertionError(
                f"embed_dims should be exactly 2 times of num_feats. Found {self.embed_dims} and "
                f"{self.positional_encoding.num_feats}.")
        self._init_layers()
        self._scale_cache = {}
        self.max_per_img = self.test_cfg.get("max_per_img", self.num_query)

    def _init_layers(self):
        C = self.embed_dims
        fc_cls = nn.Linear(C, self.cls_out_channels)
        reg = []
        for _ in range(self.num_reg_fcs):
            reg += [nn.Linear(C, C), nn.ReLU()]
        reg.append(nn.Linear(C, 4))
        reg = nn.Sequential(*reg)
        num_pred = self.transformer.decoder.num_layers + 1 if self.as_two_stage else self.transformer.decoder.num_layers
        self.cls_branches = nn.ModuleList(copy.deepcopy(fc_cls) for _ in range(num_pred))
        self.reg_branches = nn.ModuleList(copy.deepcopy(reg) for _ in range(num_pred))
        # never used in forward; exists so that the published checkpoint's keys load (reference :115-118)
        self.downsample = nn.Sequential(nn.Conv2d(C, C, kernel_size=3, stride=2, padding=1), nn.GroupNorm(32, C))

    def init_weights(self):
        """mmdet DeformableDETRHead.init_weights semantics + transformer init."""
        self.transformer.init_weights()
        if self.use_sigmoid:
            prior = -math.log((1 - 0.01) / 0.01)
            for m in self.cls_branches:
                nn.init.constant_(m.bias, prior)
        for m in self.reg_branches:
            nn.init.zeros_(m[-1].weight)
            nn.init.zeros_(m[-1].bias)
        nn.init.constant_(self.reg_branches[0][-1].bias.data[2:], -2.0)
        if self.as_two_stage:
            for m in self.reg_branches:
                nn.init.constant_(m[-1].bias.data[2:], 0.0)

    def forward(self, mlvl_feats: List[torch.Tensor], img_masks: torch.Tensor, forced_topk_indices=None,
                capture=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """mlvl_feats: list of [B,C,h,w]; img_masks [B,H,W] (0 = image, 1 = padding).
        Returns boxes [B,K,4] (x1,y1,x2,y2 in input pixels), scores [B,K], labels [B,K] int64; K = max_per_img."""
        shapes = [tuple(f.shape[-2:]) for f in mlvl_feats]
        feat = torch.cat([f.flatten(2).transpose(1, 2) for f in mlvl_feats], 1)
        return self.forward_flat(feat, shapes, img_masks, forced_topk_indices, capture)

    def forward_flat(self, feat, shapes, img_masks, forced_topk_indices=None, capture=None):
        """`forward` on the flattened multi-level map feat [B, S, C] (+ the level shapes): what the token-major
        backbone/neck path hands over, no NCHW round trip."""
        Himg, Wimg = img_masks.shape[-2:]
        masks, pos = [], []
        pe = self.positional_encoding
        native_pos = feat.is_cuda and feat.dtype in (torch.float16, torch.bfloat16) and pe.num_feats % 8 == 0
        pos_flat = feat.new_empty(feat.shape[0], feat.shape[1], 2 * pe.num_feats) if native_pos else None
        B = feat.shape[0]
        mask_flat = valid_counts = None
        if feat.is_cuda:
            # one launch: level masks (already concatenated), running valid counts for the encoding, valid-ratio counts
            mask_flat, ycum, xcum, valid_counts = hip_ops.mask_pyramid(img_masks, shapes)
        else:
            m4 = img_masks.unsqueeze(1)
        start = 0
        for lvl, hw in enumerate(shapes):
            n = hw[0] * hw[1]
            if mask_flat is not None:
                m = mask_flat[:, start:start + n].view(B, hw[0], hw[1])
                cums = hip_ops.level_cums(ycum, xcum, B, start, hw)
            else:
                m = F.interpolate(m4, size=tuple(hw)).to(torch.bool).squeeze(1)  # nearest
                cums = None
            masks.append(m)
            if native_pos:
                # one kernel per level: encoding + level embedding straight into lvl_pos_embed[:, start:start+HW]
                hip_ops.sine_pos_tokens_into(m, pos_flat, start, self.transformer.level_embeds[lvl], pe.num_feats,
                                             pe.temperature, pe.scale, pe.eps, pe.offset, pe.normalize, cums=cums)
            else:
                pos.append(pe.forward_tokens(m, dtype=feat.dtype))
            start += n
        if native_pos and mask_flat is not None and pe.num_feats == 128:
            # the encoder's projections can re-generate this tensor's rows from the running sums instead of reading them
            # (codetr_encoder_projections_posgen_*: bit-identical operand); the recipe rides on the tensor object
            le = self.transformer.level_embeds
            le = le if le.dtype == pos_flat.dtype else le.to(pos_flat.dtype)
            st_, cums_ = 0, []
            for hw in shapes:
                cums_.append(hip_ops.level_cums(ycum, xcum, B, st_, hw))
                st_ += hw[0] * hw[1]
            pos_flat._codetr_posgen = {"B": B, "S": feat.shape[1], "cums": cums_, "shapes": [tuple(int(v) for v in hw) for hw in shapes],
                                       "level_embed": le.detach().contiguous(), "temperature": pe.temperature, "scale": pe.scale,
                                       "eps": pe.eps, "offset": pe.offset, "normalize": pe.normalize}
        state, refs = self.transformer.forward_flat(feat, shapes, masks, pos, reg_branches=self.reg_branches,
                                                    cls_branches=self.cls_branches if self.as_two_stage else None,
                                                    forced_topk_indices=forced_topk_indices, capture=capture,
                                                    lvl_pos_embed_flat=pos_flat, mask_flat=mask_flat,
                                                    valid_counts=valid_counts)
        lvl = len(self.transformer.decoder.layers) - 1
        cls_head = self.cls_branches[lvl]
        cls = hip_ops.linear(state, cls_head.weight, cls_head.bias)  # [B,Nq,classes]
        if refs.shape[-1] == 4:
            tmp = run_mlp(self.reg_branches[lvl], state, residual=refs)   # `tmp += refs` in the last Linear's epilogue
        else:
            if refs.shape[-1] != 2:
                raise AssertionError("reference points must be 2-d or 4-d")
            tmp = run_mlp(self.reg_branches[lvl], state)
            tmp = torch.cat((tmp[..., :2] + refs, tmp[..., 2:]), -1)
        B = tmp.shape[0]
        if capture is not None:
            capture.update(final_state=state, final_refs_unact=refs, outputs_classes=cls, outputs_coords=hip_ops.sigmoid(tmp))
        if self.use_sigmoid:
            scores, idx = hip_ops.topk(hip_ops.sigmoid(cls).view(B, -1), self.max_per_img)
            if hip_ops.decode_boxes_supported(tmp, idx):
                # label = idx % C, query = idx // C, sigmoid, cxcywh -> xyxy, scale, clamp: one launch
                boxes, labels = hip_ops.decode_boxes(tmp, idx, self.num_classes, Wimg, Himg)
                return boxes, scores, labels
            labels = idx % self.num_classes
            q = idx // self.num_classes
        else:
            s, labels_all = F.softmax(cls, dim=-1)[..., :-1].max(-1)
            scores, q = hip_ops.topk(s, self.max_per_img)
            labels = torch.gather(labels_all, 1, q)
        coords = tmp.sigmoid()
        boxes = bbox_cxcywh_to_xyxy(torch.gather(coords, 1, q.unsqueeze(-1).expand(-1, -1, 4)))
        key = (Wimg, Himg, boxes.dtype, str(boxes.device))
        scale = self._scale_cache.get(key)
        if scale is None:  # built once per image size: no host->device copy in the steady state / under capture
            scale = self._scale_cache[key] = boxes.new_tensor([Wimg, Himg, Wimg, Himg])
        boxes = torch.minimum((boxes * scale).clamp(min=0), scale)
        return boxes, scores, labels
