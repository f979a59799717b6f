"""Transformer building blocks: ``FFN``, ``MultiheadAttention``, ``BaseTransformerLayer``,
``DetrTransformerDecoderLayer`` -- host-side mirrors of reference codetr/transformer_mmcv.py:319-749
and codetr/transformer.py:233-277 (same constructor kwargs incl. the deprecated
``feedforward_channels`` / ``ffn_dropout`` / ``ffn_num_fcs`` spellings the configs use, same
parameter names so mmdet checkpoints load).

Internally every layer is batch-first ``[B, N, C]`` and eval-only (dropout / DropPath are the
identity); the public ``forward`` of each class accepts the reference's sequence-first layout.
"""
import copy
import warnings

import torch
import torch.nn as nn

from . import hip_ops
from .multi_scale_deformable_attention import DeferredOutputProj, MultiScaleDeformableAttention


def build_norm(cfg, dim):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cfg.pop("requires_grad", None)
    if t == "LN":
        return nn.LayerNorm(dim, **cfg)
    if t == "GN":
        return nn.GroupNorm(num_channels=dim, **cfg)
    raise NotImplementedError(f"norm type {t}")


def _act_name(act_cfg):
    t = dict(act_cfg)["type"]
    if t == "ReLU":
        return "relu"
    if t == "GELU":
        return "gelu"
    raise NotImplementedError(f"activation {t}")


def _act_module(name):
    return nn.ReLU(inplace=True) if name == "relu" else nn.GELU()


class FFN(nn.Module):
    """Linear -> act -> Linear (+ identity).  Parameter names: ``layers.0.0.*`` and ``layers.1.*``."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type="ReLU", inplace=True),
                 ffn_drop=0.0, dropout_layer=None, add_identity=True, init_cfg=None, layer_scale_init_value=0.0):
        super().__init__()
        if num_fcs != 2:
            raise NotImplementedError("only the 2-layer FFN of the Co-DETR configs is built")
        if layer_scale_init_value > 0:
            raise NotImplementedError("LayerScale is not used by the Co-DETR configs")
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        self.act = _act_name(act_cfg)
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), _act_module(self.act), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims),
            nn.Dropout(ffn_drop),
        )
        self.add_identity = add_identity

    def fused_supported(self, x, identity=None):
        fc1, fc2 = self.layers[0][0], self.layers[1]
        return (self.add_identity and identity is None and fc1.bias is not None and fc2.bias is not None
                and hip_ops.ffn_fused_supported(x, fc1.weight, fc2.weight, self.act))

    # fp8 inference mode (codetr/fp8.py): None | "calibrate" (fp16 forward that records the absolute maxima of the FFN
    # input and of the hidden activation) | "run" (both products on the e4m3 MFMA path, static scales _fp8_scales)
    fp8_mode = None

    def _fp8_observe(self, x, ln_in):
        fc1 = self.layers[0][0]
        x1 = x if ln_in is None else hip_ops.layer_norm(x, *ln_in)
        h = hip_ops.linear(x1, fc1.weight, fc1.bias, act=self.act)
        amax = self.__dict__.setdefault("_fp8_amax", {})
        for k, t in (("x", x1), ("h", h)):
            a = t.detach().abs().amax().float()
            amax[k] = a if k not in amax else torch.maximum(amax[k], a)

    def _fp8_ready(self, x):
        fc1, fc2 = self.layers[0][0], self.layers[1]
        return (self.fp8_mode == "run" and hasattr(self, "_fp8_scales") and not torch.is_grad_enabled()
                and hip_ops.ffn_fp8_supported(x, fc1.weight, fc2.weight, self.act))

    def takes_output_proj(self, x):
        """True when forward_norm can fold a preceding attention's `output_proj(attn) + identity` into its launch
        (the fp16 / bf16 fused kernel; not the fp8 mode, whose kernel has no such form)"""
        return hip_ops.FFN_OPROJ and self.fp8_mode not in ("calibrate", "run") and self.fused_supported(x)

    def forward_norm(self, x, norm, pos=None, norm_in=None, oproj=None):
        """LayerNorm(x + ffn(x)) -- and, with `pos`, also that + pos -- in the fused kernel's epilogue
        (call only when fused_supported(x)).  norm_in: a LayerNorm applied to x first, inside the kernel.
        oproj = (output_proj Linear, identity): x is an attention output and the kernel starts with
        identity + output_proj(x) (call only when takes_output_proj(x))."""
        fc1, fc2 = self.layers[0][0], self.layers[1]
        ln_in = None if norm_in is None else (norm_in.weight, norm_in.bias, norm_in.eps)
        if oproj is not None:
            lin, identity = oproj
            return hip_ops.ffn_oproj_fused(x, lin.weight, lin.bias, identity, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                                           ln=(norm.weight, norm.bias, norm.eps), pos=pos, ln_in=ln_in)
        if self.fp8_mode == "calibrate":
            self._fp8_observe(x, ln_in)
        elif self._fp8_ready(x):
            sc = self._fp8_scales
            return hip_ops.ffn_fp8(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, sc["x"], sc["h"],
                                   ln=(norm.weight, norm.bias, norm.eps), pos=pos, ln_in=ln_in)
        return hip_ops.ffn_fused(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ln=(norm.weight, norm.bias, norm.eps),
                                 pos=pos, ln_in=ln_in)

    def forward(self, x, identity=None):
        fc1, fc2 = self.layers[0][0], self.layers[1]
        if self.fused_supported(x, identity):
            # encoder / decoder FFN (256 -> 2048 -> 256, ReLU): one kernel, the hidden activation never reaches HBM
            if self.fp8_mode == "calibrate":
                self._fp8_observe(x, None)
            elif self._fp8_ready(x):
                sc = self._fp8_scales
                return hip_ops.ffn_fp8(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, sc["x"], sc["h"])
            return hip_ops.ffn_fused(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        h = hip_ops.linear(x, fc1.weight, fc1.bias, act=self.act)
        if not self.add_identity:
            return hip_ops.linear(h, fc2.weight, fc2.bias)
        return hip_ops.linear(h, fc2.weight, fc2.bias, residual=x if identity is None else identity)


class MultiheadAttention(nn.Module):
    """Self-attention with positional encodings added to q and k, residual included.
    Parameters live in ``self.attn`` (an ``nn.MultiheadAttention``, for checkpoint key parity:
    ``attn.in_proj_weight``, ``attn.in_proj_bias``, ``attn.out_proj.*``)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type="Dropout", drop_prob=0.0), init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        if "dropout" in kwargs:  # deprecated spelling used by the configs
            attn_drop = kwargs.pop("dropout")
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)

    def forward_bf(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                   key_padding_mask=None):
        if attn_mask is not None or key_padding_mask is not None:
            raise NotImplementedError("masked dense attention is not on the Co-DETR inference path")
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        q = hip_ops.add(query, query_pos) if query_pos is not None else query
        if key is query and key_pos is query_pos:
            k = q  # self-attention: one add, and q | k projected by one GEMM below
        else:
            k = hip_ops.add(key, key_pos) if key_pos is not None else key
        C = self.embed_dims
        W, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        if q is k:
            qk = hip_ops.linear(q, W[: 2 * C], b[: 2 * C])
            qp, kp = qk[..., :C], qk[..., C:]
        else:
            qp = hip_ops.linear(q, W[:C], b[:C])
            kp = hip_ops.linear(k, W[C: 2 * C], b[C: 2 * C])
        vp = hip_ops.linear(value, W[2 * C:], b[2 * C:])
        o = hip_ops.mha_self_attention(qp, kp, vp, self.num_heads)
        return hip_ops.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=identity)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        if self.batch_first:
            return self.forward_bf(query, key, value, identity, query_pos, key_pos, attn_mask, key_padding_mask)
        t = lambda a: None if a is None else a.transpose(0, 1)  # noqa: E731
        return self.forward_bf(t(query), t(key), t(value), t(identity), t(query_pos), t(key_pos), attn_mask,
                               key_padding_mask).transpose(0, 1)


class BaseTransformerLayer(nn.Module):
    """Interpreter of an ``operation_order`` over attentions / norms / FFNs (post-norm or pre-norm)."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0.0,
                               act_cfg=dict(type="ReLU", inplace=True)),
                 operation_order=None, norm_cfg=dict(type="LN"), init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for old, new in (("feedforward_channels", "feedforward_channels"), ("ffn_dropout", "ffn_drop"),
                         ("ffn_num_fcs", "num_fcs")):
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        allowed = {"self_attn", "norm", "ffn", "cross_attn"}
        if not set(operation_order) <= allowed:
            raise AssertionError(f"operation_order of {type(self).__name__} may only contain {sorted(allowed)}")
        self.batch_first = batch_first
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            if num_attn != len(attn_cfgs):
                raise AssertionError(f"{len(attn_cfgs)} attn_cfgs for {num_attn} attentions in {operation_order}")
            attn_cfgs = [copy.deepcopy(c) for c in attn_cfgs]
        self.num_attn = num_attn
        self.operation_order = tuple(operation_order)
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == "norm"
        self.attentions = nn.ModuleList()
        for cfg in attn_cfgs:
            cfg = dict(cfg)
            cfg.setdefault("batch_first", batch_first)
            kind = cfg.pop("type")
            if kind == "MultiheadAttention":
                self.attentions.append(MultiheadAttention(**cfg))
            elif kind == "MultiScaleDeformableAttention":
                self.attentions.append(MultiScaleDeformableAttention(**cfg))
            else:
                raise NotImplementedError(f"Not implemented {kind}")
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count("ffn")):
            cfg = dict(copy.deepcopy(ffn_cfgs))
            cfg.setdefault("embed_dims", self.embed_dims)
            if cfg.pop("type", "FFN") != "FFN":
                raise NotImplementedError("only FFN")
            self.ffns.append(FFN(**cfg))
        self.norms = nn.ModuleList(build_norm(norm_cfg, self.embed_dims) for _ in range(operation_order.count("norm")))

    def forward_bf(self, query, key=None, value=None, query_pos=None, key_pos=None, query_key_padding_mask=None,
                   key_padding_mask=None, query_plus_pos=None, want_plus_pos=False, want_pos_output=True, **kw):
        """batch-first walk of operation_order; `kw` carries reference_points / spatial_shapes /
        level_start_index for the deformable attentions.
        query_plus_pos: `query + query_pos` when the caller already has it (the previous layer's fused FFN epilogue);
        want_plus_pos: return a pair (out, out + query_pos); the second member is produced when the layer ends in
        'ffn', 'norm', the fused kernel applies and want_pos_output is set -- None otherwise."""
        ni = ai = fi = 0
        identity = query
        plus_pos_out = None
        ops = self.operation_order
        skip_norm = False
        norm_in = None  # a LayerNorm deferred into the fused FFN kernel
        oproj = None    # (output_proj, identity) of a deformable attention, deferred into the same kernel
        for oi, op in enumerate(ops):
            if skip_norm and op == "norm":
                skip_norm = False
                ni += 1
                continue
            if op in ("self_attn", "cross_attn"):
                att = self.attentions[ai]
                res = identity if self.pre_norm else None
                if isinstance(att, MultiScaleDeformableAttention):
                    val = query if op == "self_attn" else value
                    mask = query_key_padding_mask if op == "self_attn" else key_padding_mask
                    qpp = query_plus_pos if (ai == 0 and oi == 0) else None  # valid for the layer's input only
                    vp = kw.get("value_projected") if op == "cross_attn" else None
                    # (attn, norm, ffn, norm) of a post-norm layer whose FFN runs fused: output_proj + identity go there too
                    defer = (not self.pre_norm and op == "self_attn" and tuple(ops[oi + 1:oi + 4]) == ("norm", "ffn", "norm")
                             and att.output_proj.bias is not None and not torch.is_grad_enabled()
                             and all(isinstance(n, nn.LayerNorm) and n.weight is not None and n.weight.dtype == query.dtype
                                     for n in self.norms[ni:ni + 2])
                             and self.ffns[fi].takes_output_proj(query))
                    query = att.forward_bf(query, val, query if res is None else res, query_pos, mask,
                                           kw["reference_points"], kw["spatial_shapes"], kw["level_start_index"],
                                           query_plus_pos=qpp, value_projected=vp, defer_output_proj=defer)
                    if isinstance(query, DeferredOutputProj):
                        oproj = (att.output_proj, query.identity)
                        query = query.attn
                else:
                    if op == "self_attn":
                        query = att.forward_bf(query, query, query, res, query_pos, query_pos)
                    else:
                        query = att.forward_bf(query, key, value, res, query_pos, key_pos)
                ai += 1
                identity = query
            elif op == "norm":
                n = self.norms[ni]
                if (not self.pre_norm and oi + 2 < len(ops) and ops[oi + 1] == "ffn" and ops[oi + 2] == "norm"
                        and isinstance(n, nn.LayerNorm) and n.weight is not None and n.weight.dtype == query.dtype
                        and isinstance(self.norms[ni + 1], nn.LayerNorm) and self.norms[ni + 1].weight is not None
                        and self.norms[ni + 1].weight.dtype == query.dtype
                        and self.ffns[fi].fused_supported(query)):
                    # (norm, ffn, norm): this norm's output is read by the FFN alone (operand and identity), so the
                    # fused FFN kernel applies it to its input rows in registers
                    norm_in = n
                else:
                    if oproj is not None:
                        raise AssertionError("an output projection was deferred into a fused FFN that does not follow")
                    query = hip_ops.layer_norm(query, n.weight, n.bias, n.eps)
                ni += 1
            else:  # ffn
                ffn = self.ffns[fi]
                last_pair = oi + 2 == len(ops) and ops[oi + 1] == "norm"
                if (not self.pre_norm and oi + 1 < len(ops) and ops[oi + 1] == "norm" and ffn.fused_supported(query)
                        and isinstance(self.norms[ni], nn.LayerNorm) and self.norms[ni].weight is not None
                        and self.norms[ni].weight.dtype == query.dtype):
                    # post-norm layer: the LayerNorm that follows (and, at the end of the layer, the next layer's
                    # `+ query_pos`) ride in the fused FFN kernel's epilogue
                    pos = query_pos if (want_plus_pos and want_pos_output and last_pair and query_pos is not None
                                        and query_pos.shape == query.shape) else None
                    r = ffn.forward_norm(query, self.norms[ni], pos, norm_in, oproj)
                    norm_in = oproj = None
                    if pos is not None:
                        query, plus_pos_out = r
                    else:
                        query = r
                    skip_norm = True
                else:
                    if norm_in is not None or oproj is not None:
                        raise AssertionError("a LayerNorm / output projection was deferred into a fused FFN that did not run")
                    query = ffn(query, identity if self.pre_norm else None)
                fi += 1
        if want_plus_pos:
            return query, plus_pos_out
        return query

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        if attn_masks is not None:
            raise NotImplementedError("attn_masks are not used on the Co-DETR inference path")
        if self.batch_first:
            return self.forward_bf(query, key, value, query_pos, key_pos, query_key_padding_mask, key_padding_mask,
                                   **kwargs)
        t = lambda a: None if a is None else a.transpose(0, 1)  # noqa: E731
        return self.forward_bf(t(query), t(key), t(value), t(query_pos), t(key_pos), query_key_padding_mask,
                               key_padding_mask, **kwargs).transpose(0, 1)


class DetrTransformerDecoderLayer(BaseTransformerLayer):
    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels, ffn_dropout=ffn_dropout,
                         operation_order=operation_order, norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        if len(operation_order) != 6 or set(operation_order) != {"self_attn", "norm", "cross_attn", "ffn"}:
            raise AssertionError("decoder layer needs ('self_attn','norm','cross_attn','norm','ffn','norm')")
