"""Deformable encoder, DINO decoder and ``CoDinoTransformer`` -- host-side mirrors of reference
codetr/transformer.py:16-93, 120-230, 280-400, 403-582 (same class names, constructor kwargs,
parameter names and return values), batch-first inside and eval-only.

Geometry that depends only on the feature-pyramid shape and the padding mask (reference points,
proposal widths, level ids) is computed with torch on the device once per call; nothing here
reads device data back to the host, so a whole forward is hipGraph-capturable.
"""
import math

import torch
import torch.nn as nn

from . import hip_ops
from .multi_scale_deformable_attention import MultiScaleDeformableAttention
from .transformer_layers import BaseTransformerLayer, DetrTransformerDecoderLayer, build_norm


# Route switches (plain module attributes; tools/ab_host_routes.py patches them, hip_ops.nondefault_switches lists them):
DEC_FUSED = True   # False = the decoder as separate launches instead of codetr_decoder_layer_f16
DEC_VPROJ = True   # False = every decoder layer projects its own value map


class DetrTransformerEncoder(nn.Module):
    def __init__(self, post_norm_cfg=dict(type="LN"), with_cp=-1, transformerlayers=None, num_layers=None,
                 init_cfg=None):
        super().__init__()
        if not isinstance(transformerlayers, dict):
            raise AssertionError("transformerlayers must be a dict")
        layer_cfg = dict(transformerlayers)
        if layer_cfg.pop("type") != "BaseTransformerLayer":
            raise AssertionError("encoder layers must be BaseTransformerLayer")
        self.num_layers = num_layers
        self.layers = nn.ModuleList(BaseTransformerLayer(**layer_cfg) for _ in range(num_layers))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm
        if post_norm_cfg is not None:
            self.post_norm = build_norm(post_norm_cfg, self.embed_dims) if self.pre_norm else None
        else:
            if self.pre_norm:
                raise AssertionError("pre-norm encoder needs post_norm_cfg")
            self.post_norm = None

    def forward_bf(self, query, query_pos, query_key_padding_mask, **kw):
        plus_pos = None
        last = len(self.layers) - 1
        # `out + query_pos` for the next layer: not needed where that layer's attention adds query_pos inside its
        # (offsets | logits) GEMM
        att0 = self.layers[0].attentions[0] if getattr(self.layers[0], "attentions", None) else None
        pos_in_gemm = (att0 is not None and hasattr(att0, "takes_pos_in_gemm") and query_pos is not None
                       and att0.takes_pos_in_gemm(query, query_pos, query))
        for i, layer in enumerate(self.layers):
            # every layer but the last also hands over `out + query_pos`, the next layer's attention input, from its
            # fused FFN + LayerNorm epilogue (None when that kernel does not apply: the next layer then adds itself)
            query, plus_pos = layer.forward_bf(query, None, None, query_pos=query_pos,
                                               query_key_padding_mask=query_key_padding_mask,
                                               query_plus_pos=plus_pos, want_plus_pos=True, want_pos_output=i < last and not pos_in_gemm,
                                               **kw)
        return query

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None, query_key_padding_mask=None,
                key_padding_mask=None, **kwargs):
        """sequence-first in / out, as the reference (:52-92)."""
        kwargs.pop("valid_ratios", None)
        qp = None if query_pos is None else query_pos.transpose(0, 1)
        return self.forward_bf(query.transpose(0, 1), qp, query_key_padding_mask, **kwargs).transpose(0, 1)


def build_MLP(input_dim, hidden_dim, output_dim, num_layers):
    if num_layers <= 1:
        raise AssertionError(f"num_layers should be greater than 1 but got {num_layers}")
    dims = [input_dim] + [hidden_dim] * (num_layers - 1)
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        layers += [nn.Linear(a, b), nn.ReLU()]
    layers.append(nn.Linear(hidden_dim, output_dim))
    return nn.Sequential(*layers)


def run_mlp(seq, x, residual=None):
    """nn.Sequential of Linear/ReLU through hip_ops (ReLU fused into the producing linear); `residual` is added to the
    last Linear's (rounded) output in its epilogue: `seq(x) + residual`."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        lin = mods[i]
        fused = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        nxt = i + (2 if fused else 1)
        x = hip_ops.linear(x, lin.weight, lin.bias, act="relu" if fused else None,
                           residual=residual if nxt >= len(mods) else None)
        i = nxt
    return x


def pack_fragment_major(w, rows=None):
    """[N][K] row-major nn.Linear weight -> the fragment-major order codetr_decoder_layer_f16 streams (include/codetr_hip.h):
    N/16 x K/32 blocks of 64 x 8 elements, block (tile, ks) at ((tile * K/32 + ks) * 64 + lane) * 8 with lane = 16 g + r
    holding w[16 tile + r][32 ks + 8 g .. + 7]; rows padded with zeros up to `rows`.  Returns a flat tensor."""
    w = w.detach()
    N, K = w.shape
    if rows is not None and rows > N:
        w = torch.cat((w, w.new_zeros(rows - N, K)), 0)
        N = rows
    if N % 16 or K % 32:
        raise ValueError("pack_fragment_major needs N % 16 == 0 and K % 32 == 0")
    return w.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


class DinoTransformerDecoder(nn.Module):
    def __init__(self, return_intermediate=False, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if not isinstance(transformerlayers, dict):
            raise AssertionError("transformerlayers must be a dict")
        layer_cfg = dict(transformerlayers)
        if layer_cfg.pop("type") != "DetrTransformerDecoderLayer":
            raise AssertionError("decoder layers must be DetrTransformerDecoderLayer")
        self.num_layers = num_layers
        self.layers = nn.ModuleList(DetrTransformerDecoderLayer(**layer_cfg) for _ in range(num_layers))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm
        self.return_intermediate = return_intermediate
        self.ref_point_head = build_MLP(self.embed_dims * 2, self.embed_dims, self.embed_dims, 2)
        self.norm = nn.LayerNorm(self.embed_dims)

    @staticmethod
    def gen_sineembed_for_position(pos_tensor, pos_feat):
        """pos_tensor [..., 2|4] in [0,1] -> [..., len*pos_feat], blocks ordered (y, x[, w, h]),
        temperature 10000 (reference :157-190)."""
        i = torch.arange(pos_feat, dtype=pos_tensor.dtype, device=pos_tensor.device)
        dim_t = 10000 ** (2 * (i // 2) / pos_feat)

        def emb(c):
            e = (c * (2 * math.pi))[..., None] / dim_t
            return torch.stack((e[..., 0::2].sin(), e[..., 1::2].cos()), dim=-1).flatten(-2)

        n = pos_tensor.size(-1)
        if n == 2:
            return torch.cat((emb(pos_tensor[..., 1]), emb(pos_tensor[..., 0])), dim=-1)
        if n == 4:
            return torch.cat((emb(pos_tensor[..., 1]), emb(pos_tensor[..., 0]), emb(pos_tensor[..., 2]),
                              emb(pos_tensor[..., 3])), dim=-1)
        raise ValueError(f"Unknown pos_tensor shape(-1):{n}")

    # ------------------------------------------------------------------ one launch per layer (csrc/decoder_layer.hip)
    def _fused_weights(self, reg_branches):
        """Packed weights of codetr_decoder_layer_f16 for every layer, or None when the decoder is not the shape that
        kernel serves (post-norm DINO layer: MultiheadAttention(256, 8), MultiScaleDeformableAttention(8 heads of 32),
        ReLU FFN, box refinement through a 3-layer reg branch, 2-layer ref_point_head).  Built once per parameter set
        (hip_ops.derived): tail blob per layer, head blob per layer, the shared ref_point_head blob, the output norm."""
        from . import _cabi
        from . import multi_scale_deformable_attention as _msda_mod
        from .multi_scale_deformable_attention import MultiScaleDeformableAttention
        from .transformer_layers import MultiheadAttention
        if not DEC_FUSED or _msda_mod.HEAD_MAJOR_VALUE or reg_branches is None:
            return None
        C = self.embed_dims
        if len(reg_branches) < len(self.layers):
            return None
        order = ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")
        lin = lambda m, o, i: isinstance(m, nn.Linear) and m.weight.shape == (o, i) and m.bias is not None  # noqa: E731
        rp = list(self.ref_point_head)
        if not (len(rp) == 3 and lin(rp[0], C, 2 * C) and isinstance(rp[1], nn.ReLU) and lin(rp[2], C, C)):
            return None
        geo = None
        params = []
        for lid, layer in enumerate(self.layers):
            if layer.operation_order != order or layer.pre_norm:
                return None
            sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
            if not (isinstance(sa, MultiheadAttention) and isinstance(ca, MultiScaleDeformableAttention)):
                return None
            if sa.attn.in_proj_weight is None or sa.attn.in_proj_bias is None or sa.attn.out_proj.bias is None:
                return None
            fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
            g = (ca.num_heads, ca.num_levels, ca.num_points, fc1.out_features)
            if geo is None:
                geo = g
            rb = list(reg_branches[lid])
            if not (g == geo and sa.num_heads == 8 and ffn.act == "relu" and ffn.add_identity and ffn.fp8_mode is None
                    and lin(fc1, g[3], C) and lin(fc2, C, g[3]) and lin(ca.output_proj, C, C)
                    and ca.sampling_offsets.bias is not None and ca.attention_weights.bias is not None
                    and len(rb) == 5 and lin(rb[0], C, C) and lin(rb[2], C, C) and lin(rb[4], 4, C)
                    and isinstance(rb[1], nn.ReLU) and isinstance(rb[3], nn.ReLU)
                    and all(isinstance(n, nn.LayerNorm) and n.weight is not None and n.eps == self.norm.eps
                            for n in layer.norms)):
                return None
            params += [p for p in layer.parameters()] + [p for p in reg_branches[lid].parameters()]
        params += list(self.ref_point_head.parameters()) + list(self.norm.parameters())
        if any(p.dtype != params[0].dtype or not p.is_cuda for p in params) or params[0].dtype not in (torch.float16, torch.bfloat16):
            return None
        M, L, P, F = geo
        if not _cabi.decoder_layer_supported(C, M, L, P, F, 4, C // 2):
            return None

        def build():
            cat = lambda ts: torch.cat([t.detach().reshape(-1) for t in ts]).contiguous()  # noqa: E731

            frag = pack_fragment_major

            tails, heads = [], []
            for lid, layer in enumerate(self.layers):
                sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
                n1, n2, n3 = layer.norms
                fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
                rb = list(reg_branches[lid])
                pad = rb[4].bias.new_zeros(4)
                wol = torch.cat((ca.sampling_offsets.weight, ca.attention_weights.weight), 0)
                # matrices first (fragment-major), then the small vectors (the kernel copies those to LDS once)
                tails.append(cat([frag(sa.attn.out_proj.weight), frag(wol, 512), frag(ca.output_proj.weight),
                                  frag(fc1.weight), frag(fc2.weight), frag(rb[0].weight), frag(rb[2].weight),
                                  frag(rb[4].weight, 16),
                                  sa.attn.out_proj.bias, n1.weight, n1.bias, ca.sampling_offsets.bias,
                                  ca.attention_weights.bias, ca.output_proj.bias, n2.weight, n2.bias, fc1.bias, fc2.bias,
                                  n3.weight, n3.bias, rb[0].bias, rb[2].bias, rb[4].bias, pad]))
                Wi, bi = sa.attn.in_proj_weight, sa.attn.in_proj_bias               # rows [q | k | v]
                heads.append(cat([frag(Wi[: 2 * C]), frag(Wi[2 * C:]), bi]))
            pos = cat([frag(rp[0].weight), frag(rp[2].weight), rp[0].bias, rp[2].bias])
            fin = cat([self.norm.weight, self.norm.bias])
            sizes = [_cabi.decoder_layer_blob_halfs(w, L, P, F) for w in range(4)]
            if (any(t.numel() != sizes[0] for t in tails) or any(h.numel() != sizes[1] for h in heads)
                    or pos.numel() != sizes[2] or fin.numel() != sizes[3]):
                raise AssertionError("decoder_layer weight blobs do not match the library's layout")
            return dict(tails=tails, heads=heads, pos=pos, fin=fin, L=L, P=P, F=F)

        return hip_ops.derived(params, "_codetr_decoder_blobs", build)

    def _forward_fused(self, blobs, query, v_all, reference_points, valid_ratios, spatial_shapes, level_start_index):
        """6 x (self-attention core, one codetr_decoder_layer_f16 launch) + the head-only launch in front."""
        from . import _cabi
        B, Nq, C = query.shape
        S = v_all[0].shape[1] if not callable(v_all) else v_all(-1)
        L, P, F = blobs["L"], blobs["P"], blobs["F"]
        vr32 = valid_ratios._codetr_f32.contiguous()
        dev = query.device
        new = lambda *shape: torch.empty(shape, dtype=query.dtype, device=dev)  # noqa: E731
        x, ref = query.contiguous(), reference_points.contiguous()
        qpos, qk, v = new(B, Nq, C), new(B, Nq, 2 * C), new(B, Nq, C)
        eps, nl = self.norm.eps, len(self.layers)
        with torch.cuda.device(dev):
            _cabi.decoder_layer(x, None, None, ref, vr32, None, None, None, None, blobs["pos"], blobs["heads"][0], None,
                                None, None, qpos, qk, v, B, Nq, S, L, P, F, eps, 10000.0)
            for lid in range(nl):
                attn = new(B, Nq, C)
                _cabi.mha_attention(qk[..., :C], qk[..., C:], v, 8, attn)
                last = lid + 1 == nl
                x_out, ref_out = new(B, Nq, C), new(B, Nq, 4)
                qpos2, qk2, v2 = (None, None, None) if last else (new(B, Nq, C), new(B, Nq, 2 * C), new(B, Nq, C))
                v_l = v_all(lid) if callable(v_all) else v_all[lid]
                _cabi.decoder_layer(x, attn, qpos, ref, vr32, v_l.contiguous(), spatial_shapes, level_start_index,
                                    blobs["tails"][lid], None if last else blobs["pos"],
                                    None if last else blobs["heads"][lid + 1], blobs["fin"] if last else None,
                                    x_out, ref_out, qpos2, qk2, v2, B, Nq, S, L, P, F, eps, 10000.0)
                x, ref, qpos, qk, v = x_out, ref_out, qpos2, qk2, v2
        return x, ref

    @staticmethod
    def _fused_inputs_ok(query, kw):
        """What _forward_fused assumes beyond dtype / shape of the query: at most 1024 queries (the
        self-attention core keeps a head's keys in LDS), no self-attention mask of any kind (the fused path would ignore
        it), and both index tensors present as int64 device tensors (they travel as raw pointers)."""
        ss, ls = kw.get("spatial_shapes"), kw.get("level_start_index")
        if ss is None or ls is None or query.shape[1] > 1024:
            return False
        if any(kw.get(k) is not None for k in ("query_key_padding_mask", "attn_masks", "attn_mask", "self_attn_mask")):
            return False
        return all(t.dtype == torch.int64 and t.is_cuda and t.is_contiguous() for t in (ss, ls))

    def forward_bf(self, query, value, key_padding_mask, reference_points, valid_ratios, reg_branches, **kw):
        """query [B,Nq,C], value [B,S,C], reference_points [B,Nq,4] unactivated."""
        out = query
        vr = None   # (only the ATen formulation below needs the tiled valid ratios)
        v_all = self._project_values(value, key_padding_mask)
        if (query.is_cuda and query.dtype in (torch.float16, torch.bfloat16) and reference_points.shape[-1] == 4
                and reference_points.dtype == query.dtype and getattr(valid_ratios, "_codetr_f32", None) is not None
                and hip_ops.MSDA_FP32_REF and not torch.is_grad_enabled() and self._fused_inputs_ok(query, kw)):
            blobs = self._fused_weights(reg_branches)
            if blobs is not None and valid_ratios.shape[1] == blobs["L"]:
                if v_all is None:   # (small memories: each layer's own value projection, mask folded in, right before its use)
                    def v_all(lid):
                        if lid < 0:
                            return value.shape[1]
                        vp = self.layers[lid].attentions[1].value_proj
                        return hip_ops.linear(value, vp.weight, vp.bias, row_mask=key_padding_mask)
                return self._forward_fused(blobs, query, v_all, reference_points, valid_ratios, kw["spatial_shapes"],
                                           kw["level_start_index"])
        for lid, layer in enumerate(self.layers):
            if hip_ops.query_sine_embed_supported(reference_points, valid_ratios, self.embed_dims // 2):
                ref_in, sine = hip_ops.query_sine_embed(reference_points, valid_ratios, self.embed_dims // 2)
            else:
                if vr is None:
                    vr = torch.cat((valid_ratios, valid_ratios), -1) if reference_points.shape[-1] == 4 else valid_ratios
                ref_in = reference_points[:, :, None].sigmoid() * vr[:, None]  # [B,Nq,L,4]
                sine = self.gen_sineembed_for_position(ref_in[:, :, 0, :], self.embed_dims // 2)
            qpos = run_mlp(self.ref_point_head, sine)
            out = layer.forward_bf(out, None, value, query_pos=qpos, key_padding_mask=key_padding_mask,
                                   reference_points=ref_in, value_projected=None if v_all is None else v_all[lid], **kw)
            if reg_branches is not None:
                if reference_points.shape[-1] != 4:
                    raise AssertionError("box refinement needs 4-d reference points")
                reference_points = run_mlp(reg_branches[lid], out, residual=reference_points)  # no detach / sigmoid
        out = hip_ops.layer_norm(out, self.norm.weight, self.norm.bias, self.norm.eps)
        return out, reference_points

    def _cross_attentions(self):
        from .multi_scale_deformable_attention import MultiScaleDeformableAttention
        atts = []
        for layer in self.layers:
            ai = 0
            for op in layer.operation_order:
                if op in ("self_attn", "cross_attn"):
                    if op == "cross_attn":
                        atts.append(layer.attentions[ai])
                    ai += 1
        ok = len(atts) == len(self.layers) and all(isinstance(a, MultiScaleDeformableAttention) for a in atts)
        return atts if ok else None

    def _project_values(self, memory, key_padding_mask):
        """value_proj(memory) of EVERY layer's cross-attention as one GEMM (N = layers * 256): the memory [B,S,256] is
        read once instead of once per layer; the output is laid out [layer][B*S][256], so each layer's value map is a
        contiguous [B,S,M,D] tensor.  None when the fused form does not apply (then every layer projects itself)."""
        atts = self._cross_attentions()
        if (atts is None or not memory.is_cuda or memory.dtype not in (torch.float16, torch.bfloat16)
                or torch.is_grad_enabled()
                or not DEC_VPROJ):
            return None
        C = atts[0].value_proj.out_features
        if (any(a.value_proj.out_features != C or a.value_proj.in_features != memory.shape[-1] or a.value_proj.bias is None
                for a in atts) or C % 64 != 0 or memory.shape[-1] not in (192, 256) or len(atts) * C > 1536
                or memory.shape[0] * memory.shape[1] < 128 * 256):
            return None
        ps = [p for a in atts for p in (a.value_proj.weight, a.value_proj.bias)]
        hit = hip_ops.derived(ps, "_codetr_vproj_all", lambda: (torch.cat([a.value_proj.weight for a in atts], 0).contiguous(),
                                                                torch.cat([a.value_proj.bias for a in atts], 0).contiguous()))
        B, S, K = memory.shape
        mask = None if key_padding_mask is None else key_padding_mask.reshape(1, B * S)
        y = hip_ops.linear(memory.reshape(1, B * S, K), hit[0], hit[1], row_mask=mask, head_major=C)  # [1, layers, B*S, C]
        return [y[0, i].view(B, S, C) for i in range(len(atts))]

    def forward(self, query, *args, reference_points=None, valid_ratios=None, reg_branches=None, **kwargs):
        """reference layout: query (Nq, bs, C), value kwarg (S, bs, C); returns ((bs,Nq,C), (bs,Nq,4))."""
        value = kwargs.pop("value").transpose(0, 1)
        kwargs.pop("key", None)
        kwargs.pop("attn_masks", None)
        return self.forward_bf(query.transpose(0, 1), value, kwargs.pop("key_padding_mask", None), reference_points,
                               valid_ratios, reg_branches, **kwargs)


# ----------------------------------------------------------------------------------------------
# geometry helpers (reference :280-400)
# ----------------------------------------------------------------------------------------------
def get_valid_ratio(mask, dtype=torch.float32):
    """mask [B,H,W] bool (True = padding) -> [B,2] (w_ratio, h_ratio)."""
    _, H, W = mask.shape
    vh = torch.sum(~mask[:, :, 0], 1).to(dtype) / H
    vw = torch.sum(~mask[:, 0, :], 1).to(dtype) / W
    return torch.stack((vw, vh), -1)


_PIXEL_CENTRES = {}


def _pixel_centres(H, W, dtype, device):
    """(ys, xs) [1, H*W]: linspace(0.5, H-0.5, H) x linspace(0.5, W-0.5, W) meshgrid, flattened; built once per
    level shape (constants of the pyramid: keeps ~6 tiny kernels per level out of every forward)"""
    key = (H, W, dtype, str(device))
    hit = _PIXEL_CENTRES.get(key)
    if hit is None:
        ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device),
                                torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device), indexing="ij")
        hit = _PIXEL_CENTRES[key] = (ys.reshape(1, -1).contiguous(), xs.reshape(1, -1).contiguous())
    return hit


def get_reference_points(mlvl_feats, valid_ratios, device):
    """pixel centres of every level, normalised by the VALID extent -> [B, S, 2] (x, y).
    `mlvl_feats`: the level tensors [B,C,H,W], or a list of (H, W) tuples (dtype then follows valid_ratios)."""
    out = []
    for lvl, feat in enumerate(mlvl_feats):
        if isinstance(feat, tuple):
            (H, W), B, dt = feat, valid_ratios.shape[0], valid_ratios.dtype
        else:
            (B, _, H, W), dt = feat.shape, feat.dtype
        ys, xs = _pixel_centres(H, W, dt, device)
        y = ys / (valid_ratios[:, lvl, 1].reshape(B, 1) * H)
        x = xs / (valid_ratios[:, lvl, 0].reshape(B, 1) * W)
        out.append(torch.stack((x, y), -1))
    return torch.cat(out, 1)


_LVL_REPEATED = {}


def get_lvl_repeated(mlvl_masks, dtype=torch.float32):
    """[S] level index of every token (constant per pyramid: cached)"""
    key = (tuple(m.shape[-2] * m.shape[-1] for m in mlvl_masks), dtype, str(mlvl_masks[0].device))
    hit = _LVL_REPEATED.get(key)
    if hit is None:
        hit = _LVL_REPEATED[key] = torch.cat([torch.full((n,), float(lvl), dtype=dtype, device=mlvl_masks[0].device)
                                              for lvl, n in enumerate(key[0])], 0)
    return hit


def make_encoder_output_proposals_export(reference_points, mlvl_masks):
    """[x, y, 0.05*2^lvl, 0.05*2^lvl] -> logit space, [B,S,4]."""
    B, S = reference_points.shape[:2]
    wh = (0.05 * (2.0 ** get_lvl_repeated(mlvl_masks, dtype=reference_points.dtype))).expand(B, S).reshape(B, S, 1)
    prop = torch.cat((reference_points, wh, wh), dim=-1)
    return torch.log(prop / (1 - prop))


def make_encoder_output_proposals(reference_points, level_counts):
    B, S = reference_points.shape[:2]
    lvl = torch.repeat_interleave(
        torch.arange(level_counts.shape[0], dtype=reference_points.dtype, device=reference_points.device), level_counts)
    wh = (0.05 * (2.0 ** lvl)).expand(B, S).reshape(B, S, 1)
    prop = torch.cat((reference_points, wh, wh), dim=-1)
    return torch.log(prop / (1 - prop))


def apply_mask_to_proposal_and_memory(output_proposals, memory, memory_padding_mask):
    """proposals outside (-4.6, 4.6) or on padding -> finfo.max, their memory rows -> 0
    (multiplicative form, as the reference :365-380)."""
    dt = output_proposals.dtype
    inside = ((output_proposals > -4.6) & (output_proposals < 4.6)).to(dt).prod(-1, keepdim=True)
    keep = inside * (~memory_padding_mask).to(dt).unsqueeze(-1)
    output_proposals = output_proposals * keep + (1.0 - keep) * torch.finfo(dt).max
    return output_proposals, memory * keep + (1.0 - keep) * 0.0


_SHAPE_TENSORS = {}
_LEVEL_WH = {}


def _level_wh(shapes, dtype, device):
    """[L,2] (W_l, H_l) in `dtype`, built once per pyramid (the divisors of get_valid_ratio)"""
    key = (tuple(tuple(s) for s in shapes), dtype, str(device))
    hit = _LEVEL_WH.get(key)
    if hit is None:
        hit = _LEVEL_WH[key] = torch.tensor([[float(w), float(h)] for h, w in shapes], dtype=dtype, device=device)
    return hit


def _shape_tensors(shapes, device):
    """(spatial_shapes [L,2] int64, level_start_index [L] int64) on `device`, built once per pyramid:
    keeps host->device copies out of the steady state (and out of hipGraph capture)."""
    key = (tuple(shapes), str(device))
    hit = _SHAPE_TENSORS.get(key)
    if hit is None:
        ss = torch.as_tensor(shapes, dtype=torch.long, device=device)
        counts = ss.prod(1)
        ss._codetr_host = tuple((int(h), int(w)) for h, w in shapes)   # host copy for launch geometry (no sync)
        hit = (ss, torch.cat((ss.new_zeros((1,)), counts.cumsum(0)[:-1])))
        _SHAPE_TENSORS[key] = hit
    return hit


class CoDinoTransformer(nn.Module):
    def __init__(self, with_pos_coord=True, with_coord_feat=True, num_co_heads=1, as_two_stage=False,
                 num_feature_levels=4, two_stage_num_proposals=300, encoder=None, decoder=None, init_cfg=None):
        super().__init__()
        enc, dec = dict(encoder), dict(decoder)
        if enc.pop("type") != "DetrTransformerEncoder":
            raise AssertionError("encoder must be DetrTransformerEncoder")
        if dec.pop("type") != "DinoTransformerDecoder":
            raise AssertionError("decoder must be DinoTransformerDecoder")
        self.encoder = DetrTransformerEncoder(**enc)
        self.decoder = DinoTransformerDecoder(**dec)
        self.embed_dims = self.encoder.embed_dims
        self.with_pos_coord, self.with_coord_feat, self.num_co_heads = with_pos_coord, with_coord_feat, num_co_heads
        self.as_two_stage = as_two_stage
        self.num_feature_levels = num_feature_levels
        self.two_stage_num_proposals = two_stage_num_proposals
        self.init_layers()

    def init_layers(self):
        C = self.embed_dims
        self.level_embeds = nn.Parameter(torch.zeros(self.num_feature_levels, C))
        self.enc_output = nn.Linear(C, C)
        self.enc_output_norm = nn.LayerNorm(C)
        self.query_embed = nn.Embedding(self.two_stage_num_proposals, C)
        # training-time aux-head position transforms: unused at inference, present so that the
        # published checkpoint's keys load (reference :462-474)
        if self.with_pos_coord and self.num_co_heads > 0:
            self.aux_pos_trans = nn.ModuleList(nn.Linear(C * 2, C) for _ in range(self.num_co_heads))
            self.aux_pos_trans_norm = nn.ModuleList(nn.LayerNorm(C) for _ in range(self.num_co_heads))
            self.pos_feats_trans = nn.ModuleList()
            self.pos_feats_norm = nn.ModuleList()
            if self.with_coord_feat:
                self.pos_feats_trans.extend(nn.Linear(C, C) for _ in range(self.num_co_heads))
                self.pos_feats_norm.extend(nn.LayerNorm(C) for _ in range(self.num_co_heads))

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.query_embed.weight)

    def forward(self, mlvl_feats, mlvl_masks, mlvl_pos_embeds, reg_branches=None, cls_branches=None,
                forced_topk_indices=None, capture=None, **kwargs):
        """feats / pos_embeds: lists of [B,C,H,W]; masks: list of [B,H,W] bool.
        Returns (final_state [B,Nq,C], final_references_unact [B,Nq,4]).

        Two test hooks that do not exist in the reference: ``forced_topk_indices`` [B,Nq] replaces the
        proposal top-k (parity checks on random weights, where top-k is unstable: reference
        tests/test_export.py:638-655) and ``capture`` (a dict) receives intermediates."""
        shapes = [tuple(f.shape[-2:]) for f in mlvl_feats]
        feat = torch.cat([f.flatten(2).transpose(1, 2) for f in mlvl_feats], 1)  # [B,S,C]
        pos_tokens = [p.flatten(2).transpose(1, 2) for p in mlvl_pos_embeds]
        return self.forward_flat(feat, shapes, mlvl_masks, pos_tokens, reg_branches, cls_branches, forced_topk_indices,
                                 capture)

    def forward_flat(self, feat, shapes, mlvl_masks, mlvl_pos_tokens, reg_branches=None, cls_branches=None,
                     forced_topk_indices=None, capture=None, lvl_pos_embed_flat=None, mask_flat=None,
                     valid_counts=None):
        """Same computation on inputs that are already in the transformer's layout: feat [B,S,C] (levels
        concatenated), shapes [(H_l, W_l)], masks list of [B,H_l,W_l] bool, positional encodings list of
        [B, H_l*W_l, C] -- or `lvl_pos_embed_flat` [B,S,C], the encodings with the level embeddings already added.
        `mask_flat` [B,S] bool / `valid_counts` [B,L,2] fp32 (hip_ops.mask_pyramid): the concatenated level masks and
        the valid pixel counts of each level's first row / column, when the caller already has them."""
        if not self.as_two_stage:
            raise AssertionError("as_two_stage must be True for DINO")
        dev = feat.device
        mask = mask_flat if mask_flat is not None else torch.cat([m.flatten(1) for m in mlvl_masks], 1)  # [B,S]
        if lvl_pos_embed_flat is not None:
            pos = lvl_pos_embed_flat
        else:
            pos = torch.cat([p + self.level_embeds[l].view(1, 1, -1) for l, p in enumerate(mlvl_pos_tokens)], 1)
        spatial_shapes, level_start_index = _shape_tensors(shapes, dev)
        if valid_counts is not None:
            # sum(~mask[:, 0, :]) / W, sum(~mask[:, :, 0]) / H of get_valid_ratio, from the counts of the pyramid kernel
            valid_ratios = hip_ops.valid_ratios(valid_counts, _level_wh(shapes, feat.dtype, dev))  # [B,L,2]
        else:
            valid_ratios = torch.stack([get_valid_ratio(m, dtype=feat.dtype) for m in mlvl_masks], 1)  # [B,L,2]
        native_geom = feat.is_cuda and feat.dtype in (torch.float16, torch.bfloat16) and mask.dtype == torch.bool
        if native_geom:
            # reference points, per-level scaling, masked proposals and the keep / drop state of every token: one launch
            reference_points, ref_by_level, proposals, row_state = hip_ops.encoder_geometry(valid_ratios, mask, shapes)
            if valid_counts is not None:
                # these ARE get_reference_points x valid ratios: the encoder's MSDA kernel may recompute them in fp32
                # from the counts instead of reading the fp16 tensor (a quarter pixel of resolution on a 480-wide level)
                ref_by_level._codetr_valid_counts = valid_counts
        else:
            reference_points = get_reference_points([tuple(s) for s in shapes], valid_ratios, device=dev)  # [B,S,2]
            ref_by_level = reference_points[:, :, None] * valid_ratios[:, None]  # [B,S,L,2]

        memory = self.encoder.forward_bf(feat, pos, mask, reference_points=ref_by_level, spatial_shapes=spatial_shapes,
                                         level_start_index=level_start_index)
        B = memory.shape[0]
        if native_geom:
            # `memory * keep` rides on the GEMM: dropped rows (state 2) read as zero input rows -> bias
            out_mem = hip_ops.linear(memory, self.enc_output.weight, self.enc_output.bias, row_mask=row_state)
        else:
            proposals = make_encoder_output_proposals_export(reference_points, mlvl_masks)
            proposals, out_mem = apply_mask_to_proposal_and_memory(proposals, memory, mask)
            out_mem = hip_ops.linear(out_mem, self.enc_output.weight, self.enc_output.bias)
        out_mem = hip_ops.layer_norm(out_mem, self.enc_output_norm.weight, self.enc_output_norm.bias,
                                     self.enc_output_norm.eps)
        last = self.decoder.num_layers  # branch index 6 = the two-stage proposal head
        cls_head = cls_branches[last]
        enc_cls = hip_ops.linear(out_mem, cls_head.weight, cls_head.bias)
        if forced_topk_indices is None:
            topk = hip_ops.topk(hip_ops.row_max(enc_cls), self.two_stage_num_proposals, want_values=False)[1]
        else:
            topk = forced_topk_indices
        # the box branch is row-wise: run it on the selected rows only (900 instead of all S tokens)
        sel = hip_ops.gather_rows(out_mem, topk)
        topk_coords = run_mlp(reg_branches[last], sel, residual=hip_ops.gather_rows(proposals, topk))
        qw = self.query_embed.weight
        if B == 1 or not qw.is_cuda or torch.is_grad_enabled():
            query = qw[None].expand(B, -1, -1)   # (a view: keeps the autograd path to query_embed.weight)
        else:
            # inference: one materialised [B, Nq, C] copy per batch size, kept on the parameter (the decoder's first layer
            # uses it as a GEMM residual, which needs real rows): no per-forward broadcast copy.  Read-only by contract:
            # every forward gets the same storage
            query = hip_ops.derived((qw,), f"_codetr_query_b{B}", lambda: qw.detach()[None].expand(B, -1, -1).contiguous())
        if capture is not None:
            # the reference's all-rows form (:555-557), for inspection only: what feeds the decoder is `topk_coords`
            enc_coord = run_mlp(reg_branches[last], out_mem) + proposals
            capture["topk_coords_unact"] = topk_coords
            capture.update(memory=memory, enc_outputs_class=enc_cls, enc_outputs_coord_unact=enc_coord,
                           topk_indices=topk, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                           valid_ratios=valid_ratios)
        return self.decoder.forward_bf(query, memory, mask, topk_coords, valid_ratios, reg_branches,
                                       spatial_shapes=spatial_shapes, level_start_index=level_start_index)
