"""``MultiScaleDeformableAttention`` module -- host-side mirror of the reference's
codetr/multi_scale_deformable_attention.py:15-218 (same constructor kwargs, parameter names,
forward signature, error behaviour), computing on MI355X through ``codetr.hip_ops``.

Differences by design:
* CPU tensors take ``_forward_cpu`` -- plain ``torch.nn.functional`` linears + ``ops.multi_scale_deformable_attention_
  pytorch`` -- as the reference's module does (:203-210); everything on a HIP device goes through ``hip_ops`` and the
  hand-written kernels, never through that branch.
* internally batch-first: ``forward_bf`` is what the encoder/decoder call; ``forward`` keeps the
  reference's sequence-first default and just permutes around it.
"""
import math
import warnings
from typing import Optional

import torch
import torch.nn as nn

from . import hip_ops


# Value-map layout used between the value projection and the fused gather kernel (both ours, so free to choose):
# head-major [B, M, S, D] makes the two horizontal neighbours of a sample one 128-byte line.  A/B on MI355X at the
# 1920x1280 encoder shape (tools/bench_msda.py --fused): 616 us vs 665 us per launch with +-3 px synthetic offsets,
# 1236 us vs 1447 us with uniformly random locations -- but no difference end to end on the model (22.6 vs 22.5
# ms/image: freshly initialised offsets are a few pixels), so the op's own [B, S, M, D] layout stays the default.
HEAD_MAJOR_VALUE = False   # route switch (module attribute, patched by tools/ab_host_routes.py)


class DeferredOutputProj:
    """what forward_bf(defer_output_proj=True) returns when it leaves `output_proj(attn) + identity` to the caller"""
    __slots__ = ("attn", "identity")

    def __init__(self, attn, identity):
        self.attn, self.identity = attn, identity


class MultiScaleDeformableAttention(nn.Module):
    def __init__(
        self,
        embed_dims: int = 256,
        num_heads: int = 8,
        num_levels: int = 4,
        num_points: int = 4,
        im2col_step: int = 64,
        dropout: float = 0.1,
        batch_first: bool = False,
        norm_cfg: Optional[dict] = None,
        init_cfg: Optional[dict] = None,
        value_proj_ratio: float = 1.0,
    ):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        dim_per_head = embed_dims // num_heads
        if dim_per_head & (dim_per_head - 1):
            warnings.warn(
                "MultiScaleDeformableAttention: a per-head dimension that is not a power of two takes the "
                "scalar HIP kernel instead of the tiled one"
            )
        self.norm_cfg = norm_cfg
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)  # identity at inference; kept for state/arg parity
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        value_proj_size = int(embed_dims * value_proj_ratio)
        self.value_proj = nn.Linear(embed_dims, value_proj_size)
        self.output_proj = nn.Linear(value_proj_size, embed_dims)
        self.init_weights()

    def init_weights(self) -> None:
        """Directional grid for the offset bias, zero attention logits, Xavier projections
        (same scheme as reference :90-115)."""
        nn.init.zeros_(self.sampling_offsets.weight)
        theta = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack((theta.cos(), theta.sin()), -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        grid = grid.view(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        grid = grid * torch.arange(1, self.num_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        for proj in (self.value_proj, self.output_proj):
            nn.init.xavier_uniform_(proj.weight)
            nn.init.zeros_(proj.bias)

    def _fused_projection(self):
        """(sampling_offsets | attention_weights) as ONE [M*L*P*3, C] weight: the two Linears read the same
        input, so they run as a single GEMM.  Rebuilt only when a parameter tensor changes."""
        ws = (self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight,
              self.attention_weights.bias)
        return hip_ops.derived(ws, "_codetr_fused_proj", lambda: (torch.cat((ws[0], ws[2]), 0).contiguous(),
                                                                  torch.cat((ws[1], ws[3]), 0).contiguous()))

    def _packed_projection(self):
        """the same two Linears with their rows permuted into the lane-major packed layout of the round-5 encoder
        kernel ([64 M, C]; hip_ops.msda_packed_projection)"""
        ws = (self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight,
              self.attention_weights.bias)
        return hip_ops.derived(ws, "_codetr_packed_proj", lambda: hip_ops.msda_packed_projection(
            *ws, self.num_heads, self.num_levels, self.num_points))

    def _encoder_projections(self):
        """value_proj and the packed (offsets | logits) projection as one [C + 64 M, C] weight + bias (value rows first):
        the operand of the one-launch form (hip_ops.encoder_projections)"""
        ws = (self.value_proj.weight, self.value_proj.bias, self.sampling_offsets.weight, self.sampling_offsets.bias,
              self.attention_weights.weight, self.attention_weights.bias)

        def build():
            Wp, bp = hip_ops.msda_packed_projection(*ws[2:], self.num_heads, self.num_levels, self.num_points)
            return torch.cat((ws[0], Wp), 0).contiguous(), torch.cat((ws[1], bp), 0).contiguous()

        return hip_ops.derived(ws, "_codetr_enc_projections", build)

    _WINDOW_CACHE_SHAPES = 8       # pyramids whose windows are kept per layer (variable input sizes: the cache must not grow)

    def _windows_lru(self, key, build):
        """Windows per (pyramid shape, settings) derived from the offset bias: ONE derived attribute on the bias tensor
        holding an ordered dict of at most _WINDOW_CACHE_SHAPES entries, least recently used dropped (one
        attribute per shape string grew by M * L * 4 ints with every new input size).  hip_ops.derived rebuilds the
        dict -- empty -- whenever the bias tensor changes."""
        from collections import OrderedDict

        b = self.sampling_offsets.bias
        table = hip_ops.derived((b,), "_codetr_enc_windows_lru", OrderedDict)
        hit = table.get(key)
        if hit is None:
            hit = table[key] = build(b)
            while len(table) > self._WINDOW_CACHE_SHAPES:
                table.popitem(last=False)
        else:
            table.move_to_end(key)
        return hit

    def _encoder_windows_packed(self, host_shapes):
        shapes = tuple((int(h), int(w)) for h, w in host_shapes)
        key = ("v4", shapes, hip_ops.MSDA_V4_THREADS, tuple(hip_ops.MSDA_V4_REGION), hip_ops.MSDA_V4_LDS_BUDGET,
               hip_ops.MSDA_V4_MARGIN_CAP)
        return self._windows_lru(key, lambda b: hip_ops.msda_encoder_windows_packed(
            b, list(shapes), self.num_heads, self.num_levels, self.num_points))

    # ------------------------------------------------------------------ batch-first core
    def forward_bf(self, query, value, identity, query_pos, key_padding_mask, reference_points, spatial_shapes,
                   level_start_index, query_plus_pos=None, value_projected=None, defer_output_proj=False):
        """query [B,Nq,C]; value [B,S,C]; returns output_proj(msda(...)) + identity, [B,Nq,C].
        query_plus_pos: `query + query_pos` if the caller already holds it.  value_projected [B,S,C]: this module's
        value_proj(value) with the padding mask applied, if the caller already computed it (the decoder projects the
        memory for all its layers in one GEMM).  defer_output_proj: where the packed encoder kernel served the call, return
        DeferredOutputProj(attention output, identity) instead -- the caller folds `output_proj(.) + identity` into its next
        kernel (transformer_layers.BaseTransformerLayer: the fused FFN)."""
        if not query.is_cuda:
            return self._forward_cpu(query if query_plus_pos is None else None, query_plus_pos, value, identity, query_pos,
                                     key_padding_mask, reference_points, spatial_shapes, value_projected)
        pos_in_gemm = None   # query_pos still to be added: folded into the (offsets | logits) GEMM where that applies
        if query_plus_pos is not None:
            query = query_plus_pos
        elif query_pos is not None:
            if self.takes_pos_in_gemm(query, query_pos, value):
                pos_in_gemm = query_pos
            else:
                query = hip_ops.add(query, query_pos)
        B, Nq, _ = query.shape
        S = value.shape[1]
        H, L, P = self.num_heads, self.num_levels, self.num_points
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError(
                f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead."
            )
        hd = self.value_proj.out_features // H
        if (HEAD_MAJOR_VALUE and query.is_cuda and self.value_proj.in_features % 64 == 0
                and hip_ops.msda_head_major_supported(value.dtype, hd, L, P)):
            # value map written head-major [B, M, S, D] by the projection's epilogue (padding mask folded in too):
            # the x0/x1 neighbours of every sample are then one 128-byte line for the gather kernel
            v = hip_ops.linear(value, self.value_proj.weight, self.value_proj.bias, row_mask=key_padding_mask,
                               head_major=hd)
            Wc, bc = self._fused_projection()
            proj = hip_ops.linear(query, Wc, bc)
            out = hip_ops.msda_fused(v, spatial_shapes, level_start_index, proj, 0, H * L * P * 2, reference_points,
                                     L, P, head_major=True)
            return hip_ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=identity)
        host_shapes = getattr(spatial_shapes, "_codetr_host", None)
        counts = getattr(reference_points, "_codetr_valid_counts", None)
        if (query.is_cuda and value_projected is None and host_shapes is not None and Nq == S and counts is not None
                and reference_points.shape[-1] == 2 and hip_ops.MSDA_FP32_REF
                and hip_ops.msda_encoder_packed_supported(value.dtype, hd, L, P)):
            # encoder self-attention, round-5 kernel (csrc/msda_encoder4.hip).  Both producers are our own GEMMs, so both
            # layouts are the gather kernel's choice: the value projection writes the HEAD-MAJOR map [B, M, S, 32] (a staged
            # window row is one contiguous run), the (offsets | logits) projection the lane-major packed rows (its weight
            # rows permuted once; two 16-byte loads per lane).  Padding mask folded into the value GEMM (reference :173-176).
            hm = hip_ops.MSDA_V4_HEAD_MAJOR and self.value_proj.in_features % 64 == 0
            if (hm and pos_in_gemm is not None and value is query and hip_ops.ENC_PROJ_FUSED
                    and self.value_proj.bias is not None and self.value_proj.out_features % 64 == 0):
                # self-attention (value IS query): both projections in one launch, x and pos read once
                Wc, bc = self._encoder_projections()
                both = hip_ops.encoder_projections(query, pos_in_gemm, Wc, bc, key_padding_mask,
                                                   self.value_proj.out_features, hd)
                if both is not None:
                    out = hip_ops.msda_encoder_packed(both[0], host_shapes, both[1], P,
                                                      self._encoder_windows_packed(host_shapes), counts, True)
                    if out is not None:
                        if defer_output_proj:
                            return DeferredOutputProj(out.view(B, Nq, -1), identity)
                        return hip_ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=identity)
            if value.dtype == torch.bfloat16:
                # bf16 model: the kernel's value map is FP16 (packed-half blend; the projection's fp32 accumulators keep
                # three more mantissa bits than a bf16 store), head-major; offsets / logits / output stay bf16
                v = hip_ops.value_projection_f16(value, self.value_proj.weight, self.value_proj.bias, key_padding_mask, hd)
                hm = True
            else:
                v = hip_ops.linear(value, self.value_proj.weight, self.value_proj.bias, row_mask=key_padding_mask,
                                   head_major=hd if hm else None)
                if not hm:
                    v = v.view(B, S, H, -1)
            out = None
            if v is not None:
                Wp, bp = self._packed_projection()
                packed = (hip_ops.linear_xadd(query, pos_in_gemm, Wp, bp) if pos_in_gemm is not None
                          else hip_ops.linear(query, Wp, bp))
                out = hip_ops.msda_encoder_packed(v, host_shapes, packed, P, self._encoder_windows_packed(host_shapes), counts, hm)
            if out is not None:
                if defer_output_proj:
                    return DeferredOutputProj(out.view(B, Nq, -1), identity)
                return hip_ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=identity)
            if v is not None and hm and v.dtype == value.dtype:
                value_projected = v.permute(0, 2, 1, 3).reshape(B, S, -1)   # (declined shape: the general kernel's layout)
        # value projection with the padding mask folded into the GEMM epilogue (reference :173-176)
        if value_projected is not None:
            v = value_projected
        else:
            v = hip_ops.linear(value, self.value_proj.weight, self.value_proj.bias, row_mask=key_padding_mask)
        v = v.view(B, S, H, -1)
        if query.is_cuda and hip_ops.msda_fused_supported(v.dtype, v.shape[-1], L, P):
            # one GEMM for (offsets | logits); softmax and location arithmetic happen inside the MSDA kernel
            Wc, bc = self._fused_projection()
            proj = hip_ops.linear_xadd(query, pos_in_gemm, Wc, bc) if pos_in_gemm is not None else hip_ops.linear(query, Wc, bc)
            # (encoder calls the packed kernel turned down -- other level / point counts, head widths -- run here too)
            out = hip_ops.msda_fused(v, spatial_shapes, level_start_index, proj, 0, H * L * P * 2, reference_points, L, P)
            return hip_ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=identity)
        offsets = hip_ops.linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias)
        offsets = offsets.view(B, Nq, H, L, P, 2)
        weights = hip_ops.linear(query, self.attention_weights.weight, self.attention_weights.bias)
        weights = weights.view(B, Nq, H, L * P).softmax(-1).view(B, Nq, H, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack((spatial_shapes[..., 1], spatial_shapes[..., 0]), -1).to(offsets.dtype)
            loc = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = (reference_points[:, :, None, :, None, :2]
                   + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5)
        else:
            raise ValueError(
                f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead."
            )
        out = hip_ops.msda(v.contiguous(), spatial_shapes, level_start_index, loc.contiguous(), weights.contiguous(),
                           self.im2col_step)
        return hip_ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=identity)

    def _forward_cpu(self, query, query_plus_pos, value, identity, query_pos, key_padding_mask, reference_points,
                     spatial_shapes, value_projected=None):
        """CPU tensors only (reference :161-218 with the PyTorch formulation of the op, :207-210)."""
        import torch.nn.functional as F

        from .ops import multi_scale_deformable_attention_pytorch

        q = query_plus_pos if query_plus_pos is not None else (query + query_pos if query_pos is not None else query)
        B, Nq, _ = q.shape
        S = value.shape[1]
        H, L, P = self.num_heads, self.num_levels, self.num_points
        if int(spatial_shapes.prod(1).sum()) != S:
            raise AssertionError("spatial_shapes do not add up to the number of keys")
        if value_projected is not None:
            v = value_projected
        else:
            v = F.linear(value, self.value_proj.weight, self.value_proj.bias)
            if key_padding_mask is not None:
                v = v.masked_fill(key_padding_mask[..., None], 0.0)
        v = v.view(B, S, H, -1)
        off = F.linear(q, self.sampling_offsets.weight, self.sampling_offsets.bias).view(B, Nq, H, L, P, 2)
        aw = F.linear(q, self.attention_weights.weight, self.attention_weights.bias).view(B, Nq, H, L * P)
        aw = aw.softmax(-1).view(B, Nq, H, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack((spatial_shapes[..., 1], spatial_shapes[..., 0]), -1).to(off.dtype)
            loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        else:
            loc = reference_points[:, :, None, :, None, :2] + off / P * reference_points[:, :, None, :, None, 2:] * 0.5
        out = multi_scale_deformable_attention_pytorch(v, spatial_shapes, loc, aw)
        return F.linear(out, self.output_proj.weight, self.output_proj.bias) + identity

    def takes_pos_in_gemm(self, query, query_pos, value):
        """True when forward_bf adds `query_pos` inside the (offsets | logits) GEMM (so nobody needs to materialise
        `query + query_pos`): the general fused-projection path + the short-K kernel's shapes."""
        if query_pos is None or not query.is_cuda or HEAD_MAJOR_VALUE:
            return False
        H, L, P = self.num_heads, self.num_levels, self.num_points
        hd = self.value_proj.out_features // H
        if not hip_ops.msda_fused_supported(value.dtype, hd, L, P):
            return False
        Wc, _ = self._fused_projection()
        return hip_ops.linear_xadd_supported(query, query_pos, Wc)   # (the packed projection, N = 64 M, takes the same kernel)

    # ------------------------------------------------------------------ reference signature
    def forward(
        self,
        query: torch.Tensor,
        key: Optional[torch.Tensor] = None,
        value: Optional[torch.Tensor] = None,
        identity: Optional[torch.Tensor] = None,
        query_pos: Optional[torch.Tensor] = None,
        key_padding_mask: Optional[torch.Tensor] = None,
        reference_points: Optional[torch.Tensor] = None,
        spatial_shapes: Optional[torch.Tensor] = None,
        level_start_index: Optional[torch.Tensor] = None,
        **kwargs,
    ) -> torch.Tensor:
        """query ``(num_query, bs, embed_dims)`` (or batch-first if ``batch_first``), value likewise with
        ``num_key``; reference_points ``(bs, num_query, num_levels, 2|4)``; returns query-shaped tensor
        including the residual (``identity`` defaults to the un-positioned query)."""
        if value is None:
            value = query
        if identity is None:
            identity = query
        if not self.batch_first:
            query, value, identity = query.permute(1, 0, 2), value.permute(1, 0, 2), identity.permute(1, 0, 2)
            if query_pos is not None:
                query_pos = query_pos.permute(1, 0, 2)
        out = self.forward_bf(query, value, identity, query_pos, key_padding_mask, reference_points, spatial_shapes,
                              level_start_index)
        return out if self.batch_first else out.permute(1, 0, 2)
