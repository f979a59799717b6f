"""``torch.ops.codetr.*`` on MI355X: schema, meta kernel and the HIP implementation.

Host-side mirror of the reference's operator layer for the inference hot path:

* schema strings are the reference's, character for character
  (reference codetr/csrc/deformable_attention_torch.cpp:16-24);
* the implementation is registered for dispatch key ``CUDA`` -- the key PyTorch-ROCm uses for
  HIP tensors -- exactly as the reference registers its CUDA kernel
  (deformable_attention_torch.cpp:28-31).  No other key is registered: calling the op with CPU
  tensors fails in the dispatcher, as it does in the reference.  The reference's second, device-agnostic entry
  point -- ``multi_scale_deformable_attention_pytorch`` (ops.py:129-186), which its ``MultiScaleDeformableAttention``
  module dispatches to for CPU tensors (multi_scale_deformable_attention.py:203-210) -- exists here too, with the same
  signature, written independently as an explicit corner gather (not ``grid_sample``, and not the oracle: nothing in
  this package imports ``oracle/``).  It serves CPU tensors ONLY; a HIP tensor never reaches it, and a missing
  ``libcodetr_hip.so`` is still an ImportError of the whole package;
* the fake/meta kernel performs the reference's rank / dtype / shape checks and returns an
  empty ``(bs, num_queries, embed_dims)`` tensor (reference codetr/ops.py:19-87);
* the argument contract enforced before the launch is the reference's AT_ASSERTM list
  (ms_deform_attn.cu:902-933): contiguous, on device, ``batch % min(batch, im2col_step) == 0``.

The backward op is implemented too (csrc/msda_backward.hip) and wired into autograd exactly as
the reference does (reference codetr/ops.py:90-126): zero-filled gradient tensors, one call of
``multi_scale_deformable_attention_backward``, gradients for value / sampling_loc / attn_weight.
"""
import torch
from torch import Tensor

from . import _cabi

__all__ = ["multi_scale_deformable_attention", "multi_scale_deformable_attention_pytorch"]

_FWD_SCHEMA = (
    "multi_scale_deformable_attention(Tensor value, Tensor spatial_shapes, "
    "Tensor level_start_index, Tensor sampling_loc, Tensor attn_weight, "
    "int im2col_step) -> Tensor"
)
_BWD_SCHEMA = (
    "multi_scale_deformable_attention_backward(Tensor value, Tensor "
    "spatial_shapes, Tensor level_start_index, Tensor sampling_loc, Tensor "
    "attn_weight, Tensor grad_output, Tensor(a!) grad_value, Tensor(b!) "
    "grad_sampling_loc, Tensor(c!) grad_attn_weight, int im2col_step) -> ()"
)

_lib = torch.library.Library("codetr", "DEF")
_lib.define(_FWD_SCHEMA)
_lib.define(_BWD_SCHEMA)

_FLOAT_DTYPES = (torch.float16, torch.bfloat16, torch.float32, torch.float64)


def _check_contract(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    # reference ms_deform_attn.cu:902-912 (contiguity + device), plugin.cpp:163-245 (ranks/dtypes)
    named = (
        ("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
        ("sampling_loc", sampling_loc), ("attn_weight", attn_weight),
    )
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if t.device != value.device:
            raise RuntimeError(f"{name} must be on the same device as value ({t.device} vs {value.device})")
    if value.dtype not in _FLOAT_DTYPES:
        raise RuntimeError(f"unsupported value dtype {value.dtype}")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("value, sampling_loc and attn_weight must share one dtype")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64")
    if value.dim() != 4 or spatial_shapes.dim() != 2 or level_start_index.dim() != 1:
        raise RuntimeError("expected value[B,S,M,D], spatial_shapes[L,2], level_start_index[L]")
    if sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected sampling_loc[B,Nq,M,L,P,2], attn_weight[B,Nq,M,L,P]")
    B, _, M, _ = value.shape
    L = spatial_shapes.shape[0]
    Nq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if spatial_shapes.shape[1] != 2 or level_start_index.shape[0] != L:
        raise RuntimeError("spatial_shapes must be [L,2] and level_start_index [L]")
    if tuple(sampling_loc.shape) != (B, Nq, M, L, P, 2):
        raise RuntimeError(f"sampling_loc shape {tuple(sampling_loc.shape)} != {(B, Nq, M, L, P, 2)}")
    if tuple(attn_weight.shape) != (B, Nq, M, L, P):
        raise RuntimeError(f"attn_weight shape {tuple(attn_weight.shape)} != {(B, Nq, M, L, P)}")


def _msda_forward_hip(
    value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor, sampling_loc: Tensor, attn_weight: Tensor,
    im2col_step: int,
) -> Tensor:
    _check_contract(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    B, _, M, D = value.shape
    Nq = sampling_loc.shape[1]
    # torch.empty, not zeros: the kernel writes every element (the reference zero-fills twice,
    # ms_deform_attn.cu:936, 968)
    out = torch.empty((B, Nq, M * D), dtype=value.dtype, device=value.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(value.device):
        _cabi.msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, out)
    return out


def _msda_backward_hip(
    value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor, sampling_loc: Tensor, attn_weight: Tensor,
    grad_output: Tensor, grad_value: Tensor, grad_sampling_loc: Tensor, grad_attn_weight: Tensor, im2col_step: int,
) -> None:
    """reference ms_deform_attn_backward (ms_deform_attn.cu:975-1028): accumulates into the caller's zero-filled
    gradient tensors"""
    _check_contract(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    for name, t, ref in (("grad_output", grad_output, None), ("grad_value", grad_value, value),
                         ("grad_sampling_loc", grad_sampling_loc, sampling_loc),
                         ("grad_attn_weight", grad_attn_weight, attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
        if not t.is_cuda or t.device != value.device or t.dtype != value.dtype:
            raise RuntimeError(f"{name} must be a CUDA tensor of value's dtype on value's device")
        if ref is not None and t.shape != ref.shape:
            raise RuntimeError(f"{name} must have the shape of its primal {tuple(ref.shape)}")
    B, _, M, D = value.shape
    Nq = sampling_loc.shape[1]
    if tuple(grad_output.shape) != (B, Nq, M * D):
        raise RuntimeError(f"grad_output shape {tuple(grad_output.shape)} != {(B, Nq, M * D)}")
    if grad_output.numel() == 0:
        return
    with torch.cuda.device(value.device):
        _cabi.msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            grad_value, grad_sampling_loc, grad_attn_weight, im2col_step)


_lib.impl("multi_scale_deformable_attention", _msda_forward_hip, "CUDA")
_lib.impl("multi_scale_deformable_attention_backward", _msda_backward_hip, "CUDA")


@torch.library.register_fake("codetr::multi_scale_deformable_attention")
def _multi_scale_deformable_attention_fake(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                           im2col_step):
    torch._check(value.dim() == 4)
    torch._check(spatial_shapes.dim() == 2)
    torch._check(level_start_index.dim() == 1)
    torch._check(sampling_loc.dim() == 6)
    torch._check(attn_weight.dim() == 5)
    torch._check(value.dtype == attn_weight.dtype)
    torch._check(value.dtype == sampling_loc.dtype)
    torch._check(spatial_shapes.dtype == torch.int64)
    torch._check(level_start_index.dtype == torch.int64)
    bs, _, num_heads, dim_per_head = value.shape
    num_levels = spatial_shapes.shape[0]
    torch._check(spatial_shapes.shape[1] == 2)
    torch._check(level_start_index.shape[0] == num_levels)
    torch._check(sampling_loc.shape[0] == bs)
    num_queries = sampling_loc.shape[1]
    torch._check(sampling_loc.shape[2] == num_heads)
    torch._check(sampling_loc.shape[3] == num_levels)
    num_points = sampling_loc.shape[4]
    torch._check(sampling_loc.shape[5] == 2)
    torch._check(attn_weight.shape[0] == bs)
    torch._check(attn_weight.shape[1] == num_queries)
    torch._check(attn_weight.shape[2] == num_heads)
    torch._check(attn_weight.shape[3] == num_levels)
    torch._check(attn_weight.shape[4] == num_points)
    return torch.empty((bs, num_queries, num_heads * dim_per_head), dtype=value.dtype, device=value.device)


def _msda_autograd_backward(ctx, grad):
    # reference codetr/ops.py:90-113
    value, spatial_shapes, level_start_index, sampling_loc, attn_weight = ctx.saved_tensors
    grad_value = torch.zeros_like(value)
    grad_sampling_loc = torch.zeros_like(sampling_loc)
    grad_attn_weight = torch.zeros_like(attn_weight)
    torch.ops.codetr.multi_scale_deformable_attention_backward(
        value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad.contiguous(), grad_value,
        grad_sampling_loc, grad_attn_weight, im2col_step=ctx.im2col_step)
    return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


def _msda_autograd_setup(ctx, inputs, output):
    # reference codetr/ops.py:116-120
    value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step = inputs
    ctx.im2col_step = im2col_step
    ctx.save_for_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


torch.library.register_autograd("codetr::multi_scale_deformable_attention", _msda_autograd_backward,
                                setup_context=_msda_autograd_setup)


def multi_scale_deformable_attention(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                     im2col_step=64):
    """Convenience alias of ``torch.ops.codetr.multi_scale_deformable_attention``."""
    return torch.ops.codetr.multi_scale_deformable_attention(
        value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step
    )


def multi_scale_deformable_attention_pytorch(value: Tensor, value_spatial_shapes: Tensor, sampling_locations: Tensor,
                                             attention_weights: Tensor) -> Tensor:
    """Device-agnostic PyTorch formulation with the signature of reference codetr/ops.py:129-186 (the path the
    reference's module takes for CPU tensors).  value [bs, num_keys, M, D]; value_spatial_shapes [L, 2] (h, w);
    sampling_locations [bs, Nq, M, L, P, 2] (x, y in [0, 1]); attention_weights [bs, Nq, M, L, P] -> [bs, Nq, M*D].

    Written as the kernel computes it (ms_deform_attn.cu:31-77, 246-252) rather than through ``grid_sample``: pixel
    coordinates ``loc * size - 0.5``, the (-1, size) range gate, four corners with per-corner bounds tests, weights
    ``hh*hw, hh*lw, lh*hw, lh*lw``, attention-weighted sum over levels and points.  Differentiable."""
    bs, _, M, D = value.shape
    _, Nq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes.tolist()]
    out = value.new_zeros(bs, Nq, M, D)
    b_idx = torch.arange(bs, device=value.device).view(bs, 1, 1, 1)
    m_idx = torch.arange(M, device=value.device).view(1, 1, M, 1)
    start = 0
    for lvl, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]                        # [bs, H*W, M, D]
        start += H * W
        x = sampling_locations[:, :, :, lvl, :, 0] * W - 0.5     # [bs, Nq, M, P]
        y = sampling_locations[:, :, :, lvl, :, 1] * H - 0.5
        inside = (y > -1) & (x > -1) & (y < H) & (x < W)
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        x0, y0 = x0.long(), y0.long()
        acc = 0
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                xi, yi = x0 + dx, y0 + dy
                ok = inside & (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
                pix = yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)
                corner = v[b_idx, pix, m_idx]                    # [bs, Nq, M, P, D]
                acc = acc + (wy * wx * ok.to(value.dtype)).unsqueeze(-1) * corner
        out = out + (acc * attention_weights[:, :, :, lvl, :].unsqueeze(-1)).sum(3)
    return out.reshape(bs, Nq, M * D)
