"""Swin Transformer backbone -- host-side mirror of reference codetr/swin.py:23-749 (+ the
PatchEmbed / PatchMerging the reference imports from mmdet, equivalent source at
codetr/transformer_mmcv.py:100-316).  Same class names, constructor kwargs and parameter names
(``stages.{i}.blocks.{j}.attn.w_msa.qkv`` ...), eval-only, computing through ``codetr.hip_ops``.

Things that depend only on shapes -- the gathered relative-position bias [nH,144,144] and the
shifted-window mask [nW,144,144] -- are built once per (shape, dtype, device) and cached, instead
of being rebuilt from -100 constants on every call (reference :191-222).
"""
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops
from .transformer_layers import FFN, build_norm


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class PatchEmbed(nn.Module):
    """Non-overlapping conv patchify with bottom/right ("corner") zero padding + optional LN."""

    def __init__(self, in_channels=3, embed_dims=768, conv_type="Conv2d", kernel_size=16, stride=16, padding="corner",
                 dilation=1, bias=True, norm_cfg=None, input_size=None, init_cfg=None):
        super().__init__()
        if padding != "corner" or dilation != 1:
            raise NotImplementedError("PatchEmbed: only 'corner' padding, dilation 1 (what Swin uses)")
        self.embed_dims = embed_dims
        self.kernel_size, self.stride = to_2tuple(kernel_size), to_2tuple(stride or kernel_size)
        self.projection = nn.Conv2d(in_channels, embed_dims, self.kernel_size, self.stride, bias=bias)
        self.norm = build_norm(norm_cfg, embed_dims) if norm_cfg is not None else None

    def forward(self, x):
        H, W = x.shape[-2:]
        kh, kw = self.kernel_size
        sh, sw = self.stride
        if hip_ops.patch_embed_supported(x, self.projection.weight, self.stride):
            # non-overlapping patches: gather + GEMM, token-major output, zero padding inside the gather
            x, hw = hip_ops.patch_embed(x, self.projection.weight, self.projection.bias)
            if self.norm is not None:
                x = hip_ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
            return x, hw
        ph = max((-(-H // sh) - 1) * sh + kh - H, 0)
        pw = max((-(-W // sw) - 1) * sw + kw - W, 0)
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))
        x = hip_ops.conv2d(x, self.projection.weight, self.projection.bias, stride=self.stride)
        hw = (x.shape[2], x.shape[3])
        x = x.flatten(2).transpose(1, 2)
        if self.norm is not None:
            x = hip_ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x, hw


class PatchMerging(nn.Module):
    """2x2 neighbourhood -> 4C (nn.Unfold channel order: c*4 + ky*2 + kx) -> LN -> Linear(4C -> out)."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=None, padding="corner", dilation=1, bias=False,
                 norm_cfg=dict(type="LN"), init_cfg=None):
        super().__init__()
        if to_2tuple(kernel_size) != (2, 2) or to_2tuple(stride or kernel_size) != (2, 2) or padding != "corner":
            raise NotImplementedError("PatchMerging: only the 2x2 / stride-2 / corner-padded form Swin uses")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm = build_norm(norm_cfg, 4 * in_channels) if norm_cfg is not None else None
        self.reduction = nn.Linear(4 * in_channels, out_channels, bias=bias)

    def forward(self, x, input_size):
        B, L, C = x.shape
        H, W = input_size
        if L != H * W:
            raise AssertionError("input feature has wrong size")
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        if x.is_cuda and not torch.is_grad_enabled():
            # Same merge with the 4C axis ordered (ky, kx, c) instead of nn.Unfold's (c, ky, kx): the gather then moves
            # whole C-vectors (>= 384 contiguous bytes) instead of single elements, and LayerNorm / Linear see their
            # parameters permuted the same way (LN is invariant under a joint permutation; the Linear's columns follow).
            nw, nb, rw = self._permuted_params(C)
            if (self.norm is not None and isinstance(self.norm, nn.LayerNorm)
                    and hip_ops.patch_merge_layernorm_supported(x, C)):
                # gather + LayerNorm in one kernel: the merged map is never written un-normalised
                x = hip_ops.patch_merge_layernorm(x.reshape(B, -1, C), (2 * H2, 2 * W2), nw, nb, self.norm.eps)
                return hip_ops.linear(x, rw, self.reduction.bias), (H2, W2)
            x = x.view(B, H2, 2, W2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H2 * W2, 4 * C)
            if self.norm is not None:
                x = hip_ops.layer_norm(x, nw, nb, self.norm.eps)
            return hip_ops.linear(x, rw, self.reduction.bias), (H2, W2)
        x = x.view(B, H2, 2, W2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H2 * W2, 4 * C)
        if self.norm is not None:
            x = hip_ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return hip_ops.linear(x, self.reduction.weight, self.reduction.bias), (H2, W2)

    def _permuted_params(self, C):
        """(norm.weight, norm.bias, reduction.weight) with the 4C axis re-ordered from (c, ky, kx) to (ky, kx, c);
        rebuilt only when a parameter tensor changes"""
        ps = [self.reduction.weight] + ([self.norm.weight, self.norm.bias] if self.norm is not None else [])

        def build():
            # new index (k, c) <- old index c*4 + k
            idx = (torch.arange(C, device=ps[0].device)[None, :] * 4 + torch.arange(4, device=ps[0].device)[:, None]).reshape(-1)
            rw = self.reduction.weight.detach()[:, idx].contiguous()
            nw = self.norm.weight.detach()[idx].contiguous() if self.norm is not None else None
            nb = self.norm.bias.detach()[idx].contiguous() if self.norm is not None else None
            return nw, nb, rw

        return hip_ops.derived(ps, "_codetr_perm", build)


class WindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None, attn_drop_rate=0.0,
                 proj_drop_rate=0.0, init_cfg=None):
        super().__init__()
        self.embed_dims, self.window_size, self.num_heads = embed_dims, to_2tuple(window_size), num_heads
        head_dim = embed_dims // num_heads
        if qk_scale is not None and qk_scale != head_dim ** -0.5:
            raise NotImplementedError("custom qk_scale")
        self.scale = head_dim ** -0.5
        Wh, Ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        # index (i, j) -> row of the table for offset (yi - yj + Wh - 1, xi - xj + Ww - 1)
        coords = torch.stack(torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing="ij")).flatten(1)
        rel = coords[:, :, None] - coords[:, None, :]
        index = (rel[0] + Wh - 1) * (2 * Ww - 1) + (rel[1] + Ww - 1)
        self.register_buffer("relative_position_index", index.contiguous())
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = nn.Linear(embed_dims, embed_dims)

    def init_weights(self):
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def relative_position_bias(self):
        """[nH, N, N] gathered bias, cached on the table parameter until it changes (eval: never)."""
        t = self.relative_position_bias_table
        N = self.window_size[0] * self.window_size[1]
        return hip_ops.derived((t,), "_codetr_rel_bias", lambda: t[self.relative_position_index.view(-1)].view(N, N, -1)
                               .permute(2, 0, 1).contiguous().detach())

    def forward(self, x, mask=None):
        """x [nW*B, N, C] -> [nW*B, N, C]."""
        qkv = hip_ops.linear(x, self.qkv.weight, self.qkv.bias)
        o = hip_ops.window_attention(qkv, self.relative_position_bias(), mask, self.num_heads)
        return hip_ops.linear(o, self.proj.weight, self.proj.bias)


_MASK_CACHE = {}


def shifted_window_mask(Hp, Wp, ws, shift, dtype, device):
    """[nW, N, N] additive mask (0 / -100) separating the 9 regions of a cyclically shifted map."""
    key = (Hp, Wp, ws, shift, dtype, str(device))
    m = _MASK_CACHE.get(key)
    if m is None:
        region = torch.zeros(Hp, Wp, dtype=torch.float32)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                region[hs, wsl] = cnt
                cnt += 1
        r = region.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
        diff = r[:, None, :] - r[:, :, None]
        m = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).to(device=device, dtype=dtype)
        _MASK_CACHE[key] = m
    return m


class ShiftWindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None,
                 attn_drop_rate=0, proj_drop_rate=0, dropout_layer=dict(type="DropPath", drop_prob=0.0), init_cfg=None):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        if not 0 <= shift_size < window_size:
            raise AssertionError("shift_size must be in [0, window_size)")
        self.w_msa = WindowMSA(embed_dims, num_heads, to_2tuple(window_size), qkv_bias, qk_scale, attn_drop_rate,
                               proj_drop_rate)

    def takes_norm(self, query, hw_shape):
        """True when forward runs the fused window-attention path (the only one that accepts `pre_norm`)"""
        return hip_ops.swin_window_attention_supported(query, self.w_msa.embed_dims, self.w_msa.num_heads, self.window_size)

    def forward(self, query, hw_shape, identity=None, pre_norm=None):
        """query [B, H*W, C] (already normalised) -> attention branch output (+ `identity` if given).
        Padding to a multiple of the window is applied AFTER norm1: pad tokens are exact zeros, come
        out of qkv as the bias and take part in the softmax as ordinary keys (only the shift mask
        exists), then are cropped (reference :191-247)."""
        B, L, C = query.shape
        H, W = hw_shape
        if L != H * W:
            raise AssertionError("input feature has wrong size")
        ws, sh = self.window_size, self.shift_size
        m = self.w_msa
        if hip_ops.swin_window_attention_supported(query, C, m.num_heads, ws):
            # fused path: qkv GEMM on real tokens only, one kernel for everything between qkv and proj,
            # residual folded into the proj GEMM's epilogue
            if pre_norm is not None:   # LayerNorm of the rows inside the qkv GEMM (SwinBlock.forward)
                qkv = hip_ops.linear_ln(query, pre_norm[0], pre_norm[1], pre_norm[2], m.qkv.weight, m.qkv.bias)
            else:
                qkv = hip_ops.linear(query, m.qkv.weight, m.qkv.bias)
            o = hip_ops.swin_window_attention(qkv, m.qkv.bias, m.relative_position_bias(), hw_shape, m.num_heads, ws, sh)
            return hip_ops.linear(o, m.proj.weight, m.proj.bias, residual=identity)
        x = query.view(B, H, W, C)
        pad_r, pad_b = (-W) % ws, (-H) % ws
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, W + pad_r
        mask = None
        if sh > 0:
            x = torch.roll(x, shifts=(-sh, -sh), dims=(1, 2))
            mask = shifted_window_mask(Hp, Wp, ws, sh, query.dtype, query.device)
        win = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
        win = self.w_msa(win, mask)
        x = win.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if sh > 0:
            x = torch.roll(x, shifts=(sh, sh), dims=(1, 2))
        if pad_r or pad_b:
            x = x[:, :H, :W, :]
        x = x.reshape(B, H * W, C)
        return x if identity is None else x + identity


class SwinBlock(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, act_cfg=dict(type="GELU"),
                 norm_cfg=dict(type="LN"), with_cp=False, init_cfg=None):
        super().__init__()
        self.norm1 = build_norm(norm_cfg, embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias,
                                   qk_scale, attn_drop_rate, drop_rate)
        self.norm2 = build_norm(norm_cfg, embed_dims)
        self.ffn = FFN(embed_dims=embed_dims, feedforward_channels=feedforward_channels, num_fcs=2, ffn_drop=drop_rate,
                       act_cfg=act_cfg, add_identity=True)

    def forward(self, x, hw_shape):
        n1, n2 = self.norm1, self.norm2
        mode = getattr(self, "fp8_mode", None)
        if mode is not None and x.is_cuda and x.dtype == torch.float16 and self.attn.takes_norm(x, hw_shape):
            rows = x.shape[0] * x.shape[1]
            if mode == "calibrate":
                return self._forward_fp8_calibrate(x, hw_shape)
            if mode == "run" and all(hip_ops.linear_fp8_supported(rows, w) for w in self._fp8_weights()):
                return self._forward_fp8(x, hw_shape)
            if mode == "mx":
                ops = getattr(self, "fp8_ops", None) or ("qkv", "proj", "fc1", "fc2")
                ws = dict(zip(("qkv", "proj", "fc1", "fc2"), self._fp8_weights()))
                if ops and all(hip_ops.linear_fp8_supported(rows, ws[o]) for o in ops):
                    return self._forward_fp8mx(x, hw_shape)
        if (isinstance(n1, nn.LayerNorm) and self.attn.takes_norm(x, hw_shape)
                and hip_ops.linear_ln_supported(x, n1.weight, self.attn.w_msa.qkv.weight)):
            # norm1 folded into the qkv GEMM's operand load (only that GEMM reads the normalised rows)
            x = self.attn(x, hw_shape, identity=x, pre_norm=(n1.weight, n1.bias, n1.eps))
        else:
            h = hip_ops.layer_norm(x, n1.weight, n1.bias, n1.eps)
            x = self.attn(h, hw_shape, identity=x)
        fc1, fc2 = self.ffn.layers[0][0], self.ffn.layers[1]
        if (isinstance(n2, nn.LayerNorm) and self.ffn.add_identity and fc1.bias is not None and fc2.bias is not None
                and hip_ops.swin_mlp_supported(x, n2.weight, fc1.weight, fc2.weight, self.ffn.act)):
            # stages 0 / 1: norm2, fc1, GELU, fc2 and the identity in ONE launch (the hidden activation stays on-chip)
            return hip_ops.swin_mlp(x, n2.weight, n2.bias, n2.eps, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        if (isinstance(n2, nn.LayerNorm) and self.ffn.add_identity
                and hip_ops.linear_ln_supported(x, n2.weight, fc1.weight)):
            # norm2 folded into fc1 (+ GELU); fc2 adds the identity as before
            h = hip_ops.linear_ln(x, n2.weight, n2.bias, n2.eps, fc1.weight, fc1.bias, act=self.ffn.act)
            return hip_ops.linear(h, fc2.weight, fc2.bias, residual=x)
        h = hip_ops.layer_norm(x, n2.weight, n2.bias, n2.eps)
        return self.ffn(h, identity=x)


    # ---- fp8 (BASELINE config 5): the block's four Linears on the e4m3 MFMA path, see codetr/fp8.py -------------
    def _fp8_weights(self):
        m = self.attn.w_msa
        return (m.qkv.weight, m.proj.weight, self.ffn.layers[0][0].weight, self.ffn.layers[1].weight)

    def _forward_fp8_calibrate(self, x, hw_shape):
        """the fp16 block, unfused, recording the absolute maxima of the four GEMM inputs (running max over calls)"""
        n1, n2, m = self.norm1, self.norm2, self.attn.w_msa
        fc1, fc2 = self.ffn.layers[0][0], self.ffn.layers[1]
        amax = self.__dict__.setdefault("_fp8_amax", {})

        def rec(key, t):
            v = t.detach().abs().amax().float()
            amax[key] = v if key not in amax else torch.maximum(amax[key], v)

        h = hip_ops.layer_norm(x, n1.weight, n1.bias, n1.eps)
        rec("ln1", h)
        qkv = hip_ops.linear(h, m.qkv.weight, m.qkv.bias)
        o = hip_ops.swin_window_attention(qkv, m.qkv.bias, m.relative_position_bias(), hw_shape, m.num_heads,
                                          self.attn.window_size, self.attn.shift_size)
        rec("attn", o)
        x = hip_ops.linear(o, m.proj.weight, m.proj.bias, residual=x)
        h = hip_ops.layer_norm(x, n2.weight, n2.bias, n2.eps)
        rec("ln2", h)
        g = hip_ops.linear(h, fc1.weight, fc1.bias, act=self.ffn.act)
        rec("act", g)
        return hip_ops.linear(g, fc2.weight, fc2.bias, residual=x)

    def _forward_fp8(self, x, hw_shape):
        """norm1 -> e4m3 | qkv (fp8 GEMM, fp16 out) | window attention (fp16 in, e4m3 out) | proj (fp8, + identity) |
        norm2 -> e4m3 | fc1 (fp8, GELU, e4m3 out) | fc2 (fp8, + identity): the residual stream stays fp16"""
        n1, n2, m = self.norm1, self.norm2, self.attn.w_msa
        fc1, fc2 = self.ffn.layers[0][0], self.ffn.layers[1]
        sc = self._fp8_scales
        h8 = hip_ops.layer_norm_fp8(x, n1.weight, n1.bias, n1.eps, sc["ln1"])
        qkv = hip_ops.linear_fp8(h8, sc["ln1"], m.qkv.weight, m.qkv.bias)
        o8 = hip_ops.swin_window_attention(qkv, m.qkv.bias, m.relative_position_bias(), hw_shape, m.num_heads,
                                           self.attn.window_size, self.attn.shift_size, out_scale=sc["attn"])
        x = hip_ops.linear_fp8(o8, sc["attn"], m.proj.weight, m.proj.bias, residual=x)
        h8 = hip_ops.layer_norm_fp8(x, n2.weight, n2.bias, n2.eps, sc["ln2"])
        g8 = hip_ops.linear_fp8(h8, sc["ln2"], fc1.weight, fc1.bias, act=self.ffn.act, out_scale=sc["act"])
        return hip_ops.linear_fp8(g8, sc["act"], fc2.weight, fc2.bias, residual=x)


    def _forward_fp8mx(self, x, hw_shape):
        """the same chain with MX block scales on every activation (one e8m0 exponent per 32 channels, chosen by the
        producer from the block's own maximum; applied by the scaled MFMA in hardware): no calibration, no static scale.
        ``self.fp8_ops`` (default: all four) names the GEMMs of this block that run in e4m3 -- the others stay on the fp16
        kernels, with the producer in front of each chosen accordingly (codetr/fp8.py: the shipped selection comes from the
        sensitivity map, profiles/r04_fp8_sensitivity.json)"""
        n1, n2, m = self.norm1, self.norm2, self.attn.w_msa
        fc1, fc2 = self.ffn.layers[0][0], self.ffn.layers[1]
        ops = getattr(self, "fp8_ops", None) or ("qkv", "proj", "fc1", "fc2")
        if "qkv" in ops:
            h8, hs = hip_ops.layer_norm_fp8mx(x, n1.weight, n1.bias, n1.eps)
            qkv = hip_ops.linear_fp8mx(h8, hs, m.qkv.weight, m.qkv.bias)
        elif hip_ops.linear_ln_supported(x, n1.weight, m.qkv.weight):
            qkv = hip_ops.linear_ln(x, n1.weight, n1.bias, n1.eps, m.qkv.weight, m.qkv.bias)
        else:
            qkv = hip_ops.linear(hip_ops.layer_norm(x, n1.weight, n1.bias, n1.eps), m.qkv.weight, m.qkv.bias)
        if "proj" in ops:
            o8, os_ = hip_ops.swin_window_attention(qkv, m.qkv.bias, m.relative_position_bias(), hw_shape, m.num_heads,
                                                    self.attn.window_size, self.attn.shift_size, out_mx=True)
            x = hip_ops.linear_fp8mx(o8, os_, m.proj.weight, m.proj.bias, residual=x)
        else:
            o = hip_ops.swin_window_attention(qkv, m.qkv.bias, m.relative_position_bias(), hw_shape, m.num_heads,
                                              self.attn.window_size, self.attn.shift_size)
            x = hip_ops.linear(o, m.proj.weight, m.proj.bias, residual=x)
        if "fc1" in ops:
            h8, hs = hip_ops.layer_norm_fp8mx(x, n2.weight, n2.bias, n2.eps)
            if "fc2" in ops:
                g8, gs = hip_ops.linear_fp8mx(h8, hs, fc1.weight, fc1.bias, act=self.ffn.act, out_mx=True)
            else:
                g = hip_ops.linear_fp8mx(h8, hs, fc1.weight, fc1.bias, act=self.ffn.act)
        else:
            if hip_ops.linear_ln_supported(x, n2.weight, fc1.weight):
                g = hip_ops.linear_ln(x, n2.weight, n2.bias, n2.eps, fc1.weight, fc1.bias, act=self.ffn.act)
            else:
                g = hip_ops.linear(hip_ops.layer_norm(x, n2.weight, n2.bias, n2.eps), fc1.weight, fc1.bias, act=self.ffn.act)
            if "fc2" in ops:
                g8, gs = hip_ops.cast_fp8mx(g)
        if "fc2" in ops:
            return hip_ops.linear_fp8mx(g8, gs, fc2.weight, fc2.bias, residual=x)
        return hip_ops.linear(g, fc2.weight, fc2.bias, residual=x)


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size=7, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, downsample=None, act_cfg=dict(type="GELU"),
                 norm_cfg=dict(type="LN"), with_cp=False, init_cfg=None):
        super().__init__()
        self.blocks = nn.ModuleList(
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, shift=bool(i % 2), qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, act_cfg=act_cfg,
                      norm_cfg=norm_cfg) for i in range(depth))
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for blk in self.blocks:
            x = blk(x, hw_shape)
        if self.downsample is not None:
            x_down, down_hw = self.downsample(x, hw_shape)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                 use_abs_pos_embed=False, act_cfg=dict(type="GELU"), norm_cfg=dict(type="LN"), with_cp=False,
                 pretrained=None, convert_weights=False, frozen_stages=-1, init_cfg=None):
        super().__init__()
        if init_cfg and pretrained:
            raise AssertionError("init_cfg and pretrained cannot be specified at the same time")
        if isinstance(pretrained, str):
            warnings.warn('DeprecationWarning: pretrained is deprecated, please use "init_cfg" instead')
            init_cfg = dict(type="Pretrained", checkpoint=pretrained)
        elif pretrained is not None:
            raise TypeError("pretrained must be a str or None")
        if use_abs_pos_embed:
            raise NotImplementedError("absolute position embedding is not used by the Co-DETR configs")
        self.init_cfg, self.convert_weights, self.frozen_stages = init_cfg, convert_weights, frozen_stages
        self.out_indices = tuple(out_indices)
        if strides[0] != patch_size:
            raise AssertionError("Use non-overlapping patch embed.")
        self.patch_embed = PatchEmbed(in_channels, embed_dims, "Conv2d", patch_size, strides[0],
                                      norm_cfg=norm_cfg if patch_norm else None)
        self.drop_after_pos = nn.Dropout(p=drop_rate)
        self.stages = nn.ModuleList()
        C = embed_dims
        n = len(depths)
        for i in range(n):
            down = PatchMerging(C, 2 * C, stride=strides[i + 1], norm_cfg=norm_cfg if patch_norm else None) \
                if i < n - 1 else None
            self.stages.append(SwinBlockSequence(C, num_heads[i], mlp_ratio * C, depths[i], window_size, qkv_bias,
                                                 qk_scale, drop_rate, attn_drop_rate, 0.0, down, act_cfg, norm_cfg))
            if down is not None:
                C = down.out_channels
        self.num_features = [int(embed_dims * 2 ** i) for i in range(n)]
        for i in self.out_indices:
            self.add_module(f"norm{i}", build_norm(norm_cfg, self.num_features[i]))

    def init_weights(self):
        """Random init (no checkpoint): truncated-normal linears, unit LayerNorms (reference :660-667).
        Loading published weights goes through ``codetr.checkpoint`` instead."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, WindowMSA):
                m.init_weights()

    def load_pretrained(self, checkpoint):
        """Backbone initialisation from a pretrained checkpoint (reference :670-723): ``state_dict`` / ``model`` /
        bare dict; official-Swin key names converted when ``convert_weights`` (swin_converter); ``backbone.`` and
        ``module.`` prefixes stripped; relative-position bias tables of another window size resized bicubically
        (:705-720: [L1, nH] -> [1, nH, S1, S1] -> bicubic to S2 x S2 -> [L2, nH]); non-strict load."""
        sd = checkpoint.get("state_dict", checkpoint.get("model", checkpoint)) if isinstance(checkpoint, dict) else checkpoint
        if self.convert_weights:
            sd = swin_converter(sd)
        state = OrderedDict()
        for k, v in sd.items():
            if k.startswith("backbone."):
                state[k[9:]] = v
        if not state:   # a bare backbone state dict
            state = OrderedDict(sd)
        if state and next(iter(state)).startswith("module."):
            state = OrderedDict((k[7:], v) for k, v in state.items())
        own = self.state_dict()
        for key in [k for k in state if "relative_position_bias_table" in k and k in own]:
            pre, cur = state[key], own[key]
            (L1, nH1), (L2, nH2) = pre.shape, cur.shape
            if nH1 != nH2:
                warnings.warn(f"Error in loading {key}, pass")
                state.pop(key)
            elif L1 != L2:
                S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
                r = F.interpolate(pre.permute(1, 0).reshape(1, nH1, S1, S1).float(), size=(S2, S2), mode="bicubic")
                state[key] = r.view(nH2, L2).permute(1, 0).contiguous().to(pre.dtype)
        # the index buffers are functions of the window size alone (rebuilt by the constructor): a checkpoint's copy
        # for another window size cannot be loaded and is not needed
        for key in [k for k in state if k.endswith("relative_position_index") and k in own
                    and tuple(state[k].shape) != tuple(own[k].shape)]:
            state.pop(key)
        return self.load_state_dict(state, strict=False)

    def forward_tokens(self, x):
        """[B,3,H,W] -> list of (tokens [B, H_i*W_i, C_i], (H_i, W_i)) for i in out_indices: the stage outputs
        after their output norm, in the token-major layout they are computed in."""
        x, hw = self.patch_embed(x)
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw, out, out_hw = stage(x, hw)
            if i in self.out_indices:
                n = getattr(self, f"norm{i}")
                outs.append((hip_ops.layer_norm(out, n.weight, n.bias, n.eps), out_hw))
        return outs

    def forward(self, x):
        """[B,3,H,W] -> list of [B, C_i, H/2^(i+2), W/2^(i+2)] for i in out_indices."""
        return [t.view(-1, *hw, t.shape[-1]).permute(0, 3, 1, 2).contiguous() for t, hw in self.forward_tokens(x)]


def swin_converter(ckpt):
    """Official Swin checkpoint -> mmdet key names / unfold channel order (reference :752-803):
    ``layers``->``stages``, ``attn.``->``attn.w_msa.``, ``mlp.fc1/fc2``->``ffn.layers.0.0/1``,
    ``patch_embed.proj``->``projection``; PatchMerging reduction/norm channels regrouped from
    (x0,x1,x2,x3) concatenation order to nn.Unfold's c*4 + k order."""
    out = OrderedDict()

    def regroup(t, last_dim):
        C4 = t.shape[-1] if last_dim else t.shape[0]
        g = t.reshape(*t.shape[:-1], 4, C4 // 4) if last_dim else t.reshape(4, C4 // 4)
        g = g[..., [0, 2, 1, 3], :] if last_dim else g[[0, 2, 1, 3], :]
        return g.transpose(-1, -2).reshape(t.shape)

    for k, v in ckpt.items():
        if k.startswith("head"):
            continue
        nk, nv = k, v
        if k.startswith("layers"):
            if "attn." in k:
                nk = k.replace("attn.", "attn.w_msa.")
            elif "mlp.fc1." in k:
                nk = k.replace("mlp.fc1.", "ffn.layers.0.0.")
            elif "mlp.fc2." in k:
                nk = k.replace("mlp.fc2.", "ffn.layers.1.")
            elif "mlp." in k:
                nk = k.replace("mlp.", "ffn.")
            elif "downsample" in k:
                if "reduction." in k:
                    nv = regroup(v, last_dim=True)
                elif "norm." in k:
                    nv = regroup(v, last_dim=False)
            nk = nk.replace("layers", "stages", 1)
        elif k.startswith("patch_embed") and "proj" in k:
            nk = k.replace("proj", "projection")
        out["backbone." + nk] = nv
    return out
