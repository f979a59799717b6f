"""``ChannelMapper`` neck -- restated from mmdet v3.3.0 semantics (third party for the reference,
built through ``MODELS.build`` at reference codetr/codetr.py:53-54 from configs lsj:40-47 + swin:29):
one k x k conv (no bias when a norm follows) + GroupNorm per input level, and ``num_outs - len(in)``
extra 3x3 stride-2 conv + GroupNorm levels, the first of which reads the RAW last input.
Parameter names follow mmcv's ConvModule: ``convs.{i}.conv.weight``, ``convs.{i}.gn.{weight,bias}``,
``extra_convs.{i}.conv.weight``, ``extra_convs.{i}.gn.*``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops


class _ConvNorm(nn.Module):
    def __init__(self, cin, cout, k, stride, padding, norm_cfg, act_cfg, bias):
        super().__init__()
        if act_cfg is not None:
            raise NotImplementedError("ChannelMapper with activation is not used by the Co-DETR configs")
        with_norm = norm_cfg is not None
        self.stride, self.padding = stride, padding
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, bias=(not with_norm) if bias == "auto" else bias)
        self.groups = None
        if with_norm:
            cfg = dict(norm_cfg)
            if cfg.pop("type") != "GN":
                raise NotImplementedError("ChannelMapper: only GroupNorm")
            self.groups = cfg["num_groups"]
            self.gn = nn.GroupNorm(self.groups, cout)

    def forward(self, x):
        y = hip_ops.conv2d(x, self.conv.weight, self.conv.bias, self.stride, self.padding)
        if self.groups is not None:
            y = hip_ops.group_norm(y, self.groups, self.gn.weight, self.gn.bias, self.gn.eps)
        return y


class ChannelMapper(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), bias="auto", num_outs=None, init_cfg=None):
        super().__init__()
        if num_outs is None:
            num_outs = len(in_channels)
        self.convs = nn.ModuleList(
            _ConvNorm(c, out_channels, kernel_size, 1, (kernel_size - 1) // 2, norm_cfg, act_cfg, bias)
            for c in in_channels)
        self.extra_convs = None
        if num_outs > len(in_channels):
            self.extra_convs = nn.ModuleList()
            for i in range(len(in_channels), num_outs):
                cin = in_channels[-1] if i == len(in_channels) else out_channels
                self.extra_convs.append(_ConvNorm(cin, out_channels, 3, 2, 1, norm_cfg, act_cfg, bias))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def tokens_supported(self, token_feats):
        """True when every level can take the token-major path: 1x1 convs followed by GN with 8 channels per group."""
        x0 = token_feats[0][0]
        convs = list(self.convs) + list(self.extra_convs or [])
        return (len(self.extra_convs or []) <= 1
                and all(c.conv.kernel_size == (1, 1) for c in self.convs)
                and all(c.groups is not None and c.conv.bias is None
                        and hip_ops.groupnorm_tokens_supported(x0.new_empty(1, 1, c.conv.out_channels), c.groups)
                        for c in convs))

    def forward_tokens(self, token_feats):
        """token_feats: list of (tokens [B, HW_i, C_i], (H_i, W_i)) from the backbone ->
        (flat [B, S, out_channels] with the levels concatenated in order, list of (H_l, W_l)).
        The 1x1 convs run as linears over tokens (native GEMM), GroupNorm writes each level directly into its
        slice of `flat`; only the stride-2 3x3 extra level(s) go through NCHW (1/16 of the pixels of level 0)."""
        if len(token_feats) != len(self.convs):
            raise AssertionError("ChannelMapper: wrong number of input levels")
        shapes = [hw for _, hw in token_feats]
        n_extra = len(self.extra_convs) if self.extra_convs else 0
        h, w = shapes[-1]
        for _ in range(n_extra):
            h, w = (h + 1) // 2, (w + 1) // 2
            shapes.append((h, w))
        B = token_feats[0][0].shape[0]
        Cout = self.convs[0].conv.out_channels
        S = sum(a * b for a, b in shapes)
        flat = token_feats[0][0].new_empty(B, S, Cout)
        start = 0
        for conv, (t, hw) in zip(self.convs, token_feats):
            y = hip_ops.linear(t, conv.conv.weight.view(Cout, -1), None)
            hip_ops.groupnorm_tokens_into(y, conv.gn.weight, conv.gn.bias, conv.groups, conv.gn.eps, flat, start)
            start += hw[0] * hw[1]
        if n_extra:
            # 3x3 stride-2 conv on the RAW last backbone level as im2col + native GEMM (K = 9*C_in), then the same GN
            t, hw = token_feats[-1]
            conv = self.extra_convs[0]
            # patches gathered in token layout, K ordered (ky, kx, c): nine strided slices of whole C-vectors, one cat
            # (F.unfold's (c, ky, kx) order needs an NCHW round trip and an element-granular transpose: 85 us vs ~20)
            Ho, Wo = (hw[0] + 1) // 2, (hw[1] + 1) // 2
            cols = hip_ops.im2col_tokens(t.view(B, hw[0], hw[1], -1), 3, 2, 1)
            y = hip_ops.linear(cols, self._extra_weight_kkc(conv.conv.weight), None)
            hip_ops.groupnorm_tokens_into(y, conv.gn.weight, conv.gn.bias, conv.groups, conv.gn.eps, flat, start)
        return flat, shapes

    def _extra_weight_kkc(self, w):
        """[Cout, Cin, 3, 3] conv weight as a [Cout, 9*Cin] GEMM weight with K ordered (ky, kx, c); cached"""
        return hip_ops.derived((w,), "_codetr_kkc", lambda: w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())

    def forward(self, inputs):
        if len(inputs) != len(self.convs):
            raise AssertionError("ChannelMapper: wrong number of input levels")
        outs = [conv(x) for conv, x in zip(self.convs, inputs)]
        if self.extra_convs:
            for i, conv in enumerate(self.extra_convs):
                outs.append(conv(inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)
