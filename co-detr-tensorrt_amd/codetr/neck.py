"""``ChannelMapper`` neck -- restated from mmdet v3.3.0 semantics (third party for the reference,
built through ``MODELS.build`` at reference codetr/codetr.py:53-54 from configs lsj:40-47 + swin:29):
one k x k conv (no bias when a norm follows) + GroupNorm per input level, and ``num_outs - len(in)``
extra 3x3 stride-2 conv + GroupNorm levels, the first of which reads the RAW last input.
Parameter names follow mmcv's ConvModule: ``convs.{i}.conv.weight``, ``convs.{i}.gn.{weight,bias}``,
``extra_convs.{i}.conv.weight``, ``extra_convs.{i}.gn.*``."""
import torch.nn as nn

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

    def forward(self, inputs):
        if len(inputs) != len(self.convs):
            raise AssertionError("ChannelMapper: wrong number of input levels")
        outs = [conv(x) for conv, x in zip(self.convs, inputs)]
        if self.extra_convs:
            for i, conv in enumerate(self.extra_convs):
                outs.append(conv(inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)
