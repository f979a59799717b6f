"""ResNet-50 backbone for config 1 ("Co-DINO 5-scale R50").  The reference cannot instantiate it
(``CoDETR.__init__`` asserts a Swin backbone, reference codetr/codetr.py:51); the R50 model exists
there only as config (configs lsj:30-39: mmdet ``ResNet`` depth 50, ``style='pytorch'``, frozen BN,
out_indices 0-3).  Restated from mmdet v3.3.0 / torchvision semantics with their parameter names
(``conv1``, ``bn1``, ``layer{1..4}.{b}.conv{1,2,3}/bn{1,2,3}/downsample.{0,1}``).  Inference only:
BatchNorm always uses its running statistics."""
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # style='pytorch': stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    @staticmethod
    def _bn(bn, x):
        return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)

    def forward(self, x):
        idt = x
        y = F.relu(self._bn(self.bn1, hip_ops.conv2d(x, self.conv1.weight)))
        y = F.relu(self._bn(self.bn2, hip_ops.conv2d(y, self.conv2.weight, None, self.stride, 1)))
        y = self._bn(self.bn3, hip_ops.conv2d(y, self.conv3.weight))
        if self.downsample is not None:
            idt = self._bn(self.downsample[1], hip_ops.conv2d(x, self.downsample[0].weight, None, self.stride, 0))
        return F.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_cfg=None,
                 norm_eval=True, style="pytorch", init_cfg=None, **kwargs):
        super().__init__()
        if depth != 50 or style != "pytorch" or num_stages != 4:
            raise NotImplementedError("only ResNet-50, style='pytorch' (the Co-DETR R50 config)")
        self.out_indices = tuple(out_indices)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and li > 0) else 1
                down = None
                if b == 0:
                    down = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
                blocks.append(_Bottleneck(cin, planes, stride, down))
                cin = planes * 4
            self.add_module(f"layer{li + 1}", nn.Sequential(*blocks))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = F.relu(_Bottleneck._bn(self.bn1, hip_ops.conv2d(x, self.conv1.weight, None, 2, 3)))
        x = F.max_pool2d(x, 3, 2, 1)
        outs = []
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in self.out_indices:
                outs.append(x)
        return outs
