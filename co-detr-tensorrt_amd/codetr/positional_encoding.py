"""``SinePositionalEncoding`` -- mirror of reference codetr/positional_encoding.py:11-103."""
import math

import torch
import torch.nn as nn


class SinePositionalEncoding(nn.Module):
    def __init__(self, num_feats: int, temperature: int = 10000, normalize: bool = False,
                 scale: float = 2 * math.pi, eps: float = 1e-6, offset: float = 0.0, init_cfg=None):
        super().__init__()
        if normalize and not isinstance(scale, (float, int)):
            raise AssertionError(f"when normalize is set, scale should be float or int, found {type(scale)}")
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """mask [B,H,W], non-zero = padding -> [B, 2*num_feats, H, W] (y features first).
        The running sums are accumulated in `dtype`, as the reference does (:78-79)."""
        B, H, W = mask.shape
        return self.forward_tokens(mask, dtype).view(B, H, W, -1).permute(0, 3, 1, 2)

    def forward_tokens(self, mask: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """Same encoding in the layout the transformer consumes: [B, H*W, 2*num_feats]."""
        B, H, W = mask.shape
        valid = 1 - mask.to(torch.int)
        y = valid.cumsum(1, dtype=dtype)
        x = valid.cumsum(2, dtype=dtype)
        if self.normalize:
            y = (y + self.offset) / (y[:, -1:, :] + self.eps) * self.scale
            x = (x + self.offset) / (x[:, :, -1:] + self.eps) * self.scale
        i = torch.arange(self.num_feats, dtype=dtype, device=mask.device)
        dim_t = self.temperature ** (2 * (i // 2) / self.num_feats)
        px = x[..., None] / dim_t
        py = y[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((py, px), dim=3).view(B, H * W, -1)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(num_feats={self.num_feats}, temperature={self.temperature}, "
                f"normalize={self.normalize}, scale={self.scale}, eps={self.eps})")
