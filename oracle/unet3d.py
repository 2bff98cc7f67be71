"""Oracle: the UNet3D baseline model (TEST INFRASTRUCTURE).

Restates ``/root/reference/src/cryovit/models/unet3d.py``:
  * ``UNet3D.__init__``        l.15-47   three analysis blocks (1->16->64->256), bottom (256->384->256), three synthesis
                                         blocks, 1x1x1 output layer, PAD = 16
  * ``UNet3D.forward_volume``  l.49-71   skips popped in reverse; upconv -> channel concat -> layers; clip(+-5)
  * ``UNet3D.forward``         l.73-96   [B,D,C,H,W] -> [B,C,D,H,W], zero-pad every axis to a multiple of 16, crop, sigmoid
  * ``AnalysisBlock``          l.113-148 (Conv3d k3 + InstanceNorm3d(eps 1e-3, affine) + GELU) x2; pool = Conv3d k2 s2 + IN + GELU
  * ``SynthesisBlock``         l.151-191 upconv = ConvTranspose3d k2 s2 + IN + GELU; layers = Linear over channels + IN + GELU +
                                         Conv3d k3 + IN + GELU
  * ``LinearProjection``       l.194-216
Attribute names match, so ``state_dict`` keys are the reference's.  ``widths`` builds a narrow member of the same family for
fast CPU fixtures; the default is the reference's (16, 64, 256 | 384).
"""

from __future__ import annotations

import math

import torch
from torch import Tensor, nn

REF_WIDTHS = ((16, 64, 256), 384)
NARROW_WIDTHS = ((8, 16, 32), 48)


def _in(c: int) -> nn.InstanceNorm3d:
    return nn.InstanceNorm3d(c, eps=1e-3, affine=True)


class AnalysisBlock(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.pool = nn.Sequential(nn.Conv3d(cout, cout, 2, stride=2), _in(cout), nn.GELU())
        self.layers = nn.Sequential(nn.Conv3d(cin, cout, 3, padding="same"), _in(cout), nn.GELU(),
                                    nn.Conv3d(cout, cout, 3, padding="same"), _in(cout), nn.GELU())

    def forward(self, x: Tensor):
        x = self.layers(x)
        return self.pool(x), x


class LinearProjection(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.proj = nn.Linear(cin, cout)

    def forward(self, x: Tensor) -> Tensor:
        return self.proj(x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)


class SynthesisBlock(nn.Module):
    def __init__(self, cin: int, cskip: int, cout: int) -> None:
        super().__init__()
        self.upconv = nn.Sequential(nn.ConvTranspose3d(cin, cout, 2, stride=2), _in(cout), nn.GELU())
        self.layers = nn.Sequential(LinearProjection(cout + cskip, cout), _in(cout), nn.GELU(),
                                    nn.Conv3d(cout, cout, 3, padding="same"), _in(cout), nn.GELU())


class UNet3D(nn.Module):
    def __init__(self, widths=REF_WIDTHS) -> None:
        super().__init__()
        (a1, a2, a3), cb = widths
        self.bottom_layer = nn.Sequential(nn.Conv3d(a3, cb, 3, padding="same"), _in(cb), nn.GELU(),
                                          nn.Conv3d(cb, a3, 3, padding="same"), _in(a3), nn.GELU())
        self.analysis_layers = nn.ModuleList([AnalysisBlock(1, a1), AnalysisBlock(a1, a2), AnalysisBlock(a2, a3)])
        self.synthesis_layers = nn.ModuleList([SynthesisBlock(a3, a3, a2), SynthesisBlock(a2, a2, a1), SynthesisBlock(a1, a1, a1)])
        self.output_layer = nn.Conv3d(a1, 1, 1, padding="same")
        self.PAD = max(16, 2 ** len(self.analysis_layers))

    def forward_volume(self, x: Tensor) -> Tensor:
        skips = []
        for block in self.analysis_layers:
            x, prev = block(x)
            skips.append(prev)
        x = self.bottom_layer(x)
        for block in self.synthesis_layers:
            x = block.upconv(x)
            x = torch.cat([x, skips.pop()], 1)
            x = block.layers(x)
        return torch.clip(self.output_layer(x), -5.0, 5.0)

    def forward_tomo_batch(self, tomo_batch: Tensor) -> Tensor:
        """unet3d.py:73-96 on ``batch.tomo_batch`` [B, D, C=1, H, W] -> probabilities [B, D, H, W]."""
        x = tomo_batch.permute(0, 2, 1, 3, 4)
        D, H, W = x.shape[-3:]
        new = [self.PAD * math.ceil(d / self.PAD) for d in (D, H, W)]
        if new != [D, H, W]:
            xp = torch.zeros(*x.shape[:-3], *new, dtype=x.dtype)
            xp[..., :D, :H, :W] = x
            x = xp
        x = self.forward_volume(x)[..., :D, :H, :W]
        return torch.sigmoid(x.squeeze(1))


def rescaled_init_(model: UNet3D, seed: int) -> None:
    """Seeded variance-preserving init with non-trivial norm affines (PyTorch's default leaves every InstanceNorm at weight 1 /
    bias 0 and the output near 0.5 everywhere)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel() if not name.startswith("synthesis_layers") or "upconv.0" not in name else p.shape[0] * 8 / 8
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(2.0 / max(1.0, float(fan_in))))
            elif name.endswith(("1.weight", "4.weight")) and p.dim() == 1 and "proj" not in name:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        model.output_layer.weight.mul_(3.0)
