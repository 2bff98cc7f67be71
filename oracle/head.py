"""Oracle: CryoVIT 3D-conv segmentation head (TEST INFRASTRUCTURE).

Restates ``/root/reference/src/cryovit/models/cryovit.py``:
  * ``CryoVIT.__init__``      l.13-34  layer stack
  * ``CryoVIT.forward_volume`` l.36-40 layers -> output_layer -> clip(+-5)
  * ``CryoVIT.forward``       l.42-49  permute, squeeze, sigmoid
  * ``SynthesisBlock``        l.52-83  GN(max(8,c1//8), eps 1e-3) -> dilated
    Conv3d+GELU x2 -> ConvTranspose3d (1,2,2)+GELU
Attribute names match so ``state_dict`` keys are the reference's
(``layers.0``, ``layers.{2..5}.layers.{0,1,3,5}``, ``output_layer.{0,2}``).
``widths`` lets tests build a narrow member of the same family; the default is
the reference's.
"""

from __future__ import annotations

import math

import torch
from torch import Tensor, nn

# (c_in, [(c1, c2, c3, d1, d2) x4], c_tail) -- cryovit.py:18-34
REF_WIDTHS = (1536, ((1024, 192, 128, 32, 24), (128, 64, 32, 16, 12), (32, 32, 32, 8, 4), (32, 16, 8, 2, 1)), 8)
# narrow variant for fast CPU fixtures: same dilations/structure, ~channels / 8
# (multiples of 8 so GroupNorm group sizes stay integral; c_in = the tiny ViT's dim)
NARROW_WIDTHS = (128, ((128, 24, 16, 32, 24), (16, 8, 8, 16, 12), (8, 8, 8, 8, 4), (8, 8, 8, 2, 1)), 8)


class SynthesisBlock(nn.Module):
    """cryovit.py:52-83."""

    def __init__(self, c1: int, c2: int, c3: int, d1: int, d2: int) -> None:
        super().__init__()
        self.layers = nn.Sequential(
            nn.GroupNorm(max(8, c1 // 8), c1, eps=1e-3),
            nn.Conv3d(c1, c2, 3, padding="same", dilation=(d1, 1, 1)),
            nn.GELU(),
            nn.Conv3d(c2, c2, 3, padding="same", dilation=(d2, 1, 1)),
            nn.GELU(),
            nn.ConvTranspose3d(c2, c3, (1, 2, 2), stride=(1, 2, 2)),
            nn.GELU(),
        )

    def forward(self, x: Tensor) -> Tensor:
        return self.layers(x)


class CryoVITHead(nn.Module):
    """cryovit.py:10-49 without the Lightning base class."""

    def __init__(self, widths=REF_WIDTHS, block_cls=SynthesisBlock) -> None:
        super().__init__()
        c_in, blocks, c_tail = widths
        assert blocks[-1][2] == c_tail
        self.layers = nn.Sequential(
            nn.Conv3d(c_in, blocks[0][0], 1, padding="same"),
            nn.GELU(),
            *[block_cls(*b) for b in blocks],
        )
        self.output_layer = nn.Sequential(
            nn.Conv3d(c_tail, c_tail, 3, padding="same"),
            nn.GELU(),
            nn.Conv3d(c_tail, 1, 3, padding="same"),
        )

    def forward_volume(self, x: Tensor) -> Tensor:  # [B,C,D,h,w] -> [B,1,D,16h,16w]
        x = self.layers(x)
        x = self.output_layer(x)
        return torch.clip(x, -5.0, 5.0)

    def forward_tomo_batch(self, tomo_batch: Tensor) -> Tensor:  # [B,D,C,h,w] -> probs [B,D,H,W]
        x = tomo_batch.permute(0, 2, 1, 3, 4)
        x = self.forward_volume(x)
        return torch.sigmoid(x.squeeze(1))


def rescaled_init_(head: nn.Module, seed: int, out_gain: float = 2.0) -> None:
    """Variance-preserving synthetic init (SURVEY App. E / BASELINE.md s.3).

    PyTorch's default init gives logits ~ -0.03 +- 0.005 => all-background and
    a degenerate Dice == 0.  Here every conv gets N(0, gain/fan_in) weights
    (gain 2 ahead of a GELU), small biases, GroupNorm affine ~ 1 + N(0,0.1);
    the last conv is scaled so logits have O(1) spread with both signs.
    """
    g = torch.Generator().manual_seed(seed)
    convs = [m for m in head.modules() if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d))]
    for m in head.modules():
        if isinstance(m, nn.Conv3d):
            fan_in = m.in_channels * math.prod(m.kernel_size)
        elif isinstance(m, nn.ConvTranspose3d):
            fan_in = m.in_channels  # kernel == stride: one tap per output voxel
        elif isinstance(m, nn.GroupNorm):
            with torch.no_grad():
                m.weight.normal_(1.0, 0.1, generator=g)
                m.bias.normal_(0.0, 0.1, generator=g)
            continue
        else:
            continue
        gain = out_gain if m is convs[-1] else 2.0
        with torch.no_grad():
            m.weight.normal_(0.0, math.sqrt(gain / fan_in), generator=g)
            m.bias.normal_(0.0, 0.05, generator=g)


@torch.inference_mode()
def forward_volume_16bit_storage(head: CryoVITHead, x: Tensor, dtype=torch.float16) -> Tensor:
    """The same network with every inter-layer activation and every GEMM-side weight rounded to a 16-bit type (fp32 math
    inside a layer, fp32 GroupNorm statistics, fp32 last conv) -- i.e. what an implementation with this storage plan
    computes when its arithmetic is exact.  Separates "kernel is wrong" from "16-bit storage differs from fp32" in the
    tests, and shows why the HIP head stores fp16 (like the reference's fp16 autocast) and not bf16: on the narrow fixture
    bf16 storage alone moves logits by 0.21 max / 0.014 mean, fp16 storage by ~0.02 / 0.002."""
    import torch.nn.functional as F

    def bf(t):
        return t.to(dtype).float()

    L = head.layers
    a = bf(F.gelu(F.conv3d(bf(x), bf(L[0].weight), L[0].bias)))
    for blk in list(L)[2:]:
        gn, c1, c2, ct = blk.layers[0], blk.layers[1], blk.layers[3], blk.layers[5]
        a = bf(F.group_norm(a, gn.num_groups, gn.weight, gn.bias, gn.eps))
        a = bf(F.gelu(F.conv3d(a, bf(c1.weight), c1.bias, padding="same", dilation=c1.dilation)))
        a = bf(F.gelu(F.conv3d(a, bf(c2.weight), c2.bias, padding="same", dilation=c2.dilation)))
        a = bf(F.gelu(F.conv_transpose3d(a, bf(ct.weight), ct.bias, stride=(1, 2, 2))))
    o0, o2 = head.output_layer[0], head.output_layer[2]
    a = bf(F.gelu(F.conv3d(a, bf(o0.weight), o0.bias, padding="same")))
    return torch.clip(F.conv3d(a, o2.weight, o2.bias, padding="same"), -5.0, 5.0)


def forward_volume_bf16_storage(head: CryoVITHead, x: Tensor) -> Tensor:
    return forward_volume_16bit_storage(head, x, torch.bfloat16)
