"""Oracle: tomogram -> ViT input (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``/root/reference/src/cryovit/datasets/vit_dataset.py``:
  * ``_load_tomogram``  l.71-88   uint8 -> float32 / 255, floats as-is
  * ``_dino_transform`` l.90-123  edge-pad H,W to x16, 3 equal channels,
    bicubic resize by 14/16 (align_corners=False, no antialias).
No ImageNet normalisation is applied on this path (the ``Normalize`` object at
l.39 is never called).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

DINO_PATCH_SIZE = 14  # reference: src/cryovit/config.py:17


def load_scale(data: np.ndarray) -> np.ndarray:
    """vit_dataset.py:86-88 -- uint8 volumes are scaled to [0,1]."""
    if data.dtype == np.uint8:
        data = data.astype(np.float32) / 255.0
    return data


def pad_to_16(data: np.ndarray) -> np.ndarray:
    """vit_dataset.py:101-112 -- edge-pad bottom/right up to a multiple of 16."""
    _, h, w = data.shape
    H = int(np.ceil(h / 16) * 16)
    W = int(np.ceil(w / 16) * 16)
    if h != H or w != W:
        data = np.pad(data, ((0, 0), (0, H - h), (0, W - w)), mode="edge")
    return data


def dino_transform(data: np.ndarray) -> torch.Tensor:
    """[D,h,w] float -> float32 [D,3,H*14/16,W*14/16]  (vit_dataset.py:90-123)."""
    scale = (DINO_PATCH_SIZE / 16, DINO_PATCH_SIZE / 16)
    data = pad_to_16(data)
    x = np.expand_dims(data, axis=1)
    x = np.repeat(x, 3, axis=1)
    x = torch.from_numpy(x).float()
    return F.interpolate(x, scale_factor=scale, mode="bicubic")


def cubic_weights(t: float, A: float = -0.75):
    """Keys cubic-convolution taps for fractional offset t."""

    def w1(x):  # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def w2(x):  # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A

    return (w2(t + 1.0), w1(t), w1(1.0 - t), w2(2.0 - t))


def bicubic_14_16_closed_form(img: np.ndarray) -> np.ndarray:
    """Closed form of the resize for ONE [H,W] slice, float64 loops.

    Source coordinate s = (dst + 0.5) * 16/14 - 0.5, taps floor(s)-1..+2 with
    index clamping, separable.  Small inputs only; used to pin the formula the
    HIP kernel implements against ``F.interpolate``.
    """
    H, W = img.shape
    Ho, Wo = H * 14 // 16, W * 14 // 16
    out = np.zeros((Ho, Wo), dtype=np.float64)
    for oy in range(Ho):
        sy = (oy + 0.5) * (16.0 / 14.0) - 0.5
        iy = int(np.floor(sy))
        wy = cubic_weights(sy - iy)
        for ox in range(Wo):
            sx = (ox + 0.5) * (16.0 / 14.0) - 0.5
            ix = int(np.floor(sx))
            wx = cubic_weights(sx - ix)
            acc = 0.0
            for a in range(4):
                yy = min(max(iy - 1 + a, 0), H - 1)
                row = 0.0
                for b in range(4):
                    xx = min(max(ix - 1 + b, 0), W - 1)
                    row += wx[b] * float(img[yy, xx])
                acc += wy[a] * row
            out[oy, ox] = acc
    return out
