"""CPU restatement of the training-side pieces of the head (TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import this; the product path never does).  SURVEY.md s.8f row N4.

  dice_loss      /root/reference/src/cryovit/models/losses.py:17-32 (applied to the masked predictions of
                 models/base_model.py:91-112)
  random_crop    /root/reference/src/cryovit/datasets/tomo_dataset.py:148-178
  adamw_step     torch.optim.AdamW (the optimizer models/base_model.py:57-63 builds; torch is a dependency of the reference, its
                 single-tensor update restated here from torch/optim/adamw.py of the installed torch 2.10 and checked against
                 torch.optim.AdamW itself in tests/test_cpu_oracle.py)

Pinned against the reference's own code (AST-extracted ``DiceLoss`` and ``TomoDataset._random_crop``) by
``oracle/make_golden_train.py`` -> tests/golden/train_pieces.npz.
"""

from __future__ import annotations

import numpy as np
import torch


def dice_loss(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """losses.py:27-32 on already masked tensors."""
    intersection = torch.sum(y_true * y_pred)
    denom = torch.sum(y_true) + torch.sum(y_pred)
    return 1 - (2 * intersection) / (denom + 1e-3)


def masked_dice_loss(probs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """base_model.py:96-110 + losses.py: voxels with label > -1 only."""
    mask = labels > -1.0
    return dice_loss(torch.masked_select(probs, mask).view(-1, 1), torch.masked_select(labels, mask).view(-1, 1).to(probs.dtype))


def sigmoid_focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: float = 0.25, gamma: float = 2, reduction: str = "none") -> torch.Tensor:
    """torchvision.ops.sigmoid_focal_loss (torchvision/ops/focal_loss.py, BSD-3; not installed here: restated from the published
    implementation, parity unpinned against the package itself)."""
    p = torch.sigmoid(inputs)
    ce_loss = torch.nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
        loss = alpha_t * loss
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def focal_loss(y_pred: torch.Tensor, y_true: torch.Tensor, gamma: float = 2) -> torch.Tensor:
    """losses.py:56-64 on already masked tensors."""
    weight = (y_true.numel() - y_true.sum()) / y_true.numel()
    return sigmoid_focal_loss(y_pred, y_true, alpha=weight.item(), gamma=gamma, reduction="mean")


def random_crop(data: dict, input_key: str) -> None:
    """tomo_dataset.py:148-178, in place; draws from the global ``np.random`` state like the reference."""
    max_depth = 128
    side = 32 if input_key == "dino_features" else 512
    d, h, w = data["input"].shape[-3:]
    x, y, z = min(d, max_depth), side, side
    if (d, h, w) == (x, y, z):
        return
    delta_d, delta_h, delta_w = d - x + 1, h - y + 1, w - z + 1
    di = np.random.choice(delta_d) if delta_d > 0 else 0
    hi = np.random.choice(delta_h) if delta_h > 0 else 0
    wi = np.random.choice(delta_w) if delta_w > 0 else 0
    data["input"] = data["input"][..., di : di + x, hi : hi + y, wi : wi + z]
    if input_key == "dino_features":
        hi, wi, y, z = 16 * np.array([hi, wi, y, z])
    data["label"] = data["label"][di : di + x, hi : hi + y, wi : wi + z]


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int) -> None:
    """One AdamW update in place (fp32 tensors), torch's single-tensor order of operations."""
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1**step, 1 - beta2**step
    denom = (v.sqrt() / (bc2**0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
