"""Dice loss on the device (mirror of ``/root/reference/src/cryovit/models/losses.py:8-32``), SURVEY.md s.8f row N4.

``DiceLoss()(y_pred, y_true)`` has the reference's call shape -- probabilities and labels of one shape, already masked
(``BaseModel._masked_predict``, models/base_model.py:91-112) -- and additionally accepts UNMASKED volumes whose labels carry
-1 = ignore: the mask is then applied inside the kernel (no ``masked_select`` copies of a 33-Mvoxel volume).  The value and
its gradient come from ``cvx_dice_loss_forward`` / ``cvx_dice_loss_backward`` (csrc/train.hip) through a
``torch.autograd.Function``; there is no CPU path.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from cryovit_amd.engine import ops


class _DiceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred: Tensor, labels_i8: Tensor) -> Tensor:
        probs = y_pred.detach().float().contiguous().view(-1)
        out4 = torch.empty(4, dtype=torch.float32, device=probs.device)
        ops.dice_loss_forward(probs, labels_i8, out4)
        ctx.save_for_backward(labels_i8, out4)
        ctx.shape, ctx.dtype = y_pred.shape, y_pred.dtype
        return out4[3].clone()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        labels_i8, out4 = ctx.saved_tensors
        grad = torch.empty(labels_i8.numel(), dtype=torch.float32, device=labels_i8.device)
        ops.dice_loss_backward(None, None, labels_i8, out4, float(grad_out), grad, through_sigmoid=False)
        return grad.view(ctx.shape).to(ctx.dtype), None


class DiceLoss(nn.Module):
    """Dice loss for imbalanced foreground / background segmentation (losses.py:8-32): ``1 - 2 sum(y p) / (sum y + sum p + 1e-3)``."""

    def __init__(self) -> None:
        super().__init__()
        self.name = "DiceLoss"

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        if not y_pred.is_cuda:
            raise ops._lib.CvxError("DiceLoss: predictions must live on a HIP device (no CPU fallback)")
        if y_pred.numel() != y_true.numel():
            raise ValueError(f"DiceLoss: {tuple(y_pred.shape)} predictions vs {tuple(y_true.shape)} labels")
        labels = y_true.detach().to(torch.int8).contiguous().view(-1)
        return _DiceLossFn.apply(y_pred, labels)


def dice_loss_and_logit_grad(probs: Tensor, logits: Tensor | None, labels_i8: Tensor, grad_out: float = 1.0):
    """Fused training-step form: loss (device scalar) and d loss / d logit for the head's output layer, with the sigmoid and the
    +-5 clip of ``CryoVIT.forward`` (models/cryovit.py:39,49) folded into the same pass over the volume."""
    out4 = torch.empty(4, dtype=torch.float32, device=probs.device)
    p = probs.contiguous().view(-1)
    ops.dice_loss_forward(p, labels_i8.contiguous().view(-1), out4)
    grad = torch.empty_like(p)
    ops.dice_loss_backward(p, None if logits is None else logits.contiguous().view(-1), labels_i8.contiguous().view(-1), out4, grad_out,
                           grad, through_sigmoid=True)
    return out4[3], grad.view(probs.shape)


class _FocalLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred: Tensor, labels_i8: Tensor, gamma: float) -> Tensor:
        x = y_pred.detach().float().contiguous().view(-1)
        out4 = torch.empty(4, dtype=torch.float32, device=x.device)
        ops.focal_loss_forward(x, labels_i8, gamma, out4)
        ctx.save_for_backward(x, labels_i8, out4)
        ctx.shape, ctx.dtype, ctx.gamma = y_pred.shape, y_pred.dtype, gamma
        return out4[3].clone()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        x, labels_i8, out4 = ctx.saved_tensors
        grad = torch.empty_like(x)
        ops.focal_loss_backward(x, labels_i8, ctx.gamma, out4, float(grad_out), grad)
        return grad.view(ctx.shape).to(ctx.dtype), None, None


class FocalLoss(nn.Module):
    """Focal loss (losses.py:35-64): ``torchvision.ops.sigmoid_focal_loss(y_pred, y_true, alpha=(n - sum y) / n, gamma, "mean")``.
    Like the reference, the model output is handed to the formula as its "logits" (the reference passes probabilities there).
    torchvision is not installed in the build image: the formula is restated from its published implementation
    (``torchvision/ops/focal_loss.py``) -- parity unpinned against torchvision itself."""

    def __init__(self, gamma=2, **kwargs) -> None:
        super().__init__()
        self.gamma = gamma
        self.name = "FocalLoss"

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        if not y_pred.is_cuda:
            raise ops._lib.CvxError("FocalLoss: predictions must live on a HIP device (no CPU fallback)")
        if y_pred.numel() != y_true.numel():
            raise ValueError(f"FocalLoss: {tuple(y_pred.shape)} predictions vs {tuple(y_true.shape)} labels")
        return _FocalLossFn.apply(y_pred, y_true.detach().to(torch.int8).contiguous().view(-1), float(self.gamma))
