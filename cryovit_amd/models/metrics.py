"""Masked Dice on the device (mirror of ``/root/reference/src/cryovit/models/metrics.py:8-53``; the masking of
``models/base_model.py:99-110`` is folded in: voxels with label <= -1 are ignored)."""

from __future__ import annotations

import numpy as np
import torch

from cryovit_amd.engine import ops


def dice_from_sums(inter: float, ysum: float, psum: float) -> float:
    return 2.0 * inter / (ysum + psum + 1e-3)  # metrics.py:41-43


class DiceMetric:
    higher_is_better = True

    def __init__(self, threshold: float = 0.5, **_):
        self.name, self.thresh = "DiceMetric", threshold
        self.reset()

    def reset(self) -> None:
        self.dice_score, self.total = 0.0, 0

    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> None:
        """y_pred: probabilities, y_true: labels with -1 = ignore; any matching shapes (device tensors)."""
        probs = y_pred.detach().float().contiguous()
        labels = y_true.detach().to(torch.int8).contiguous()
        sums = torch.zeros(3, dtype=torch.float32, device=probs.device)
        ops.dice_sums(probs.view(-1), labels.view(-1), sums, self.thresh)
        i, sy, sp = sums.cpu().tolist()
        self.dice_score += dice_from_sums(i, sy, sp)
        self.total += 1

    def compute(self) -> float:
        return self.dice_score / self.total if self.total > 0 else 0.0

    def __call__(self, y_pred, y_true) -> float:
        """``forward`` of a torchmetrics Metric: value of this batch alone, accumulated into the running mean."""
        before = self.dice_score
        self.update(y_pred, y_true)
        return self.dice_score - before


class F1Metric:
    """metrics.py:56-93: ``p_hat = p > 0.5`` (strict, unlike DiceMetric's ``>=``), precision / recall with 1e-6 guards.
    tp / fp / fn follow from the same three masked sums the Dice kernel produces: the strict comparison is the kernel's
    ``p >= t`` with t = the float32 successor of 0.5."""

    higher_is_better = True
    _THRESH = float(np.nextafter(np.float32(0.5), np.float32(1.0)))

    def __init__(self, **_):
        self.name = "F1Metric"
        self.reset()

    def reset(self) -> None:
        self.f1, self.total = 0.0, 0

    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> None:
        probs = y_pred.detach().float().contiguous()
        labels = y_true.detach().to(torch.int8).contiguous()
        sums = torch.zeros(3, dtype=torch.float32, device=probs.device)
        ops.dice_sums(probs.view(-1), labels.view(-1), sums, self._THRESH)
        tp, ysum, psum = sums.cpu().tolist()
        fp, fn = psum - tp, ysum - tp
        precision, recall = tp / (tp + fp + 1e-6), tp / (tp + fn + 1e-6)
        self.f1 += 2 * (precision * recall) / (precision + recall + 1e-6)
        self.total += 1

    def compute(self) -> float:
        return self.f1 / self.total if self.total > 0 else 0.0

    def __call__(self, y_pred, y_true) -> float:
        before = self.f1
        self.update(y_pred, y_true)
        return self.f1 - before
