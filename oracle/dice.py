"""Oracle: masked Dice (TEST INFRASTRUCTURE).

Restates
  * ``BaseModel._masked_predict``  /root/reference/src/cryovit/models/base_model.py:91-112
    (``mask = labels > -1``; ``masked_select`` of predictions and labels)
  * ``DiceMetric.update/compute``  /root/reference/src/cryovit/models/metrics.py:30-53
    (``p_hat = where(p < thr, 0, 1)``; ``2*sum(y*p_hat) / (sum(y)+sum(p_hat)+1e-3)``),
    threshold 0.5 from ``configs/model/metrics/dice_metric.yaml:3``.
``test_step`` calls the metric once per tomogram then ``reset()``
(base_model.py:225-228), so the value is per tomogram.
"""

from __future__ import annotations

import torch
from torch import Tensor


def masked_select_pair(preds: Tensor, labels: Tensor) -> tuple[Tensor, Tensor]:
    mask = labels > -1.0
    return torch.masked_select(preds, mask).view(-1, 1), torch.masked_select(labels, mask).view(-1, 1)


def dice_sums(preds: Tensor, labels: Tensor, thresh: float = 0.5) -> tuple[float, float, float]:
    """(sum y*p_hat, sum y, sum p_hat) over voxels with label > -1, in float64."""
    y_pred, y_true = masked_select_pair(preds, labels.to(preds.dtype))
    p_hat = torch.where(y_pred < thresh, 0.0, 1.0).double()
    y = y_true.double()
    return float((y * p_hat).sum()), float(y.sum()), float(p_hat.sum())


def dice_metric(preds: Tensor, labels: Tensor, thresh: float = 0.5) -> float:
    inter, sy, sp = dice_sums(preds, labels, thresh)
    return 2.0 * inter / (sy + sp + 1e-3)
