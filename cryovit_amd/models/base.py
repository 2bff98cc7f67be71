"""Evaluation protocol shared by the model mirrors (``/root/reference/src/cryovit/models/base_model.py:91-112, 115-175, 176-241``):
masked prediction, the per-batch loss / metric bookkeeping of ``_do_step`` and ``test_step``'s ``BatchedModelResult``.  The
models provide ``forward(batch) -> probabilities [B, D, H, W]`` on ``self._device`` plus ``loss_fns`` / ``metric_fns`` dicts."""

from __future__ import annotations

import torch
from torch import Tensor


class EvalProtocol:
    @torch.inference_mode()
    def _masked_predict(self, batch, use_mito_mask: bool = False) -> dict[str, Tensor]:
        """Predictions and labels restricted to labelled voxels (``labels > -1``), optionally also to the ``labels/mito``
        mask of the batch's aux data; everything stays on the device."""
        y_true = batch.labels.to(self._device)
        y_pred_full = self.forward(batch)  # (B, D, H, W) probabilities
        mask = y_true > -1.0
        if use_mito_mask:
            assert batch.aux_data is not None and "labels/mito" in batch.aux_data, "Batch aux_data must contain 'labels/mito' key for mito masking."
            mask = mask & (torch.as_tensor(batch.aux_data["labels/mito"][0]).to(self._device) > 0)  # eval batch size is 1
        return {"preds": torch.masked_select(y_pred_full, mask).view(-1, 1), "labels": torch.masked_select(y_true, mask).view(-1, 1),
                "preds_full": y_pred_full}

    @torch.inference_mode()
    def _predict_for_metrics(self, batch, use_mito_mask: bool = False):
        """What ``_masked_predict`` computes, WITHOUT its two ``masked_select`` gathers over a 33-Mvoxel volume: the metric and loss
        kernels (``cvx_dice_sums``, ``cvx_dice_loss_forward``, ...) ignore voxels whose label is -1 themselves, so the mask is folded
        into the labels (a mito mask turns excluded voxels into -1) and the full volumes are handed over.  Same sums, no copies."""
        y_true = batch.labels.to(self._device)
        y_pred_full = self.forward(batch)
        if use_mito_mask:
            assert batch.aux_data is not None and "labels/mito" in batch.aux_data, "Batch aux_data must contain 'labels/mito' key for mito masking."
            mito = torch.as_tensor(batch.aux_data["labels/mito"][0]).to(self._device) > 0  # eval batch size is 1
            y_true = torch.where(mito, y_true, torch.full_like(y_true, -1.0))
        return y_pred_full, y_true

    @torch.inference_mode()
    def test_step(self, batch, batch_idx: int = 0):
        """One evaluation batch -> ``BatchedModelResult`` (base_model.py:176-241): predictions, per-tomogram metrics (each
        metric is called once, then reset: a per-batch value), file metadata for the writers.  Losses are training-side
        (SURVEY s.8f N4) and are computed only if ``losses`` holds callables."""
        from cryovit_amd.types import BatchedModelResult

        assert batch.aux_data is not None and "data" in batch.aux_data, "Batch aux_data must contain 'data' key for testing."
        use_mito_mask = bool("labels/mito" in batch.aux_data and len(batch.aux_data["labels/mito"]))
        y_pred_full, y_true = self._predict_for_metrics(batch, use_mito_mask=use_mito_mask)
        y_pred = y_pred_full  # (masked inside the metric / loss kernels: label -1 = ignore)
        samples, tomo_names = batch.metadata.identifiers
        split_id = batch.metadata.split_id
        metrics = {}
        for name, m_fn in self.metric_fns.items():
            metrics[name] = float(m_fn(y_pred, y_true))
            m_fn.reset()
        losses = {k: float(fn(y_pred, y_true)) for k, fn in self.loss_fns.items()}
        losses["total"] = sum(losses.values())  # (compute_losses always adds it, 0 without loss functions: base_model.py:126-133)
        return BatchedModelResult(
            num_tomos=batch.num_tomos, samples=samples, tomo_names=tomo_names,
            split_id=None if split_id is None else [int(s) for s in split_id],
            data=batch.aux_data["data"], label=[t.cpu().numpy() for t in batch.labels],
            preds=[t.float().cpu().numpy() for t in y_pred_full], losses=losses, metrics=metrics, aux_data=None)


    @torch.inference_mode()
    def predict_step(self, batch, batch_idx: int = 0):
        """One prediction batch -> ``BatchedModelResult`` (base_model.py:243-273): the probabilities of ``self(batch)`` per tomogram
        as float32 numpy arrays with the file metadata; no losses, no metrics, ``split_id`` None.  ``PredictionWriter``
        (callbacks.py:100-102, ``run/writers.py``) thresholds them; ``predict_mask`` below is the fused form of both."""
        from cryovit_amd.types import BatchedModelResult

        assert batch.aux_data is not None and "data" in batch.aux_data, "Batch aux_data must contain 'data' key for prediction."
        preds = self.forward(batch)
        samples, tomo_names = batch.metadata.identifiers
        return BatchedModelResult(
            num_tomos=batch.num_tomos, samples=samples, tomo_names=tomo_names, split_id=None, data=batch.aux_data["data"],
            label=[t.cpu().numpy() for t in batch.labels], preds=[t.float().cpu().numpy() for t in preds], losses={}, metrics={},
            aux_data=None)

    @torch.inference_mode()
    def _do_step(self, batch, batch_idx: int = 0, prefix: str = "val") -> float:
        """base_model.py:153-165 without the Lightning logging: masked prediction -> every loss -> every metric updated; returns
        the total loss of the batch (forward only: there is no backward through the HIP forward passes)."""
        y_pred, y_true = self._predict_for_metrics(batch)
        losses = {k: float(fn(y_pred, y_true)) for k, fn in self.loss_fns.items()}
        for m_fn in self.metric_fns.values():
            m_fn(y_pred, y_true)
        return sum(losses.values())

    def validation_step(self, batch, batch_idx: int = 0) -> float:
        return self._do_step(batch, batch_idx, "val")
