"""End-to-end segmentation runner: (optional) DINOv2 features -> CryoVIT head -> probabilities, masked Dice, result files.

This is the hot path of BASELINE.json configs[2] (head + Dice on precomputed features) and configs[3] (features + head on
a list of tomograms sharded over the GPUs of a node).  It replaces the Lightning evaluation loop of
``/root/reference/src/cryovit/run/eval_model.py:143-197`` / ``models/base_model.py:176-241`` for inference: one tomogram
per step (the reference's batch size for evaluation), metric rows appended to the per-sample CSV, predictions optionally
written in the reference's formats (``run/writers.py``).  Tomograms are independent: each rank processes its shard, rank 0
merges the CSV rows (``run/sharding.py``) -- no data-path collective.
"""

from __future__ import annotations

import logging
from pathlib import Path

import numpy as np
import torch

from cryovit_amd import io
from cryovit_amd.engine import ops
from cryovit_amd.run import writers
from cryovit_amd.run.sharding import gather_rows, shard_records, world_info


@torch.inference_mode()
def segment_tomogram(path: Path, head, label_key: str | None, encoder=None, batch_size: int = 128, input_key: str = "dino_features"):
    """One tomogram file -> (probs fp32 [D,H,W] device tensor, dice | None, labels | None).

    With ``encoder`` the features are computed from ``data`` and handed to the head in channels-last fp16 without leaving
    HBM; otherwise ``input_key`` (float16 [C,D,h,w], the feature stage's output) is read from the file."""
    labels = None
    if label_key is not None:
        labels = torch.from_numpy(io.read_dataset(path, f"labels/{label_key}").astype(np.int8))
    if encoder is not None:
        vol = io.read_dataset(path, "data")
        vol_t = torch.from_numpy(vol if vol.dtype == np.uint8 else vol.astype(np.float32))
        _, cl = encoder.features_from_raw(vol_t, batch_size, want_f16=False, want_cl=True)
        D = vol.shape[0]
        hp, wp, *_ = encoder.engine.geometry(vol.shape[1], vol.shape[2])
    else:
        feats = torch.from_numpy(io.read_dataset(path, input_key)).to(head._device)
        C, D, hp, wp = feats.shape
        cl = torch.zeros(ops.alloc_rows(D * hp * wp), C, dtype=torch.float16, device=head._device)
        if feats.dtype == torch.float16:
            ops.features_to_channels_last(feats.contiguous(), cl)
        else:
            cl[: D * hp * wp] = feats.permute(1, 2, 3, 0).reshape(-1, C).to(torch.float16)
    probs, dice = head.predict_with_dice(cl, D, hp, wp, labels)
    return probs, dice, labels


def run_segmentation(records: list[tuple[str, Path]], head, label_key: str | None, results_dir=None, encoder=None,
                     batch_size: int = 128, save_predictions: bool = False, threshold: float = 0.5, weights=None) -> list[dict]:
    """records: (sample, tomogram path).  Returns the metric rows of ALL ranks (on every rank)."""
    rank, _, world = world_info()
    rows = []
    for i in shard_records(records, rank, world, weights):
        sample, path = records[i]
        probs, dice, labels = segment_tomogram(Path(path), head, label_key, encoder, batch_size)
        row = {"sample": sample, "tomo_name": Path(path).name, "dice_metric": dice}
        rows.append(row)
        logging.info("[rank %d] %s/%s dice %s", rank, sample, Path(path).name, "n/a" if dice is None else f"{dice:.4f}")
        if save_predictions and results_dir is not None:
            data = io.read_dataset(path, "data")
            preds = probs.cpu().numpy()
            if labels is not None:
                writers.write_test_prediction(Path(results_dir) / "predictions", sample, Path(path).name, label_key, data,
                                              labels.numpy(), preds)
            else:
                writers.write_prediction(Path(results_dir) / "predictions" / sample, Path(path).name, label_key or "seg", data, preds,
                                         threshold)
    all_rows = gather_rows(rows, world)
    if rank == 0 and results_dir is not None:
        for r in all_rows:
            if r["dice_metric"] is not None:
                writers.update_metrics_csv(Path(results_dir) / "results", r["sample"], r["tomo_name"], {"dice_metric": r["dice_metric"]})
    return all_rows
