"""Output files of the evaluation / inference flows (mirror of the formats written by
``/root/reference/src/cryovit/models/callbacks.py``; SURVEY.md App. C, "next" row N1).

  write_test_prediction  TestPredictionWriter l.15-58:  ``data`` contiguous, ``<label>`` gzip, ``<label>_preds`` fp32 gzip at
                         <results_dir>/<sample>/<tomo_name>
  write_prediction       PredictionWriter l.61-109:     ``data`` fp32 gzip, ``<label>_preds`` uint8 (preds >= threshold) gzip
                         at <results_dir>/<tomo stem>.hdf
  update_metrics_csv     CsvWriter l.112-206:           <results_dir>/<sample>[_<split>].csv, columns sample, tomo_name,
                         <metrics...>[, split_id]; an existing row for the same tomogram is replaced
"""

from __future__ import annotations

import csv
import logging
from pathlib import Path

import numpy as np

from cryovit_amd import io


def write_test_prediction(results_dir, sample: str, tomo_name: str, label_key: str, data: np.ndarray, labels: np.ndarray,
                          preds: np.ndarray) -> Path:
    out = Path(results_dir) / sample / tomo_name
    with io.FileWriter(out) as fh:
        fh.create_dataset("data", data)
        fh.create_dataset(label_key, labels, compression="gzip")
        fh.create_dataset(f"{label_key}_preds", preds.astype(np.float32, copy=False), compression="gzip")
    return out


def write_prediction(results_dir, tomo_name: str, label_key: str, data: np.ndarray, preds: np.ndarray, threshold: float) -> Path:
    out = (Path(results_dir) / tomo_name).with_suffix(".hdf")
    with io.FileWriter(out) as fh:
        fh.create_dataset("data", data.astype(np.float32), compression="gzip")
        fh.create_dataset(f"{label_key}_preds", (preds >= threshold).astype(np.uint8), compression="gzip")
    return out


def write_segmentation(results_dir, tomo_name: str, label_key: str, data: np.ndarray, segs: np.ndarray) -> Path:
    """``write_prediction`` for a segmentation that was already thresholded on the GPU (uint8 {0,1})."""
    out = (Path(results_dir) / tomo_name).with_suffix(".hdf")
    with io.FileWriter(out) as fh:
        fh.create_dataset("data", data.astype(np.float32), compression="gzip")
        fh.create_dataset(f"{label_key}_preds", segs.astype(np.uint8, copy=False), compression="gzip")
    return out


def update_metrics_csv(results_dir, sample: str, tomo_name: str, metrics: dict[str, float], split_id=None) -> Path:
    results_dir = Path(results_dir)
    results_dir.mkdir(parents=True, exist_ok=True)
    path = results_dir / f"{sample}{'' if split_id is None else f'_{split_id}'}.csv"
    columns = ["sample", "tomo_name", *metrics] + (["split_id"] if split_id is not None else [])
    rows: list[dict] = []
    if path.exists():
        with open(path, newline="") as f:
            rows = list(csv.DictReader(f))
    def same(r):
        return r.get("tomo_name") == tomo_name and r.get("sample") == sample and (split_id is None or str(r.get("split_id")) == str(split_id))
    n_old = sum(same(r) for r in rows)
    if n_old:
        logging.warning("Data with sample %s, name %s, and split %s already has an entry. Replacing %d rows...", sample, tomo_name,
                        split_id, n_old)
        rows = [r for r in rows if not same(r)]
    new = {"sample": sample, "tomo_name": tomo_name, **{k: repr(float(v)) for k, v in metrics.items()}}
    if split_id is not None:
        new["split_id"] = split_id
    rows.append(new)
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=columns, extrasaction="ignore")
        w.writeheader()
        for r in rows:
            w.writerow(r)
    return path


class TestPredictionWriter:
    """Callback form of ``write_test_prediction`` (callbacks.py:15-58): the ``_target_`` of ``configs/callbacks/test_pred_writer.yaml``."""

    __test__ = False  # not a pytest class

    def __init__(self, results_dir, label_key: str, **_):
        self.results_dir, self.label_key = Path(results_dir), label_key

    def on_test_batch_end(self, trainer, pl_module, outputs, batch=None, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        for n in range(outputs.num_tomos):
            write_test_prediction(self.results_dir, outputs.samples[n], outputs.tomo_names[n], self.label_key, outputs.data[n],
                                  outputs.label[n], outputs.preds[n])


class CsvWriter:
    """Callback form of ``update_metrics_csv`` (callbacks.py:112-206): one row per tomogram in ``<sample>[_<split>].csv``."""

    def __init__(self, results_dir, **_):
        self.results_dir = Path(results_dir)
        self.results_dir.mkdir(parents=True, exist_ok=True)

    def on_test_batch_end(self, trainer, pl_module, outputs, batch=None, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        assert outputs.num_tomos == 1, "CsvWriter only supports single-tomogram batches."
        split_id = outputs.split_id[0] if outputs.split_id is not None else None
        update_metrics_csv(self.results_dir, outputs.samples[0], outputs.tomo_names[0], outputs.metrics, split_id)
