"""Loader feeding the segmentation head (mirror of the inference half of
``/root/reference/src/cryovit/datasets/tomo_dataset.py:89-146`` and ``datamodules/utils.py:13-121``).

Training-only behaviour (random crops, l.148-178) is out of scope.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from cryovit_amd import io
from cryovit_amd.types import BatchedTomogramData, TomogramData


class TomoDataset(Dataset):
    """records: list of (sample, tomo_name); reads ``input_key`` (default ``dino_features``) and ``labels/<label_key>``."""

    def __init__(self, records, input_key: str, label_key: str, data_root, aux_keys=(), **_):
        self.records = list(records)
        self.input_key, self.label_key, self.aux_keys = input_key, label_key, list(aux_keys)
        self.data_root = Path(data_root)

    def __len__(self) -> int:
        return len(self.records)

    def __getitem__(self, idx: int) -> TomogramData:
        sample, tomo_name = self.records[idx]
        return self._load_tomogram(sample, tomo_name)

    def _load_tomogram(self, sample: str, tomo_name: str) -> TomogramData:
        path = self.data_root / sample / tomo_name
        data = io.read_dataset(path, self.input_key)
        if data.dtype == np.uint8:  # tomo_dataset.py: raw input is scaled, features pass through
            data = data.astype(np.float32) / 255.0
        if data.ndim == 3:
            data = data[np.newaxis]
        label = io.read_dataset(path, f"labels/{self.label_key}")
        aux = {k: io.read_dataset(path, k) for k in self.aux_keys}
        return TomogramData(sample=sample, tomo_name=tomo_name, data=torch.from_numpy(np.ascontiguousarray(data)),
                            label=torch.from_numpy(np.ascontiguousarray(label)), aux_data=aux)


def collate_fn(batch: list[TomogramData]) -> BatchedTomogramData:
    """Stack tomograms: depth padded to the longest (data 0, label -1), cast to fp32, ``[B,D,C,h,w]``
    (datamodules/utils.py:13-121; the reference's B>1 label-padding quirk, SURVEY App. D-4, is not reproduced)."""
    dmax = max(t.data.shape[1] for t in batch)
    datas, labels = [], []
    for t in batch:
        d = t.data.float()
        lab = t.label.float()
        pad = dmax - d.shape[1]
        if pad:
            d = torch.nn.functional.pad(d, (0, 0, 0, 0, 0, pad), value=0.0)
            lab = torch.nn.functional.pad(lab, (0, 0, 0, 0, 0, pad), value=-1.0)
        datas.append(d.permute(1, 0, 2, 3))
        labels.append(lab)
    return BatchedTomogramData(
        tomo_batch=torch.stack(datas), labels=torch.stack(labels),
        tomo_sizes=torch.tensor([t.data.shape[1] for t in batch]), min_slices=min(t.data.shape[1] for t in batch),
        metadata={"samples": [t.sample for t in batch], "tomo_names": [t.tomo_name for t in batch]},
        aux_data={k: [t.aux_data[k] for t in batch] for k in batch[0].aux_data} if batch[0].aux_data else None,
    )
