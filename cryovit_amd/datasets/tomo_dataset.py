"""Loader feeding the segmentation head (mirror of the inference half of
``/root/reference/src/cryovit/datasets/tomo_dataset.py:16-146`` and ``datamodules/utils.py:13-121``).

``train=True`` adds the training-side random crop of l.148-178 (SURVEY s.8f row N4): depth at most 128, 32 x 32 feature
positions (512 x 512 voxels of the label), drawn with ``numpy.random.choice`` in the reference's order so that a seeded
run crops the same windows.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from cryovit_amd import io
from cryovit_amd.types import BatchedTomogramData, BatchedTomogramMetadata, TomogramData


def _as_records(records) -> list[dict]:
    """A pandas DataFrame (the reference's type), a list of dicts (csv rows) or of (sample, tomo_name) pairs."""
    if hasattr(records, "to_dict"):
        return records.to_dict("records")
    out = []
    for r in records:
        out.append(dict(r) if isinstance(r, dict) else {"sample": r[0], "tomo_name": r[1]})
    return out


class TomoDataset(Dataset):
    """One item per record (``sample``, ``tomo_name`` [, ``<split_key>``]): ``<data_root>/<sample>/<tomo_name>`` holds
    ``<input_key>`` (``dino_features`` fp16 [C,D,h,w], or a raw [D,H,W] volume), ``labels/<label_key>`` and optional
    auxiliary datasets (``aux_keys``, e.g. ``data``; keys absent from the file are skipped like upstream)."""

    def __init__(self, records, input_key: str, label_key: str, split_key: str | None = "split_id", data_root=".", aux_keys=None,
                 train: bool = False) -> None:
        self.records = _as_records(records)
        self.input_key, self.label_key, self.split_key = input_key, label_key, split_key
        self.aux_keys = list(aux_keys or [])
        self.data_root = Path(data_root)
        self.train = train

    def __len__(self) -> int:
        return len(self.records)

    def __getitem__(self, idx: int) -> TomogramData:
        if idx >= len(self):
            raise IndexError
        record = self.records[idx]
        data = self._load_tomogram(record)
        if self.train:
            self._random_crop(data)
        return TomogramData(sample=record["sample"], tomo_name=record["tomo_name"], split_id=data.get("split_id"),
                            data=torch.from_numpy(np.ascontiguousarray(data["input"])),
                            label=torch.from_numpy(np.ascontiguousarray(data["label"])),
                            aux_data={k: data[k] for k in self.aux_keys if k in data})

    def _load_tomogram(self, record: dict) -> dict:
        """tomo_dataset.py:89-146: uint8 input scaled to [0,1], a 3-D input gets a channel axis, labels as stored."""
        path = self.data_root / record["sample"] / record["tomo_name"]
        out = {"sample": record["sample"], "tomo_name": record["tomo_name"]}
        if self.split_key is not None and record.get(self.split_key) not in (None, ""):
            out["split_id"] = int(float(record[self.split_key]))
        top = io.list_keys(path, "/")
        assert self.input_key in top, f"Input key '{self.input_key}' not found in {path}."
        assert "labels" in top and self.label_key in io.list_keys(path, "labels"), f"Label key '{self.label_key}' not found in {path}/labels."
        data = io.read_dataset(path, self.input_key)
        if data.dtype == np.uint8:
            data = data.astype(np.float32) / 255.0
        if data.ndim == 3:
            data = data[np.newaxis]  # channel axis
        out["input"] = data
        out["label"] = io.read_dataset(path, f"labels/{self.label_key}")
        for key in self.aux_keys:
            if key == "sam_features":
                raise NotImplementedError("cached SAM2 features as aux data feed the SAM2 video model, which is out of scope")
            node, ok = "/", True
            for part in key.split("/"):  # "data", "labels/mito"
                if part not in io.list_keys(path, node):
                    ok = False
                    break
                node = part if node == "/" else f"{node}/{part}"
            if ok:
                out[key] = io.read_dataset(path, key)
        return out


def random_crop_window(shape_dhw, input_key: str, rng=np.random):
    """tomo_dataset.py:155-170: (di, hi, wi, depth, side, side) of the crop in INPUT coordinates, or None when the input already
    has the crop's size.  Three ``choice`` draws (depth, height, width), each only when that axis has room."""
    max_depth = 128
    side = 32 if input_key == "dino_features" else 512
    d, h, w = (int(v) for v in shape_dhw)
    x, y, z = min(d, max_depth), side, side
    if (d, h, w) == (x, y, z):
        return None
    delta_d, delta_h, delta_w = d - x + 1, h - y + 1, w - z + 1
    di = rng.choice(delta_d) if delta_d > 0 else 0
    hi = rng.choice(delta_h) if delta_h > 0 else 0
    wi = rng.choice(delta_w) if delta_w > 0 else 0
    return int(di), int(hi), int(wi), x, y, z


def _random_crop(self, data: dict) -> None:
    """In place on ``data['input']`` ([..., D, h, w]) and ``data['label']`` ([D, H, W]); with ``dino_features`` the label window
    is the feature window times 16 (tomo_dataset.py:172-178)."""
    win = random_crop_window(data["input"].shape[-3:], self.input_key)
    if win is None:
        return
    di, hi, wi, x, y, z = win
    data["input"] = data["input"][..., di : di + x, hi : hi + y, wi : wi + z]
    if self.input_key == "dino_features":
        hi, wi, y, z = 16 * hi, 16 * wi, 16 * y, 16 * z
    data["label"] = data["label"][di : di + x, hi : hi + y, wi : wi + z]


TomoDataset._random_crop = _random_crop


def collate_fn(batch: list[TomogramData]) -> BatchedTomogramData:
    """Stack tomograms: depth padded to the longest (data 0, label -1), cast to fp32, ``[B,C,D,h,w] -> [B,D,C,h,w]``, metadata
    as index pairs into the distinct samples / names (datamodules/utils.py:13-121).  The reference pads the LABEL of a
    shorter tomogram with a padded copy of its DATA (l.83-85): that is a bug, not a contract, and only
    triggers for B > 1 with unequal depths (evaluation uses B = 1); here the label is padded with -1 = ignore."""
    sizes = torch.tensor([t.data.shape[-3] for t in batch], dtype=torch.int)
    dmax = int(sizes.max())
    C, _, hp, wp = batch[0].data.shape
    H, W = batch[0].label.shape[-2:]
    tomo_batch = torch.empty(len(batch), C, dmax, hp, wp, dtype=torch.float)
    labels = torch.empty(len(batch), dmax, H, W, dtype=torch.float)
    aux: dict[str, list] = {k: [] for k in (batch[0].aux_data or {})}
    samples: dict[str, None] = {}
    names: dict[str, None] = {}
    ident = torch.empty(len(batch), 2, dtype=torch.long)
    split = torch.empty(len(batch), dtype=torch.int)
    use_splits = True
    for i, t in enumerate(batch):
        d = int(sizes[i])
        tomo_batch[i, :, :d] = t.data
        labels[i, :d] = t.label.reshape(-1, H, W)[:d] if t.label.dim() == 4 else t.label
        if d < dmax:
            tomo_batch[i, :, d:] = 0.0
            labels[i, d:] = -1.0
        for k, v in (t.aux_data or {}).items():
            aux[k].append(v)
        samples[t.sample] = None
        names[t.tomo_name] = None
        ident[i, 0], ident[i, 1] = list(samples).index(t.sample), list(names).index(t.tomo_name)
        if t.split_id is not None and use_splits:
            split[i] = t.split_id
        else:
            use_splits = False
    meta = BatchedTomogramMetadata(samples=list(samples), tomo_names=list(names), unique_id=ident, split_id=split if use_splits else None)
    return BatchedTomogramData(tomo_batch=tomo_batch.permute(0, 2, 1, 3, 4), labels=labels, tomo_sizes=sizes,
                               min_slices=int(sizes.min()), metadata=meta, aux_data=aux or None)
