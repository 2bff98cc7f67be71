"""Dataset feeding the feature stage (mirror of ``/root/reference/src/cryovit/datasets/vit_dataset.py``).

Same constructor (``data_root, use_sam, records`` -- ``configs/datamodule/dataset/vit.yaml``) and the same
``_load_tomogram`` semantics (HDF5 ``data``; uint8 is scaled to [0,1], floats pass through -- l.71-88).  The difference
is WHERE the resize happens: the reference edge-pads, replicates to 3 channels and bicubically resizes on the host
(l.90-123, 308 MB fp32 per tomogram); here ``__getitem__`` returns the raw ``[D,H,W]`` volume (uint8 or float32) and
the pad + x14/16 bicubic resize + patch cut run fused in the encoder's first HIP kernel (``cvx_preprocess_patches``).
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from cryovit_amd import io


class VITDataset(Dataset):
    def __init__(self, data_root, use_sam: bool, records: list[str]) -> None:
        self.root = data_root if isinstance(data_root, Path) else Path(data_root)
        # use_sam: the reference's _sam_transform (l.125-142) only adds a batch axis and replicates the slice to 3 channels;
        # both happen inside the encoder's first kernel here (cvx_sam_patches), so the item is the same raw volume.
        self.use_sam = use_sam
        self.records = records

    def __len__(self) -> int:
        return len(self.records)

    def __getitem__(self, idx: int) -> torch.Tensor:
        if idx >= len(self):
            raise IndexError
        return torch.from_numpy(self._load_tomogram(self.records[idx]))

    def _load_tomogram(self, record: str) -> np.ndarray:
        """``data`` as stored: uint8 stays uint8 (the /255 of vit_dataset.py:86-88 happens in the kernel), floating point
        is converted to float32."""
        data = io.read_dataset(self.root / record, "data")
        if data.ndim != 3:
            raise ValueError(f"{record}: expected a [D,H,W] volume, got shape {data.shape}")
        if data.dtype != np.uint8:
            data = data.astype(np.float32, copy=False)
        return np.ascontiguousarray(data)
