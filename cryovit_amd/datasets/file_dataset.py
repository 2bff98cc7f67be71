"""Dataset over user-supplied tomogram files for the script-level flows ``cryovit features`` / ``cryovit infer`` (mirror of
``/root/reference/src/cryovit/datasets/file_dataset.py``; same constructor, same item contract).

  predict / eval (``train=False``)   l.99-117   ``data`` = the model's input key ([C,D,h,w] features or [1,D,H,W] raw),
                                                 ``label`` = the label volume or zeros, ``aux_data["data"]`` = the raw tomogram
  feature extraction (``for_dino``)  l.79-97    ``data`` = the encoder input, ``aux_data["data"]`` = the raw tomogram

Difference, by design: for ``for_dino`` the reference resizes on the host, and its ``_dino_transform`` (l.196-233) keeps ONE
ImageNet-normalised channel which a 3-channel patch embedding rejects (that path cannot run as written);
the parity target is the Hydra path's ``VITDataset`` behaviour (3 equal channels, no normalisation).  Here the item carries
the raw ``[D,H,W]`` volume and the edge-pad + x14/16 bicubic resize + patch cut happen in the encoder's first HIP kernel.
Random crops (``train=True``, l.160-194) belong to training and are out of scope.
"""

from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from cryovit_amd.types import FileData, TomogramData
from cryovit_amd.utils import load_data, load_labels


class FileDataset(Dataset):
    def __init__(self, files: list[FileData], input_key: str | None, label_key: str | None, train: bool = False,
                 for_dino: bool = False, use_sam: bool = False) -> None:
        if train:
            raise NotImplementedError("training crops are out of scope of this build (SURVEY s.8f N4)")
        self.files = files
        self.input_key, self.label_key = input_key, label_key
        self.train, self.for_dino, self.use_sam = train, for_dino, use_sam
        self._key_cache: dict = {}

    def __len__(self) -> int:
        return len(self.files)

    def __getitem__(self, idx: int) -> TomogramData:
        if idx >= len(self):
            raise IndexError
        fd = self.files[idx]
        data = self._load_tomogram(fd)
        if self.for_dino:
            raw = data["input"]
            if raw.shape[0] != 1:
                raise ValueError(f"{fd.tomo_path}: feature extraction needs a single-channel [D,H,W] tomogram, got {raw.shape}")
            vol = np.ascontiguousarray(raw[0], dtype=np.float32)
            return TomogramData(sample=fd.sample, tomo_name=fd.tomo_path.name, data=torch.from_numpy(vol),
                                label=torch.zeros(raw.shape, dtype=torch.bool), aux_data={"data": vol})
        aux = {"data": load_data(fd.tomo_path, key="data")[0].squeeze(0) if self.input_key != "data" else data["input"].squeeze(0)}
        return TomogramData(sample=fd.sample, tomo_name=fd.tomo_path.name, data=torch.from_numpy(np.ascontiguousarray(data["input"])),
                            label=torch.from_numpy(np.ascontiguousarray(data["label"])), aux_data=aux)

    def _load_tomogram(self, fd: FileData) -> dict:
        """input: ``load_data`` of the tomogram under ``input_key`` (found key cached per file, l.133-138); label: the
        requested label volume, or int8 zeros of shape [1, D, H, W] of the input when no label file is given (l.139-157)."""
        if fd.tomo_path in self._key_cache:
            data, _ = load_data(fd.tomo_path, key=self._key_cache[fd.tomo_path])
        else:
            data, key = load_data(fd.tomo_path, key=self.input_key)
            self._key_cache[fd.tomo_path] = key
        labels = (load_labels(fd.label_path, label_keys=fd.labels, key=self.label_key)
                  if fd.label_path is not None and fd.labels is not None else None)
        assert data is not None, f"Failed to load data from {fd.tomo_path}"
        label = labels[self.label_key] if labels is not None and self.label_key is not None else np.zeros((1, *data.shape[1:]), np.int8)
        return {"input": data, "label": label}
