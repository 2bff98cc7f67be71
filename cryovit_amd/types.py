"""Data contracts of the hot path (mirror of ``/root/reference/src/cryovit/types.py``; tensordict is not required).

``Sample`` is the closed list of sample directory names the reference accepts for ``sample=`` (types.py:15-46); the
values are dataset vocabulary, reproduced so the same command lines and directory layouts work.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from pathlib import Path
from typing import Any

import numpy as np
import torch

_SAMPLE_VALUES = {
    "BACHD": "BACHD", "BACHD_Microtubules": "BACHD Microtubules", "dN17_BACHD": "dN17 BACHD", "Q109": "Q109",
    "Q109_Microtubules": "Q109 Microtubules", "Q18": "Q18", "Q18_Microtubules": "Q18 Microtubules", "Q20": "Q20", "Q53": "Q53",
    "Q53_KD": "Q53 PIAS1", "Q66": "Q66", "Q66_GRFS1": "Q66 GRFS1", "Q66_KD": "Q66 PIAS1", "WT": "Wild Type",
    "WT_Microtubules": "Wild Type Microtubules", "cancer": "Cancer", "AD": "AD", "AD_Abeta": "AD Abeta", "Aged": "Aged",
    "Young": "Young", "RGC_CM": "RGC CM", "RGC_control": "RGC Control", "RGC_naPP": "RGC naPP", "RGC_PP": "RGC PP",
    "CZI_Algae": "Algae", "CZI_Campy_C": "Campy C", "CZI_Campy_CDel": "Campy C-Deletion", "CZI_Campy_F": "Campy F",
    "CZI_Fibroblast": "Mouse Fibroblast",
}
Sample = Enum("Sample", _SAMPLE_VALUES)
Sample.__doc__ = "Enum of all valid CryoET samples (directory names under data_dir/<tomo_name>/)."


class ModelType(Enum):
    CRYOVIT = "cryovit"
    UNET3D = "unet3d"
    SAM2 = "sam2"
    MEDSAM = "medsam"


@dataclass
class FileData:
    """File description of one tomogram for the script-level flows (types.py:62-76)."""

    tomo_path: Path
    label_path: Path | None = None
    labels: list[str] | None = None
    sample: str | None = None


@dataclass
class TomogramData:
    """One loaded tomogram (types.py:79-99): ``data`` fp16/fp32 [C,D,h,w] features or [1,D,H,W] raw, ``label`` [D,H,W]."""

    sample: str | None
    tomo_name: str
    data: torch.Tensor
    label: torch.Tensor
    aux_data: dict[str, Any] = field(default_factory=dict)
    split_id: int | None = None


@dataclass
class BatchedTomogramMetadata:
    """Who is in a batch (types.py:102-123): the distinct samples / file names and, per tomogram, its (sample index, name
    index) pair; ``split_id`` only when every tomogram of the batch carries one."""

    samples: list[str]
    tomo_names: list[str]
    unique_id: torch.Tensor  # long [B,2]
    split_id: torch.Tensor | None = None  # int [B]

    @property
    def identifiers(self) -> tuple[list[str], list[str]]:
        return ([self.samples[int(i[0])] for i in self.unique_id], [self.tomo_names[int(i[1])] for i in self.unique_id])


@dataclass
class BatchedTomogramData:
    """Collated batch (types.py:126-189): ``tomo_batch`` fp32 [B,D,C,h,w], ``labels`` fp32 [B,D,H,W]."""

    tomo_batch: torch.Tensor
    labels: torch.Tensor
    tomo_sizes: torch.Tensor
    min_slices: int = 0
    metadata: BatchedTomogramMetadata | None = None
    aux_data: dict[str, Any] | None = None

    @property
    def num_tomos(self) -> int:
        return int(self.tomo_batch.shape[0])

    @property
    def num_slices(self) -> int:
        return int(self.tomo_batch.shape[1])


@dataclass
class BatchedModelResult:
    """Per-batch evaluation result, organised per tomogram (types.py:192-219): what ``test_step`` hands to the writers."""

    num_tomos: int
    samples: list[str]
    tomo_names: list[str]
    split_id: list[int] | None
    data: list[np.ndarray]
    label: list[np.ndarray]
    preds: list[np.ndarray]
    losses: dict[str, float]
    metrics: dict[str, float]
    aux_data: dict[str, Any] | None = None
