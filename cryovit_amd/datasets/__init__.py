from cryovit_amd.datasets.file_dataset import FileDataset
from cryovit_amd.datasets.tomo_dataset import TomoDataset, collate_fn
from cryovit_amd.datasets.vit_dataset import VITDataset

__all__ = ["VITDataset", "TomoDataset", "FileDataset", "collate_fn"]
