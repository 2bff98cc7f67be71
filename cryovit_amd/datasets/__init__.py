from cryovit_amd.datasets.tomo_dataset import TomoDataset, collate_fn
from cryovit_amd.datasets.vit_dataset import VITDataset

__all__ = ["VITDataset", "TomoDataset", "collate_fn"]
