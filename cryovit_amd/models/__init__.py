from cryovit_amd.models.cryovit import CryoVIT
from cryovit_amd.models.encoder import DinoEncoder, load_encoder
from cryovit_amd.models.losses import DiceLoss
from cryovit_amd.models.metrics import DiceMetric, F1Metric
from cryovit_amd.models.sam_encoder import SamImageEncoder, load_sam_encoder
from cryovit_amd.models.unet3d import UNet3D

__all__ = ["CryoVIT", "DinoEncoder", "load_encoder", "DiceLoss", "DiceMetric", "F1Metric", "SamImageEncoder", "load_sam_encoder", "UNet3D"]
