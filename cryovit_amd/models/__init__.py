from cryovit_amd.models.cryovit import CryoVIT
from cryovit_amd.models.encoder import DinoEncoder, load_encoder
from cryovit_amd.models.metrics import DiceMetric

__all__ = ["CryoVIT", "DinoEncoder", "load_encoder", "DiceMetric"]
