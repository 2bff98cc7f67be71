from cryovit_amd.run.dino_features import run_trainer as run_dino_trainer

__all__ = ["run_dino_trainer"]
