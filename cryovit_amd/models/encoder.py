"""The frozen DINOv2 encoder object the feature stage drives (replaces the ``torch.hub.load`` result of
``/root/reference/src/cryovit/run/dino_features.py:255-256,335-337``).

It satisfies the reference's duck-typed protocol -- ``.cuda()``, ``.eval()``,
``.forward_features(x)["x_norm_patchtokens"]`` (SURVEY s.8b) -- and additionally exposes ``features_from_raw`` which
the runner uses to skip the host-side resize.
"""

from __future__ import annotations

from pathlib import Path

import torch

from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict

# upstream checkpoint file names (facebookresearch/dinov2 hub), looked up under model_dir
CHECKPOINT_FILES = {name: name + "4_pretrain.pth" for name in VIT_CONFIGS}


class DinoEncoder:
    def __init__(self, name: str, state_dict: dict, device="cuda:0"):
        if name not in VIT_CONFIGS:
            raise ValueError(f"unknown encoder {name!r}; known: {sorted(VIT_CONFIGS)}")
        self.name, self.cfg = name, VIT_CONFIGS[name]
        self.engine = VitEngine(self.cfg, state_dict, device)
        self.device = self.engine.device
        self.embed_dim = self.cfg.dim

    # protocol no-ops: the engine lives on the GPU and has no training mode
    def cuda(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def forward_features(self, x: torch.Tensor) -> dict:
        return self.engine.forward_features(x)

    @torch.inference_mode()
    def features_from_raw(self, volume: torch.Tensor, batch_size: int, want_f16=True, want_cl=False):
        """volume: [D,H,W] uint8 / float32 (host or device).  Returns (feats_f16 [C,D,h,w] | None, feats_cl | None) on
        the device; slices go through the encoder ``batch_size`` at a time like the reference's loop."""
        from cryovit_amd.engine import ops

        vol = volume.to(self.device).contiguous()
        D, H, W = vol.shape
        hp, wp, _, _, _ = self.engine.geometry(H, W)
        C = self.cfg.dim
        f16 = torch.empty(C, D, hp, wp, dtype=torch.float16, device=self.device) if want_f16 else None
        cl = torch.zeros(ops.alloc_rows(D * hp * wp), C, dtype=torch.float16, device=self.device) if want_cl else None
        for d0 in range(0, D, batch_size):
            b = min(batch_size, D - d0)
            self.engine.features(vol[d0 : d0 + b], feats_f16=f16, d_total=D, d0=d0,
                                 feats_cl=None if cl is None else cl[d0 * hp * wp :])
        return f16, cl


def load_encoder(name: str, model_dir=None, checkpoint=None, synthetic_seed=None, device="cuda:0") -> DinoEncoder:
    """Weights come from a local state_dict file (``torch.load(weights_only=True)``); there is no network fetch.
    ``synthetic_seed`` builds seeded random weights instead (offline smoke runs)."""
    if synthetic_seed is not None:
        sd = random_state_dict(VIT_CONFIGS[name], int(synthetic_seed), device=device)
        return DinoEncoder(name, sd, device)
    path = Path(checkpoint) if checkpoint else Path(model_dir) / CHECKPOINT_FILES[name]
    if not path.exists():
        raise FileNotFoundError(
            f"DINOv2 checkpoint {path} not found. Place the upstream state_dict there (this build never downloads), or "
            "set encoder.synthetic_seed=<int> for a random-weight run."
        )
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return DinoEncoder(name, sd, device)
