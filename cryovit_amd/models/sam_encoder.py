"""The SAM2 image-encoder object the feature stage drives when ``use_sam=True`` (replaces the ``SAM2`` Lightning module
built by ``create_sam_model_from_weights``, ``/root/reference/src/cryovit/models/sam2.py:802-842``, for the one method the
feature path calls: ``forward_features``, ``sam2.py:190-209``).

Protocol: ``.cuda()``, ``.eval()``, ``.forward_features(data [b,d,3,h,w]) -> {"vision_features", "vision_pos_enc",
"backbone_fpn"}`` (lists ordered fine -> coarse, tensors ``[b*d, 256, h_l, w_l]``); ``features_from_raw`` additionally takes
the raw ``[D,H,W]`` volume so that the channel replication and the resize happen inside the first kernel.  Prompt encoder,
memory attention and mask decoder (the video-segmentation model) are outside the feature path and not built.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from cryovit_amd.engine.hiera import HIERA_CONFIGS, HieraEngine, random_state_dict

# upstream checkpoint files (facebook/sam2.1-hiera-large, models/sam2.py:32-35), looked up under model_dir
CHECKPOINT_FILES = {"sam2.1_hiera_l": "sam2.1_hiera_large.pt"}
MODEL_NAMES = {"SAM2": "sam2.1_hiera_l"}  # configs/model/sam2.yaml `name` -> encoder variant


class SamImageEncoder:
    def __init__(self, name: str, state_dict: dict, device="cuda:0", slice_batch: int = 64, fold_ln: bool = True):
        if name not in HIERA_CONFIGS:
            raise ValueError(f"unknown SAM2 encoder {name!r}; known: {sorted(HIERA_CONFIGS)}")
        self.name, self.cfg = name, HIERA_CONFIGS[name]
        self.engine = HieraEngine(self.cfg, state_dict, device, fold_ln=fold_ln)
        self.device = self.engine.device
        self.image_size = self.cfg.image_size
        self.slice_batch = slice_batch

    def cuda(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def _outs(self, D: int) -> list[torch.Tensor]:
        return [torch.empty(D, self.cfg.d_model, g, g, dtype=torch.float16, device=self.device)
                for g in self.engine.grids[: self.engine.n_levels()]]

    @torch.inference_mode()
    def forward_features(self, data: torch.Tensor) -> dict:
        """data: float [b,d,3,h,w] -> the ``image_encoder`` dict for the flattened b*d slices (float16 device tensors)."""
        b, d, c, h, w = data.shape
        if c != 3:
            raise ValueError("forward_features expects 3-channel slices [b,d,3,h,w]")
        flat = data.reshape(b * d, c, h, w).to(self.device, torch.float32).contiguous()
        outs = self._outs(b * d)
        for d0 in range(0, b * d, self.slice_batch):
            self.engine.encode(flat[d0 : d0 + self.slice_batch], outs, d0)
        pos = [self.engine.pos_enc(l).to(self.device)[None].expand(b * d, -1, -1, -1) for l in range(len(outs))]
        return {"vision_features": outs[-1], "vision_pos_enc": pos, "backbone_fpn": outs}

    @torch.inference_mode()
    def features_from_raw(self, volume: torch.Tensor, batch_size: int | None = None) -> dict[str, list[np.ndarray]]:
        """volume: [D,H,W] uint8 / float32 (host or device) -> what ``_sam_features`` returns: every key except
        ``vision_features`` as a list of float16 arrays ``[D,256,h_l,w_l]`` (host)."""
        vol = volume.to(self.device).contiguous()
        if vol.dtype != torch.uint8:
            vol = vol.float()
        D = vol.shape[0]
        outs = self._outs(D)
        step = min(batch_size or self.slice_batch, self.slice_batch)
        for d0 in range(0, D, step):
            self.engine.encode(vol[d0 : d0 + step], outs, d0)
        feats = []
        for o in outs:
            host = torch.empty(o.shape, dtype=o.dtype, pin_memory=True)
            host.copy_(o, non_blocking=True)
            feats.append(host)
        torch.cuda.current_stream(self.device).synchronize()
        pos = [np.broadcast_to(self.engine.pos_enc(l).numpy()[None], (D, *self.engine.pos_enc(l).shape)) for l in range(len(outs))]
        return {"vision_pos_enc": pos, "backbone_fpn": [f.numpy() for f in feats]}


def load_sam_encoder(name: str = "sam2.1_hiera_l", model_dir=None, checkpoint=None, synthetic_seed=None, device="cuda:0",
                     slice_batch: int = 64, fold_ln: bool = True) -> SamImageEncoder:
    """Weights: the upstream ``sam2.1_hiera_large.pt`` (``{"model": state_dict}``, keys ``image_encoder.*``) read with
    ``torch.load(weights_only=True)``; never downloaded.  ``synthetic_seed`` builds seeded random weights instead."""
    name = MODEL_NAMES.get(name, name)
    if synthetic_seed is not None:
        return SamImageEncoder(name, random_state_dict(HIERA_CONFIGS[name], int(synthetic_seed), device=device), device, slice_batch, fold_ln)
    path = Path(checkpoint) if checkpoint else Path(model_dir) / CHECKPOINT_FILES[name]
    if not path.exists():
        raise FileNotFoundError(f"SAM2 checkpoint {path} not found. Place the upstream file there (this build never downloads), "
                                "or set encoder.synthetic_seed=<int> for a random-weight run.")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = sd.get("model", sd)
    sd = {k: v for k, v in sd.items() if k.startswith(("image_encoder.", "trunk.", "neck."))}
    return SamImageEncoder(name, sd, device, slice_batch, fold_ln)
