"""Oracle: per-tomogram feature extraction loop (TEST INFRASTRUCTURE).

Restates ``_dino_features`` --
``/root/reference/src/cryovit/run/dino_features.py:31-64``: slice batches of
``batch_size`` through ``model.forward_features(...)["x_norm_patchtokens"]``,
``reshape(b, H'/14, W'/14, C).permute(3,0,1,2)``, cast to float16 on the host,
concatenate on axis 1 -> ``float16 [C, D, H/16, W/16]`` (the
reference names the grid ``w, h`` but it is (rows, cols)).
"""

from __future__ import annotations

import numpy as np
import torch


@torch.inference_mode()
def dino_features(data: torch.Tensor, model, batch_size: int) -> np.ndarray:
    hp, wp = (np.array(data.shape[-2:]) // 14).tolist()
    out = []
    n = len(data)
    for i in range(0, n, batch_size):
        vec = data[i : min(i + batch_size, n)]
        f = model.forward_features(vec)["x_norm_patchtokens"]
        f = f.reshape(f.shape[0], hp, wp, -1).permute([3, 0, 1, 2]).contiguous()
        out.append(f.to("cpu").half().numpy())
    return np.concatenate(out, axis=1)


def collate_features(feats_f16: np.ndarray) -> torch.Tensor:
    """``collate_fn`` for B=1 -- /root/reference/src/cryovit/datamodules/utils.py:13-121:
    ``[C,D,h,w]`` fp16 -> fp32 ``tomo_batch`` ``[1,D,C,h,w]`` (permute at :105-107)."""
    x = torch.from_numpy(feats_f16.astype(np.float32))
    return x.permute(1, 0, 2, 3).unsqueeze(0).contiguous()
