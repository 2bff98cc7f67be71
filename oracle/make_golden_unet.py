"""tests/golden/unet3d_narrow.npz: the reference's own ``UNet3D`` classes (models/unet3d.py:12-216), executed where they lie by
AST extraction (see oracle/make_golden.py) with the channel widths patched to the narrow family, against the oracle restatement
on the same seeded weights and input (asserted bit-equal before the fixture is written).  BUILD container only.

    python -m oracle.make_golden_unet
"""

from __future__ import annotations

import ast
import math

import numpy as np
import torch
from torch import Tensor, nn

from oracle import unet3d as ou
from oracle.make_golden import GOLD, _exec_nodes, _find, _parse


def ref_unet_classes():
    tree = _parse("models/unet3d.py")

    class BaseModel(nn.Module):
        def __init__(self, **kwargs):
            super().__init__()

    ns = {"torch": torch, "nn": nn, "Tensor": Tensor, "math": math, "BaseModel": BaseModel, "BatchedTomogramData": object}
    _exec_nodes([_find(tree, n) for n in ("AnalysisBlock", "LinearProjection", "SynthesisBlock", "UNet3D")], ns, "ref:models/unet3d.py")
    return ns


def main() -> None:
    ns = ref_unet_classes()
    # 1) reference widths: same state_dict keys and shapes as the restatement
    ref_full = ns["UNet3D"]()
    or_full = ou.UNet3D(ou.REF_WIDTHS)
    a, b = ref_full.state_dict(), or_full.state_dict()
    assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a), "state_dict layout differs from the reference"
    # 2) numerics on the full-width model, small volume: copy the weights, compare outputs bit for bit
    ou.rescaled_init_(or_full, seed=11)
    ref_full.load_state_dict(or_full.state_dict(), strict=True)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(1, 1, 16, 16, 32, generator=g)
    with torch.no_grad():
        ya, yb = ref_full.forward_volume(x), or_full.forward_volume(x)
    assert torch.equal(ya, yb), float((ya - yb).abs().max())

    # 3) the narrow family for the GPU parity test: the reference classes compose the same way, so build the reference model and
    # swap its blocks for narrow ones made from the reference's own block classes
    A, S = ns["AnalysisBlock"], ns["SynthesisBlock"]
    (a1, a2, a3), cb = ou.NARROW_WIDTHS
    ref = ns["UNet3D"]()
    ref.bottom_layer = nn.Sequential(nn.Conv3d(a3, cb, 3, padding="same"), nn.InstanceNorm3d(cb, eps=1e-3, affine=True), nn.GELU(),
                                     nn.Conv3d(cb, a3, 3, padding="same"), nn.InstanceNorm3d(a3, eps=1e-3, affine=True), nn.GELU())
    ref.analysis_layers = nn.ModuleList([A(1, a1), A(a1, a2), A(a2, a3)])
    ref.synthesis_layers = nn.ModuleList([S(a3, a3, a2), S(a2, a2, a1), S(a1, a1, a1)])
    ref.output_layer = nn.Conv3d(a1, 1, 1, padding="same")
    orc = ou.UNet3D(ou.NARROW_WIDTHS)
    ou.rescaled_init_(orc, seed=13)
    ref.load_state_dict(orc.state_dict(), strict=True)
    vol = torch.rand(1, 20, 1, 40, 37, generator=g)  # [B, D, C, H, W]: every axis needs padding to 32 / 48 / 48

    class Batch:
        tomo_batch = vol

    with torch.no_grad():
        pa = ref.forward(Batch())
        pb = orc.forward_tomo_batch(vol)
        lg = orc.forward_volume(torch.nn.functional.pad(vol.permute(0, 2, 1, 3, 4), (0, 48 - 37, 0, 48 - 40, 0, 32 - 20)))
    assert torch.equal(pa, pb), float((pa - pb).abs().max())
    np.savez_compressed(GOLD / "unet3d_narrow.npz", vol=vol.numpy(), probs=pb.numpy(), logits_padded=lg.numpy(), seed=np.int64(13))
    print("wrote", GOLD / "unet3d_narrow.npz", tuple(pb.shape), "probs range", float(pb.min()), float(pb.max()), "frac>0.5", float((pb > 0.5).float().mean()))


if __name__ == "__main__":
    main()
