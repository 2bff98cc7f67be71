"""tests/golden/train_pieces.npz: the reference's own ``DiceLoss`` (models/losses.py:8-32) and ``TomoDataset._random_crop``
(datasets/tomo_dataset.py:148-178), executed where they lie by AST extraction (see oracle/make_golden.py), on seeded inputs;
the oracle restatement is asserted equal before the fixture is written.  BUILD container only.

    python -m oracle.make_golden_train
"""

from __future__ import annotations

import types

import numpy as np
import torch
from torch import Tensor, nn

from oracle import train_pieces as tp
from oracle.make_golden import GOLD, _exec_nodes, _find, _parse


def ref_dice_loss():
    ns = {"torch": torch, "nn": nn, "Tensor": Tensor}
    _exec_nodes([_find(_parse("models/losses.py"), "DiceLoss")], ns, "ref:models/losses.py")
    return ns["DiceLoss"]()


def ref_random_crop():
    fn = _find(_find(_parse("datasets/tomo_dataset.py"), "TomoDataset"), "_random_crop")
    ns = {"np": np, "Any": object}
    _exec_nodes([fn], ns, "ref:datasets/tomo_dataset.py")
    return lambda data, input_key: ns["_random_crop"](types.SimpleNamespace(input_key=input_key), data)


def main() -> None:
    out = {}
    # ---- Dice loss + its autograd gradient, on a masked volume ----
    g = torch.Generator().manual_seed(71)
    probs = torch.rand(6, 20, 24, generator=g, dtype=torch.float32)
    labels = torch.randint(-1, 2, (6, 20, 24), generator=g).float()
    mask = labels > -1.0
    p_req = probs.clone().requires_grad_(True)
    loss_ref = ref_dice_loss()(torch.masked_select(p_req, mask).view(-1, 1), torch.masked_select(labels, mask).view(-1, 1))
    loss_ref.backward()
    loss_or = tp.masked_dice_loss(probs, labels)
    assert torch.equal(loss_ref.detach(), loss_or), (loss_ref, loss_or)
    out.update(dice_probs=probs.numpy(), dice_labels=labels.numpy().astype(np.int8), dice_loss=np.float32(loss_ref.item()),
               dice_grad=p_req.grad.numpy())
    # ---- random crop: features [C, D, h, w] + label [D, 16h, 16w], and a raw volume ----
    crop = ref_random_crop()
    cases = [("dino_features", (4, 140, 40, 37)), ("dino_features", (4, 100, 32, 48)), ("dino_features", (2, 128, 32, 32)), ("data", (1, 130, 520, 512))]
    wins = []
    for k, (key, shp) in enumerate(cases):
        C, D, h, w = shp
        up = 16 if key == "dino_features" else 1
        # index volumes: the crop's origin and extent are readable from the corner values
        inp = np.arange(D * h * w, dtype=np.int64).reshape(1, D, h, w).repeat(C, 0)
        lab = np.arange(D * h * up * w * up, dtype=np.int64).reshape(D, h * up, w * up)
        for impl, store in ((crop, "ref"), (tp.random_crop, "or")):
            np.random.seed(1000 + k)
            d = {"input": inp, "label": lab}
            impl(d, key)
            i0, l0 = int(d["input"][0, 0, 0, 0]), int(d["label"][0, 0, 0])
            rec = [i0 // (h * w), (i0 // w) % h, i0 % w, *d["input"].shape[-3:], l0 // (h * up * w * up), (l0 // (w * up)) % (h * up), l0 % (w * up), *d["label"].shape]
            if store == "ref":
                wins.append(rec)
            else:
                assert rec == wins[-1], (rec, wins[-1])
    out["crop_cases"] = np.array([[c[1][0], c[1][1], c[1][2], c[1][3], 1 if c[0] == "dino_features" else 0] for c in cases], dtype=np.int64)
    out["crop_windows"] = np.array(wins, dtype=np.int64)  # di, hi, wi, x, y, z (input) | dl, hl, wl, D, H, W (label)
    np.savez_compressed(GOLD / "train_pieces.npz", **out)
    print("wrote", GOLD / "train_pieces.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
