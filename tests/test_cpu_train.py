"""Training-side pieces (SURVEY s.8f row N4), CPU part: the oracle restatements against the fixture written by executing the
reference's own DiceLoss / _random_crop (oracle/make_golden_train.py) and against torch.optim.AdamW; the host mirror of the
random crop against the same fixture."""

import numpy as np
import torch

from oracle import train_pieces as tp


def test_oracle_dice_loss_matches_reference_fixture(gold):
    g = gold("train_pieces.npz")
    probs = torch.from_numpy(g["dice_probs"]).requires_grad_(True)
    labels = torch.from_numpy(g["dice_labels"].astype(np.float32))
    loss = tp.masked_dice_loss(probs, labels)
    assert float(loss.detach()) == float(g["dice_loss"])  # same expression, same reduction order: bit-equal
    loss.backward()
    assert torch.equal(probs.grad, torch.from_numpy(g["dice_grad"]))
    # the closed form the HIP kernel evaluates: -2 y / den + 2 I / den^2 on labelled voxels, 0 elsewhere
    p, y = probs.detach(), labels
    m = y > -1
    inter, den = (y * p)[m].sum(), y[m].sum() + p[m].sum() + 1e-3
    closed = torch.where(m, -2 * y / den + 2 * inter / den**2, torch.zeros_like(p))
    assert torch.allclose(closed, probs.grad, rtol=1e-5, atol=1e-9)


def test_oracle_adamw_matches_torch():
    gen = torch.Generator().manual_seed(5)
    for shape, kw in (((37,), dict(lr=1e-4, weight_decay=1e-2)), ((8, 5, 3), dict(lr=3e-3, weight_decay=0.0, betas=(0.8, 0.95), eps=1e-6))):
        p0 = torch.randn(*shape, generator=gen)
        ref_p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([ref_p], foreach=False, **kw)
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        b1, b2 = kw.get("betas", (0.9, 0.999))
        for step in range(1, 8):
            g = torch.randn(*shape, generator=gen) * (0.1 if step % 2 else 3.0)
            ref_p.grad = g.clone()
            opt.step()
            tp.adamw_step(p, g, m, v, lr=kw["lr"], beta1=b1, beta2=b2, eps=kw.get("eps", 1e-8), weight_decay=kw["weight_decay"], step=step)
            assert torch.equal(p, ref_p.detach()), step  # the restatement IS torch's update, operation for operation


def test_random_crop_mirror_matches_reference_fixture(gold):
    from cryovit_amd.datasets.tomo_dataset import TomoDataset, random_crop_window

    g = gold("train_pieces.npz")
    for k, (case, want) in enumerate(zip(g["crop_cases"], g["crop_windows"])):
        C, D, h, w, is_feat = (int(v) for v in case)
        key, up = ("dino_features", 16) if is_feat else ("data", 1)
        np.random.seed(1000 + k)
        win = random_crop_window((D, h, w), key)
        if win is None:
            assert list(want[:6]) == [0, 0, 0, D, h, w]
        else:
            assert list(win) == [int(v) for v in want[:6]]
        # the dataset method on index volumes: label window = 16 x the feature window
        inp = np.arange(D * h * w, dtype=np.int64).reshape(1, D, h, w).repeat(C, 0)
        lab = np.arange(D * h * up * w * up, dtype=np.int64).reshape(D, h * up, w * up)
        ds = TomoDataset.__new__(TomoDataset)
        ds.input_key = key
        d = {"input": inp, "label": lab}
        np.random.seed(1000 + k)
        ds._random_crop(d)
        i0, l0 = int(d["input"][0, 0, 0, 0]), int(d["label"][0, 0, 0])
        got = [i0 // (h * w), (i0 // w) % h, i0 % w, *d["input"].shape[-3:], l0 // (h * up * w * up), (l0 // (w * up)) % (h * up), l0 % (w * up), *d["label"].shape]
        assert got == [int(v) for v in want]


def test_unet3d_oracle_reproduces_reference_fixture_and_mirror_layout(gold):
    """The UNet3D oracle (pinned bit-for-bit against the reference's classes when the fixture was written) reproduces the
    fixture here; the product's parameter container has the reference's state_dict keys and shapes, loads strictly, and the
    Hydra-style config instantiates it."""
    from cryovit_amd.config import compose, instantiate
    from cryovit_amd.models import UNet3D
    from oracle import unet3d as ou

    g = gold("unet3d_narrow.npz")
    orc = ou.UNet3D(ou.NARROW_WIDTHS)
    ou.rescaled_init_(orc, seed=int(g["seed"]))
    with torch.no_grad():
        probs = orc.forward_tomo_batch(torch.from_numpy(g["vol"]))
    assert torch.allclose(probs, torch.from_numpy(g["probs"]), atol=2e-6)
    ref, mine = ou.UNet3D(ou.REF_WIDTHS).state_dict(), UNet3D(device="cpu").state_dict()
    assert list(ref) == list(mine) and all(ref[k].shape == mine[k].shape for k in ref) and len(ref) == 82
    UNet3D(device="cpu", widths=ou.NARROW_WIDTHS).load_state_dict(orc.state_dict(), strict=True)
    cfg = compose("eval_model", ["model=unet3d", "datamodule=single", "datamodule.sample=[Q109]", "datamodule.split_id=1", "label_key=mito",
                                 "paths.model_dir=/m", "paths.data_dir=/d", "paths.exp_dir=/e"])
    model = instantiate(cfg.model, device="cpu")
    assert isinstance(model, UNet3D) and model.input_key == "data" and model.lr == 3e-3 and set(model.metric_fns) == {"dice_metric", "f1_metric"}
