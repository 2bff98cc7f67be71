"""Drop-in surface on the GPU: the ``python -m ...training.dino_features`` entry point on a synthetic data directory
(BASELINE configs[0] plumbing: 64x256x256 uint8 tomogram, ViT-S/14-reg), the ``CryoVIT`` module with a
reference-layout state_dict, and ``DiceMetric`` -- each against the CPU oracle."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_entry_point_config1_vits(gpu, tmp_path):
    """configs[0]: single 64x256x256 synthetic HDF5 tomogram, ViT-S/14-reg, full Hydra-style plumbing.
    Seeded random weights are generated on the device by the entry point (encoder.synthetic_seed) and re-generated
    here for the oracle, so both sides see identical parameters."""
    from cryovit_amd import io
    from cryovit_amd.engine.vit import VIT_CONFIGS, random_state_dict
    from cryovit_amd.training import dino_features
    from oracle import dinov2 as o
    from oracle import features as ofe
    from oracle import preprocess as opre

    rng = np.random.default_rng(0)
    vol = rng.integers(0, 256, size=(64, 256, 256), dtype=np.uint8)
    lab = rng.integers(-1, 2, size=(64, 256, 256)).astype(np.int8)
    src = tmp_path / "processed" / "Q109"
    src.mkdir(parents=True)
    with io.FileWriter(src / "tomo_a.hdf") as f:
        f.create_dataset("data", vol, compression="gzip")
        f.create_dataset("labels/mito", lab, compression="gzip")
    dino_features.main([f"paths.model_dir={tmp_path}", f"paths.data_dir={tmp_path}", f"paths.exp_dir={tmp_path / 'exp'}",
                        "paths.feature_name=processed", "sample=Q109", "batch_size=24", "encoder.name=dinov2_vits14_reg",
                        "encoder.synthetic_seed=7"])
    out = tmp_path / "tomograms" / "Q109" / "tomo_a.hdf"  # inverted naming: reads feature_name, writes tomo_name (App. D-1)
    assert out.exists(), "entry point did not produce the output tomogram (see logged traceback)"
    flat = io.read_all_flat(out)
    assert sorted(flat) == ["data", "dino_features", "mito"]
    assert np.array_equal(flat["data"], vol) and np.array_equal(flat["mito"], lab)
    feats = flat["dino_features"]
    assert feats.dtype == np.float16 and feats.shape == (384, 64, 16, 16)
    # oracle on a subset of slices (the whole volume takes minutes on the CPU): slices are independent in the ViT
    sd = {k: v.cpu() for k, v in random_state_dict(VIT_CONFIGS["dinov2_vits14_reg"], 7, device=gpu).items()}
    pick = [0, 23, 24, 63]  # both sides of a slice-batch boundary and the ends
    ref = ofe.dino_features(opre.dino_transform(opre.load_scale(vol[pick])), o.OracleDino(o.VITS14_REG, sd), 4)
    err = np.abs(feats[:, pick].astype(np.float32) - ref.astype(np.float32))
    assert err.max() <= 1e-1 and err.mean() <= 1e-2, (err.max(), err.mean())


def test_cryovit_module_forward(gpu, gold):
    """CryoVIT(_target_ of configs/model/cryovit.yaml): load a reference-layout state_dict, forward(batch) -> probs."""
    from cryovit_amd.models import CryoVIT
    from cryovit_amd.types import BatchedTomogramData
    from oracle import features as ofe
    from oracle import head as oh

    ref = oh.CryoVITHead()
    oh.rescaled_init_(ref, seed=5)
    model = CryoVIT(input_key="dino_features", lr=1e-3, weight_decay=1e-3, losses={}, metrics={}, name="CryoVIT", device=gpu)
    model.load_state_dict(ref.state_dict())
    feats = torch.randn(1536, 6, 3, 2, generator=torch.Generator().manual_seed(3)).half()
    tomo_batch = ofe.collate_features(feats.numpy())  # fp32 [1,D,C,h,w] like collate_fn
    batch = BatchedTomogramData(tomo_batch=tomo_batch, labels=torch.zeros(1, 6, 48, 32), tomo_sizes=torch.tensor([6]))
    probs = model.forward(batch).cpu()
    with torch.inference_mode():
        want = ref.forward_tomo_batch(tomo_batch)
    assert tuple(probs.shape) == (1, 6, 48, 32)
    lg, lw = torch.logit(probs.double()), torch.logit(want.double())
    err = (lg - lw).abs()
    assert float(err.max()) <= 2.5e-1 and float(err.mean()) <= 2e-2, (float(err.max()), float(err.mean()))
    assert float(probs.min()) >= 0.0066 and float(probs.max()) <= 0.9934  # sigmoid(+-5) bounds (App. D-8)


def test_dice_metric(gpu, gold):
    from cryovit_amd.models import DiceMetric

    g = gold("dice.npz")
    m = DiceMetric(threshold=0.5)
    v = m(torch.from_numpy(g["preds"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu))
    assert abs(v - float(g["dice"])) < 1e-6 and abs(m.compute() - float(g["dice"])) < 1e-6
    m.reset()
    assert m.compute() == 0.0
