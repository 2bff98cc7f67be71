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


def test_end_to_end_runner_configs_2_and_3(gpu, tmp_path):
    """configs[2]/[3] plumbing at small size: three synthetic tomograms -> run_segmentation twice, once from raw ``data``
    through the ViT-g encoder (features never leave HBM) and once from the ``dino_features`` written by the feature
    stage; both must agree with each other, with the CSV, and with the prediction files."""
    import csv
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    from cryovit_amd import io
    from cryovit_amd.models import CryoVIT, load_encoder
    from cryovit_amd.run.dino_features import _dino_features, _save_data
    from cryovit_amd.run.segment import run_segmentation

    enc = load_encoder("dinov2_vitg14_reg", synthetic_seed=2, device=gpu)
    model = CryoVIT(device=gpu)
    model.load_state_dict({k: v.cpu() for k, v in bench.synthetic_head_state_dict(5, gpu).items()})
    rng = np.random.default_rng(5)
    recs = []
    for i in range(3):
        vol = rng.integers(0, 256, size=(6 + i, 64, 48), dtype=np.uint8)
        lab = rng.integers(-1, 2, size=vol.shape).astype(np.int8)
        feats = _dino_features(torch.from_numpy(vol), enc, 4)
        assert feats.shape == (1536, vol.shape[0], 4, 3) and feats.dtype == np.float16
        _save_data({"data": vol, "mito": lab}, feats, f"t{i}.hdf", tmp_path / "tomograms" / "Q109")
        recs.append(("Q109", tmp_path / "tomograms" / "Q109" / f"t{i}.hdf"))
    rows_e2e = run_segmentation(recs, model, "mito", results_dir=tmp_path / "res_e2e", encoder=enc, batch_size=4)
    rows_feat = run_segmentation(recs, model, "mito", results_dir=tmp_path / "res_feat", save_predictions=True)
    assert [r["tomo_name"] for r in rows_e2e] == ["t0.hdf", "t1.hdf", "t2.hdf"]
    for a, b in zip(rows_e2e, rows_feat):
        assert 0.0 < a["dice_metric"] < 1.0
        assert a["dice_metric"] == b["dice_metric"]  # the file and the in-HBM hand-over carry the same fp16 features
    got = {r["tomo_name"]: float(r["dice_metric"]) for r in csv.DictReader(open(tmp_path / "res_feat" / "results" / "Q109.csv"))}
    assert got == {r["tomo_name"]: r["dice_metric"] for r in rows_feat}
    pred = tmp_path / "res_feat" / "predictions" / "Q109" / "t1.hdf"
    assert sorted(io.list_keys(pred)) == ["data", "mito", "mito_preds"]
    p = io.read_dataset(pred, "mito_preds")
    assert p.dtype == np.float32 and p.shape == (7, 64, 48) and 0.0066 <= p.min() and p.max() <= 0.9934


def test_run_inference_against_oracle(gpu, tmp_path):
    """``cryovit infer`` (run/infer_model.py:18-85 + PredictionWriter, callbacks.py:81-109) on files that hold
    ``dino_features``: result files, key names, dtypes, and the uint8 segmentation against the fp32 oracle head.
    The fp16 head stays within 5e-2 of the fp32 logits (tests/test_gpu_model.py), so disagreement is only allowed for voxels
    whose oracle logit lies that close to the threshold."""
    from typer.testing import CliRunner

    from cryovit_amd import io
    from cryovit_amd.cli import cli
    from cryovit_amd.run.infer_model import run_inference
    from cryovit_amd.types import ModelType
    from cryovit_amd.utils import save_model_from_weights
    from oracle import head as oh

    ref = oh.CryoVITHead()
    oh.rescaled_init_(ref, seed=5)
    torch.save(ref.state_dict(), tmp_path / "weights.pt")
    save_model_from_weights("demo", "mito", ModelType.CRYOVIT, tmp_path / "weights.pt", tmp_path / "demo.model")
    rng = np.random.default_rng(9)
    files, feats_all, vols = [], [], []
    (tmp_path / "in").mkdir()
    for i, D in enumerate((6, 9)):
        vol = rng.integers(0, 256, size=(D, 48, 32), dtype=np.uint8)
        feats = rng.standard_normal((1536, D, 3, 2)).astype(np.float16)
        with io.FileWriter(tmp_path / "in" / f"tomo{i}.hdf") as f:
            f.create_dataset("data", vol, compression="gzip")
            f.create_dataset("dino_features", feats)
        files.append(tmp_path / "in" / f"tomo{i}.hdf")
        feats_all.append(feats)
        vols.append(vol)
    thr = 0.4
    paths = run_inference(files, tmp_path / "demo.model", tmp_path / "out", threshold=thr)
    assert paths == [tmp_path / "out" / "tomo0.hdf", tmp_path / "out" / "tomo1.hdf"]
    logit_thr = float(np.log(thr / (1 - thr)))
    for path, feats, vol in zip(paths, feats_all, vols):
        assert sorted(io.list_keys(path)) == ["data", "mito_preds"]
        data = io.read_dataset(path, "data")
        assert data.dtype == np.float32 and np.array_equal(data, vol.astype(np.float32) / 255.0)
        seg = io.read_dataset(path, "mito_preds")
        assert seg.dtype == np.uint8 and seg.shape == vol.shape and set(np.unique(seg)) <= {0, 1}
        with torch.no_grad():
            logits = ref.forward_volume(torch.from_numpy(feats).float()[None])[0, 0].numpy()
        want = (1.0 / (1.0 + np.exp(-logits)) >= thr).astype(np.uint8)
        bad = seg != want
        assert bad.mean() <= 0.01 and np.all(np.abs(logits[bad] - logit_thr) < 0.05), (bad.mean(), np.abs(logits[bad] - logit_thr).max())
        assert 0.02 < seg.mean() < 0.98  # the synthetic head produces both classes
    # the same through the command line (cli/infer_cli.py): folder argument, --model, --result-folder, --threshold
    res = CliRunner().invoke(cli, ["infer", str(tmp_path / "in"), "--model", str(tmp_path / "demo.model"), "--result-folder",
                                   str(tmp_path / "out_cli"), "--threshold", str(thr)])
    assert res.exit_code == 0, res.output
    for p in paths:
        assert np.array_equal(io.read_dataset(tmp_path / "out_cli" / p.name, "mito_preds"), io.read_dataset(p, "mito_preds"))
    res = CliRunner().invoke(cli, ["infer", str(tmp_path / "in"), "--model", str(tmp_path / "weights.pt")])
    assert res.exit_code != 0  # "Model path does not exist or is not a .model file."


def test_features_cli_then_infer_and_on_the_fly_encoder(gpu, tmp_path):
    """``cryovit features`` (run_dino, run/dino_features.py:211-299) on an .mrc and an .hdf tomogram, then ``infer`` on the
    produced files; and the build's shortcut -- ``run_inference(..., encoder=)`` on the raw files -- must give the same
    segmentation (both hand fp16 features to the head; the two copies come from the same LayerNorm in one kernel)."""
    import struct

    from typer.testing import CliRunner

    import bench
    from cryovit_amd import io
    from cryovit_amd.cli import cli
    from cryovit_amd.models import CryoVIT, load_encoder
    from cryovit_amd.run.infer_model import run_inference
    from cryovit_amd.utils import save_model

    rng = np.random.default_rng(21)
    (tmp_path / "raw").mkdir()
    vol_a = rng.integers(0, 256, size=(5, 64, 48), dtype=np.uint8)
    with io.FileWriter(tmp_path / "raw" / "a.hdf") as f:
        f.create_dataset("data", vol_a, compression="gzip")
        f.create_dataset("mito", rng.integers(-1, 2, size=vol_a.shape).astype(np.int8))
    vol_b = rng.random((4, 64, 48)).astype(np.float32)
    hdr = bytearray(1024)
    hdr[0:16] = struct.pack("<4i", 48, 64, 4, 2)
    hdr[208:216] = b"MAP " + bytes([0x44, 0x44, 0, 0])
    (tmp_path / "raw" / "b.mrc").write_bytes(bytes(hdr) + vol_b.tobytes())
    res = CliRunner().invoke(cli, ["features", str(tmp_path / "raw"), str(tmp_path / "feat"), "--batch-size", "3", "--synthetic-seed", "2"])
    assert res.exit_code == 0, res.output
    fa, fb = tmp_path / "feat" / "a.hdf", tmp_path / "feat" / "b.hdf"
    assert sorted(io.list_keys(fa)) == ["data", "dino_features"]  # FileDataset hands only `data` to _save_data (file_dataset.py:96)
    assert sorted(io.list_keys(fb)) == ["data", "dino_features"]
    feats_b = io.read_dataset(fb, "dino_features")
    assert feats_b.dtype == np.float16 and feats_b.shape == (1536, 4, 4, 3)
    assert np.array_equal(io.read_dataset(fb, "data"), vol_b)  # float MRC data passes through un-normalised (utils.py:216-219)
    enc = load_encoder("dinov2_vitg14_reg", synthetic_seed=2, device=gpu)
    from cryovit_amd.run.dino_features import _dino_features

    direct = _dino_features(torch.from_numpy(vol_a.astype(np.float32) / 255.0), enc, 5)
    assert np.array_equal(io.read_dataset(fa, "dino_features"), direct)  # slice batching (3 vs 5) does not change features
    model = CryoVIT(device=gpu)
    model.load_state_dict({k: v.cpu() for k, v in bench.synthetic_head_state_dict(5, gpu).items()})
    save_model("e2e", "mito", model, {"_target_": "cryovit_amd.models.CryoVIT", "name": "CryoVIT", "input_key": "dino_features"},
               tmp_path / "e2e.model")
    from_files = run_inference([fa, fb], tmp_path / "e2e.model", tmp_path / "seg_files")
    on_the_fly = run_inference([tmp_path / "raw" / "a.hdf", tmp_path / "raw" / "b.mrc"], tmp_path / "e2e.model", tmp_path / "seg_raw",
                               encoder=enc, batch_size=3)
    assert [p.name for p in on_the_fly] == ["a.hdf", "b.hdf"]
    for p, q in zip(from_files, on_the_fly):
        s0, s1 = io.read_dataset(p, "mito_preds"), io.read_dataset(q, "mito_preds")
        assert s0.shape == s1.shape and np.array_equal(s0, s1)  # identical fp16 features either way -> identical segmentation
        assert np.array_equal(io.read_dataset(p, "data"), io.read_dataset(q, "data"))
    # without an encoder a file lacking dino_features fails like the reference: KeyError from the HDF5 lookup
    with pytest.raises(KeyError):
        run_inference([tmp_path / "raw" / "a.hdf"], tmp_path / "e2e.model", tmp_path / "seg_fail")


def test_graph_replay_equals_eager_config1_size(gpu):
    """hipGraph replay of the whole per-tomogram sequence (resize + ViT + head + Dice) is bit-identical to eager launches,
    for two different tomograms through ONE captured graph.  BASELINE configs[0] geometry (64x256x256) on ViT-g so that the
    1536-channel head can follow; depth cut to 16 slices to keep the test short."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.graph import GraphedTomogram
    from cryovit_amd.engine.head import HeadEngine
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict

    cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
    vit = VitEngine(cfg, random_state_dict(cfg, seed=2, device=gpu), gpu)
    head = HeadEngine(bench.synthetic_head_state_dict(5, gpu), gpu)
    D, H, W = 16, 256, 256
    g = GraphedTomogram(vit, head, D, H, W, slice_batch=8, mask_threshold=0.5)
    rng = np.random.default_rng(3)
    for seed in (1, 2):
        vol = torch.from_numpy(rng.integers(0, 256, (D, H, W), dtype=np.uint8)).to(gpu)
        labels = torch.from_numpy(rng.integers(-1, 2, (D, H, W)).astype(np.int8)).to(gpu)
        out = g.run(vol, labels)
        probs, dice, mask, f16 = out["probs"].clone(), out["dice_sums"].clone(), out["mask"].clone(), out["feats_f16"].clone()
        f16_e = torch.zeros_like(f16)
        cl = torch.zeros(ops.alloc_rows(D * 16 * 16), cfg.dim, dtype=torch.float16, device=gpu)
        for d0 in range(0, D, 8):
            vit.features(vol[d0 : d0 + 8], feats_f16=f16_e, d_total=D, d0=d0, feats_cl=cl[d0 * 256 :])
        ref = head.forward(cl, D, 16, 16, labels=labels, mask_threshold=0.5)
        assert torch.equal(f16, f16_e) and torch.equal(probs, ref["probs"]) and torch.equal(mask, ref["mask"])
        assert torch.equal(dice, ref["dice_sums"]) and float(dice[1]) > 0
    with pytest.raises(ValueError):
        g.run(torch.zeros(D, H, W + 16, dtype=torch.uint8, device=gpu))


def test_eval_model_entry_point_against_oracle(gpu, tmp_path):
    """``python -m cryovit_amd.training.eval_model`` (reference: training/eval_model.py:16-44, run/eval_model.py:143-197)
    starting from files on disk: ``weights.pt`` under the experiment directory, ``csv/splits.csv``, tomograms holding
    ``dino_features`` + ``labels/mito`` + ``data``.  Checks the experiment-directory contract, that records go through
    ``TomoDataset`` -> ``collate_fn`` -> ``CryoVIT.test_step``, the prediction files (TestPredictionWriter layout) and the
    metrics CSV (CsvWriter) against the fp32 CPU oracle head on the same weights: Dice within 1e-3, F1 within 1e-3,
    probabilities within sigmoid'(0) * 5e-2 of the oracle's."""
    import csv

    from cryovit_amd import io
    from cryovit_amd.training import eval_model as entry
    from oracle import dice as od
    from oracle import head as oh

    ref = oh.CryoVITHead()
    oh.rescaled_init_(ref, seed=5)
    data_dir, exp_dir = tmp_path / "data", tmp_path / "exp"
    name = "single_any_cryovit_mito"
    (exp_dir / name / "Q109" / "split_1").mkdir(parents=True)
    torch.save(ref.state_dict(), exp_dir / name / "Q109" / "split_1" / "weights.pt")
    (data_dir / "csv").mkdir(parents=True)
    rng = np.random.default_rng(31)
    rows, truth = [], {}
    for i, (D, split) in enumerate(((6, 1), (9, 0), (7, 1))):
        vol = rng.integers(0, 256, size=(D, 48, 32), dtype=np.uint8)
        feats = rng.standard_normal((1536, D, 3, 2)).astype(np.float16)
        lab = rng.integers(-1, 2, size=(D, 48, 32)).astype(np.int8)
        p = data_dir / "tomograms" / "Q109" / f"t{i}.hdf"
        p.parent.mkdir(parents=True, exist_ok=True)
        with io.FileWriter(p) as f:
            f.create_dataset("data", vol, compression="gzip")
            f.create_dataset("dino_features", feats)
            f.create_dataset("labels/mito", lab, compression="gzip")
        rows.append({"sample": "Q109", "tomo_name": f"t{i}.hdf", "split_id": split})
        truth[f"t{i}.hdf"] = (vol, feats, lab, split)
    with open(data_dir / "csv" / "splits.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample", "tomo_name", "split_id"])
        w.writeheader()
        w.writerows(rows)
    entry.main(["model=cryovit", "datamodule=single", "datamodule.sample=Q109", "datamodule.split_id=1", "label_key=mito",
                f"paths.model_dir={tmp_path}", f"paths.data_dir={data_dir}", f"paths.exp_dir={exp_dir}",
                f"paths.results_dir={tmp_path / 'results'}"])
    # split_id=1, no test_sample -> the test records are the validation split: t0 and t2
    got = list(csv.DictReader(open(tmp_path / "results" / "results" / name / "Q109_1.csv")))
    assert [r["tomo_name"] for r in got] == ["t0.hdf", "t2.hdf"] and list(got[0]) == ["sample", "tomo_name", "dice_metric", "f1_metric", "split_id"]
    for r in got:
        vol, feats, lab, split = truth[r["tomo_name"]]
        assert int(r["split_id"]) == split == 1
        with torch.no_grad():
            probs = torch.sigmoid(ref.forward_volume(torch.from_numpy(feats).float()[None])[0, 0])
        labt = torch.from_numpy(lab).float()
        want_dice = od.dice_metric(probs, labt)
        m = labt > -1
        ph, y = (probs[m] > 0.5).double(), labt[m].double()
        tp, fp, fn = (y * ph).sum(), ((1 - y) * ph).sum(), (y * (1 - ph)).sum()
        pr, rc = tp / (tp + fp + 1e-6), tp / (tp + fn + 1e-6)
        want_f1 = float(2 * pr * rc / (pr + rc + 1e-6))
        assert abs(float(r["dice_metric"]) - want_dice) <= 1e-3, (r, want_dice)
        assert abs(float(r["f1_metric"]) - want_f1) <= 1e-3, (r, want_f1)
        pred = tmp_path / "results" / "predictions" / name / "Q109" / r["tomo_name"]
        assert sorted(io.list_keys(pred)) == ["data", "mito", "mito_preds"]
        pp = io.read_dataset(pred, "mito_preds")
        assert pp.dtype == np.float32 and pp.shape == lab.shape
        assert np.abs(pp - probs.numpy()).max() <= 5e-2 / 4 + 1e-4
        assert np.array_equal(io.read_dataset(pred, "mito"), lab.astype(np.float32))  # collated labels are float (utils.py:40)
        assert np.array_equal(io.read_dataset(pred, "data"), vol)  # aux "data" as stored in the source file
    assert not (tmp_path / "results" / "predictions" / name / "Q109" / "t1.hdf").exists()
    # the experiment-directory contract: a missing directory is an (logged, swallowed) error, nothing is written
    entry.main(["model=cryovit", "datamodule=single", "datamodule.sample=Q18", "label_key=mito", f"paths.model_dir={tmp_path}",
                f"paths.data_dir={data_dir}", f"paths.exp_dir={exp_dir}", f"paths.results_dir={tmp_path / 'results2'}"])
    assert not (tmp_path / "results2" / "results").exists()


def test_checkpoint_file_loader_upstream_layout(gpu, tmp_path):
    """The path real weights take (reference load site run/dino_features.py:25-28,336): a state_dict file
    in the UPSTREAM key layout under ``<model_dir>/<name>4_pretrain.pth`` -> ``load_encoder(model_dir=...)`` (weights_only
    load, key-by-key packing) -> features equal to those of an engine built from the in-memory dict, and within tolerance
    of the CPU oracle on the same weights.  Extra keys the upstream file carries (``mask_token``) are accepted; a file
    lacking a block tensor is refused with the key's name."""
    from cryovit_amd.models import load_encoder
    from cryovit_amd.models.encoder import CHECKPOINT_FILES
    from oracle import dinov2 as o
    from oracle import preprocess as opre

    sd = o.init_state_dict(o.VITS14_REG, 71)
    assert CHECKPOINT_FILES["dinov2_vits14_reg"] == "dinov2_vits14_reg4_pretrain.pth"
    torch.save(sd, tmp_path / CHECKPOINT_FILES["dinov2_vits14_reg"])
    enc = load_encoder("dinov2_vits14_reg", model_dir=tmp_path, device=gpu)
    vol = np.random.default_rng(72).integers(0, 256, size=(2, 64, 96), dtype=np.uint8)
    f16, _ = enc.features_from_raw(torch.from_numpy(vol), 2)
    mem = load_encoder("dinov2_vits14_reg", checkpoint=tmp_path / CHECKPOINT_FILES["dinov2_vits14_reg"], device=gpu)
    f16b, _ = mem.features_from_raw(torch.from_numpy(vol), 2)
    assert torch.equal(f16, f16b)
    ref = o.forward_features(o.VITS14_REG, sd, opre.dino_transform(opre.load_scale(vol)))["x_norm_patchtokens"]  # [2, 24, 384]
    got = f16.float().cpu().permute(1, 2, 3, 0).reshape(2, -1, 384)
    err = (got - ref).abs()
    assert float(err.max()) <= 1e-1 and float(err.mean()) <= 1e-2, (float(err.max()), float(err.mean()))
    broken = {k: v for k, v in sd.items() if k != "blocks.3.mlp.fc2.weight"}
    torch.save(broken, tmp_path / "broken.pth")
    with pytest.raises(KeyError, match="blocks.3.mlp.fc2.weight"):
        load_encoder("dinov2_vits14_reg", checkpoint=tmp_path / "broken.pth", device=gpu)
    with pytest.raises(FileNotFoundError):
        load_encoder("dinov2_vits14_reg", model_dir=tmp_path / "nowhere", device=gpu)
