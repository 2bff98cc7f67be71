"""CPU suite, part 4: host-side mirror of the reference interface -- config composition / validation, the on-disk
contract of ``_save_data``, dataset loading, tomogram sharding incl. a world_size-2 gloo run."""

import logging
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_compose_defaults_and_overrides():
    from cryovit_amd.config import compose, missing_keys

    cfg = compose("dino_features", ["paths.model_dir=/m", "paths.data_dir=/d", "paths.exp_dir=/e", "sample=Q109", "batch_size=64"])
    assert cfg.batch_size == 64 and cfg.sample == "Q109" and cfg.use_sam is False and cfg.export_features is False
    assert cfg.model_dir == "/m/DINOv2" and cfg.paths.results_dir == "/e"  # ${paths.*} interpolation
    assert cfg.paths.tomo_name == "tomograms" and cfg.paths.feature_name == "dino_features" and cfg.paths.csv_name == "csv"
    assert cfg.datamodule.dataset._target_ == "cryovit_amd.datasets.VITDataset" and cfg.datamodule.dataset._partial_ is True
    assert cfg.datamodule.dataset.data_root == "/d/tomograms"
    # dino.yaml overrides the dataloader defaults: no collation, no workers (reference configs/datamodule/dino.yaml)
    assert cfg.datamodule.dataloader.batch_size is None and cfg.datamodule.dataloader.num_workers == 0
    assert missing_keys(cfg) == []


def test_missing_keys_exit_1(caplog):
    from cryovit_amd.training import dino_features

    with caplog.at_level(logging.ERROR), pytest.raises(SystemExit) as e:
        dino_features.main(["sample=Q109"])
    assert e.value.code == 1
    assert "paths.data_dir" in caplog.text


def test_runtime_errors_are_logged_not_raised(tmp_path, caplog, monkeypatch):
    """training/dino_features.py:33-37 of the reference: any exception is logged and the process returns normally."""
    from cryovit_amd.run import dino_features as run_mod
    from cryovit_amd.training import dino_features

    def boom(cfg):
        raise RuntimeError("no GPU here")

    monkeypatch.setattr(run_mod, "run_trainer", boom)
    with caplog.at_level(logging.ERROR):
        dino_features.main([f"paths.model_dir={tmp_path}", f"paths.data_dir={tmp_path}", f"paths.exp_dir={tmp_path}"])
    assert "RuntimeError: no GPU here" in caplog.text


def test_instantiate_partial_dataset(tmp_path):
    from cryovit_amd import io
    from cryovit_amd.config import compose, instantiate

    vol = np.random.default_rng(0).integers(0, 256, size=(5, 20, 24), dtype=np.uint8)
    (tmp_path / "tomograms" / "Q109").mkdir(parents=True)
    with io.FileWriter(tmp_path / "tomograms" / "Q109" / "t0.hdf") as f:
        f.create_dataset("data", vol, compression="gzip")
    cfg = compose("dino_features", [f"paths.model_dir={tmp_path}", f"paths.data_dir={tmp_path}", f"paths.exp_dir={tmp_path}"])
    ds = instantiate(cfg.datamodule.dataset, data_root=tmp_path / "tomograms" / "Q109", use_sam=False)(records=["t0.hdf"])
    x = ds[0]
    assert x.dtype == torch.uint8 and tuple(x.shape) == (5, 20, 24) and np.array_equal(x.numpy(), vol)
    with pytest.raises(IndexError):
        ds[1]
    # float volumes come back as float32, unscaled (vit_dataset.py:86-88 only rescales uint8)
    with io.FileWriter(tmp_path / "tomograms" / "Q109" / "t1.hdf") as f:
        f.create_dataset("data", (vol / 255.0).astype(np.float64))
    ds = instantiate(cfg.datamodule.dataset, data_root=tmp_path / "tomograms" / "Q109", use_sam=False)(records=["t1.hdf"])
    assert ds[0].dtype == torch.float32


def test_save_data_layout(tmp_path):
    """App. C: ``data`` gzip, every other source leaf under ``labels/`` gzip, ``dino_features`` contiguous float16;
    stale source features are dropped."""
    from cryovit_amd import io
    from cryovit_amd.io.hdf5 import H5Reader
    from cryovit_amd.run.dino_features import _save_data

    rng = np.random.default_rng(1)
    src = {"data": rng.integers(0, 256, size=(6, 32, 32), dtype=np.uint8), "mito": rng.integers(-1, 2, size=(6, 32, 32)).astype(np.int8),
           "dino_features": np.zeros((4, 6, 2, 2), np.float16)}
    feats = rng.standard_normal((8, 6, 2, 2)).astype(np.float16)
    _save_data(src, feats, "t.hdf", tmp_path / "out" / "Q109")
    p = tmp_path / "out" / "Q109" / "t.hdf"
    flat = io.read_all_flat(p)
    assert sorted(flat) == ["data", "dino_features", "mito"]
    assert np.array_equal(flat["data"], src["data"]) and np.array_equal(flat["mito"], src["mito"])
    assert flat["dino_features"].dtype == np.float16 and np.array_equal(flat["dino_features"], feats)
    if not io.HAVE_H5PY:
        with H5Reader(p) as f:
            assert f["data"]._layout[0] == "chunked" and f["labels/mito"]._layout[0] == "chunked"
            assert f["dino_features"]._layout[0] == "contiguous" and f["dino_features"]._filters == []


def test_collate_fn_layout():
    from cryovit_amd.datasets import collate_fn
    from cryovit_amd.types import TomogramData

    t = TomogramData("Q109", "a.hdf", torch.zeros(16, 5, 2, 3, dtype=torch.float16), torch.zeros(5, 32, 48, dtype=torch.int8))
    u = TomogramData("Q109", "b.hdf", torch.zeros(16, 3, 2, 3, dtype=torch.float16), torch.zeros(3, 32, 48, dtype=torch.int8))
    b = collate_fn([t, u])
    assert tuple(b.tomo_batch.shape) == (2, 5, 16, 2, 3) and b.tomo_batch.dtype == torch.float32
    assert tuple(b.labels.shape) == (2, 5, 32, 48) and torch.all(b.labels[1, 3:] == -1)  # padded depth is "ignore"
    assert b.min_slices == 3 and b.metadata.identifiers == (["Q109", "Q109"], ["a.hdf", "b.hdf"]) and b.metadata.split_id is None


def test_cryovit_state_dict_is_reference_compatible():
    from cryovit_amd.models import CryoVIT
    from oracle import head as oh

    ref = oh.CryoVITHead()
    mine = CryoVIT()
    assert [(k, tuple(v.shape)) for k, v in ref.state_dict().items()] == [(k, tuple(v.shape)) for k, v in mine.state_dict().items()]
    mine.load_state_dict(ref.state_dict())  # strict


def test_shard_records_partition():
    from cryovit_amd.run.sharding import shard_records

    recs = [f"t{i}" for i in range(11)]
    for world in (1, 2, 3, 8):
        parts = [shard_records(recs, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(11))
    w = [5, 1, 1, 1, 9, 2, 2, 8, 1, 1, 1]
    parts = [shard_records(recs, r, 2, w) for r in range(2)]
    assert sorted(i for p in parts for i in p) == list(range(11))
    loads = [sum(w[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 1  # greedy balance


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from cryovit_amd.run.sharding import gather_rows, shard_records, world_info
rank, local, world = world_info()
dist.init_process_group("gloo")
recs = [f"t{i}.hdf" for i in range(7)]
mine = shard_records(recs, rank, world)
rows = gather_rows([(recs[i], rank, float(i) / 10) for i in mine], world)
if rank == 0:
    assert sorted(r[0] for r in rows) == sorted(recs), rows
    assert {r[1] for r in rows} == {0, 1}
    print("GATHER_OK", len(rows))
dist.destroy_process_group()
"""


def test_two_rank_gloo_shard_and_gather(tmp_path):
    """The N>1 path: two processes (gloo), disjoint shards, rows gathered on rank 0 -- no data-path collective."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), str(ROOT)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GATHER_OK 7" in r.stdout


def test_metrics_csv_semantics(tmp_path, caplog):
    """CsvWriter (callbacks.py:112-206): per-sample file, header sample,tomo_name,<metrics>, re-evaluated rows replaced."""
    import csv

    from cryovit_amd.run.writers import update_metrics_csv

    p = update_metrics_csv(tmp_path, "Q109", "a.hdf", {"dice_metric": 0.5})
    update_metrics_csv(tmp_path, "Q109", "b.hdf", {"dice_metric": 0.25})
    with caplog.at_level(logging.WARNING):
        update_metrics_csv(tmp_path, "Q109", "a.hdf", {"dice_metric": 0.75})
    assert "already has an entry" in caplog.text
    rows = list(csv.DictReader(open(p)))
    assert p.name == "Q109.csv" and list(rows[0].keys()) == ["sample", "tomo_name", "dice_metric"]
    assert {r["tomo_name"]: float(r["dice_metric"]) for r in rows} == {"a.hdf": 0.75, "b.hdf": 0.25}
    p2 = update_metrics_csv(tmp_path, "Q109", "a.hdf", {"dice_metric": 0.1}, split_id=3)
    assert p2.name == "Q109_3.csv" and list(csv.DictReader(open(p2)))[0]["split_id"] == "3"


def test_prediction_file_formats(tmp_path):
    from cryovit_amd import io
    from cryovit_amd.run.writers import write_prediction, write_test_prediction

    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, size=(4, 16, 16), dtype=np.uint8)
    lab = rng.integers(-1, 2, size=(4, 16, 16)).astype(np.int8)
    preds = rng.random((4, 16, 16)).astype(np.float32)
    p = write_test_prediction(tmp_path, "Q109", "t.hdf", "mito", data, lab, preds)
    assert p == tmp_path / "Q109" / "t.hdf" and sorted(io.list_keys(p)) == ["data", "mito", "mito_preds"]
    assert np.array_equal(io.read_dataset(p, "mito_preds"), preds) and io.read_dataset(p, "mito_preds").dtype == np.float32
    q = write_prediction(tmp_path / "inf", "t.mrc", "mito", data, preds, 0.5)
    assert q.name == "t.hdf"
    seg = io.read_dataset(q, "mito_preds")
    assert seg.dtype == np.uint8 and np.array_equal(seg, (preds >= 0.5).astype(np.uint8))
    assert io.read_dataset(q, "data").dtype == np.float32


def test_sam_feature_file_layout_and_config(tmp_path):
    """``_save_data`` with SAM features (run/dino_features.py:133-146): ``sam_features/<key>/<level>`` uncompressed, the source's
    ``dino_features`` kept; ``sam_features`` config (configs/sam_features.yaml: use_sam, model = sam2, SAM2 model dir)."""
    import numpy as np

    from cryovit_amd import io
    from cryovit_amd.config import compose
    from cryovit_amd.engine.hiera import HIERA_CONFIGS
    from cryovit_amd.run.dino_features import _save_data
    from oracle import sam2_hiera as oh

    cfg = compose("sam_features", ["paths.model_dir=/m", "paths.data_dir=/d", "paths.exp_dir=/e", "sample=Q109"])
    assert cfg.use_sam is True and cfg.model.name == "SAM2" and cfg.model_dir == "/m/SAM2" and cfg.batch_size == 128
    assert HIERA_CONFIGS["sam2.1_hiera_l"].block_plan() == oh.HIERA_L.block_plan()  # engine plan == oracle plan (pinned vs HF)
    rng = np.random.default_rng(0)
    data = {"data": rng.integers(0, 255, (3, 8, 8), dtype=np.uint8), "mito": rng.integers(-1, 2, (3, 8, 8)).astype(np.int8),
            "dino_features": rng.standard_normal((4, 3, 1, 1)).astype(np.float16)}
    feats = {"vision_pos_enc": [np.broadcast_to(rng.standard_normal((1, 6, 2, 2)).astype(np.float16), (3, 6, 2, 2)),
                                rng.standard_normal((3, 6, 1, 1)).astype(np.float16)],
             "backbone_fpn": [rng.standard_normal((3, 6, 2, 2)).astype(np.float16), rng.standard_normal((3, 6, 1, 1)).astype(np.float16)]}
    _save_data(data, feats, "t.hdf", tmp_path)
    out = tmp_path / "t.hdf"
    assert sorted(io.list_keys(out)) == ["data", "dino_features", "labels", "sam_features"]
    assert sorted(io.list_keys(out, "sam_features")) == ["backbone_fpn", "vision_pos_enc"]
    assert io.list_keys(out, "sam_features/backbone_fpn") == ["0", "1"]
    for key, arrs in feats.items():
        for i, a in enumerate(arrs):
            got = io.read_dataset(out, f"sam_features/{key}/{i}")
            assert got.dtype == np.float16 and np.array_equal(got, a)
    assert np.array_equal(io.read_dataset(out, "dino_features"), data["dino_features"])
    assert np.array_equal(io.read_dataset(out, "labels/mito"), data["mito"])


def test_ops_reject_host_and_mixed_device_operands():
    """The op wrappers launch on the stream of the device their operands live on: host tensors (no CPU path) and operands
    spread over two devices are refused before any C call (ranks >= 1 must never fall through to GPU 0)."""
    import pytest
    import torch

    from cryovit_amd._lib import CvxError
    from cryovit_amd.engine import ops

    with pytest.raises(CvxError, match="no CPU path"):
        ops._dev_check(torch.zeros(4))

    class FakeDev:  # stands in for a device tensor (no GPU in the CPU suite)
        is_cuda = True

        def __init__(self, index):
            self.device, self.shape = torch.device("cuda", index), (4,)

        def is_contiguous(self):
            return True

    assert ops._dev_check(FakeDev(1), None, FakeDev(1)) == torch.device("cuda", 1)
    with pytest.raises(CvxError, match="different devices"):
        ops._dev_check(FakeDev(0), FakeDev(1))
    with pytest.raises(CvxError, match="without any device tensor"):
        ops._dev_check(None)


def test_select_device_follows_local_rank(monkeypatch):
    """Under torch.distributed.run the rank's device is cuda:LOCAL_RANK whatever the config asks for; it must be made the
    ACTIVE device (set_device) before anything is allocated."""
    import torch

    from cryovit_amd.run import sharding

    calls = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: calls.append(i))
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setenv("RANK", "5")
    assert sharding.select_device("cuda:0") == "cuda:5" and calls == [5]
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert sharding.select_device("cuda:2") == "cuda:2" and calls == [5, 2]
    assert sharding.select_device(None) == "cuda:0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    import pytest

    with pytest.raises(RuntimeError, match="only 1 device"):
        sharding.select_device("cuda:3")


def _write_tomo(path, D=4, with_data=True, seed=0):
    from cryovit_amd import io

    rng = np.random.default_rng(seed)
    path.parent.mkdir(parents=True, exist_ok=True)
    feats = rng.standard_normal((16, D, 2, 3)).astype(np.float16)
    lab = rng.integers(-1, 2, size=(D, 32, 48)).astype(np.int8)
    vol = rng.integers(0, 256, size=(D, 32, 48), dtype=np.uint8)
    with io.FileWriter(path) as f:
        if with_data:
            f.create_dataset("data", vol, compression="gzip")
        f.create_dataset("dino_features", feats)
        f.create_dataset("labels/mito", lab, compression="gzip")
        f.create_dataset("labels/granule", lab, compression="gzip")
    return feats, lab, vol


def test_tomo_dataset_and_datamodule_records(tmp_path):
    """tomo_dataset.py:89-146 + single/multi_sample_datamodule.py: record selection, item contents (input as stored, uint8
    raw input scaled and given a channel axis, aux keys that exist, split id), collate metadata, train=True."""
    import csv

    import pytest

    from cryovit_amd.config import compose, instantiate
    from cryovit_amd.datamodules import MultiSampleDataModule, SingleSampleDataModule
    from cryovit_amd.datasets import TomoDataset, collate_fn

    root = tmp_path / "tomograms"
    truth = {}
    rows = []
    for s, n, split in (("Q109", "a.hdf", 0), ("Q109", "b.hdf", 1), ("Q18", "c.hdf", 1), ("Q18", "d.hdf", 0)):
        truth[(s, n)] = _write_tomo(root / s / n, D=3 + len(rows), with_data=(n != "d.hdf"), seed=len(rows))
        rows.append({"sample": s, "tomo_name": n, "split_id": split})
    (tmp_path / "csv").mkdir()
    with open(tmp_path / "csv" / "splits.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample", "tomo_name", "split_id"])
        w.writeheader()
        w.writerows(rows)
    cfg = compose("eval_model", ["model=cryovit", "datamodule=single", "datamodule.sample=[Q109]", "datamodule.split_id=1", "label_key=mito",
                                 "paths.model_dir=/m", f"paths.data_dir={tmp_path}", f"paths.exp_dir={tmp_path}"])
    dataset_fn = instantiate(cfg.datamodule.dataset)
    dm = instantiate({k: v for k, v in cfg.datamodule.items() if k not in ("dataset", "dataloader")})(
        split_file=tmp_path / "csv" / "splits.csv", dataset_fn=dataset_fn, dataloader_fn=None)
    assert isinstance(dm, SingleSampleDataModule)
    assert [r["tomo_name"] for r in dm.test_df()] == ["b.hdf"] and [r["tomo_name"] for r in dm.train_df()] == ["a.hdf"]
    assert [r["tomo_name"] for r in dm.predict_df()] == ["a.hdf", "b.hdf"] and "split_id" not in dm.predict_df()[0]
    ds = dm.test_dataset()
    assert isinstance(ds, TomoDataset) and len(ds) == 1 and ds.aux_keys == ["data"]
    item = ds[0]
    feats, lab, vol = truth[("Q109", "b.hdf")]
    assert item.sample == "Q109" and item.tomo_name == "b.hdf" and item.split_id == 1
    assert item.data.dtype == torch.float16 and np.array_equal(item.data.numpy(), feats)
    assert item.label.dtype == torch.int8 and np.array_equal(item.label.numpy(), lab)
    assert np.array_equal(item.aux_data["data"], vol)  # aux data as stored (uint8)
    batch = collate_fn([item])
    assert batch.metadata.identifiers == (["Q109"], ["b.hdf"]) and batch.metadata.split_id.tolist() == [1]
    assert tuple(batch.tomo_batch.shape) == (1, 4, 16, 2, 3) and batch.labels.dtype == torch.float32 and batch.aux_data["data"][0] is item.aux_data["data"]
    with pytest.raises(IndexError):
        ds[1]
    # test on another sample: the whole sample, no split ids; missing aux keys are skipped (d.hdf has no "data")
    multi = MultiSampleDataModule(["Q109"], None, "split_id", test_sample=["Q18"], split_file=tmp_path / "csv" / "splits.csv",
                                  dataset_fn=dataset_fn)
    ds2 = multi.test_dataset()
    assert [r["tomo_name"] for r in multi.test_df()] == ["c.hdf", "d.hdf"] and ds2[0].split_id is None
    assert "data" in ds2[0].aux_data and ds2[1].aux_data == {}
    assert multi.val_df() == multi.train_df()  # split_id None: validate on the train set
    # raw input key: uint8 -> /255 float32 with a channel axis (l.118-121); other label key
    raw = TomoDataset([("Q109", "a.hdf")], input_key="data", label_key="granule", data_root=root, aux_keys=["labels/mito", "nope"])[0]
    assert raw.data.dtype == torch.float32 and tuple(raw.data.shape) == (1, 3, 32, 48)
    assert np.array_equal(raw.data[0].numpy(), truth[("Q109", "a.hdf")][2].astype(np.float32) / 255.0)
    assert list(raw.aux_data) == ["labels/mito"]
    with pytest.raises(AssertionError, match="Label key 'cristae' not found"):
        TomoDataset([("Q109", "a.hdf")], input_key="data", label_key="cristae", data_root=root)[0]
    # train=True: the random crop of tomo_dataset.py:148-178 (a 3 x 32 x 48 raw volume is smaller than 512 x 512: numpy slicing
    # past the end keeps what is there, like the reference)
    np.random.seed(3)
    tr = TomoDataset([("Q109", "a.hdf")], input_key="data", label_key="mito", data_root=root, train=True)[0]
    assert tuple(tr.data.shape) == (1, 3, 32, 48) and tuple(tr.label.shape) == (3, 32, 48)
    with pytest.raises(ValueError, match="No testing data"):
        SingleSampleDataModule(["Q53"], None, "split_id", split_file=tmp_path / "csv" / "splits.csv", dataset_fn=dataset_fn).test_dataset()


def test_eval_config_validation_and_exp_dir(tmp_path):
    """config.py:234-286 (missing keys / invalid samples -> exit 1) and run/eval_model.py:100-140 (experiment directory)."""
    import pytest

    from cryovit_amd.config import compose, validate_experiment_config
    from cryovit_amd.run.eval_model import setup_exp_dir

    with pytest.raises(SystemExit) as e:
        validate_experiment_config(compose("eval_model", ["model=cryovit", "datamodule=single", "datamodule.sample=Q109"]))  # label_key, paths
    assert e.value.code == 1
    base = ["model=cryovit", "datamodule=multi", "label_key=mito", "paths.model_dir=/m", f"paths.data_dir={tmp_path}", f"paths.exp_dir={tmp_path}"]
    with pytest.raises(SystemExit):
        validate_experiment_config(compose("eval_model", base + ["datamodule.sample=[Q109,NotASample]"]))
    cfg = compose("eval_model", base + ["datamodule.sample=[Q18,Q109]", "datamodule.split_id=3"])
    validate_experiment_config(cfg)
    assert cfg.name == "multi_any_cryovit_mito" and cfg.callbacks.csv_writer.results_dir == f"{tmp_path}/results/multi_any_cryovit_mito"
    with pytest.raises(AssertionError, match="Run training first"):
        setup_exp_dir(cfg)
    d = tmp_path / "multi_any_cryovit_mito" / "Q109_Q18" / "split_3"
    d.mkdir(parents=True)
    cfg = setup_exp_dir(cfg)
    assert cfg.paths.exp_dir == d and cfg.ckpt_path == d / "weights.pt"
