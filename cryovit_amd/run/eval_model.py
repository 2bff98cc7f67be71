"""Evaluation runner (mirror of ``/root/reference/src/cryovit/run/eval_model.py:100-197``): experiment directory layout,
``weights.pt`` from disk, the test records of the datamodule, one tomogram per step through
``TomoDataset`` -> ``collate_fn`` -> ``CryoVIT.test_step`` (HIP head + masked metrics), results handed to the configured
callbacks (``TestPredictionWriter``, ``CsvWriter``).

What replaces ``pytorch_lightning.Trainer.test``: the loop below.  A reader thread loads and collates tomogram i+1 while the
GPU works on i; under ``torch.distributed.run`` the test records are sharded over the ranks (tomograms are independent, no
data-path collective) and every rank writes the files of its own tomograms; the CSV rows are appended by rank 0 after an
object gather so two ranks never rewrite one CSV file concurrently.
"""

from __future__ import annotations

import logging
import random
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from cryovit_amd.config import instantiate
from cryovit_amd.datasets import collate_fn
from cryovit_amd.run.sharding import gather_rows, select_device, shard_records, world_info


def setup_exp_dir(cfg):
    """l.100-140: ``exp_dir/<name>/<sample>[/split_<id>][/test_<sample>]`` must exist (training wrote it); ``ckpt_path``
    defaults to its ``weights.pt``."""
    p = cfg.paths
    p.model_dir, p.data_dir, p.exp_dir, p.results_dir = Path(p.model_dir), Path(p.data_dir), Path(p.exp_dir), Path(p.results_dir)

    def joined(s):
        return "_".join(sorted(s)) if isinstance(s, (list, tuple)) else s

    sample, test_sample = joined(cfg.datamodule.sample), joined(cfg.datamodule.test_sample)
    new_exp_dir = p.exp_dir / cfg.name / sample
    if cfg.datamodule.split_id is not None:
        new_exp_dir = new_exp_dir / f"split_{cfg.datamodule.split_id}"
    if "Fractional" in cfg.datamodule._target_ and test_sample is not None:
        new_exp_dir = new_exp_dir / f"test_{test_sample}"
    p.results_dir.mkdir(parents=True, exist_ok=True)
    assert new_exp_dir.exists(), f"Experiment directory {new_exp_dir} does not exist. Run training first."
    p.exp_dir = new_exp_dir
    cfg.ckpt_path = Path(cfg.ckpt_path) if cfg.get("ckpt_path") is not None else new_exp_dir / "weights.pt"
    return cfg


def test_loop(model, dataset, callbacks) -> list:
    """``trainer.test``: every tomogram of ``dataset`` owned by this rank -> ``test_step`` -> callbacks."""
    rank, _, world = world_info()
    mine = shard_records(list(range(len(dataset))), rank, world)
    file_cbs = [cb for cb in callbacks if not hasattr(cb, "results_dir") or type(cb).__name__ != "CsvWriter"]
    csv_cbs = [cb for cb in callbacks if type(cb).__name__ == "CsvWriter"]
    results = []

    def load(i):
        return collate_fn([dataset[i]])

    with ThreadPoolExecutor(max_workers=1) as reader:
        nxt = reader.submit(load, mine[0]) if mine else None
        for k, i in enumerate(mine):
            batch = nxt.result()
            nxt = reader.submit(load, mine[k + 1]) if k + 1 < len(mine) else None
            out = model.test_step(batch, k)
            logging.info("[rank %d] %s/%s %s", rank, out.samples[0], out.tomo_names[0],
                         " ".join(f"{m}={v:.4f}" for m, v in out.metrics.items()))
            for cb in file_cbs:
                cb.on_test_batch_end(None, model, out, batch, k)
            out.data, out.label, out.preds = [], [], []  # the volumes are on disk now; keep only the small fields
            results.append(out)
    for out in gather_rows(results, world) if rank == 0 or world > 1 else results:
        if rank == 0:
            for cb in csv_cbs:
                cb.on_test_batch_end(None, model, out, None, 0)
    return results


def run_trainer(cfg) -> None:
    random.seed(cfg.random_seed)
    np.random.seed(cfg.random_seed)
    torch.manual_seed(cfg.random_seed)
    cfg = setup_exp_dir(cfg)
    assert cfg.ckpt_path is not None and cfg.ckpt_path.exists(), f"{cfg.paths.exp_dir} does not contain a checkpoint."

    dataset_fn = instantiate(cfg.datamodule.dataset)
    split_file = cfg.paths.data_dir / cfg.paths.csv_name / cfg.paths.split_name
    dm_node = {k: v for k, v in cfg.datamodule.items() if k not in ("dataset", "dataloader")}
    datamodule = instantiate(dm_node)(split_file=split_file, dataloader_fn=None, dataset_fn=dataset_fn)
    logging.info("Setup dataset.")

    callbacks = [instantiate(cb_cfg) for cb_cfg in cfg.callbacks.values()]
    device = select_device((cfg.get("trainer") or {}).get("device"))
    if cfg.model._target_.rsplit(".", 1)[-1] not in ("CryoVIT", "UNet3D"):
        raise NotImplementedError(f"{cfg.model._target_}: the CryoVIT head and the UNet3D baseline are built (SAM2 / MedSAM segmentation are out of scope)")
    model = instantiate(cfg.model, device=device)
    if cfg.ckpt_path.suffix == ".pt":
        # weights_only: a state_dict needs nothing else, and nothing from the file is ever executed
        model.load_state_dict(torch.load(cfg.ckpt_path, map_location="cpu", weights_only=True))
    elif cfg.ckpt_path.suffix == ".ckpt":
        raise ValueError("Lightning .ckpt files pickle arbitrary objects; export the state_dict to weights.pt instead")
    else:
        raise ValueError(f"Unsupported checkpoint format: {cfg.ckpt_path.suffix}. Use .pt or .ckpt files.")
    logging.info("Setup model.")

    logging.info("Starting testing.")
    test_loop(model, datamodule.test_dataset(), callbacks)
