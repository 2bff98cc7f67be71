"""``cryovit infer`` flow (mirror of ``/root/reference/src/cryovit/run/infer_model.py:18-85``).

``run_inference(data_files, model_path, result_dir, threshold)`` keeps the reference's signature and result: one
``<tomogram stem>.hdf`` per input under ``result_dir`` holding ``data`` (float32, gzip) and ``<label_key>_preds`` (uint8 =
probabilities >= threshold, gzip) -- ``PredictionWriter.write_on_batch_end`` (``models/callbacks.py:81-109``) -- and the list
of written paths.  What replaces the Lightning ``trainer.predict`` loop: one tomogram per step through
``FileDataset`` -> ``collate_fn`` -> the HIP head; the threshold is applied in the head's last kernel so 1 byte per voxel
leaves the GPU; a writer thread gzips tomogram i-1 while the GPU runs i.  With ``torch.distributed.run`` the files are
sharded over the ranks (no collective on the data path; rank 0 learns the other ranks' paths through one object gather).

Extension (not in the reference): ``encoder=`` -- a loaded DINOv2 encoder.  The reference requires ``dino_features`` to be
in every input file (produced by ``cryovit features``); with an encoder, files that only hold ``data`` are encoded on the
fly and the features go to the head in HBM without a round trip through the file system.
"""

from __future__ import annotations

import logging
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from cryovit_amd import io
from cryovit_amd.config import compose, instantiate
from cryovit_amd.datasets import collate_fn
from cryovit_amd.run import writers
from cryovit_amd.run.sharding import gather_rows, select_device, shard_records, world_info
from cryovit_amd.types import FileData
from cryovit_amd.utils import load_data, load_model


def _has_key(path: Path, key: str) -> bool:
    if path.suffix not in (".h5", ".hdf", ".hdf5"):
        return False
    try:
        node = "/"
        for part in key.split("/"):
            if part not in io.list_keys(path, node):
                return False
            node = part if node == "/" else f"{node}/{part}"
        return True
    except Exception:  # noqa: BLE001
        return False


@torch.inference_mode()
def _predict_file(model, dataset, idx: int, threshold: float, encoder, batch_size: int):
    """(raw data [D,H,W] float32, uint8 segmentation [D,H,W]) of one file."""
    fd = dataset.files[idx]
    if encoder is not None and model.input_key == "dino_features" and not _has_key(fd.tomo_path, "dino_features"):
        raw = load_data(fd.tomo_path, key="data")[0].squeeze(0)
        vol = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float32))
        _, cl = encoder.features_from_raw(vol, batch_size, want_f16=False, want_cl=True)
        hp, wp, *_ = encoder.engine.geometry(vol.shape[1], vol.shape[2])
        out = model.engine().forward(cl, vol.shape[0], hp, wp, want_probs=False, mask_threshold=threshold)
        return raw, out["mask"]
    item = dataset[idx]
    batch = collate_fn([item])
    mask = model.predict_mask(batch, threshold)[0]
    return item.aux_data["data"], mask


def run_inference(data_files: list[Path], model_path: Path, result_dir: Path, threshold: float = 0.5, *, encoder=None,
                  batch_size: int = 128, device: str | None = None) -> list[Path]:
    rank, _, world = world_info()
    device = select_device(device)
    model, model_type, model_name, label_key = load_model(model_path, device=device)
    assert model is not None, "Loaded model is None."
    cfg = compose("infer_model", [f"name={model_name}", f"label_key={label_key}", f"model={model_type.value}", "datamodule=file"])
    cfg.paths.results_dir = result_dir
    input_key = cfg.model.input_key if cfg.model.input_key == "dino_features" else None  # else: find available data instead
    dataset_fn = instantiate(cfg.datamodule.dataset, input_key=input_key, label_key=label_key)
    files = [FileData(tomo_path=Path(f)) for f in data_files]
    if len(files) == 0:
        raise ValueError("No prediction data provided.")
    dataset = dataset_fn(files, train=False)
    logging.info("Setup dataset.")
    result_dir = Path(result_dir)
    mine = shard_records(files, rank, world)
    logging.info("Starting prediction.")
    paths: list[tuple[int, str]] = []
    with ThreadPoolExecutor(max_workers=2) as writer:
        pending = []
        for i in mine:
            raw, mask = _predict_file(model, dataset, i, threshold, encoder, batch_size)
            host = torch.empty(mask.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(mask, non_blocking=True)
            torch.cuda.current_stream(mask.device).synchronize()
            pending.append((i, writer.submit(writers.write_segmentation, result_dir, files[i].tomo_path.name, label_key, raw, host.numpy())))
            while len(pending) > 2:
                j, fut = pending.pop(0)
                paths.append((j, str(fut.result())))
        paths += [(j, str(fut.result())) for j, fut in pending]
    rows = gather_rows([{"i": j, "p": p} for j, p in paths], world)
    return [Path(r["p"]) for r in sorted(rows, key=lambda r: r["i"])]
