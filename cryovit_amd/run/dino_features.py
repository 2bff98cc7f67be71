"""Feature-stage runner (mirror of ``/root/reference/src/cryovit/run/dino_features.py``): same functions, same on-disk
result, arithmetic on the HIP engine.

  _dino_features   l.31-64    slices -> encoder -> float16 [C, D, H/16, W/16]
  _save_data       l.109-153  output HDF5: ``data`` + ``labels/<leaf>`` gzip, ``dino_features`` contiguous fp16
  _process_sample  l.156-205  enumerate records, per tomogram: features -> re-read source -> save
  run_trainer      l.304-350  paths (incl. the inverted src/dst naming), model load, sample loop
Multi-GPU: records of a sample are sharded over the ranks of a ``torch.distributed.run`` launch (run/sharding.py).
"""

from __future__ import annotations

import csv
import logging
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from cryovit_amd import io
from cryovit_amd.config import instantiate, samples, tomogram_exts
from cryovit_amd.models.encoder import load_encoder
from cryovit_amd.models.sam_encoder import load_sam_encoder
from cryovit_amd.run.sharding import select_device, shard_records, world_info


@torch.inference_mode()
def _dino_features(data: torch.Tensor, model, batch_size: int) -> np.ndarray:
    """Patch features of one tomogram as float16 ``[C, D, H/16, W/16]`` (C-contiguous, owned by the caller).

    ``data`` is either the raw volume ``[D,H,W]`` (uint8 / float32: the fused path -- resize happens on the GPU) or the
    reference's pre-resized ``[D,3,H',W']`` float32 tensor (protocol path through ``model.forward_features``)."""
    if data.dim() == 3:
        f16, _ = model.features_from_raw(data, batch_size, want_f16=True, want_cl=False)
        host = torch.empty(f16.shape, dtype=f16.dtype, pin_memory=True)  # pinned: the 403 MB D2H runs at PCIe speed
        host.copy_(f16, non_blocking=True)
        torch.cuda.current_stream(f16.device).synchronize()  # the copy runs on the stream of the TENSOR's device
        return host.numpy()
    hp, wp = data.shape[-2] // 14, data.shape[-1] // 14
    chunks = []
    for i in range(0, len(data), batch_size):
        vec = data[i : i + batch_size]
        f = model.forward_features(vec)["x_norm_patchtokens"]
        f = f.reshape(f.shape[0], hp, wp, -1).permute(3, 0, 1, 2).contiguous()
        chunks.append(f.half().cpu().numpy())
    return np.concatenate(chunks, axis=1)


class _PinnedRing:
    """A few page-locked host buffers of one shape, handed out round-robin and returned by the writer that consumed them: the
    403-MB device-to-host copy of tomogram i runs on a side stream while the GPU already works on tomogram i+1."""

    def __init__(self):
        self._free: dict[tuple, list[torch.Tensor]] = {}
        self._streams: dict[torch.device, torch.cuda.Stream] = {}
        import threading

        self._lock = threading.Lock()

    def take(self, shape, dtype) -> torch.Tensor:
        with self._lock:
            lst = self._free.setdefault((tuple(shape), dtype), [])
            return lst.pop() if lst else torch.empty(tuple(shape), dtype=dtype, pin_memory=True)

    def give(self, t: torch.Tensor) -> None:
        with self._lock:
            lst = self._free.setdefault((tuple(t.shape), t.dtype), [])
            if len(lst) < 4:
                lst.append(t)

    def stream(self, device) -> torch.cuda.Stream:
        with self._lock:
            if device not in self._streams:
                self._streams[device] = torch.cuda.Stream(device)
            return self._streams[device]


_ring = _PinnedRing()


@torch.inference_mode()
def _dino_features_async(data: torch.Tensor, model, batch_size: int):
    """The fused path of ``_dino_features`` WITHOUT its synchronisation: returns ``(host fp16 tensor, event, release)``; the array is
    valid once ``event.synchronize()`` returns, ``release()`` hands the pinned buffer back.  Used by ``_process_sample``'s pipeline
    (the writer thread waits for the event), so the launch thread never blocks on the copy of the tomogram it has just issued."""
    f16, _ = model.features_from_raw(data, batch_size, want_f16=True, want_cl=False)
    host = _ring.take(f16.shape, f16.dtype)
    cur = torch.cuda.current_stream(f16.device)
    side = _ring.stream(f16.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        host.copy_(f16, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    f16.record_stream(side)  # the caching allocator must not hand this block out again before the copy has read it
    return host, ev, (lambda: _ring.give(host))


@torch.inference_mode()
def _sam_features(data: torch.Tensor, model, batch_size: int) -> dict[str, list[np.ndarray]]:
    """SAM2 image-encoder features of one tomogram (mirror of l.67-106): every key of the encoder output except
    ``vision_features`` as a list (fine -> coarse) of float16 arrays ``[D, 256, h_l, w_l]``.

    ``data`` is the raw volume ``[D,H,W]`` (fused path) or the reference's ``[1,D,3,H,W]`` float tensor (protocol path
    through ``model.forward_features``; ``len(data) == 1``, so the whole tomogram is one call like upstream)."""
    if data.dim() == 3:
        return model.features_from_raw(data, batch_size)
    all_features: dict[str, list[np.ndarray]] = {}
    for i in range(0, len(data), batch_size):
        backbone = model.forward_features(data[i : i + batch_size])
        for key, feats in backbone.items():
            if key == "vision_features":
                continue
            arrs = [f.to("cpu").half().numpy() for f in feats]
            all_features[key] = arrs if key not in all_features else [np.concatenate([a, b], axis=0) for a, b in zip(all_features[key], arrs)]
    return all_features


def _save_data(data: dict[str, np.ndarray], features, tomo_name: str, dst_dir: Path) -> None:
    """``data`` (gzip), every other source leaf under ``labels/`` (gzip); DINO: ``dino_features`` uncompressed; SAM (dict of
    lists): the source's ``dino_features`` kept (gzip) and ``sam_features/<key>/<level>`` uncompressed (l.119-153)."""
    dst_dir.mkdir(parents=True, exist_ok=True)
    with io.FileWriter(dst_dir / tomo_name) as fh:
        for key, arr in data.items():
            if key == "dino_features":
                continue  # stale features of the source are dropped (re-added below for the SAM case)
            if key == "data":
                fh.create_dataset("data", arr, compression="gzip")
            else:
                fh.create_dataset(f"labels/{key}", arr, compression="gzip")
        if isinstance(features, dict):
            if "dino_features" in data:
                fh.create_dataset("dino_features", data["dino_features"], compression="gzip")
            for key, feats in features.items():
                for i, feat in enumerate(feats):
                    fh.create_dataset(f"sam_features/{key}/{i}", feat)
        else:
            fh.create_dataset("dino_features", features)


def _list_records(tomo_dir: Path, csv_file: Path) -> list[str]:
    if csv_file.exists():
        with open(csv_file, newline="") as f:
            return [row["tomo_name"] for row in csv.DictReader(f)]
    return sorted(f.name for f in tomo_dir.glob("*") if f.suffix in tomogram_exts)


def _process_sample(src_dir: Path, dst_dir: Path, csv_dir: Path, model, sample: str, datamodule, batch_size: int,
                    image_dir, use_sam: bool = False) -> list[str]:
    tomo_dir, result_dir = src_dir / sample, dst_dir / sample
    records = _list_records(tomo_dir, csv_dir / f"{sample}.csv")
    rank, _, world = world_info()
    mine = [records[i] for i in shard_records(records, rank, world)]
    dataset = instantiate(datamodule.dataset, data_root=tomo_dir, use_sam=use_sam)(records=mine)
    done = []
    # Three-stage host pipeline around the GPU (the reference runs these serially, SURVEY s.8a a3): a reader thread
    # decompresses tomogram i+1 while the GPU works on i, a writer thread gzips and writes i-1 (zlib drops the GIL).
    feature_fn = _sam_features if use_sam else _dino_features

    def save(i, features, ready=None, release=None):
        try:
            if ready is not None:
                ready.synchronize()  # the device-to-host copy of THIS tomogram (issued on a side stream) has landed
                features = features.numpy()
            _save_data(io.read_all_flat(tomo_dir / mine[i]), features, mine[i], result_dir)
        finally:
            if release is not None:
                release()
        shapes = features.shape if not isinstance(features, dict) else {k: [f.shape for f in v] for k, v in features.items()}
        logging.info("[rank %d] %s/%s -> %s %s", rank, sample, mine[i], "sam_features" if use_sam else "dino_features", shapes)
        return mine[i]

    with ThreadPoolExecutor(max_workers=1) as reader, ThreadPoolExecutor(max_workers=2) as writer:
        nxt = reader.submit(dataset.__getitem__, 0) if len(dataset) else None
        pending = []
        for i in range(len(dataset)):
            x = nxt.result()
            nxt = reader.submit(dataset.__getitem__, i + 1) if i + 1 < len(dataset) else None
            if not use_sam and torch.is_tensor(x) and x.dim() == 3 and hasattr(model, "features_from_raw"):
                host, ev, release = _dino_features_async(x, model, batch_size)  # no host synchronisation on the launch thread
                pending.append(writer.submit(save, i, host, ev, release))
            else:
                pending.append(writer.submit(save, i, feature_fn(x, model, batch_size)))
            while len(pending) > 2:  # bound the number of 400-MB feature arrays waiting to be written
                done.append(pending.pop(0).result())
        done += [f.result() for f in pending]
    if image_dir is not None:
        logging.warning("export_features=True: PCA colour maps are plotting (out of scope of this build) -- skipped")
    return done


def _load_model(cfg, enc: dict, device: str):
    """The frozen encoder: SAM2 image encoder for ``use_sam`` (l.325-331), else DINOv2 (l.332-336); weights from local files."""
    if cfg.use_sam:
        assert cfg.get("model") is not None, "SAM model configuration must be provided."
        return load_sam_encoder(enc.get("name") or cfg.model.get("name", "SAM2"), model_dir=cfg.model_dir, checkpoint=enc.get("checkpoint"),
                                synthetic_seed=enc.get("synthetic_seed"), device=device).cuda().eval()
    return load_encoder(enc.get("name", "dinov2_vitg14_reg"), model_dir=cfg.model_dir, checkpoint=enc.get("checkpoint"),
                        synthetic_seed=enc.get("synthetic_seed"), device=device).cuda().eval()


def run_trainer(cfg) -> None:
    paths = cfg.paths
    data_dir, exp_dir = Path(paths.data_dir), Path(paths.exp_dir)
    src_dir = data_dir / paths.feature_name  # sic: the reference reads from feature_name ...
    dst_dir = data_dir / paths.tomo_name     # ... and writes to tomo_name (run/dino_features.py:316-317)
    csv_dir = data_dir / paths.csv_name
    image_dir = exp_dir / "dino_images"
    sample = cfg.sample
    sample_names = [getattr(sample, "name", sample)] if sample is not None else [s for s in samples if (src_dir / s).exists()]
    enc = cfg.get("encoder", {}) or {}
    device = select_device(enc.get("device"))  # cuda:LOCAL_RANK under torch.distributed.run; sets the active device
    model = _load_model(cfg, enc, device)
    for sample_name in sample_names:
        _process_sample(src_dir, dst_dir, csv_dir, model, sample_name, cfg.datamodule, cfg.batch_size,
                        image_dir if cfg.export_features else None, cfg.use_sam)


## For scripts (``cryovit features``)


def run_dino(train_data: list[Path], result_dir: Path, batch_size: int, use_sam: bool = False, visualize: bool = False, *,
             encoder: dict | None = None) -> None:
    """Feature extraction over a list of tomogram files of any supported format (mirror of l.211-299): one
    ``<result_dir>/<stem>.hdf`` per input with ``data``, ``dino_features`` and the source's other datasets under ``labels/``.
    ``encoder`` overrides keys of the config's ``encoder`` block (name / checkpoint / synthetic_seed / device)."""
    from cryovit_amd.config import compose
    from cryovit_amd.types import FileData

    cfg = compose("dino_features" if not use_sam else "sam_features",
                  [f"batch_size={batch_size}", "sample=null", "export_features=False", "datamodule/dataset=file"])
    enc = dict(cfg.get("encoder", {}) or {})
    enc.update(encoder or {})
    rank, _, world = world_info()
    device = select_device(enc.get("device"))
    model = _load_model(cfg, enc, device)
    feature_fn = _sam_features if use_sam else _dino_features
    assert len(train_data) > 0, "No valid tomogram files found in the specified training data path."
    files = [FileData(tomo_path=Path(f)) for f in train_data]
    dataset = instantiate(cfg.datamodule.dataset, input_key=None, label_key=None)(files, for_dino=True, use_sam=use_sam)
    result_list = [Path(result_dir) / f"{Path(f).stem}.hdf" for f in train_data]
    if visualize:
        logging.warning("visualize=True: PCA colour maps are plotting (out of scope of this build) -- skipped")
    try:
        with ThreadPoolExecutor(max_workers=2) as writer:
            pending = []
            for i in shard_records(files, rank, world):
                x = dataset[i]
                features = feature_fn(x.data, model, cfg.batch_size)
                result_path = result_list[i].with_suffix(".hdf")
                pending.append(writer.submit(_save_data, x.aux_data, features, result_path.name, result_path.parent))
                while len(pending) > 2:
                    pending.pop(0).result()
            for f in pending:
                f.result()
    except torch.OutOfMemoryError:
        print(f"Ran out of GPU memory during DINO feature extraction. Try reducing the batch size. Current batch size is {cfg.batch_size}.")
        return
