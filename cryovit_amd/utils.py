"""File loading / model container utilities of the inference flow (mirror of ``/root/reference/src/cryovit/utils.py``;
SURVEY.md s.8f row N2).  Same function names, arguments, return values and error behaviour:

  read_hdf / read_mrc / read_tiff   l.115-183   -> (data, FileMetadata); HDF5 without a key: the dataset with the most unique values
  load_data                          l.186-225   any supported file -> float32-normalised [C, D, H, W] + the key used
  load_labels / _match_label_keys_to_data  l.228-301
  load_files_from_path               l.304-330   directory (recursive) or .txt listing
  save_model / save_model_from_weights / load_model   l.336-468

The reference reads .mrc through ``mrcfile`` and .tif through ``tifffile``; neither is a dependency here -- the MRC2014
header and baseline TIFF (strips; none / deflate / PackBits) are parsed directly.  ``.model`` files: the reference pickles a
dataclass holding Hydra config objects (l.354-381); unpickling executes code from the file, so this build uses a
container that ``torch.load(weights_only=True)`` can read (plain dict: name, model_type, label_key, model_cfg, weights)
and refuses pickled ``.model`` files with instructions to re-export them from ``weights.pt``.
"""

from __future__ import annotations

import logging
import struct
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np
import torch

from cryovit_amd import io
from cryovit_amd.config import tomogram_exts
from cryovit_amd.types import ModelType

#### Data loading ####


@dataclass
class FileMetadata:
    drange: tuple[float, float]
    dshape: tuple[int, ...]
    dtype: np.dtype
    nunique: int = 0


def _metadata(data: np.ndarray) -> FileMetadata:
    return FileMetadata(drange=(float(np.min(data)), float(np.max(data))), dshape=data.shape, dtype=data.dtype,
                        nunique=len(np.unique(data)))


def _walk_hdf(group, prefix: str = "") -> dict[str, np.ndarray]:
    out = {}
    for key in group.keys():
        obj = group[key]
        if isinstance(obj, io.H5Group) or (io.HAVE_H5PY and isinstance(obj, io.h5py.Group)):
            out.update(_walk_hdf(obj, f"{prefix}{key}/"))
        else:
            out[f"{prefix}{key}"] = obj.read() if isinstance(obj, io.H5Dataset) else obj[()]
    return out


def read_hdf(hdf_file, key: str | None = None) -> tuple[str, np.ndarray, FileMetadata]:
    """(key used, data, metadata).  ``key=None``: the dataset with the most unique values is taken to be the data.  A key that
    is not in the file logs a warning, reads everything and then fails with ``KeyError`` (reference l.75-92,138-143)."""
    opener = (lambda p: io.h5py.File(p, "r")) if io.HAVE_H5PY else io.H5Reader
    with opener(hdf_file) as fh:
        if key is not None:
            try:
                obj = fh[key]
                data = obj.read() if isinstance(obj, io.H5Dataset) else obj[()]
                return key, data, _metadata(data)
            except KeyError:
                logging.warning("Key %s not found in file %s. Attempting to read all keys instead.", key, hdf_file)
        datasets = _walk_hdf(fh)
    meta = {k: _metadata(v) for k, v in datasets.items()}
    if key is None:
        data_key = max(meta.items(), key=lambda kv: kv[1].nunique)[0]
        logging.info("No key specified for file %s. Assuming data is the key with the most unique values, and using key '%s' "
                     "with %d unique values. If this is incorrect, please specify the `data_key` manually as a `/`-separated "
                     "string.", hdf_file, data_key, meta[data_key].nunique)
    else:
        data_key = key
    return data_key, datasets[data_key], meta[data_key]


_MRC_MODES = {0: "i1", 1: "i2", 2: "f4", 6: "u2", 12: "f2"}


def _read_mrc_array(path) -> np.ndarray:
    """MRC2014: 1024-byte header (nx, ny, nz, mode at words 1-4; nsymbt = bytes of extended header at word 24; machine
    stamp at byte 212: 0x44 little-endian, 0x11 big-endian), then nz*ny*nx samples, x fastest.  Returned like
    ``mrcfile.read``: [nz, ny, nx], or [ny, nx] for a single image."""
    with open(path, "rb") as f:
        header = f.read(1024)
        if len(header) < 1024:
            raise ValueError(f"{path}: shorter than an MRC header")
        order = ">" if header[212] == 0x11 else "<"
        nx, ny, nz, mode = struct.unpack(order + "4i", header[:16])
        if not (0 < nx < 1 << 20 and 0 < ny < 1 << 20 and 0 < nz < 1 << 20):  # wrong guess or a pre-2000 file without a stamp
            order = ">" if order == "<" else "<"
            nx, ny, nz, mode = struct.unpack(order + "4i", header[:16])
        if mode not in _MRC_MODES:
            raise ValueError(f"{path}: unsupported MRC mode {mode} (supported: {sorted(_MRC_MODES)})")
        (nsymbt,) = struct.unpack(order + "i", header[92:96])
        f.seek(1024 + max(nsymbt, 0))
        dt = np.dtype(order + _MRC_MODES[mode])
        count = nx * ny * nz
        raw = f.read(count * dt.itemsize)
    if len(raw) != count * dt.itemsize:
        raise ValueError(f"{path}: data block truncated ({len(raw)} of {count * dt.itemsize} bytes)")
    data = np.frombuffer(raw, dtype=dt).astype(dt.newbyteorder("="), copy=True)
    return data.reshape(ny, nx) if nz == 1 else data.reshape(nz, ny, nx)


def read_mrc(mrc_file) -> tuple[np.ndarray, FileMetadata]:
    data = _read_mrc_array(mrc_file)
    return data, _metadata(data)


_TIFF_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d", 16: "Q", 17: "q"}


def _unpackbits(buf: bytes) -> bytes:
    out, i = bytearray(), 0
    while i < len(buf):
        n = buf[i]
        i += 1
        if n < 128:
            out += buf[i : i + n + 1]
            i += n + 1
        elif n > 128:
            out += buf[i : i + 1] * (257 - n)
            i += 1
    return bytes(out)


def _read_tiff_array(path) -> np.ndarray:
    """Baseline (and Big-) TIFF, one grayscale image per IFD, data in strips; compression none (1), deflate (8, 32946) or
    PackBits (32773).  Pages are stacked: [pages, H, W], a single page gives [H, W] (what ``tifffile.imread`` returns)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] not in (b"II", b"MM"):
        raise ValueError(f"{path}: not a TIFF file")
    bo = "<" if buf[:2] == b"II" else ">"
    (magic,) = struct.unpack(bo + "H", buf[2:4])
    big = magic == 43
    if magic not in (42, 43):
        raise ValueError(f"{path}: bad TIFF magic {magic}")
    (ifd,) = struct.unpack(bo + "Q", buf[8:16]) if big else struct.unpack(bo + "I", buf[4:8])
    pages = []
    while ifd:
        (n,) = struct.unpack(bo + ("Q" if big else "H"), buf[ifd : ifd + (8 if big else 2)])
        pos, esz, inl = ifd + (8 if big else 2), (20 if big else 12), (8 if big else 4)
        tags: dict[int, tuple] = {}
        for e in range(n):
            ent = buf[pos + e * esz : pos + (e + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", ent[:4])
            (cnt,) = struct.unpack(bo + ("Q" if big else "I"), ent[4 : 4 + inl])
            fmt = _TIFF_TYPES.get(typ)
            if fmt is None:
                continue
            size = struct.calcsize("=" + fmt) * cnt
            if size <= inl:
                raw = ent[4 + inl : 4 + inl + size]
            else:
                (off,) = struct.unpack(bo + ("Q" if big else "I"), ent[4 + inl : 4 + 2 * inl])
                raw = buf[off : off + size]
            tags[tag] = raw if typ == 2 else struct.unpack(bo + fmt * cnt, raw)
        nxt = buf[pos + n * esz : pos + n * esz + inl]
        (ifd,) = struct.unpack(bo + ("Q" if big else "I"), nxt)
        w, h = tags[256][0], tags[257][0]
        spp = tags.get(277, (1,))[0]
        bits = tags.get(258, (1,))[0]
        comp = tags.get(259, (1,))[0]
        fmt_code = tags.get(339, (1,))[0]
        if spp != 1 or 322 in tags or tags.get(317, (1,))[0] != 1:
            raise ValueError(f"{path}: only single-sample strip TIFFs without predictor are supported")
        kind = {1: "u", 2: "i", 3: "f"}.get(fmt_code, "u")
        if bits not in (8, 16, 32, 64):
            raise ValueError(f"{path}: unsupported bit depth {bits}")
        dt = np.dtype(f"{bo}{kind}{bits // 8}")
        chunks = []
        for off, cnt in zip(tags[273], tags[279]):
            s = buf[off : off + cnt]
            if comp in (8, 32946):
                s = zlib.decompress(s)
            elif comp == 32773:
                s = _unpackbits(s)
            elif comp != 1:
                raise ValueError(f"{path}: unsupported TIFF compression {comp}")
            chunks.append(s)
        img = np.frombuffer(b"".join(chunks)[: w * h * dt.itemsize], dtype=dt).reshape(h, w)
        pages.append(img.astype(dt.newbyteorder("=")))
    if not pages:
        raise ValueError(f"{path}: no images")
    return pages[0] if len(pages) == 1 else np.stack(pages)


def read_tiff(tiff_file) -> tuple[np.ndarray, FileMetadata]:
    data = _read_tiff_array(tiff_file)
    return data, _metadata(data)


_UNSUPPORTED = ("Unsupported file format for file {}. Supported formats are .h5, .hdf, .hdf5, .mrc, .mrcs, .tiff, .tif, "
                "and image folders.")


def load_data(file_path, key: str | None = None) -> tuple[np.ndarray, str]:
    """Data of one tomogram file as ``[C, D, H, W]`` (3-D data gets a channel axis); 8/16-bit integer data is scaled by
    1/255 to float32, float data (already normalised, or DINO features) is returned as stored (reference l.186-225)."""
    file_path = Path(file_path)
    found_key = ""
    if not file_path.exists():
        raise FileNotFoundError(f"File {file_path} does not exist.")
    if file_path.suffix in (".h5", ".hdf", ".hdf5"):
        found_key, data, metadata = read_hdf(file_path, key=key)
    elif file_path.suffix in (".mrc", ".mrcs"):
        data, metadata = read_mrc(file_path)
    elif file_path.suffix in (".tiff", ".tif"):
        data, metadata = read_tiff(file_path)
    else:
        raise ValueError(_UNSUPPORTED.format(file_path))
    if metadata.dtype in (np.uint8, np.int8, np.uint16, np.int16):
        data = data.astype(np.float32) / 255.0
    if data.ndim == 3:
        data = data[np.newaxis, ...]
    return data, found_key


def _match_label_keys_to_data(data: np.ndarray, label_keys: list[str], metadata: FileMetadata) -> dict[str, np.ndarray]:
    """One int8 {-1, 0, 1} volume per label name from a multi-valued label volume; ``label_keys`` in ascending value
    order; -1 stays "unlabelled"; a 0 that is not named is background (reference l.228-254)."""
    values = np.unique(data).tolist()
    nunique = metadata.nunique if metadata.drange[0] >= 0 else metadata.nunique - 1
    if nunique == len(label_keys):
        label_values = sorted(values)
    elif nunique == len(label_keys) + 1 and 0 in values:
        logging.debug("Assuming 0 is the background class in label data and hasn't been specified in label_keys.")
        label_values = sorted(v for v in values if v > 0)
    else:
        raise ValueError(f"Number of unique values in label data ({metadata.nunique}) does not match number of provided label "
                         f"keys ({len(label_keys)}).")
    if len(label_values) != len(label_keys):  # zip(strict=True) of the reference
        raise ValueError("zip() argument 2 is " + ("shorter" if len(label_keys) < len(label_values) else "longer") + " than argument 1")
    labels = {}
    for v, name in zip(label_values, label_keys):
        label = np.where((data != v) & (data != -1), 0, data)
        labels[name] = np.where(label == v, 1, label).astype(np.int8)
    return labels


def load_labels(file_path, label_keys: list[str], key: str | None) -> dict[str, np.ndarray]:
    assert key is None or key in label_keys, f"Label key {key} must be one of the specified label keys {label_keys} or None."
    file_path = Path(file_path)
    if not file_path.exists():
        raise FileNotFoundError(f"File {file_path} does not exist.")
    labels: dict[str, np.ndarray] = {}
    if file_path.suffix in (".h5", ".hdf", ".hdf5"):
        _, data, metadata = read_hdf(file_path, key=key)
        if len(label_keys) > 1:
            labels.update(_match_label_keys_to_data(data, label_keys, metadata))
        else:
            labels[key] = data.astype(np.int8)
    elif file_path.suffix in (".mrc", ".mrcs"):
        data, metadata = read_mrc(file_path)
        labels.update(_match_label_keys_to_data(data, label_keys, metadata))
    elif file_path.suffix in (".tiff", ".tif"):
        data, metadata = read_tiff(file_path)
        labels.update(_match_label_keys_to_data(data, label_keys, metadata))
    else:
        raise ValueError(_UNSUPPORTED.format(file_path))
    return labels


def load_files_from_path(path: Path) -> list[Path]:
    """Tomogram files under a directory (recursive, sorted) or listed one per line in a .txt file (reference l.304-330)."""
    path = Path(path)
    if path.is_dir():
        file_paths = sorted(f for f in path.rglob("*") if f.suffix in tomogram_exts)
    elif path.is_file() and path.suffix == ".txt":
        with open(path) as f:
            file_paths = [Path(line.strip()) for line in f if line.strip()]
    else:
        raise ValueError("Data path must be a directory or a .txt file listing data files.")
    assert len(file_paths) > 0, f"No valid tomogram files found in {path}."
    return file_paths


#### Model container ####

_FORMAT = "cryovit_amd.model.v1"


def _plain(obj: Any) -> Any:
    """Config objects -> containers ``torch.load(weights_only=True)`` accepts (dict / list / str / number / None)."""
    if hasattr(obj, "to_dict"):
        obj = obj.to_dict()
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if isinstance(obj, (str, int, float, bool)) or obj is None:
        return obj
    return str(obj)


def save_model(model_name: str, label_key: str, model: torch.nn.Module, model_cfg, save_path) -> None:
    """Write the ``.model`` container: what ``SavedModel`` holds in the reference (l.336-381), as a plain dict."""
    cfg = _plain(model_cfg)
    model_type = ModelType(str(cfg.get("name", "cryovit")).lower())
    weights = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"format": _FORMAT, "name": model_name, "model_type": model_type.value, "label_key": label_key, "model_cfg": cfg,
                "weights": weights}, save_path)


def save_model_from_weights(model_name: str, label_key: str, model_type: ModelType, weights_path, save_path, **kwargs) -> None:
    """``weights.pt`` (a state_dict) -> ``.model`` container; kwargs override model config keys (``a__b`` -> ``a.b``)."""
    from cryovit_amd.config import compose

    if not Path(weights_path).exists():
        raise FileNotFoundError(f"Weights file {weights_path} does not exist.")
    weights = torch.load(weights_path, map_location="cpu", weights_only=True)
    cfg = compose("infer_model", [f"model={ModelType(model_type).value}", f"label_key={label_key}"]
                  + [f"model.{k.replace('__', '.')}={v}" for k, v in kwargs.items()])
    torch.save({"format": _FORMAT, "name": model_name, "model_type": ModelType(model_type).value, "label_key": label_key,
                "model_cfg": _plain(cfg.model), "weights": {k: v.detach().cpu() for k, v in weights.items()}}, save_path)


def load_model(model_path, load_model: bool = True, device="cuda:0"):
    """``(model | None, model_type, name, label_key)`` (reference l.431-468)."""
    from cryovit_amd.config import Cfg, instantiate

    if not Path(model_path).exists():
        raise FileNotFoundError(f"Model file {model_path} does not exist.")
    try:
        saved = torch.load(model_path, map_location="cpu", weights_only=True)
    except Exception as e:  # noqa: BLE001 -- a reference-made pickle: never unpickled
        raise ValueError(
            f"{model_path} is not a cryovit_amd model container (the reference's .model files are pickles of Python objects and "
            "are not loaded because unpickling executes code). Re-export it from the training weights with "
            "cryovit_amd.utils.save_model_from_weights(name, label_key, ModelType.CRYOVIT, 'weights.pt', 'x.model').") from e
    if not isinstance(saved, dict) or saved.get("format") != _FORMAT:
        raise ValueError(f"{model_path}: unknown model container")
    model = None
    if load_model:
        cfg = Cfg(saved["model_cfg"]) if not isinstance(saved["model_cfg"], Cfg) else saved["model_cfg"]
        target = str(cfg.get("_target_", ""))
        if not target.endswith((".CryoVIT", ".UNet3D")):
            raise NotImplementedError(f"model target {target!r}: the CryoVIT head and the UNet3D baseline run on this build (SURVEY s.8)")
        model = instantiate(cfg, device=device)
        model.load_state_dict(saved["weights"])
    return model, ModelType(saved["model_type"]), saved["name"], saved["label_key"]
