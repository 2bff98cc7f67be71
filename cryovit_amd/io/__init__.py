"""Tomogram file I/O on either side of the hot path (HDF5 layout of SURVEY.md App. C).

Backend: h5py when importable, else the pure-Python subset implementation in ``cryovit_amd.io.hdf5``.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np

try:  # pragma: no cover - h5py is absent in the build image
    import h5py  # type: ignore

    HAVE_H5PY = True
except Exception:  # noqa: BLE001
    h5py = None
    HAVE_H5PY = False

from cryovit_amd.io.hdf5 import H5Dataset, H5Error, H5Group, H5Reader, H5Writer  # noqa: E402,F401


def read_dataset(path, key: str) -> np.ndarray:
    """One dataset as a numpy array (``fh[key][()]``)."""
    if HAVE_H5PY:
        with h5py.File(path, "r") as fh:
            return fh[key][()]
    with H5Reader(path) as fh:
        return fh[key].read()


def read_all_flat(path) -> dict[str, np.ndarray]:
    """Every dataset of the file with one level of groups flattened to leaf names -- what ``_process_sample`` does
    before re-writing a tomogram (/root/reference/src/cryovit/run/dino_features.py:193-200)."""
    out: dict[str, np.ndarray] = {}
    if HAVE_H5PY:
        with h5py.File(path, "r") as fh:
            for key in fh:
                if isinstance(fh[key], h5py.Group):
                    for sub in fh[key]:
                        out[sub] = fh[key][sub][()]
                else:
                    out[key] = fh[key][()]
        return out
    with H5Reader(path) as fh:
        for key in fh.keys():
            obj = fh[key]
            if isinstance(obj, H5Group):
                for sub in obj.keys():
                    leaf = obj[sub]
                    if isinstance(leaf, H5Dataset):
                        out[sub] = leaf.read()
            else:
                out[key] = obj.read()
    return out


def list_keys(path, group: str = "/") -> list[str]:
    if HAVE_H5PY:
        with h5py.File(path, "r") as fh:
            return list(fh[group].keys())
    with H5Reader(path) as fh:
        g = fh if group in ("/", "") else fh[group]
        return g.keys()


class FileWriter:
    """``with FileWriter(path) as f: f.create_dataset("labels/mito", arr, compression="gzip")`` on either backend."""

    def __init__(self, path):
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        self._w = h5py.File(path, "w") if HAVE_H5PY else H5Writer(path)

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if HAVE_H5PY:
            self._w.close()
        elif et is None:
            self._w.close()

    def create_dataset(self, name: str, data: np.ndarray, compression=None) -> None:
        if HAVE_H5PY:
            self._w.create_dataset(name, data=data, shape=data.shape, dtype=data.dtype, compression=compression)
        else:
            self._w.create_dataset(name, data, compression=compression)
