"""File formats and loaders of the script-level flows (``cryovit_amd/utils.py``, SURVEY s.8f N2) against the behaviour of
``/root/reference/src/cryovit/utils.py`` (cited per test).  MRC files are assembled by hand from the MRC2014 header layout,
TIFF files are written by Pillow -- both independent of the readers under test."""

import logging
import pickle
import struct

import numpy as np
import pytest
import torch

from cryovit_amd import io
from cryovit_amd import utils as U
from cryovit_amd.types import FileData, ModelType


def _mrc_bytes(arr: np.ndarray, mode: int, big: bool = False, ext: bytes = b"") -> bytes:
    bo = ">" if big else "<"
    nz, ny, nx = arr.shape if arr.ndim == 3 else (1, *arr.shape)
    h = bytearray(1024)
    h[0:16] = struct.pack(bo + "4i", nx, ny, nz, mode)
    h[28:40] = struct.pack(bo + "3i", nx, ny, nz)
    h[64:76] = struct.pack(bo + "3i", 1, 2, 3)
    h[92:96] = struct.pack(bo + "i", len(ext))
    h[208:212] = b"MAP "
    h[212:216] = bytes([0x11, 0x11, 0, 0]) if big else bytes([0x44, 0x44, 0, 0])
    return bytes(h) + ext + arr.astype(arr.dtype.newbyteorder(bo)).tobytes()


@pytest.mark.parametrize("mode,dtype", [(0, np.int8), (1, np.int16), (2, np.float32), (6, np.uint16), (12, np.float16)])
@pytest.mark.parametrize("big", [False, True])
def test_read_mrc_modes(tmp_path, mode, dtype, big):
    rng = np.random.default_rng(mode)
    arr = (rng.standard_normal((3, 5, 7)) * 50).astype(dtype)
    p = tmp_path / "t.mrc"
    p.write_bytes(_mrc_bytes(arr, mode, big, ext=b"x" * 160))  # extended header must be skipped
    data, meta = U.read_mrc(p)
    assert data.dtype == np.dtype(dtype) and np.array_equal(data, arr)
    assert meta.dshape == (3, 5, 7) and meta.nunique == len(np.unique(arr))
    assert meta.drange == (float(arr.min()), float(arr.max()))


def test_read_mrc_errors(tmp_path):
    p = tmp_path / "bad.mrc"
    p.write_bytes(b"\0" * 100)
    with pytest.raises(ValueError):
        U.read_mrc(p)
    p.write_bytes(_mrc_bytes(np.zeros((2, 2, 2), np.float32), 2)[:-4])
    with pytest.raises(ValueError, match="truncated"):
        U.read_mrc(p)
    p.write_bytes(_mrc_bytes(np.zeros((2, 2, 2), np.float32), 4))
    with pytest.raises(ValueError, match="mode"):
        U.read_mrc(p)
    single = tmp_path / "one.mrc"
    single.write_bytes(_mrc_bytes(np.arange(12, dtype=np.float32).reshape(3, 4), 2))
    assert U.read_mrc(single)[0].shape == (3, 4)  # mrcfile: a single image is 2-D


@pytest.mark.parametrize("dtype,mode", [(np.uint8, "L"), (np.uint16, "I;16"), (np.float32, "F")])
@pytest.mark.parametrize("compression", [None, "tiff_adobe_deflate", "packbits"])
def test_read_tiff_pages(tmp_path, dtype, mode, compression):
    from PIL import Image

    rng = np.random.default_rng(3)
    vol = (rng.random((4, 9, 13)) * 200).astype(dtype)
    vol[:, 2:5] = vol[0, 0, 0]  # runs, so PackBits emits both literal and repeat packets
    p = tmp_path / "v.tif"
    pages = [Image.fromarray(s) for s in vol]
    assert pages[0].mode == mode
    kw = {"compression": compression} if compression else {}
    pages[0].save(p, save_all=True, append_images=pages[1:], **kw)
    data, meta = U.read_tiff(p)
    assert data.shape == vol.shape and data.dtype == np.dtype(dtype) and np.array_equal(data, vol)
    pages[0].save(p, **kw)
    assert U.read_tiff(p)[0].shape == (9, 13)


def test_read_tiff_rejects_other_files(tmp_path):
    p = tmp_path / "x.tif"
    p.write_bytes(b"not a tiff at all")
    with pytest.raises(ValueError):
        U.read_tiff(p)


def _hdf(path, **datasets):
    with io.FileWriter(path) as f:
        for k, v in datasets.items():
            f.create_dataset(k.replace("__", "/"), v)


def test_read_hdf_key_selection(tmp_path, caplog):
    """utils.py:115-143: explicit key; no key -> most unique values (nested groups flattened with '/'); a key that is
    absent warns, reads everything and raises KeyError."""
    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, (4, 8, 8), dtype=np.uint8)
    lab = rng.integers(-1, 2, (4, 8, 8)).astype(np.int8)
    p = tmp_path / "a.hdf"
    _hdf(p, data=data, labels__mito=lab)
    k, d, m = U.read_hdf(p, key="labels/mito")
    assert k == "labels/mito" and np.array_equal(d, lab) and m.nunique == 3 and m.drange == (-1.0, 1.0)
    k, d, m = U.read_hdf(p)
    assert k == "data" and np.array_equal(d, data)
    with caplog.at_level(logging.WARNING), pytest.raises(KeyError):
        U.read_hdf(p, key="dino_features")
    assert "not found" in caplog.text


def test_load_data_normalisation_and_channel_axis(tmp_path):
    """utils.py:216-224: 8/16-bit integers -> float32 / 255; floats as stored; 3-D gets a channel axis, 4-D does not."""
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, (3, 4, 5), dtype=np.uint8)
    f16 = rng.standard_normal((6, 3, 2, 2)).astype(np.float16)
    p = tmp_path / "a.hdf"
    _hdf(p, data=u8, dino_features=f16)
    d, k = U.load_data(p, key="data")
    assert k == "data" and d.dtype == np.float32 and d.shape == (1, 3, 4, 5) and np.array_equal(d[0], u8.astype(np.float32) / 255.0)
    d, k = U.load_data(p, key="dino_features")
    assert d.dtype == np.float16 and d.shape == f16.shape and np.array_equal(d, f16)
    i16 = (rng.standard_normal((2, 4, 4)) * 300).astype(np.int16)
    m = tmp_path / "b.mrc"
    m.write_bytes(_mrc_bytes(i16, 1))
    d, k = U.load_data(m)
    assert k == "" and np.array_equal(d[0], i16.astype(np.float32) / 255.0)
    with pytest.raises(FileNotFoundError):
        U.load_data(tmp_path / "nope.hdf")
    (tmp_path / "c.png").write_bytes(b"x")
    with pytest.raises(ValueError, match="Unsupported file format"):
        U.load_data(tmp_path / "c.png")


def test_match_label_keys(tmp_path):
    """utils.py:228-254: values map to names in ascending order, -1 stays -1, unnamed 0 is background, count mismatch raises."""
    lab = np.array([[-1, 0, 1, 2], [2, 2, 0, 1]], dtype=np.int8)[None]
    meta = U._metadata(lab)
    out = U._match_label_keys_to_data(lab, ["mito", "cristae"], meta)
    assert np.array_equal(out["mito"][0], [[-1, 0, 1, 0], [0, 0, 0, 1]]) and out["mito"].dtype == np.int8
    assert np.array_equal(out["cristae"][0], [[-1, 0, 0, 1], [1, 1, 0, 0]])
    with pytest.raises(ValueError):  # 0 named explicitly while -1 is present: the reference's zip(strict=True) sees 4 values
        U._match_label_keys_to_data(lab, ["bg", "mito", "cristae"], meta)
    pos = np.abs(lab)
    out = U._match_label_keys_to_data(pos, ["bg", "mito", "cristae"], U._metadata(pos))
    # reference formula (l.239-241) for value 0: where(data != 0, 0, data) is all zero, then where(label == 0, 1, .) -> all ones
    assert np.all(out["bg"] == 1) and np.array_equal(out["mito"][0], [[1, 0, 1, 0], [0, 0, 0, 1]])
    with pytest.raises(ValueError, match="does not match"):
        U._match_label_keys_to_data(lab, ["only"], meta)
    p = tmp_path / "l.hdf"
    _hdf(p, mito=lab)
    assert np.array_equal(U.load_labels(p, ["mito"], "mito")["mito"], lab)
    with pytest.raises(AssertionError):
        U.load_labels(p, ["mito"], "er")


def test_load_files_from_path(tmp_path):
    (tmp_path / "sub").mkdir()
    for n in ("b.hdf", "a.mrc", "sub/c.hdf", "notes.txt", "d.tif"):
        (tmp_path / n).write_bytes(b"")
    got = U.load_files_from_path(tmp_path)
    assert [p.relative_to(tmp_path).as_posix() for p in got] == ["a.mrc", "b.hdf", "sub/c.hdf"]  # config.py:15 tomogram_exts
    lst = tmp_path / "list.txt"
    lst.write_text(f"{tmp_path / 'b.hdf'}\n\n  {tmp_path / 'a.mrc'}  \n")
    assert U.load_files_from_path(lst) == [tmp_path / "b.hdf", tmp_path / "a.mrc"]
    with pytest.raises(ValueError):
        U.load_files_from_path(tmp_path / "b.hdf")
    (tmp_path / "empty").mkdir()
    with pytest.raises(AssertionError):
        U.load_files_from_path(tmp_path / "empty")


def test_model_container_roundtrip_and_pickle_refusal(tmp_path):
    """save_model_from_weights / load_model (utils.py:384-468) on the safe container; a pickled .model is refused unopened."""
    from oracle import head as oh

    ref = oh.CryoVITHead()
    oh.rescaled_init_(ref, seed=11)
    torch.save(ref.state_dict(), tmp_path / "weights.pt")
    U.save_model_from_weights("my_model", "mito", ModelType.CRYOVIT, tmp_path / "weights.pt", tmp_path / "m.model", lr=5e-4)
    model, mtype, name, label_key = U.load_model(tmp_path / "m.model", load_model=False)
    assert model is None and mtype is ModelType.CRYOVIT and name == "my_model" and label_key == "mito"
    saved = torch.load(tmp_path / "m.model", weights_only=True)
    assert saved["model_cfg"]["_target_"] == "cryovit_amd.models.CryoVIT" and saved["model_cfg"]["lr"] == 5e-4
    assert all(torch.equal(saved["weights"][k], v) for k, v in ref.state_dict().items())

    class Evil:
        def __reduce__(self):
            return (pytest.fail, ("a reference-style pickle was executed",))

    (tmp_path / "ref.model").write_bytes(pickle.dumps(Evil()))
    with pytest.raises(ValueError, match="not loaded because unpickling executes code"):
        U.load_model(tmp_path / "ref.model")
    with pytest.raises(FileNotFoundError):
        U.load_model(tmp_path / "absent.model")


def test_file_dataset_items(tmp_path):
    """file_dataset.py:62-157: predict item (input key, zero label [1,D,H,W], raw data in aux) and for_dino item."""
    from cryovit_amd.config import compose, instantiate
    from cryovit_amd.datasets import FileDataset, collate_fn

    rng = np.random.default_rng(2)
    u8 = rng.integers(0, 256, (3, 32, 32), dtype=np.uint8)
    f16 = rng.standard_normal((8, 3, 2, 2)).astype(np.float16)
    p = tmp_path / "t.hdf"
    _hdf(p, data=u8, dino_features=f16)
    cfg = compose("infer_model", ["model=cryovit", "label_key=mito", "datamodule=file"])
    ds = instantiate(cfg.datamodule.dataset)([FileData(tomo_path=p)], train=False)
    assert isinstance(ds, FileDataset) and ds.input_key == "dino_features" and ds.label_key == "mito" and len(ds) == 1
    item = ds[0]
    assert item.tomo_name == "t.hdf" and item.sample is None
    assert item.data.dtype == torch.float16 and tuple(item.data.shape) == (8, 3, 2, 2)
    assert tuple(item.label.shape) == (1, 3, 2, 2) and item.label.dtype == torch.int8 and not item.label.any()
    assert np.array_equal(item.aux_data["data"], u8.astype(np.float32) / 255.0)
    batch = collate_fn([item])
    assert tuple(batch.tomo_batch.shape) == (1, 3, 8, 2, 2) and batch.tomo_batch.dtype == torch.float32
    with pytest.raises(IndexError):
        ds[1]
    dino = FileDataset([FileData(tomo_path=p)], input_key=None, label_key=None, for_dino=True)[0]
    assert tuple(dino.data.shape) == (3, 32, 32) and dino.data.dtype == torch.float32
    assert np.array_equal(dino.aux_data["data"], u8.astype(np.float32) / 255.0)


def test_compose_group_choices():
    from cryovit_amd.config import compose, missing_keys

    cfg = compose("infer_model", ["label_key=mito"])
    assert "model" in missing_keys(cfg)  # `model: ???` in the defaults list stays mandatory
    cfg = compose("dino_features", ["datamodule/dataset=file", "sample=null"])
    assert cfg.datamodule.dataset._target_.endswith("FileDataset")
    cfg = compose("infer_model", ["label_key=mito", "model=cryovit", "name=abc"])
    assert cfg.name == "abc" and cfg.model._target_ == "cryovit_amd.models.CryoVIT" and cfg.datamodule.dataset.label_key == "mito"
