"""CPU suite, part 3: the pure-Python HDF5 subset (cryovit_amd/io/hdf5.py) round-trips, reads files written by the real
libhdf5 and writes files the real libhdf5 reads (library found under /opt/conda in this image; skipped if absent)."""

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from cryovit_amd.io.hdf5 import H5Dataset, H5Group, H5Reader, H5Writer

LIBHDF5 = next((p for p in ("/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so") if Path(p).exists()), None)
H5DUMP = "/opt/conda/bin/h5dump" if Path("/opt/conda/bin/h5dump").exists() else None


def sample_arrays():
    rng = np.random.default_rng(0)
    return {
        "data": rng.integers(0, 256, size=(70, 24, 40), dtype=np.uint8),  # 70 slices -> 35 chunks of 2
        "labels/mito": rng.integers(-1, 2, size=(70, 24, 40)).astype(np.int8),
        "labels/granule": rng.integers(-1, 2, size=(70, 24, 40)).astype(np.int8),
        "dino_features": rng.standard_normal((16, 70, 2, 3)).astype(np.float16),
        "aux/f32": rng.standard_normal((5, 7)).astype(np.float32),
        "aux/f64": rng.standard_normal((3,)),
        "aux/i32": rng.integers(-1000, 1000, size=(4, 4)).astype(np.int32),
    }


def write_sample(path, arrays):
    with H5Writer(path) as f:
        for k, v in arrays.items():
            f.create_dataset(k, v, compression="gzip" if k in ("data", "labels/mito", "labels/granule") else None)


def test_roundtrip(tmp_path):
    arrays = sample_arrays()
    p = tmp_path / "t.hdf"
    write_sample(p, arrays)
    with H5Reader(p) as f:
        assert sorted(f.keys()) == ["aux", "data", "dino_features", "labels"]
        assert isinstance(f["labels"], H5Group) and sorted(f["labels"].keys()) == ["granule", "mito"]
        for k, v in arrays.items():
            d = f[k]
            assert isinstance(d, H5Dataset) and d.shape == v.shape and d.dtype == v.dtype
            assert np.array_equal(d.read(), v)
        assert f["data"]._layout[0] == "chunked" and f["dino_features"]._layout[0] == "contiguous"


@pytest.mark.skipif(H5DUMP is None, reason="h5dump not available")
def test_h5dump_reads_our_file(tmp_path):
    arrays = sample_arrays()
    p = tmp_path / "t.hdf"
    write_sample(p, arrays)
    r = subprocess.run([H5DUMP, "-H", "-p", str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert 'DATASET "dino_features"' in out and "16-bit little-endian floating-point" in out and "CONTIGUOUS" in out
    assert 'GROUP "labels"' in out and 'DATASET "mito"' in out and "H5T_STD_I8LE" in out
    assert "COMPRESSION DEFLATE { LEVEL 4 }" in out and "CHUNKED" in out
    r = subprocess.run([H5DUMP, "-d", "/aux/i32", str(p)], capture_output=True, text=True)
    assert r.returncode == 0 and str(int(arrays["aux/i32"][0, 0])) in r.stdout


@pytest.fixture(scope="module")
def hdf5lib():
    if LIBHDF5 is None:
        pytest.skip("libhdf5 not available")
    lib = C.CDLL(LIBHDF5)
    lib.H5open()
    for name in ("H5Fopen", "H5Fcreate", "H5Dopen2", "H5Dcreate2", "H5Screate_simple", "H5Pcreate", "H5Gcreate2", "H5Dget_type"):
        getattr(lib, name).restype = C.c_int64
    return lib


def _h5t(lib, name):
    return C.c_int64.in_dll(lib, name + "_g").value


def test_libhdf5_reads_our_file(tmp_path, hdf5lib):
    lib = hdf5lib
    arrays = sample_arrays()
    p = tmp_path / "ours.hdf"
    write_sample(p, arrays)
    fid = lib.H5Fopen(str(p).encode(), C.c_uint(0), C.c_int64(0))
    assert fid >= 0
    native = {np.dtype("u1"): "H5T_NATIVE_UCHAR", np.dtype("i1"): "H5T_NATIVE_SCHAR", np.dtype("f4"): "H5T_NATIVE_FLOAT",
              np.dtype("f8"): "H5T_NATIVE_DOUBLE", np.dtype("i4"): "H5T_NATIVE_INT"}
    for k, v in arrays.items():
        did = lib.H5Dopen2(C.c_int64(fid), ("/" + k).encode(), C.c_int64(0))
        assert did >= 0, k
        if v.dtype == np.float16:  # no native half in C: read as float (libhdf5 converts)
            out = np.empty(v.shape, np.float32)
            mem = _h5t(lib, "H5T_NATIVE_FLOAT")
        else:
            out = np.empty(v.shape, v.dtype)
            mem = _h5t(lib, native[v.dtype])
        rc = lib.H5Dread(C.c_int64(did), C.c_int64(mem), C.c_int64(0), C.c_int64(0), C.c_int64(0), out.ctypes.data_as(C.c_void_p))
        assert rc >= 0, k
        assert np.array_equal(out.astype(v.dtype), v), k
        lib.H5Dclose(C.c_int64(did))
    lib.H5Fclose(C.c_int64(fid))


def test_we_read_libhdf5_file(tmp_path, hdf5lib):
    """A file written by the real library the way h5py's create_dataset(compression="gzip") does (chunked + deflate,
    groups, contiguous) -- chunk shape chosen by the caller, several B-tree entries."""
    lib = hdf5lib
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, size=(40, 33, 29), dtype=np.uint8)
    lab = rng.integers(-1, 2, size=(40, 33, 29)).astype(np.int8)
    feats = rng.standard_normal((8, 40, 3, 2)).astype(np.float32)
    p = tmp_path / "lib.hdf"
    H5F_ACC_TRUNC, H5P_DEFAULT = 2, 0
    fid = lib.H5Fcreate(str(p).encode(), C.c_uint(H5F_ACC_TRUNC), C.c_int64(H5P_DEFAULT), C.c_int64(H5P_DEFAULT))
    assert fid >= 0
    gid = lib.H5Gcreate2(C.c_int64(fid), b"labels", C.c_int64(0), C.c_int64(0), C.c_int64(0))

    def put(loc, name, arr, h5type, chunk=None):
        dims = (C.c_uint64 * arr.ndim)(*arr.shape)
        sid = lib.H5Screate_simple(C.c_int(arr.ndim), dims, None)
        dcpl = C.c_int64(0)
        if chunk:
            dcpl = C.c_int64(lib.H5Pcreate(C.c_int64(_h5t(lib, "H5P_CLS_DATASET_CREATE_ID"))))
            assert lib.H5Pset_chunk(dcpl, C.c_int(arr.ndim), (C.c_uint64 * arr.ndim)(*chunk)) >= 0
            assert lib.H5Pset_shuffle(dcpl) >= 0
            assert lib.H5Pset_deflate(dcpl, C.c_uint(6)) >= 0
        did = lib.H5Dcreate2(C.c_int64(loc), name, C.c_int64(_h5t(lib, h5type)), C.c_int64(sid), C.c_int64(0), dcpl, C.c_int64(0))
        assert did >= 0
        assert lib.H5Dwrite(C.c_int64(did), C.c_int64(_h5t(lib, h5type)), C.c_int64(0), C.c_int64(0), C.c_int64(0),
                            arr.ctypes.data_as(C.c_void_p)) >= 0
        lib.H5Dclose(C.c_int64(did))

    put(fid, b"data", data, "H5T_NATIVE_UCHAR", chunk=(5, 9, 8))  # 8*4*4 = 128 chunks -> multi-level chunk B-tree
    put(gid, b"mito", lab, "H5T_NATIVE_SCHAR", chunk=(40, 33, 29))
    put(fid, b"feats", feats, "H5T_NATIVE_FLOAT")
    lib.H5Gclose(C.c_int64(gid))
    lib.H5Fclose(C.c_int64(fid))
    with H5Reader(p) as f:
        assert sorted(f.keys()) == ["data", "feats", "labels"]
        assert np.array_equal(f["data"].read(), data)
        assert np.array_equal(f["labels/mito"].read(), lab)
        assert np.array_equal(f["feats"].read(), feats)


def test_read_scipy_matlab73_file():
    """A genuine HDF5 file with a 512-byte user block shipped with scipy (MATLAB v7.3)."""
    import scipy.io

    p = Path(scipy.io.__file__).parent / "matlab" / "tests" / "data" / "testhdf5_7.4_GLNX86.mat"
    if not p.exists():
        pytest.skip("scipy test data not installed")
    with H5Reader(p) as f:
        assert f.base == 512
        keys = f.keys()
        assert len(keys) > 0
        for k in keys:
            obj = f[k]
            if isinstance(obj, H5Dataset) and obj.dtype.kind in "uif":
                arr = obj.read()
                assert arr.shape == obj.shape
