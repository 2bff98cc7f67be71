"""PCIe- and file-inclusive rates of the feature stage (never the headline `value`):
  (1) _dino_features host volume -> host fp16 features for one 128x512x512 tomogram (H2D + ViT-g + D2H),
  (2) the whole entry point over N synthetic HDF5 tomograms (read + gunzip, GPU, gzip + write; 3-stage thread pipeline)."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd import io  # noqa: E402
from cryovit_amd.models import load_encoder  # noqa: E402
from cryovit_amd.run.dino_features import _dino_features  # noqa: E402
from cryovit_amd.training import dino_features as entry  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D, H, W = 128, 512, 512
enc = load_encoder("dinov2_vitg14_reg", synthetic_seed=2)
vol = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(D, H, W), dtype=np.uint8))
_dino_features(vol, enc, 128)
t0 = time.perf_counter()
for _ in range(3):
    f = _dino_features(vol, enc, 128)
dt = (time.perf_counter() - t0) / 3
print(f"(1) host->host features: {dt * 1e3:.1f} ms / tomogram = {D * H * W / dt / 1e6:.1f} Mvoxel/s (ViT only, PCIe inclusive)")
del enc
torch.cuda.empty_cache()
with tempfile.TemporaryDirectory() as tmp:
    tmp = Path(tmp)
    src = tmp / "processed" / "Q109"
    src.mkdir(parents=True)
    rng = np.random.default_rng(1)
    base = (rng.random((D, H, W)) * 255).astype(np.uint8)
    for i in range(N):
        with io.FileWriter(src / f"t{i}.hdf") as fh:
            fh.create_dataset("data", np.roll(base, i, axis=0), compression="gzip")
            fh.create_dataset("labels/mito", (np.roll(base, i, axis=0) > 128).astype(np.int8), compression="gzip")
    # (3) the host stages in isolation, one tomogram each (what the 3-stage pipeline has to hide behind the GPU)
    from cryovit_amd.run.dino_features import _save_data

    t0 = time.perf_counter()
    flat = io.read_all_flat(src / "t0.hdf")
    t_read = time.perf_counter() - t0
    feats = np.zeros((1536, D, 32, 32), dtype=np.float16)
    t0 = time.perf_counter()
    _save_data(flat, feats, "probe.hdf", tmp / "probe")
    t_save = time.perf_counter() - t0
    t0 = time.perf_counter()
    pinned = torch.empty(feats.shape, dtype=torch.float16, pin_memory=True)
    t_pin = time.perf_counter() - t0
    print(f"(3) stages alone: read+gunzip all leaves {t_read * 1e3:.0f} ms, save (gzip data+labels, write 403 MB features) {t_save * 1e3:.0f} ms, "
          f"pinned 403 MB allocation {t_pin * 1e3:.0f} ms; host cpus {__import__('os').cpu_count()}")
    import logging

    stamps = []

    class _Stamp(logging.Handler):  # completion time of every tomogram (the runner logs one line per written file)
        def emit(self, record):
            if "-> dino_features" in record.getMessage():
                stamps.append(time.perf_counter())

    logging.getLogger().addHandler(_Stamp())
    t0 = time.perf_counter()
    entry.main([f"paths.model_dir={tmp}", f"paths.data_dir={tmp}", f"paths.exp_dir={tmp / 'exp'}", "paths.feature_name=processed",
                "sample=Q109", "batch_size=128", "encoder.synthetic_seed=2"])
    dt = time.perf_counter() - t0
    outs = sorted((tmp / "tomograms" / "Q109").glob("*.hdf"))
    gaps = sorted(b - a for a, b in zip(stamps[1:-1], stamps[2:]))  # steady state: without the first two files (start-up, pipeline fill)
    steady = gaps[len(gaps) // 2] if gaps else float("nan")
    print(f"(2) entry point: {len(outs)} tomograms in {dt:.1f} s incl. start-up (weight generation, first workspace allocation); "
          f"STEADY STATE {steady * 1e3:.0f} ms / tomogram (median interval between written files, {len(gaps)} intervals) = "
          f"{D * H * W / steady / 1e6:.1f} Mvoxel/s  (HDF5 backend: {'h5py' if io.HAVE_H5PY else 'pure-Python'})")
