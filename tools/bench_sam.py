"""BASELINE configs[4]: the sam_features path -- SAM2.1 Hiera-L image encoder + FPN neck over the slices of one synthetic
128x512x512 tomogram on one MI355X (seeded random weights; inputs resident in HBM).  Prints one JSON line per run:
tomogram voxels/s of the encoder alone, ms per tomogram, algorithmic TFLOP/s (2*MACs of linear + attention products)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slice-batch", type=int, default=64)
    ap.add_argument("--depth", type=int, default=128)
    ap.add_argument("--no-fold", action="store_true", help="round 2's plan: fp32 stream + LayerNorm kernels in every stage")
    ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE passed to cvx_set_option (A/B runs)")
    a = ap.parse_args()
    from cryovit_amd import _lib
    from cryovit_amd.models import load_sam_encoder

    for o in a.opt:
        k, v = o.split("=")
        _lib.set_option(k, int(v))

    dev = torch.device("cuda:0")
    enc = load_sam_encoder("SAM2", synthetic_seed=2, device=dev, slice_batch=a.slice_batch, fold_ln=not a.no_fold)
    vol = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (a.depth, 512, 512), dtype=np.uint8)).to(dev)
    outs = enc._outs(a.depth)

    def step():
        for d0 in range(0, a.depth, a.slice_batch):
            enc.engine.encode(vol[d0 : d0 + a.slice_batch], outs, d0)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fl = enc.engine.flops(a.depth)
    print(json.dumps({"metric": "tomogram voxels/sec (sam_features: SAM2.1 Hiera-L image encoder + FPN neck)", "value": vol.numel() / dt,
                      "unit": "voxels/s", "ms_per_tomogram": dt * 1e3, "tflops": fl / dt / 1e12, "algorithmic_tflop": fl / 1e12,
                      "dtype": "bf16", "data": "synthetic", "config": {"workload": f"{a.depth}x512x512 uint8 tomogram, Hiera-L, slice batch {a.slice_batch}"}}))


if __name__ == "__main__":
    main()
