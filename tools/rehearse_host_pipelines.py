"""CPU-only rehearsal of the HOST side of ``cryovit_amd.training.dino_features`` under N concurrent ranks on one machine
(VERDICT r02 item 2): every rank runs the runner's own three-stage pipeline -- reader thread (HDF5 read + gunzip), "GPU" stage
(a sleep of --gpu-ms: the device time of one tomogram), two writer threads (``_save_data``: gzip ``data`` + labels, 403 MB of
uncompressed fp16 ``dino_features`` through the HDF5 backend) -- with synthetic feature arrays.  No GPU is touched.  Prints the
sustained tomograms/s of all ranks together: the ceiling the host puts on BASELINE configs[3] through the drop-in entry point,
whatever ``bench.py`` (volumes resident in HBM) prints.

    python tools/rehearse_host_pipelines.py --ranks 8 --tomograms 6 --gpu-ms 290 [--dir /tmp/x] [--keep]
"""
import argparse
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
D, H, W, C = 128, 512, 512, 1536


def rank_main(rank, args, root, q):
    from cryovit_amd import io
    from cryovit_amd.run.dino_features import _save_data

    src, dst = Path(root) / f"src{rank}", Path(root) / f"dst{rank}"
    feats = np.random.default_rng(rank).standard_normal((C, D, 32, 32), dtype=np.float32).astype(np.float16)
    names = [f"t{i}.hdf" for i in range(args.tomograms)]
    q.put(("ready", rank))
    while q.empty():  # crude start barrier: the parent refills the queue with "go" tokens
        time.sleep(0.01)
    stamps = []

    def save(i, flat):
        _save_data(flat, feats, names[i], dst)
        if not args.keep:
            (dst / names[i]).unlink()
        stamps.append(time.perf_counter())

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=1) as reader, ThreadPoolExecutor(max_workers=2) as writer:
        nxt = reader.submit(io.read_all_flat, src / names[0])
        pending = []
        for i in range(args.tomograms):
            flat = nxt.result()
            nxt = reader.submit(io.read_all_flat, src / names[i + 1]) if i + 1 < args.tomograms else None
            time.sleep(args.gpu_ms * 1e-3)  # the device stage
            pending.append(writer.submit(save, i, flat))
            while len(pending) > 2:
                pending.pop(0).result()
        for f in pending:
            f.result()
    q.put(("done", rank, time.perf_counter() - t0, sorted(stamps)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--tomograms", type=int, default=6)
    ap.add_argument("--gpu-ms", type=float, default=290.0)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    from cryovit_amd import io

    root = Path(tempfile.mkdtemp(dir=args.dir))
    try:
        rng = np.random.default_rng(1)
        base = (rng.random((D, H, W)) * 255).astype(np.uint8)
        for r in range(args.ranks):
            (root / f"src{r}").mkdir()
            for i in range(args.tomograms):
                with io.FileWriter(root / f"src{r}" / f"t{i}.hdf") as fh:
                    fh.create_dataset("data", np.roll(base, i + r, axis=0), compression="gzip")
                    fh.create_dataset("labels/mito", (np.roll(base, i + r, axis=0) > 128).astype(np.int8), compression="gzip")
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=rank_main, args=(r, args, str(root), q)) for r in range(args.ranks)]
        for p in procs:
            p.start()
        for _ in procs:
            assert q.get()[0] == "ready"
        q.put(("go",))
        res = []
        t0 = time.perf_counter()
        while len(res) < args.ranks:
            m = q.get()
            if m[0] == "done":
                res.append(m)
            elif m[0] == "go":
                q.put(m)
                time.sleep(0.005)
        wall = time.perf_counter() - t0
        for p in procs:
            p.join()
        per_rank = [m[2] / args.tomograms for m in res]
        # steady state: interval between written files per rank, without the first two (pipeline fill)
        gaps = sorted(b - a for m in res for a, b in zip(m[3][1:-1], m[3][2:]))
        steady = gaps[len(gaps) // 2] if gaps else float("nan")
        total = args.ranks * args.tomograms
        print(f"{args.ranks} ranks x {args.tomograms} tomograms, device stage {args.gpu_ms:.0f} ms, host cpus {os.cpu_count()}, dir {root.parent}: "
              f"wall {wall:.2f} s = {total / wall:.2f} tomograms/s over all ranks ({wall / args.tomograms * 1e3:.0f} ms per tomogram and rank incl. fill); "
              f"steady-state interval per rank median {steady * 1e3:.0f} ms, worst {gaps[-1] * 1e3 if gaps else float('nan'):.0f} ms; "
              f"device-bound would be {args.gpu_ms:.0f} ms -> host-side efficiency {args.gpu_ms * 1e-3 / steady:.2f}; "
              f"feature bytes written {total * C * D * 32 * 32 * 2 / 1e9:.1f} GB = {total * C * D * 32 * 32 * 2 / 1e9 / wall:.2f} GB/s")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
