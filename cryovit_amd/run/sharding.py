"""Tomogram-level sharding across the GPUs of a node (SURVEY s.8e): tomograms are independent units, so every rank
(one process per GPU) takes a static share of the record list and there is NO data-path collective; only the small
per-tomogram result rows (names, Dice) are gathered on rank 0 for the CSV.

The reference has no distributed code (``devices: "1"``, one sbatch job per sample); this replaces that job-level
parallelism."""

from __future__ import annotations

import os


def world_info() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def select_device(requested: str | None = None) -> str:
    """The HIP device of this rank, made the ACTIVE device of the process before anything is allocated.

    Under ``torch.distributed.run`` (WORLD_SIZE > 1) that is ``cuda:LOCAL_RANK`` whatever the config says; launched plainly it
    is ``requested`` (default ``cuda:0``).  Every later allocation, ``torch.cuda.current_stream()`` and every call into
    ``libcryovit_hip.so`` (``hipFuncSetAttribute``, launches) then refers to this GPU and not to GPU 0."""
    import torch

    _, local_rank, world = world_info()
    device = f"cuda:{local_rank}" if world > 1 else (requested or "cuda:0")
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError(f"cryovit_amd runs on HIP devices only, got {device!r}")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: cryovit_amd has no CPU path")
    index = d.index if d.index is not None else 0
    if index >= torch.cuda.device_count():
        raise RuntimeError(f"rank wants cuda:{index} but only {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(index)
    return f"cuda:{index}"


def shard_records(records: list, rank: int, world: int, weights: list[float] | None = None) -> list[int]:
    """Indices of the records this rank owns.  With ``weights`` (e.g. D*H*W voxels) a size-sorted greedy assignment
    balances the load; otherwise round-robin.  Deterministic and identical on every rank."""
    n = len(records)
    if world <= 1:
        return list(range(n))
    if weights is None:
        return list(range(rank, n, world))
    order = sorted(range(n), key=lambda i: (-weights[i], i))
    load = [0.0] * world
    owner = [0] * n
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += weights[i]
    return [i for i in range(n) if owner[i] == rank]


def gather_rows(rows: list, world: int) -> list:
    """All ranks' result rows on every rank, ordered by rank (object all-gather: a few hundred bytes per tomogram)."""
    if world <= 1:
        return list(rows)
    import torch.distributed as dist

    if not dist.is_initialized():  # result rows are host objects: gloo is enough, the data path has no collective
        dist.init_process_group("gloo")
    out: list = [None] * world
    dist.all_gather_object(out, rows)
    return [r for part in out for r in part]
