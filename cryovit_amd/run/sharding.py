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
