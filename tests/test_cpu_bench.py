"""bench.py's self-launching N-rank path, rehearsed on CPU (gloo, world_size 2): `python bench.py --gpus N` must need no
wrapper -- the parent starts the ranks itself, relays rank 0's single JSON line and fails when a rank fails."""

import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(*argv, timeout=300):
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_self_launch_two_ranks_dry_run():
    r = _run("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak" and line["unit"] == "voxels/s"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    # configs[3] sharding: rank r takes seeds 100 + r*steps + i -- disjoint, contiguous, starting at 100
    assert line["config"]["seeds_per_rank"] == [[100, 101, 102, 103], [104, 105, 106, 107]]


def test_eight_ranks_cover_the_32_tomograms():
    r = _run("--gpus", "8", "--steps", "4", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    seeds = [s for part in line["config"]["seeds_per_rank"] for s in part]
    assert sorted(seeds) == list(range(100, 132)) and line["n_gpus"] == 8


def test_single_process_dry_run_and_world_mismatch():
    r = _run("--gpus", "1", "--steps", "2", "--dry-run")
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # launched as a rank of a 2-process job but asked for 4 GPUs: refuse instead of reporting a wrong n_gpus
    import os

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, env=env,
                       timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_failing_rank_fails_the_run():
    """Without --dry-run and without a GPU every rank exits with an error: the launcher must return non-zero, print no JSON."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a GPU-less host")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_board_sampler_without_a_device():
    """The clock / power sampler of the bench line is optional evidence: without a card (or without readable sysfs files) every
    field is None and nothing raises."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b = bench.BoardSampler(0)
    b.start()
    st = b.stop()
    assert st["sclk_mhz_median"] is None and st["power_w_median"] is None and st["samples"] == 0 and st["energy_j"] is None
