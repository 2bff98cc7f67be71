"""Builds ``libcryovit_hip.so`` (gfx950) in-tree with hipcc.  No torch involved: the library is a plain C ABI.

    python -m cryovit_amd.build [--force] [--ablation]

``--ablation`` builds ``libcryovit_hip_ablation.so`` with ``-DCVX_ABLATION``: the product library plus the timing-only GEMM /
attention variants (in-kernel stamps, schedules with parts removed whose OUTPUT IS GARBAGE).  Only ``tools/`` loads it
(``CVX_ABLATION_LIB=1``); the product library does not contain those kernels and rejects their option values.

hipcc cross-compiles without a GPU, so this also runs in the (GPU-less) build container; the resulting
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""

from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIB = PKG / "libcryovit_hip.so"
BUILD_DIR = PKG / "build"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-file extras.  attention.hip: keep the MFMA accumulators in VGPRs -- its softmax does VALU math on them every tile;
# with the default AGPR form hipcc emitted 159 v_accvgpr_read/write per 64-key tile (tools/bench_attn.py: 536 -> see DESIGN.md)
FILE_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _fingerprint() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*")) + list(INCLUDE.glob("*.h"))):
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    h.update((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


ABLATION_LIB = PKG / "libcryovit_hip_ablation.so"


def build_library(force: bool = False, verbose: bool = False, ablation: bool = False) -> Path:
    lib, build_dir = (ABLATION_LIB, PKG / "build_ablation") if ablation else (LIB, BUILD_DIR)
    extra = ["-DCVX_ABLATION"] if ablation else []
    more = os.environ.get("CVX_EXTRA_DEFINES", "").split() if ablation else []  # experiments on the ablation build only (tools/)
    extra += more
    stamp = build_dir / "fingerprint"
    fp = _fingerprint() + ("+ablation" + "".join(more) if ablation else "")
    if not force and lib.exists() and stamp.exists() and stamp.read_text() == fp:
        return lib
    build_dir.mkdir(exist_ok=True)
    hipcc = hipcc_path()

    def compile_one(src: Path) -> Path:
        obj = build_dir / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, *extra, *FILE_FLAGS.get(src.name, []), "-I", str(INCLUDE), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    stamp.write_text(fp)
    return lib


if __name__ == "__main__":
    p = build_library(force="--force" in sys.argv, verbose=True, ablation="--ablation" in sys.argv)
    print("built", p)
