"""In-process A/B of the GEMM tile kernels on the ViT-g shapes (interleaved rounds, one device; guide rule 24).

    python tools/bench_gemm.py [--rounds 5] [--variants 128,0,1,2,3]
variant 128 = the 128x128 two-phase tile (gemm_core.h); 400 = the 4-wave tile (gemm4w.h); 0.. = gemm256.h schedules.
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402

os.environ.setdefault("CVX_ABLATION_LIB", "1")  # the timing-only variants live in the -DCVX_ABLATION build only
from cryovit_amd import _lib  # noqa: E402
from cryovit_amd.build import build_library  # noqa: E402

build_library(ablation=os.environ["CVX_ABLATION_LIB"] == "1")
from cryovit_amd._lib import EPI_BF16, EPI_RESID, EPI_SWIGLU  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="128,0,1,2,3")
ap.add_argument("--m", type=int, default=128 * 1032)
ap.add_argument("--group-l", default="", help="comma list: sweep the tile-order band height instead of kernel variants")
args = ap.parse_args()
dev = torch.device("cuda:0")
M = args.m
shapes = {"qk  K1536 N3072": (EPI_BF16, 1536, 3072), "proj K1536 N1536": (EPI_RESID, 1536, 1536), "w12 K1536 N8192": (EPI_SWIGLU, 1536, 8192),
          "w3  K4096 N1536": (EPI_RESID, 4096, 1536)}
g = torch.Generator(device=dev).manual_seed(0)
bufs = {}
for name, (epi, K, N) in shapes.items():
    a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    if epi == EPI_RESID:
        out = torch.zeros(ops.alloc_rows(M), N, device=dev)
    elif epi == EPI_SWIGLU:
        out = torch.zeros(ops.alloc_rows(M), N // 2, dtype=torch.bfloat16, device=dev)
    else:
        out = torch.zeros(ops.alloc_rows(M), N, dtype=torch.bfloat16, device=dev)
    bufs[name] = (a, w, bias, out, torch.ones(N, device=dev) * 1e-3)
def parse_variant(tok):  # "9s4000" = variant 9 with gemm_stagger 4000 (cycles between XCD start offsets)
    v, _, st = tok.partition("s")
    return (int(v), int(st) if st else 0)


variants = [parse_variant(v) for v in args.variants.split(",")]
groups = [int(v) for v in args.group_l.split(",")] if args.group_l else []
if groups:
    base_variant = variants[0]
    variants = groups
res = {(n, v): [] for n in shapes for v in variants}


def run(name, v, reps=4):
    epi, K, N = shapes[name]
    a, w, bias, out, gamma = bufs[name]
    if groups:
        _lib.set_option("tile_group_l", v)
        v = base_variant
    v, stag = v
    _lib.set_option("gemm_stagger", stag)
    _lib.set_option("gemm_tail_split", 0 if 1000 <= v < 2000 else 1)  # 1000 + v = schedule v without the tail split
    v = v - 1000 if 1000 <= v < 2000 else v
    _lib.set_option("use_gemm256", 0 if v == 128 else (2 if v >= 400 else 1))  # 400 + a = 4-wave tile, ablation a
    _lib.set_option("gemm256_variant", 0 if v == 128 else (v - 400 if v >= 400 else v))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.gemm(epi, a, w, out, bias, m=M, n=N, gamma=gamma if epi == EPI_RESID else None)
    s.record()
    for _ in range(reps):
        ops.gemm(epi, a, w, out, bias, m=M, n=N, gamma=gamma if epi == EPI_RESID else None)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for r in range(args.rounds):
    for name in shapes:
        for v in variants:
            res[(name, v)].append(run(name, v))
valid_rows = M
for name, (epi, K, N) in shapes.items():
    fl = 2.0 * valid_rows * K * N
    label = (lambda v: f"group_l {v}") if groups else (lambda v: f"v{v[0]}s{v[1]}")
    print(name, "  ".join(f"{label(v)}: {fl / (sorted(res[(name, v)])[len(res[(name, v)]) // 2] * 1e-3) / 1e12:7.1f} TF (min {fl / (min(res[(name, v)]) * 1e-3) / 1e12:6.1f}..)" for v in variants))
