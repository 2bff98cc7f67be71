"""Diagnostic: where one workgroup of the persistent GEMM tile (gemm256p.h) spends its cycles, tile by tile (variant 29 =
the persistent schedule + stamps; -DCVX_ABLATION build)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402

os.environ.setdefault("CVX_ABLATION_LIB", "1")
from cryovit_amd import _lib  # noqa: E402
from cryovit_amd.build import build_library  # noqa: E402

build_library(ablation=True)
from cryovit_amd._lib import EPI_BF16, EPI_RESID, EPI_SWIGLU  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 128 * 1032
g = torch.Generator(device=dev).manual_seed(0)
_lib.set_option("gemm256_variant", 29)
if len(sys.argv) > 1:
    _lib.set_option("gemm_stagger", int(sys.argv[1]))
for name, epi, K, N in (("w12/SwiGLU", EPI_SWIGLU, 1536, 8192), ("proj/Resid", EPI_RESID, 1536, 1536), ("w3/Resid", EPI_RESID, 4096, 1536),
                        ("qk/BF16", EPI_BF16, 1536, 3072)):
    a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    if epi == EPI_RESID:
        out = torch.zeros(ops.alloc_rows(M), N, device=dev)
    else:
        out = torch.zeros(ops.alloc_rows(M), N // 2 if epi == EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    gm = torch.ones(N, device=dev) * 1e-3
    for _ in range(3):
        ops.gemm(epi, a, w, out, bias, m=M, n=N, gamma=gm if epi == EPI_RESID else None)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 96)()
    _lib.check(_lib.load().cvx_debug_read_gemm256p(buf), "dbg")
    nk = K // 64
    print(f"== {name}: K tiles {nk}; columns: first K tile | K tiles 1.. (per K tile) | wait for partner group | epilogue | release barriers | tile total")
    for grp in range(2):
        for t in range(1, 7):
            s = [buf[(grp * 8 + t) * 6 + k] for k in range(6)]
            nxt = buf[(grp * 8 + t + 1) * 6 + 0]
            print(f"  group {grp} tile {t}: {s[5] - s[0]:6d} | {s[1] - s[5]:7d} ({(s[1] - s[5]) / (nk - 1):6.0f}) | {s[2] - s[1]:6d} | {s[3] - s[2]:6d} | {s[4] - s[3]:6d} | {nxt - s[0]:7d}")
