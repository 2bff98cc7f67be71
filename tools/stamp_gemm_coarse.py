"""Diagnostic: prologue / main loop / epilogue cycles of one gemm256 tile (variant 21 = default schedule + 4 stamps)."""
import ctypes as C
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

dev = torch.device("cuda:0")
M = 128 * 1032
g = torch.Generator(device=dev).manual_seed(0)
_lib.set_option("gemm256_variant", 21)
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
    buf = (C.c_ulonglong * 32)()
    _lib.check(_lib.load().cvx_debug_read_gemm256(buf), "dbg")
    v = [sum(buf[wv * 4 + i] for wv in range(8)) / 8 for i in range(4)]
    nk = K // 64
    print(f"{name:12s} prologue {v[0]:7.0f}  main {v[1]:8.0f} ({v[1] / nk:6.0f}/K-tile, ideal 2048)  epilogue {v[2]:7.0f}  total {v[3]:8.0f} cycles")
