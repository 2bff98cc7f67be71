"""Diagnostic: run gemm256 variant 20 (s_memtime-stamped) on the w12 shape and print where a phase's cycles go."""
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
from cryovit_amd._lib import EPI_BF16, EPI_SWIGLU  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N = 128 * 1032, 1536, 8192
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev, generator=g)
out = torch.zeros(ops.alloc_rows(M), N // 2, dtype=torch.bfloat16, device=dev)
_lib.set_option("gemm256_variant", 20)
for _ in range(3):
    ops.gemm(EPI_SWIGLU, a, w, out, bias, m=M, n=N)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
_lib.check(_lib.load().cvx_debug_read_gemm256(buf), "dbg")
nph = 4 * (K // 64)
print("per-phase cycle averages (block 0): wave  load  load-barrier  mma  mma-barrier  total")
for wv in range(8):
    v = [buf[wv * 4 + i] / nph for i in range(4)]
    print(f"  wave {wv}: {v[0]:7.0f} {v[1]:7.0f} {v[2]:7.0f} {v[3]:7.0f}   {sum(v):7.0f}")
