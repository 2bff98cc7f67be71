"""Launch the dominant GEMM (w12 + SwiGLU, ViT-g shape, one 128-slice batch) a few times -- target for rocprofv3 --pmc."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402

os.environ.setdefault("CVX_ABLATION_LIB", "1")  # CVX_ABLATION_LIB=0: the product library (profiles of the shipped kernel)
from cryovit_amd import _lib  # noqa: E402
from cryovit_amd.build import build_library  # noqa: E402

build_library(ablation=os.environ["CVX_ABLATION_LIB"] == "1")
from cryovit_amd._lib import EPI_SWIGLU  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N = 128 * 1032, 1536, 8192
if len(sys.argv) > 1 and int(sys.argv[1]) >= 0:
    _lib.set_option("gemm256_variant", int(sys.argv[1]))
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
out = torch.zeros(ops.alloc_rows(M), N // 2, dtype=torch.bfloat16, device=dev)
# the shipped form (round 3): LayerNorm folded into the epilogue -- bias = [2, N] (b' | column sums), row constants (rstd, -mean*rstd)
bc = torch.randn(2, N, device=dev, generator=g)
rowstat = torch.rand(ops.alloc_rows(M), 2, device=dev, generator=g)
plain = len(sys.argv) > 2 and sys.argv[2] == "plain"  # the round-2 epilogue (acc + bias) for comparison
for _ in range(6):
    if plain:
        ops.gemm(EPI_SWIGLU, a, w, out, bc[0].contiguous(), m=M, n=N)
    else:
        ops.gemm(EPI_SWIGLU, a, w, out, bc, m=M, n=N, ln_rowstat=rowstat)
torch.cuda.synchronize()
