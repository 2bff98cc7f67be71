"""LayerNorm micro-benchmark at the ViT-g shape (132096 rows x 1536) and the Hiera stage shapes: ms and GB/s (read fp32 + write bf16)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
for rows, C, ldo in [(132096, 1536, 1536), (64 * 16384, 144, 192), (64 * 4096, 288, 320), (64 * 1024, 576, 576)]:
    x = torch.randn(rows, C, device=dev)
    w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.zeros(rows, ldo, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.layernorm(x, w, b, out, rows, C, 1e-6)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.layernorm(x, w, b, out, rows, C, 1e-6)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"rows {rows} C {C}: {ms:.3f} ms  {rows * C * 6 / ms / 1e6:.0f} GB/s")
