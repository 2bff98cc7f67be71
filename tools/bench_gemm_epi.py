"""What the epilogue kind costs on the proj shape (M = 128*1032, K = N = 1536), persistent tile kernel: bf16 store / fp32 store /
fp32 read-modify-write (the residual update).  Interleaved rounds in one process."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd import _lib  # noqa: E402
from cryovit_amd._lib import EPI_BF16, EPI_F32, EPI_RESID  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N = 128 * 1032, int(sys.argv[1]) if len(sys.argv) > 1 else 1536, 1536
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev, generator=g)
gm = torch.ones(N, device=dev) * 1e-3
o16 = torch.zeros(ops.alloc_rows(M), N, dtype=torch.bfloat16, device=dev)
o32 = torch.zeros(ops.alloc_rows(M), N, device=dev)
cases = {"bf16 store": (EPI_BF16, o16), "fp32 store": (EPI_F32, o32), "fp32 RMW": (EPI_RESID, o32)}
res = {k: [] for k in cases}
for r in range(6):
    for name, (epi, out) in cases.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.gemm(epi, a, w, out, bias, m=M, n=N, gamma=gm)
        s.record()
        for _ in range(4):
            ops.gemm(epi, a, w, out, bias, m=M, n=N, gamma=gm)
        e.record()
        torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 4)
for name, v in res.items():
    t = sorted(v)[len(v) // 2]
    print(f"K={K} {name:10s}: {t:6.3f} ms  {2.0 * M * K * N / t / 1e9:7.1f} TF")
