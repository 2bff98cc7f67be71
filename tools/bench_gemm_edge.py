"""What the predicated (non-FULL) path of the persistent GEMM tile costs at the SAM2 Hiera-L stage-3 shapes (65536 rows = one 64-slice
batch; widths 576 / 1728 / 2304): N = 576 runs on 3 tile columns of 256 with a quarter of the third one valid, the same launch with
N = 768 is all interior tiles -- same MFMA work, the difference is predication + the conservative waits."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd._lib import EPI_BF16, EPI_RESID  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 65536
g = torch.Generator(device=dev).manual_seed(0)
R = ops.alloc_rows(M)


def case(epi, N, K, npad):
    a = torch.randn(R, K, device=dev, generator=g).to(torch.bfloat16)
    w = torch.zeros(npad, K, dtype=torch.bfloat16, device=dev)
    w[:N] = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
    bias, gamma = torch.randn(npad, device=dev, generator=g), torch.ones(npad + 256, device=dev)
    out = torch.zeros(R, N, device=dev) if epi == EPI_RESID else torch.zeros(R, N, dtype=torch.bfloat16, device=dev)
    fn = lambda: ops.gemm(epi, a, w, out, bias, m=M, n=N, gamma=gamma if epi == EPI_RESID else None)  # noqa: E731
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        s.record()
        for _ in range(4):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 4)
    ms = sorted(ts)[2]
    return ms, 2.0 * M * N * K / ms / 1e9


for name, epi, K in (("proj  resid K=576 ", EPI_RESID, 576), ("fc2   resid K=2304", EPI_RESID, 2304), ("qkv   bf16  K=576 ", EPI_BF16, 576)):
    for N, npad in ((576, 768), (768, 768)) if epi == EPI_RESID else ((1728, 1792), (1792, 1792)):
        ms, tf = case(epi, N, K, npad)
        print(f"{name} N={N:5d} (n_pad {npad}): {ms:7.3f} ms  {tf:7.1f} TFLOP/s of valid work")
