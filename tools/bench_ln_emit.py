"""Timing prototype for the LayerNorm fold: what does it cost the persistent residual GEMM to ALSO emit
bf16(x) (16-B stores after a lane-pair exchange, 48 instead of 32 stores per wave and tile)?  Two builds of the library, one process
each:   python tools/bench_ln_emit.py main | emit      (emit = gemm.hip compiled with -DCVX_LN_EMIT_PROTO -> libcryovit_hip_emit.so)
Prints ms per launch for the proj (K = 1536) and w3 (K = 4096) shapes at M = 128 x 1032 rows, and checks the emitted copy."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from cryovit_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "main"
if mode == "emit":
    _lib.LIB_PATH = ROOT / "cryovit_amd" / "libcryovit_hip_emit.so"
from cryovit_amd._lib import EPI_RESID  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M, N = 128 * 1032, 1536
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.zeros(ops.alloc_rows(M), N, dtype=torch.bfloat16, device=dev)
RP = ops.alloc_rows(M)
part = torch.zeros(N // 64, RP, 2, device=dev)  # P[slot][row][sum, sum of squares], one slot per 64 columns
if mode == "emit":
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    raw.cvx_debug_set_emit.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
    assert raw.cvx_debug_set_emit(xb.data_ptr(), N, part.data_ptr(), RP) == 0
for name, K in (("proj", 1536), ("w3", 4096)):
    a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
    bias, gamma = torch.randn(N, device=dev, generator=g), torch.ones(N, device=dev) * 1e-3
    x = torch.randn(ops.alloc_rows(M), N, device=dev, generator=g)
    for _ in range(3):
        ops.gemm(EPI_RESID, a, w, x, bias, m=M, n=N, gamma=gamma)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.gemm(EPI_RESID, a, w, x, bias, m=M, n=N, gamma=gamma)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    ok = ""
    if mode == "emit":
        main_rows = (M // 256 * (N // 256)) // 256 * 256 // (N // 256) * 256  # rows of the whole rounds (the tail launch does not emit)
        ok = f"  emitted copy == bf16(x): {bool(torch.equal(xb[:main_rows], x[:main_rows].to(torch.bfloat16)))} (rows 0..{main_rows})"
        xs = x[:main_rows].double().reshape(main_rows, N // 64, 64)
        want = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).permute(1, 0, 2)  # [slot][row][2]
        got = part[:, :main_rows].double()
        rel = float(((got - want).abs() / (want.abs() + 1e-3)).max())
        ok += f"; row partial sums max rel err {rel:.1e}"
    print(f"{mode:5s} {name:5s} K={K}: {ms:.4f} ms / launch  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s{ok}", flush=True)
