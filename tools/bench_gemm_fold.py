"""In-process A/B of the ViT-g GEMMs in their round-2 form (fp32 residual stream, plain epilogues) and their round-3 form (bf16
hi/lo residual stream, LayerNorm folded into the consuming epilogues), interleaved rounds on one device, product library.

    python tools/bench_gemm_fold.py [--rounds 5] [--m 132096]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd import _lib  # noqa: E402
from cryovit_amd.build import build_library  # noqa: E402

build_library()
from cryovit_amd._lib import EPI_BF16, EPI_RESID, EPI_RESID_HL, EPI_SWIGLU, EPI_VT  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--m", type=int, default=128 * 1032)
ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE passed to cvx_set_option")
args = ap.parse_args()
for o in args.opt:
    k, v = o.split("=")
    _lib.set_option(k, int(v))
dev = torch.device("cuda:0")
M, C, H = args.m, 1536, 4096
R = ops.alloc_rows(M)
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)  # noqa: E731
a_c, a_h = rb(R, C), rb(R, H)
w = {"qk": rb(2 * C, C) * C**-0.5, "v": rb(C, C) * C**-0.5, "proj": rb(C, C) * C**-0.5, "w12": rb(2 * H, C) * C**-0.5, "w3": rb(C, H) * H**-0.5}
w = {k: v.to(torch.bfloat16).contiguous() for k, v in w.items()}
bias = {k: torch.randn(v.shape[0], device=dev, generator=g) for k, v in w.items()}
bc = {k: torch.randn(2, v.shape[0], device=dev, generator=g) for k, v in w.items()}
gamma = torch.full((C,), 1e-3, device=dev)
rowstat = torch.rand(R, 2, device=dev, generator=g)
x32 = torch.zeros(R, C, device=dev)
xh, xl = torch.zeros(R, C, dtype=torch.bfloat16, device=dev), torch.zeros(R, C, dtype=torch.bfloat16, device=dev)
part = torch.zeros(C // 64, R, 2, device=dev)
o_qk, o_hid = torch.zeros(R, 2 * C, dtype=torch.bfloat16, device=dev), torch.zeros(R, H, dtype=torch.bfloat16, device=dev)
ntp, kp, heads = 1032, 1088, 24
vt = torch.zeros(M // ntp, heads, 64, kp, dtype=torch.bfloat16, device=dev)

cases = {
    "qk   plain": (lambda: ops.gemm(EPI_BF16, a_c, w["qk"], o_qk, bias["qk"], m=M, n=2 * C), 2 * C * C),
    "qk   LN   ": (lambda: ops.gemm(EPI_BF16, a_c, w["qk"], o_qk, bc["qk"], m=M, n=2 * C, ln_rowstat=rowstat), 2 * C * C),
    "v    plain": (lambda: ops.gemm(EPI_VT, a_c, w["v"], vt, bias["v"], m=M, n=C, heads=heads, ntp=ntp, kp=kp, ldc=0), C * C),
    "v    LN   ": (lambda: ops.gemm(EPI_VT, a_c, w["v"], vt, bc["v"], m=M, n=C, heads=heads, ntp=ntp, kp=kp, ldc=0, ln_rowstat=rowstat), C * C),
    "N1536 bf16 NREG (what V^T would cost row-major)": (lambda: ops.gemm(EPI_BF16, a_c, w["v"], o_qk, bc["v"], m=M, n=C, ln_rowstat=rowstat, ldc=2 * C), C * C),
    "proj fp32 ": (lambda: ops.gemm(EPI_RESID, a_c, w["proj"], x32, bias["proj"], m=M, n=C, gamma=gamma), C * C),
    "proj hi/lo": (lambda: ops.gemm(EPI_RESID_HL, a_c, w["proj"], xh, bias["proj"], m=M, n=C, gamma=gamma, out2=xl, stat_part=part), C * C),
    "w12  plain": (lambda: ops.gemm(EPI_SWIGLU, a_c, w["w12"], o_hid, bias["w12"], m=M, n=2 * H), 2 * H * C),
    "w12  LN   ": (lambda: ops.gemm(EPI_SWIGLU, a_c, w["w12"], o_hid, bc["w12"], m=M, n=2 * H, ln_rowstat=rowstat), 2 * H * C),
    "w3   fp32 ": (lambda: ops.gemm(EPI_RESID, a_h, w["w3"], x32, bias["w3"], m=M, n=C, gamma=gamma), C * H),
    "w3   hi/lo": (lambda: ops.gemm(EPI_RESID_HL, a_h, w["w3"], xh, bias["w3"], m=M, n=C, gamma=gamma, out2=xl, stat_part=part), C * H),
    "rowstat finalize": (lambda: ops.rowstat_finalize(part, rowstat, rows=M, Cdim=C, eps=1e-6), 0),
}
res = {k: [] for k in cases}


def run(fn, reps=4):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for r in range(args.rounds):
    for k, (fn, _) in cases.items():
        res[k].append(run(fn))
for k, (_, nk) in cases.items():
    ms = sorted(res[k])[len(res[k]) // 2]
    print(f"{k}: {ms:7.3f} ms" + (f"  {2.0 * M * nk / (ms * 1e-3) / 1e12:7.1f} TFLOP/s (best {2.0 * M * nk / (min(res[k]) * 1e-3) / 1e12:6.1f})" if nk else ""))
