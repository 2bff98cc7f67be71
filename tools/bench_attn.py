"""In-process A/B of attention kernel variants at the ViT-g shape (128 slices x 24 heads x 1029 tokens)."""
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
from cryovit_amd.engine import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,4,5")
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
slices, heads, nt = 128, 24, 1029
C, ntp, kp = heads * 64, 1032, 1088
M = slices * ntp
g = torch.Generator(device=dev).manual_seed(0)
qk = (torch.randn(ops.alloc_rows(M), 2 * C, device=dev, generator=g) * 0.5).to(torch.bfloat16)
vt = torch.randn(slices, heads, 64, kp, device=dev, generator=g).to(torch.bfloat16)
out = torch.zeros(ops.alloc_rows(M), C, dtype=torch.bfloat16, device=dev)
fl = 4.0 * nt * nt * 64 * heads * slices
qkv = torch.cat([qk[:, : 2 * C], (torch.randn(qk.shape[0], C, device=dev, generator=g)).to(torch.bfloat16)], dim=1).contiguous()  # Q | K | V row-major
res = {}
for r in range(args.rounds):
    for v in [int(x) for x in args.variants.split(",")]:
        if v in (1007, 1008):  # the shipped form: V row-major in the qkv buffer (cvx_attention_qkv_bf16, variant 7 arithmetic); 1008: without the half-tile skip
            _lib.set_option("attn_variant", 7)
            _lib.set_option("attn_mfma_prio", 2)
            _lib.set_option("attn_half_tile", 1 if v == 1007 else 0)
            ops.attention_qkv(qkv, out, slices=slices, heads=heads, ntok=nt, ntp=ntp)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                ops.attention_qkv(qkv, out, slices=slices, heads=heads, ntok=nt, ntp=ntp)
            e.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(s.elapsed_time(e) / 3)
            continue
        _lib.set_option("attn_variant", v % 100)
        _lib.set_option("attn_xcd_remap", 0 if 100 <= v < 200 else 1)  # variant + 100 = same kernel without the XCD block remap
        _lib.set_option("attn_mfma_prio", v // 200)                   # variant + 200 p = s_setprio 1 around the MFMA blocks (p bit 0: S^T, bit 1: O^T)
        ops.attention(qk, vt, out, slices=slices, heads=heads, ntok=nt, ntp=ntp, kp=kp)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            ops.attention(qk, vt, out, slices=slices, heads=heads, ntok=nt, ntp=ntp, kp=kp)
        e.record()
        torch.cuda.synchronize()
        res.setdefault(v, []).append(s.elapsed_time(e) / 3)
for v, ts in res.items():
    med = sorted(ts)[len(ts) // 2]
    print(f"variant {v}: {med:.3f} ms  {fl / med / 1e9:.0f} TF")
