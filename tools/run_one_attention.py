"""Launch the attention kernel (ViT-g shape: 128 slices x 24 heads x 1029 tokens, the shipped default variant, product library) a
few times -- target for rocprofv3 --pmc (tools/profile_round.sh)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd.build import build_library  # noqa: E402

build_library()
from cryovit_amd.engine import ops  # noqa: E402

dev = torch.device("cuda:0")
slices, heads, nt = 128, 24, 1029
C, ntp, kp = heads * 64, 1032, 1088
M = slices * ntp
g = torch.Generator(device=dev).manual_seed(0)
qk = (torch.randn(ops.alloc_rows(M), 2 * C, device=dev, generator=g) * 0.5).to(torch.bfloat16)
vt = torch.randn(slices, heads, 64, kp, device=dev, generator=g).to(torch.bfloat16)
out = torch.zeros(ops.alloc_rows(M), C, dtype=torch.bfloat16, device=dev)
for _ in range(4):
    ops.attention(qk, vt, out, slices=slices, heads=heads, ntok=nt, ntp=ntp, kp=kp)
torch.cuda.synchronize()
