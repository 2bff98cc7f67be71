import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from cryovit_amd.engine import ops
dev = torch.device("cuda:0")
g = np.load("tests/golden/synthesis_block.npz")
x = torch.from_numpy(g["x"])
C, D, H, W = x.shape
nv = D * H * W
xin = x.permute(1, 2, 3, 0).reshape(nv, C).to(torch.bfloat16)
w, b = torch.from_numpy(g["layers_0_weight"]), torch.from_numpy(g["layers_0_bias"])
print("w", w[:8], "b", b[:8])
for trial, (ww, bb) in enumerate([(w, b), (torch.ones(C), torch.zeros(C))]):
    stats = torch.zeros(16, device=dev)
    out = torch.zeros(nv, C, dtype=torch.bfloat16, device=dev)
    ops.groupnorm(xin.to(dev), ww.to(dev), bb.to(dev), out, stats, nvox=nv, Cdim=C, G=8, eps=1e-3)
    xf = xin.float()
    grp = xf.reshape(nv, 8, 4)
    print("gpu sums ", stats[:8].cpu().numpy())
    print("ref sums ", grp.sum(dim=(0, 2)).numpy())
    print("gpu sumsq", stats[8:].cpu().numpy())
    print("ref sumsq", (grp * grp).sum(dim=(0, 2)).numpy())
    ref = F.group_norm(xf.T.reshape(1, C, D, H, W), 8, ww, bb, 1e-3)[0].permute(1, 2, 3, 0).reshape(nv, C)
    e = (out.float().cpu() - ref).abs()
    print("trial", trial, "max err", float(e.max()), "per-channel max err", e.max(dim=0).values.numpy().round(3))
