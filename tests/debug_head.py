"""GPU debug helper (not a test): stage-by-stage comparison of the head kernels against CPU fp32 and a bf16-rounding emulation."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from cryovit_amd._lib import EPI_CONVT, EPI_BF16_GELU
from cryovit_amd.engine import ops
from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1, _pad2

dev = torch.device("cuda:0")
bf = lambda t: t.to(torch.bfloat16).float()
g = np.load("tests/golden/synthesis_block.npz")
t = lambda k: torch.from_numpy(g[k])
x, y = t("x"), t("y")
C, D, H, W = x.shape
nv = D * H * W
zero = torch.zeros(256, dtype=torch.uint8, device=dev)

def cl(v):  # [C,D,H,W] -> [nv,C]
    return v.permute(1, 2, 3, 0).reshape(nv, -1)

def stat(name, got, ref):
    e = (got - ref).abs()
    print(f"{name}: max {float(e.max()):.4f} mean {float(e.mean()):.5f} ref-absmax {float(ref.abs().max()):.3f} frac>0.1 {float((e>0.1).float().mean()):.4f}")

# CPU fp32 reference stages and bf16-emulated stages
xb = bf(x)
gn_ref = F.group_norm(xb.unsqueeze(0), 8, t("layers_0_weight"), t("layers_0_bias"), 1e-3)[0]
xin = cl(x).to(torch.bfloat16).to(dev)
stats = torch.zeros(16, device=dev)
gn = torch.zeros_like(xin)
ops.groupnorm(xin, t("layers_0_weight").to(dev), t("layers_0_bias").to(dev), gn, stats, nvox=nv, Cdim=32, G=8, eps=1e-3)
stat("gn", gn.float().cpu(), cl(gn_ref))
gn_e = bf(gn_ref)
c1_ref = F.gelu(F.conv3d(gn_e.unsqueeze(0), bf(t("layers_1_weight")), t("layers_1_bias"), padding="same", dilation=(2, 1, 1)))[0]
t1 = torch.zeros(nv, 16, dtype=torch.bfloat16, device=dev)
ops.conv3d(gn, _conv3_weight(t("layers_1_weight")).to(dev), _pad1(t("layers_1_bias"), 16).to(dev), t1, zero, Cin=32, D=D, H=H, W=W, dil=2, cout=16, act=1)
stat("conv1 (vs emulated)", t1.float().cpu(), cl(c1_ref))
c1_e = bf(c1_ref)
c2_ref = F.gelu(F.conv3d(c1_e.unsqueeze(0), bf(t("layers_3_weight")), t("layers_3_bias"), padding="same", dilation=(1, 1, 1)))[0]
t2 = torch.zeros(ops.alloc_rows(nv) * 16 + 4096, dtype=torch.bfloat16, device=dev)
ops.conv3d(t1, _conv3_weight(t("layers_3_weight")).to(dev), _pad1(t("layers_3_bias"), 16).to(dev), t2, zero, Cin=16, D=D, H=H, W=W, dil=1, cout=16, act=1)
stat("conv2 (vs emulated)", t2[: nv * 16].reshape(nv, 16).float().cpu(), cl(c2_ref))
c2_e = bf(c2_ref)
ct_ref = F.gelu(F.conv_transpose3d(c2_e.unsqueeze(0), bf(t("layers_5_weight")), t("layers_5_bias"), stride=(1, 2, 2)))[0]
wt = t("layers_5_weight")
wg = wt[:, :, 0].permute(2, 3, 1, 0).reshape(32, 16)
out = torch.zeros(D, 2 * H, 2 * W, 8, dtype=torch.bfloat16, device=dev)
ops.gemm(EPI_CONVT, torch.as_strided(t2, (ops.alloc_rows(nv), 16), (16, 1)), _pad2(wg, _npad(32), 64).to(dev), out,
         _pad1(t("layers_5_bias").repeat(4), _npad(32)).to(dev), m=nv, n=32, H=H, W=W, cout=8, act=1, ldc=8)
got = out.float().cpu().permute(3, 0, 1, 2)
stat("convT (vs emulated)", got, ct_ref)
stat("convT (vs fixture fp32)", got, y)
stat("emulated vs fixture fp32", ct_ref, y)
