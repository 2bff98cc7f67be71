"""UNet3D baseline forward on one raw 128x512x512 tomogram (synthetic weights): ms per forward; under
`rocprofv3 --kernel-trace --stats` the per-kernel split."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd.engine.unet3d import UNet3DEngine  # noqa: E402
from cryovit_amd.models import UNet3D  # noqa: E402

dev = torch.device("cuda:0")
model = UNet3D(device="cpu")
g = torch.Generator().manual_seed(6)
with torch.no_grad():
    for name, p in model.named_parameters():
        p.copy_(torch.randn(p.shape, generator=g) * ((2.0 / max(1, p[0].numel())) ** 0.5 if p.dim() > 1 else 0.1) + (1.0 if p.dim() == 1 and name.endswith(("1.weight", "4.weight")) else 0.0))
eng = UNet3DEngine(model.state_dict(), dev)
vol = torch.rand(128, 512, 512, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
for copy in ((False, True, False, True) if "--ab" in sys.argv else (False,)):  # --ab: in-place concatenation vs round 2's copy kernel
    eng.concat_copy = copy
    eng.forward(vol)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        p = eng.forward(vol)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"UNet3D forward ({'concat copy kernel' if copy else 'concatenation in place'}): {ms:.3f} ms = {vol.numel() / ms / 1e6:.2f} Gvoxel/s = "
          f"{eng.flops(*vol.shape) / ms / 1e9:.0f} TFLOP/s; mean prob {float(p.mean()):.4f}")
