"""Eager launches vs hipGraph replay of the per-tomogram sequence (cryovit_amd/engine/graph.py), ViT-g + full head:
ms per tomogram at the benchmark size and at BASELINE configs[0]'s geometry (where the path is launch-bound)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402
from cryovit_amd.engine.graph import GraphedTomogram  # noqa: E402
from cryovit_amd.engine.head import HeadEngine  # noqa: E402
from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
vit = VitEngine(cfg, random_state_dict(cfg, seed=2, device=dev), dev)
head = HeadEngine(bench.synthetic_head_state_dict(5, dev), dev)
for D, H, W, sb, reps in [(64, 256, 256, 64, 10), (16, 256, 256, 16, 20), (128, 512, 512, 128, 3)]:
    vol = (torch.rand(D, H, W, device=dev) * 255).to(torch.uint8)
    labels = torch.zeros(D, H, W, dtype=torch.int8, device=dev)
    hp, wp = H // 16, W // 16
    f16 = torch.zeros(cfg.dim, D, hp, wp, dtype=torch.float16, device=dev)
    cl = torch.zeros(ops.alloc_rows(D * hp * wp), cfg.dim, dtype=torch.float16, device=dev)

    def eager():
        for d0 in range(0, D, sb):
            vit.features(vol[d0 : d0 + sb], feats_f16=f16, d_total=D, d0=d0, feats_cl=cl[d0 * hp * wp :])
        return head.forward(cl, D, hp, wp, labels=labels)

    g = GraphedTomogram(vit, head, D, H, W, slice_batch=sb)
    res = {}
    for name, fn in (("eager", eager), ("graph", lambda: g.run(vol, labels))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps * 1e3
    print(f"{D}x{H}x{W}: eager {res['eager']:.2f} ms  graph {res['graph']:.2f} ms  ({res['eager'] / res['graph']:.3f}x)", flush=True)
    del g
