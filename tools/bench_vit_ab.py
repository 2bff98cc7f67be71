"""In-process A/B of whole-encoder plans on ONE device (boards differ by ~5 %: never compare plans across runs): ViT-g/14-reg over one
128x512x512 tomogram, interleaved rounds.  Plans: the shipped one (folded LayerNorms + one qkv GEMM), folded with split qk / V^T GEMMs,
round 2's (fp32 stream + LayerNorm launches).

    python tools/bench_vit_ab.py [--rounds 4] [--plans merged,split,r2]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd.engine import ops  # noqa: E402
from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--plans", default="merged,split,r2")
ap.add_argument("--opts", default="", help="comma list of option sets NAME=VALUE[+NAME=VALUE]: each plan is timed under each set")
args = ap.parse_args()
from cryovit_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
sd = random_state_dict(cfg, seed=2, device=dev)
kw = {"merged": dict(fold_ln=True, merge_qkv=True), "split": dict(fold_ln=True, merge_qkv=False), "r2": dict(fold_ln=False)}
plans = args.plans.split(",")
D, H, W = 128, 512, 512
vol = (torch.rand(D, H, W, generator=torch.Generator().manual_seed(100)) * 255).to(torch.uint8).to(dev)
f16 = torch.zeros(1536, D, 32, 32, dtype=torch.float16, device=dev)
optsets = [o for o in args.opts.split(",") if o] or [""]
res = {(p, o): [] for p in plans for o in optsets}
engines = {}
for r in range(args.rounds):
  for o in optsets:
    for kv in (o.split("+") if o else []):
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    for p in plans:
        if p not in engines:  # one engine alive at a time would re-pack weights every round: keep all (3 x 2.3 GB)
            engines[p] = VitEngine(cfg, sd, dev, **kw[p])
        eng = engines[p]
        eng.features(vol, feats_f16=f16, d_total=D, d0=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            eng.features(vol, feats_f16=f16, d_total=D, d0=0)
        e1.record()
        torch.cuda.synchronize()
        res[(p, o)].append(e0.elapsed_time(e1) / 2)
for (p, o), ts in res.items():
    ts = sorted(ts)
    print(f"{p:7s} {o:28s}: {ts[len(ts) // 2]:8.2f} ms per tomogram (ViT-g features alone; min {ts[0]:.2f}, max {ts[-1]:.2f})")
