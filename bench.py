"""Headline benchmark: tomogram voxels/sec for (DINOv2 ViT-g/14-reg features + CryoVIT 3D-conv head forward + Dice).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one synthetic 128x512x512 tomogram (BASELINE.json configs[1]+[2] chained = the end-to-end unit of
configs[3]) through the whole hot path on one GPU, input volume already resident in HBM.  Tomograms shard
embarrassingly (SURVEY s.8e): every rank processes its own K tomograms, no data-path collective ("weak" scaling);
RCCL is used only for the barrier and the MAX-over-ranks of the timed region.

The JSON line also carries
  roofline     -- the dominant kernel (the SwiGLU w12 GEMM, 44 % of all FLOPs): algorithmic FLOPs per launch divided
                  by its average launch duration measured live with HIP events on the launch stream
  cpu_baseline -- the torch-CPU fp32 oracle timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

D_, H_, W_ = 128, 512, 512
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def synthetic_head_state_dict(seed: int, device) -> dict:
    """Variance-preserving synthetic init of the reference-layout head (PyTorch's default init gives an all-background
    prediction and a degenerate Dice == 0 -- SURVEY App. E)."""
    from cryovit_amd.engine.head import REF_WIDTHS

    g = torch.Generator(device=device).manual_seed(seed)
    c_in, blocks, c_tail = REF_WIDTHS

    def n(shape, std, mean=0.0):
        return torch.empty(*shape, device=device).normal_(mean, std, generator=g)

    sd = {"layers.0.weight": n((blocks[0][0], c_in, 1, 1, 1), math.sqrt(2.0 / c_in)), "layers.0.bias": n((blocks[0][0],), 0.05)}
    for i, (c1, c2, c3, _, _) in enumerate(blocks):
        p = f"layers.{i + 2}.layers."
        sd[p + "0.weight"], sd[p + "0.bias"] = n((c1,), 0.1, 1.0), n((c1,), 0.1)
        sd[p + "1.weight"], sd[p + "1.bias"] = n((c2, c1, 3, 3, 3), math.sqrt(2.0 / (27 * c1))), n((c2,), 0.05)
        sd[p + "3.weight"], sd[p + "3.bias"] = n((c2, c2, 3, 3, 3), math.sqrt(2.0 / (27 * c2))), n((c2,), 0.05)
        sd[p + "5.weight"], sd[p + "5.bias"] = n((c2, c3, 1, 2, 2), math.sqrt(2.0 / c2)), n((c3,), 0.05)
    sd["output_layer.0.weight"], sd["output_layer.0.bias"] = n((c_tail, c_tail, 3, 3, 3), math.sqrt(2.0 / (27 * c_tail))), n((c_tail,), 0.05)
    sd["output_layer.2.weight"], sd["output_layer.2.bias"] = n((1, c_tail, 3, 3, 3), math.sqrt(2.0 / (27 * c_tail))), n((1,), 0.05)
    return sd


def synthetic_labels(device, seed: int) -> torch.Tensor:
    """int8 {-1,0,1}: z<16 and z>=112 unlabeled, one ellipsoid of foreground (SURVEY s.8d config 3)."""
    z = torch.arange(D_, device=device).view(-1, 1, 1).float()
    y = torch.arange(H_, device=device).view(1, -1, 1).float()
    x = torch.arange(W_, device=device).view(1, 1, -1).float()
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(3, generator=g) * 0.2 + 0.4
    lab = ((((z / D_ - c[0]) / 0.3) ** 2 + ((y / H_ - c[1]) / 0.3) ** 2 + ((x / W_ - c[2]) / 0.3) ** 2) < 1.0).to(torch.int8)
    lab[:16] = -1
    lab[112:] = -1
    return lab.contiguous()


class DominantKernelTimer:
    """HIP-event timing of one GEMM epilogue kind on the stream the kernels are launched on, through the library's
    measurement hook (the encoder is ONE C call, cvx_vit_encode, so the events are recorded inside it)."""

    def __init__(self, lib_mod, epilogue: int, capacity: int = 4096):
        import ctypes as C

        self.lib, self.epi, self.cap, self.C = lib_mod, epilogue, capacity, C
        self.starts = [torch.cuda.Event(enable_timing=True) for _ in range(capacity)]
        self.stops = [torch.cuda.Event(enable_timing=True) for _ in range(capacity)]
        for e in self.starts + self.stops:  # force creation of the underlying hipEvent_t
            e.record()
        torch.cuda.synchronize()
        self.h_start = (C.c_void_p * capacity)(*[e.cuda_event for e in self.starts])
        self.h_stop = (C.c_void_p * capacity)(*[e.cuda_event for e in self.stops])
        self.n = 0

    def start(self):
        self.lib.check(self.lib.load().cvx_set_gemm_event_hook(self.epi, self.h_start, self.h_stop, self.cap), "hook")

    def stop(self):
        self.n = self.lib.load().cvx_get_gemm_event_count()
        self.lib.check(self.lib.load().cvx_set_gemm_event_hook(-1, None, None, 0), "hook")

    def mean_ms(self) -> float:
        return sum(self.starts[i].elapsed_time(self.stops[i]) for i in range(self.n)) / max(1, self.n)


def cpu_baseline() -> dict:
    """torch-CPU fp32 oracle on a bounded sample of the same workload (SURVEY s.8d 'CPU reference timing')."""
    from oracle import dinov2 as o
    from oracle import head as oh

    threads = torch.get_num_threads()
    cfg = o.VITG14_REG
    # timing-only weights: one block's tensors shared by all 40 layers (values do not affect the time)
    one = o.init_state_dict(o.VitCfg(cfg.dim, 1, cfg.heads, cfg.ffn, cfg.ffn_hidden), seed=1)
    sd = dict(one)
    for i in range(1, cfg.depth):
        for k, v in one.items():
            if k.startswith("blocks.0."):
                sd[k.replace("blocks.0.", f"blocks.{i}.")] = v
    x = torch.rand(1, 3, 448, 448)
    t0 = time.perf_counter()
    o.forward_features(cfg, sd, x)
    t_vit = time.perf_counter() - t0  # one 512x512 slice
    head = oh.CryoVITHead()
    oh.rescaled_init_(head, seed=5)
    d_s, h_s = 8, 4  # [1,1536,8,4,4] -> 8 x 64 x 64 output voxels
    feats = torch.randn(1, 1536, d_s, h_s, h_s)
    t0 = time.perf_counter()
    with torch.inference_mode():
        head.forward_volume(feats)
    t_head = time.perf_counter() - t0
    s_per_voxel = t_vit / (H_ * W_) + t_head / (d_s * (16 * h_s) ** 2)
    return {
        "value": 1.0 / s_per_voxel, "unit": "voxels/s", "cores": threads, "kind": "port",
        "sample": f"oracle fp32: ViT-g/14-reg on 1 slice 448x448 ({t_vit:.1f} s) + head on [1,1536,{d_s},{h_s},{h_s}] "
                  f"({t_head:.1f} s), extrapolated per voxel to 128x512x512; host has {os.cpu_count()} cpus",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slice-batch", type=int, default=128, help="slices per ViT launch sequence (reference default 128)")
    ap.add_argument("--streams", type=int, default=1, help="tomograms in flight per GPU, one per HIP stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs MI355X devices"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm; only barrier + timing MAX use it

    from cryovit_amd import _lib
    from cryovit_amd.build import build_library
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict

    if rank == 0:
        build_library()
    if world > 1:
        dist.barrier()
    _lib.load()

    cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
    vit = VitEngine(cfg, random_state_dict(cfg, seed=2, device=dev), dev)
    head = HeadEngine(synthetic_head_state_dict(5, dev), dev)
    torch.cuda.empty_cache()

    vol = (torch.rand(D_, H_, W_, generator=torch.Generator().manual_seed(100 + rank)) * 255).to(torch.uint8).to(dev)
    labels = synthetic_labels(dev, 4 + rank)
    hp, wp = H_ // 16, W_ // 16
    nvox_feat = D_ * hp * wp
    feats_cl = torch.zeros(ops.alloc_rows(nvox_feat), cfg.dim, dtype=torch.float16, device=dev)
    feats_f16 = torch.zeros(cfg.dim, D_, hp, wp, dtype=torch.float16, device=dev)  # the on-disk `dino_features` tensor
    sb = args.slice_batch

    # args.streams volumes in flight, one per HIP stream (north_star: "one volume per HIP stream"): kernels of different
    # volumes fill each other's tails and epilogue bubbles.  Every context has private workspaces; weights are shared.
    S = max(1, args.streams)
    ctxs = []
    for k in range(S):
        ctxs.append({
            "vit": vit if k == 0 else vit.clone_for_stream(), "head": head if k == 0 else head.clone_for_stream(),
            "stream": torch.cuda.current_stream() if k == 0 else torch.cuda.Stream(device=dev),
            "cl": feats_cl if k == 0 else torch.zeros_like(feats_cl), "f16": feats_f16 if k == 0 else torch.zeros_like(feats_f16),
        })
    step_no = [0]

    def step():
        c = ctxs[step_no[0] % S]
        step_no[0] += 1
        with torch.cuda.stream(c["stream"]):
            for d0 in range(0, D_, sb):
                b = min(sb, D_ - d0)
                c["vit"].features(vol[d0 : d0 + b], feats_f16=c["f16"], d_total=D_, d0=d0, feats_cl=c["cl"][d0 * hp * wp :])
            return c["head"].forward(c["cl"], D_, hp, wp, labels=labels, want_probs=True)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from cryovit_amd._lib import EPI_SWIGLU

    kt = DominantKernelTimer(_lib, EPI_SWIGLU)
    for _ in range(args.warmup):
        out = step()
    sync_all()
    kt.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    kt.stop()
    k_ms = kt.mean_ms()
    n_launch = kt.n

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    i, sy, sp = out["dice_sums"].cpu().tolist()
    fg = float((out["probs"] >= 0.5).float().mean())

    if rank == 0:
        voxels = D_ * H_ * W_
        value = world * args.steps * voxels / elapsed
        rows = min(sb, D_) * (hp * wp + 1 + cfg.n_reg)  # valid tokens per launch
        k_flops = 2.0 * rows * cfg.dim * 2 * cfg.ffn_hidden
        achieved = k_flops / (k_ms * 1e-3) / 1e12
        flops_tomo = vit.flops(D_, H_, W_) + head.flops(D_, hp, wp)
        traffic = None
        tf = ROOT / "profiles" / "dominant_kernel_traffic.json"
        if tf.exists():
            traffic = json.loads(tf.read_text()).get("hbm_bytes_per_launch")
        line = {
            "metric": "tomogram voxels/sec (DINO feats + 3D seg fwd)", "value": value, "unit": "voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "one 128x512x512 uint8 tomogram per step per GPU: fused resize + DINOv2 ViT-g/14-reg "
                                   "(40 layers, N=1029 tokens/slice) -> fp16 dino_features [1536,128,32,32] + CryoVIT head "
                                   "[1,1536,128,32,32] -> probs [128,512,512] + masked Dice; synthetic weights",
                       "slice_batch": sb, "streams": args.streams, "parallelism": f"tomogram-sharded x{world}, no collectives"},
            "tflops_end_to_end": flops_tomo * world * args.steps / elapsed / 1e12,
            "frac_of_mfma_peak_end_to_end": flops_tomo * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS,
            "dice": 2 * i / (sy + sp + 1e-3), "pred_fg_fraction": fg,
            "roofline": {"bound": "mfma", "kernel": "k_gemm256_nreg<EpiSwiGLU,5> (w12 GEMM 1536->8192 + fused SiLU gate)", "achieved": achieved,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "launches_timed": n_launch, "avg_launch_ms": k_ms, "flops_per_launch": k_flops},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
