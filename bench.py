"""Headline benchmark: tomogram voxels/sec for (DINOv2 ViT-g/14-reg features + CryoVIT 3D-conv head forward + Dice).

    python bench.py --gpus N --steps K --warmup W

One "step" = one synthetic 128x512x512 tomogram (BASELINE.json configs[1]+[2] chained = the end-to-end unit of
configs[3]) through the whole hot path on one GPU, input volume already resident in HBM.  Tomograms shard
embarrassingly (SURVEY s.8e): every rank processes its own K tomograms (seeds 100 + rank*K + step: at N = 8, K = 4 these
are configs[3]'s 32 tomograms, seeds 100..131, 4 per GPU), no data-path collective ("weak" scaling); RCCL is used only
for the barrier and the MAX-over-ranks of the timed region.

Launching.  ``python bench.py --gpus N`` with N > 1 is SELF-LAUNCHING: the parent process builds the library, touches no
GPU, starts N child processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1),
relays rank 0's JSON line and exits non-zero if any child fails.  Under ``python -m torch.distributed.run ... bench.py
--gpus N`` (WORLD_SIZE already set) the process is one of the ranks and runs directly.

The JSON line also carries
  roofline     -- the dominant kernel (the SwiGLU w12 GEMM, 44 % of all FLOPs): algorithmic FLOPs per launch divided
                  by its average launch duration measured live with HIP events on the launch stream
  stages_ms    -- per-stage device time of ONE extra, untimed tomogram run op by op with HIP events between the launches
                  (ViT GEMMs by kind / attention / LayerNorm / head ...): configs[1] = `vit_ms`, configs[2] = `head_ms`
  cpu_baseline -- the torch-CPU fp32 oracle timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""

from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

D_, H_, W_ = 128, 512, 512
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_TBS = 8.0
N_TOMOGRAMS = 32  # configs[3]: seeds 100..131
TRAFFIC_FILE = ROOT / "profiles" / "dominant_kernel_traffic.json"


def synthetic_head_state_dict(seed: int, device) -> dict:
    """Variance-preserving synthetic init of the reference-layout head (PyTorch's default init gives an all-background
    prediction and a degenerate Dice == 0 -- SURVEY App. E)."""
    import torch

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


def synthetic_labels(device, seed: int):
    """int8 {-1,0,1}: z<16 and z>=112 unlabeled, one ellipsoid of foreground (SURVEY s.8d config 3)."""
    import torch

    z = torch.arange(D_, device=device).view(-1, 1, 1).float()
    y = torch.arange(H_, device=device).view(1, -1, 1).float()
    x = torch.arange(W_, device=device).view(1, 1, -1).float()
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(3, generator=g) * 0.2 + 0.4
    lab = ((((z / D_ - c[0]) / 0.3) ** 2 + ((y / H_ - c[1]) / 0.3) ** 2 + ((x / W_ - c[2]) / 0.3) ** 2) < 1.0).to(torch.int8)
    lab[:16] = -1
    lab[112:] = -1
    return lab.contiguous()


def synthetic_tomogram(device, seed: int):
    """uint8 [128,512,512], uniform 0..254, generated on the device (BASELINE configs[1]/[3]: seeds 100..131)."""
    import torch

    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.rand(D_, H_, W_, generator=g, device=device) * 255).to(torch.uint8)


class DominantKernelTimer:
    """HIP-event timing of one GEMM epilogue kind on the stream the kernels are launched on, through the library's
    measurement hook (the encoder is ONE C call, cvx_vit_encode, so the events are recorded inside it)."""

    def __init__(self, lib_mod, epilogue: int, capacity: int = 4096):
        import ctypes as C

        import torch

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


def stage_breakdown(vit, head, vol, labels, feats_cl, feats_f16, sb: int) -> dict:
    """Device time per stage of ONE tomogram (outside the timed region): the same launch list as the product path, issued op
    by op (``VitEngine._encode_py`` / ``HeadEngine._forward_py``) with a HIP event between consecutive launches on the launch
    stream.  Kernel classes are told apart by the op wrapper that launched them."""
    import torch

    from cryovit_amd import _lib
    from cryovit_amd.engine import ops

    marks = []  # (label, event recorded AFTER the op)
    names = {_lib.EPI_PATCH: "gemm_patch_embed", _lib.EPI_VT: "gemm_v", _lib.EPI_SWIGLU: "gemm_w12", _lib.EPI_BF16_GELU: "gemm_fc1"}

    def ev(label):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((label, e))

    wrapped = {}

    def wrap(name, label_fn):
        orig = getattr(ops, name)
        wrapped[name] = orig

        def f(*a, **k):
            orig(*a, **k)
            ev(label_fn(*a, **k))

        setattr(ops, name, f)

    def gemm_label(epi, a, w, *_, **k):
        if epi in names:
            return names[epi]
        if epi in (_lib.EPI_RESID, _lib.EPI_RESID_HL):
            return "gemm_proj" if w.shape[1] == vit.cfg.dim else "gemm_w3"
        return "gemm_qkv" if w.shape[0] >= 3 * vit.cfg.dim else "gemm_qk"

    hp, wp = H_ // 16, W_ // 16
    try:
        wrap("gemm", gemm_label)
        wrap("layernorm", lambda *a, **k: "layernorm")
        wrap("attention", lambda *a, **k: "attention")
        wrap("attention_qkv", lambda *a, **k: "attention")
        wrap("preprocess_patches", lambda *a, **k: "preprocess")
        wrap("init_tokens", lambda *a, **k: "init_tokens")
        wrap("final_norm_features", lambda *a, **k: "final_norm_features")
        wrap("final_norm_features_hl", lambda *a, **k: "final_norm_features")
        wrap("split_stream", lambda *a, **k: "split_stream")
        wrap("rowstat_finalize", lambda *a, **k: "layernorm")  # what is left of the 80 LayerNorm passes: 79 row-constant launches
        torch.cuda.synchronize()
        ev("_start")
        for d0 in range(0, D_, sb):
            b = min(sb, D_ - d0)
            ws = vit._workspace(b, hp, wp)
            ape = ws["ape"].view(-1)[: ws["ape"].shape[0] * 256].view(-1, 256)
            ops.preprocess_patches(vol[d0 : d0 + b], ape)
            vit._encode_py(b, hp, wp, ape, vit.w["pe_w"], feats_f16, D_, d0, feats_cl[d0 * hp * wp :], None)
    finally:
        for name, orig in wrapped.items():
            setattr(ops, name, orig)
    head.forward(feats_cl, D_, hp, wp, labels=labels, want_probs=True)
    ev("head")
    torch.cuda.synchronize()
    out: dict[str, float] = {}
    for (_, e0), (label, e1) in zip(marks[:-1], marks[1:]):
        out[label] = out.get(label, 0.0) + e0.elapsed_time(e1)
    out["vit_ms"] = sum(v for k, v in out.items() if k != "head")
    out["head_ms"] = out.pop("head")
    return {k: round(v, 3) for k, v in out.items()}


def other_configs(dev, vol) -> dict:
    """Outside the timed region, rank 0 at N = 1: BASELINE configs[4] (sam_features: SAM2.1 Hiera-L image encoder + FPN neck over
    the same tomogram) and the reference's baseline model, UNet3D, on the raw tomogram (SURVEY s.8f N4) -- one untimed warm-up,
    then HIP-event time of one tomogram each; synthetic weights."""
    import torch

    from cryovit_amd.engine.unet3d import UNet3DEngine
    from cryovit_amd.models import UNet3D, load_sam_encoder

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    out = {}
    voxels = vol.numel()
    enc = load_sam_encoder("SAM2", synthetic_seed=2, device=dev, slice_batch=64)
    outs = enc._outs(D_)

    def sam():
        for d0 in range(0, D_, 64):
            enc.engine.encode(vol[d0 : d0 + 64], outs, d0)

    ms = timed(sam)
    out["configs[4] sam_features (SAM2.1 Hiera-L + FPN), one tomogram"] = {"ms": round(ms, 3), "voxels_per_s": voxels / ms * 1e3,
                                                                           "tflops": enc.engine.flops(D_) / ms / 1e9}
    del enc, outs
    torch.cuda.empty_cache()
    model = UNet3D(device="cpu")
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * ((2.0 / max(1, p[0].numel())) ** 0.5 if p.dim() > 1 else 0.1) + (1.0 if p.dim() == 1 and name.endswith(("1.weight", "4.weight")) else 0.0))
    eng = UNet3DEngine(model.state_dict(), dev)
    volf = vol.float() / 255.0
    ms = timed(lambda: eng.forward(volf))
    fl = eng.flops(*vol.shape)
    out["UNet3D baseline (models/unet3d.py) forward, one raw tomogram"] = {
        "ms": round(ms, 3), "voxels_per_s": voxels / ms * 1e3, "algorithmic_tflop": fl / 1e12, "tflops": fl / ms / 1e9,
        "frac_of_mfma_peak": fl / ms / 1e9 / PEAK_BF16_TFLOPS,
        # every activation is written once and read once or twice by the next layer: the forward is HBM-bound, not MFMA-bound
        "roofline_note": "HBM-bound (full-resolution 32-channel fp16 activations: 2.1 GB per layer pair); see profiles/r02_unet_kernel_stats.csv"}
    return out


def head_layer_breakdown(head, feats_cl, labels, D: int, hp: int, wp: int) -> list:
    """configs[2] per layer: the head's launch sequence issued op by op (``HeadEngine._forward_py`` = what ``cvx_head_forward``
    launches) with a HIP event after each op; per entry the algorithmic FLOPs (2 x MACs) and the compulsory HBM bytes (input read
    once + output written once, fp16; GroupNorm reads its input twice) turned into TFLOP/s and TB/s.  A GroupNorm entry is
    three launches (block statistics, finalize, apply)."""
    import torch

    from cryovit_amd.engine import ops

    marks, wrapped = [], {}

    def ev(label, flops, nbytes):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((label, flops, nbytes, e))

    def wrap(name, describe):
        orig = getattr(ops, name)
        wrapped[name] = orig

        def f(*a, **k):
            orig(*a, **k)
            ev(*describe(*a, **k))

        setattr(ops, name, f)

    def d_gemm(epi, a, w, out, bias, *, m, n, cout=0, **k):
        kdim = a.shape[1]
        what = f"ConvT {kdim}->{cout} x4 +GELU" if cout else f"1x1x1 {kdim}->{n} +GELU"
        return what, 2.0 * m * kdim * n, 2.0 * m * (kdim + n)

    def d_conv(x, w, bias, out, zp, *, Cin, D, H, W, dil, cout, act):
        nv = D * H * W
        return f"conv3 {Cin}->{cout} dil {dil} +GELU @ {H}x{W}", 2.0 * nv * 27 * Cin * cout, 2.0 * nv * (Cin + cout)

    def d_gn(x, w, b, out, stats, *, nvox, Cdim, G, eps, act=0):
        return f"GroupNorm {Cdim} ch, {G} groups", 0.0, 2.0 * nvox * Cdim * 3

    def d_out(x, w, bias, logits, probs, labels, dice, *, D, H, W, **k):
        nv, c = D * H * W, head.widths[2]
        return f"conv3 {c}->1 + clip + sigmoid + Dice @ {H}x{W}", 2.0 * nv * 27 * c, nv * (2.0 * c + 4 + 1)

    try:
        wrap("gemm", d_gemm)
        wrap("conv3d", d_conv)
        wrap("groupnorm", d_gn)
        wrap("conv3_out_fused", d_out)
        torch.cuda.synchronize()
        ev("_start", 0.0, 0.0)
        head._forward_py(feats_cl, D, hp, wp, labels=labels, want_probs=True)
    finally:
        for name, orig in wrapped.items():
            setattr(ops, name, orig)
    torch.cuda.synchronize()
    rows = []
    for (_, _, _, e0), (label, fl, nb, e1) in zip(marks[:-1], marks[1:]):
        ms = e0.elapsed_time(e1)
        rows.append({"op": label, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "compulsory_TBps": round(nb / ms / 1e9, 2)})
    return rows


def cpu_baseline() -> dict:
    """torch-CPU fp32 oracle on a bounded sample of the same workload (SURVEY s.8d 'CPU reference timing'): ViT-g/14-reg on
    k = 2 slices of 448x448, the head on [1,1536,128,8,8] (FULL depth, so dilation / padding / GroupNorm behave as in the real
    volume; 1/16 of the in-plane extent), both extrapolated per voxel to 128x512x512, plus the whole T0 configuration
    (64x256x256, ViT-S/14-reg, BASELINE configs[0]) un-extrapolated."""
    import numpy as np
    import torch

    from oracle import dinov2 as o
    from oracle import features as ofe
    from oracle import head as oh
    from oracle import preprocess as opre

    threads = torch.get_num_threads()
    cfg = o.VITG14_REG
    # timing-only weights: one block's tensors shared by all 40 layers (values do not affect the time)
    one = o.init_state_dict(o.VitCfg(cfg.dim, 1, cfg.heads, cfg.ffn, cfg.ffn_hidden), seed=1)
    sd = dict(one)
    for i in range(1, cfg.depth):
        for k, v in one.items():
            if k.startswith("blocks.0."):
                sd[k.replace("blocks.0.", f"blocks.{i}.")] = v
    k_slices = 2
    x = torch.rand(k_slices, 3, 448, 448)
    t0 = time.perf_counter()
    o.forward_features(cfg, sd, x)
    t_vit = time.perf_counter() - t0
    head = oh.CryoVITHead()
    oh.rescaled_init_(head, seed=5)
    h_s = 8  # [1,1536,128,8,8] -> 128 x 128 x 128 output voxels
    feats = torch.randn(1, 1536, D_, h_s, h_s)
    t0 = time.perf_counter()
    with torch.inference_mode():
        head.forward_volume(feats)
    t_head = time.perf_counter() - t0
    s_per_voxel = t_vit / (k_slices * H_ * W_) + t_head / (D_ * (16 * h_s) ** 2)
    # T0, un-extrapolated: 64x256x256 uint8 -> resize -> ViT-S/14-reg -> fp16 [384,64,16,16]
    vol0 = np.random.default_rng(0).integers(0, 256, size=(64, 256, 256), dtype=np.uint8)
    sd_s = o.init_state_dict(o.VITS14_REG, seed=1)
    t0 = time.perf_counter()
    f0 = ofe.dino_features(opre.dino_transform(opre.load_scale(vol0)), o.OracleDino(o.VITS14_REG, sd_s), 64)
    t_t0 = time.perf_counter() - t0
    assert f0.shape == (384, 64, 16, 16)
    return {
        "value": 1.0 / s_per_voxel, "unit": "voxels/s", "cores": threads, "kind": "port",
        "sample": f"oracle fp32, {threads} torch threads on {os.cpu_count()} host cpus: ViT-g/14-reg on {k_slices} slices 448x448 "
                  f"({t_vit:.1f} s, x{D_ // k_slices} to 128 slices) + head on [1,1536,{D_},{h_s},{h_s}] ({t_head:.1f} s, x16 in-plane), "
                  f"extrapolated per voxel to 128x512x512",
        "t0_config": {"workload": "64x256x256 uint8, resize + ViT-S/14-reg features (configs[0]), whole run, not extrapolated",
                      "seconds": round(t_t0, 2), "voxels_per_s": 64 * 256 * 256 / t_t0},
    }


class BoardSampler:
    """Shader clock and board power of THIS rank's GPU over the timed region, from the amdgpu sysfs files (hwmon freq1_input /
    power1_average, the card matched by PCI bus id), sampled every 50 ms by a thread that touches no GPU API.  The MFMA peak the
    roofline is priced at (2.5 PFLOP/s) assumes 2.4 GHz; under this workload the board sits at its power limit and holds a lower
    clock (`profiles/r03_clocks_power.txt`, `profiles/r03_mfma_power.txt`), which the line reports beside the fraction.  Every
    field is None when sysfs is not readable."""

    def __init__(self, dev_index: int) -> None:
        import glob
        import threading

        import torch

        self.hw, self.mhz, self.watts, self.on, self.elapsed_s, self._t0 = None, [], [], False, None, None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for d in sorted(glob.glob("/sys/class/drm/card*/device")):
                if want in os.path.realpath(d):
                    hw = sorted(glob.glob(d + "/hwmon/hwmon*"))
                    self.hw = hw[0] if hw else None
        except (AttributeError, AssertionError, OSError, RuntimeError) as e:  # no device / a missing attribute / an unreadable sysfs only drops the optional fields
            print(f"[bench] board sampler off: {e}", file=sys.stderr)
            self.hw = None
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                v = f.read().strip()
            return float(v) if v else None
        except (OSError, ValueError):
            return None

    def _run(self) -> None:
        while self.on:
            f = self._read(f"{self.hw}/freq1_input")
            w = self._read(f"{self.hw}/power1_average") or self._read(f"{self.hw}/power1_input")
            if f:
                self.mhz.append(f / 1e6)
            if w:
                self.watts.append(w / 1e6)
            time.sleep(0.05)

    def start(self) -> None:
        self._t0 = time.perf_counter()
        if self.hw:
            self.on = True
            self._thread.start()

    def stop(self) -> dict:
        self.elapsed_s = time.perf_counter() - self._t0 if self._t0 else None
        if self.on:
            self.on = False
            self._thread.join()

        def med(xs):
            return sorted(xs)[len(xs) // 2] if xs else None

        mhz = med(self.mhz)
        return {"sclk_mhz_median": mhz, "sclk_mhz_min": min(self.mhz) if self.mhz else None, "power_w_median": med(self.watts),
                "samples": len(self.mhz), "nominal_sclk_mhz": 2400,
                "mfma_peak_at_held_clock_tflops": PEAK_BF16_TFLOPS * mhz / 2400.0 if mhz else None,
                "energy_j": (med(self.watts) * self.elapsed_s) if (self.watts and self.elapsed_s) else None,
                "note": "amdgpu sysfs over the timed region; the board holds its power limit by lowering the clock under MFMA load "
                        "(MFMA-only loops on normal(0,1) bf16 operands sustain 1.77-1.85 PFLOP/s on these boards: profiles/r03_mfma_power.txt)"}


def measured_traffic() -> dict:
    """PMC traffic of the dominant kernel, per launch, from the committed rocprofv3 --pmc passes (collected on the GPU box with
    tools/profile_round.sh, condensed by tools/summarize_profiles.py; `rocprofv3` cannot run inside this process).  The JSON names
    the commit it was measured at: a number from an older kernel is labelled as such, not passed off as this build's."""
    if not TRAFFIC_FILE.exists():
        return {"traffic": None}
    t = json.loads(TRAFFIC_FILE.read_text())
    return {
        "traffic": t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"],
        "traffic_fabric_fetch_bytes": t["fetch_bytes_per_launch"], "traffic_write_bytes": t["write_bytes_per_launch"],
        "algorithmic_bytes": t["algorithmic_bytes_per_launch"],
        "traffic_source": f"profiles/{TRAFFIC_FILE.name} (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 per the gfx950 correction: a "
                          "fabric-side counter that INCLUDES Infinity-Cache hits, i.e. an upper bound on HBM bytes)",
        "traffic_measured_at_commit": t.get("measured_at_commit", "round 2 (before the LayerNorm fold)"), "traffic_kernel": t.get("kernel"),
    }


# ---------------------------------------------------------------------------------------------------------------------------
# launcher


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """Parent of an N-GPU run: no HIP / torch.cuda call happens in this process.  One child per GPU; rank 0 inherits stdout
    (its JSON line is the run's output), the other ranks' stdout goes to stderr.  Returns the exit code for the run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            code = p.poll()
            if code not in (None, 0) and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                deadline = time.time() + 10
                for q in procs:
                    if q.poll() is None:
                        q.terminate()  # the exact PIDs this launcher started
        if deadline is not None and time.time() > deadline:
            for q in procs:
                if q.poll() is None:
                    q.kill()
        time.sleep(0.2)
    for r, p in enumerate(procs):
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slice-batch", type=int, default=128, help="slices per ViT launch sequence (reference default 128)")
    ap.add_argument("--streams", type=int, default=1, help="tomograms in flight per GPU, one per HIP stream")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="cvx_set_option(NAME, VALUE) before the run (A/B experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-stage breakdown tomogram")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launch / barrier / MAX-over-ranks / JSON plumbing over gloo (no GPU, no kernels)")
    return ap.parse_args(argv)


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.dry_run:
            from cryovit_amd.build import build_library

            build_library()  # once, before the ranks start (hipcc needs no GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    if args.dry_run:
        run_dry(args, rank, world, dist)
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs MI355X devices (use --dry-run for the CPU rehearsal of the launcher)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
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
    for kv in args.opt:
        name, value = kv.split("=")
        _lib.set_option(name, int(value))

    cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
    vit = VitEngine(cfg, random_state_dict(cfg, seed=2, device=dev), dev)
    head = HeadEngine(synthetic_head_state_dict(5, dev), dev)
    torch.cuda.empty_cache()

    K = args.steps
    seeds = [100 + (rank * K + i) % N_TOMOGRAMS for i in range(max(K, 1))]
    vols = [synthetic_tomogram(dev, s) for s in seeds]  # resident in HBM before the timed region (33.5 MB each)
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
        i = step_no[0]
        c = ctxs[i % S]
        vol = vols[i % len(vols)]
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
    step_no[0] = 0
    sync_all()
    board = BoardSampler(dev.index or 0)
    board.start()
    kt.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    kt.stop()
    board_stats = board.stop()
    k_ms = kt.mean_ms()
    n_launch = kt.n

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    i, sy, sp = out["dice_sums"].cpu().tolist()
    fg = float((out["probs"] >= 0.5).float().mean())
    voxels = D_ * H_ * W_
    per_rank = None
    if world > 1:  # every rank's own rate (its K tomograms / its own time), gathered on rank 0 for the line
        rates = [None] * world
        dist.all_gather_object(rates, args.steps * voxels / elapsed)
        per_rank = [float(r) for r in rates]

    if rank == 0:
        value = world * args.steps * voxels / elapsed_max
        rows = min(sb, D_) * (hp * wp + 1 + cfg.n_reg)  # valid tokens per launch
        k_flops = 2.0 * rows * cfg.dim * 2 * cfg.ffn_hidden
        achieved = k_flops / (k_ms * 1e-3) / 1e12
        flops_tomo = vit.flops(D_, H_, W_) + head.flops(D_, hp, wp)
        line = {
            "metric": "tomogram voxels/sec (DINO feats + 3D seg fwd)", "value": value, "unit": "voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "one 128x512x512 uint8 tomogram per step per GPU (configs[3]'s seeds 100..131, rank r takes "
                                   "100 + r*steps + i): fused resize + DINOv2 ViT-g/14-reg (40 layers, N=1029 tokens/slice) -> fp16 "
                                   "dino_features [1536,128,32,32] + CryoVIT head [1,1536,128,32,32] -> probs [128,512,512] + masked "
                                   "Dice; synthetic weights",
                       "slice_batch": sb, "streams": args.streams, "parallelism": f"tomogram-sharded x{world}, no collectives"},
            "tflops_end_to_end": flops_tomo * world * args.steps / elapsed_max / 1e12,
            "frac_of_mfma_peak_end_to_end": flops_tomo * args.steps / elapsed_max / 1e12 / PEAK_BF16_TFLOPS,
            "dice": 2 * i / (sy + sp + 1e-3), "pred_fg_fraction": fg,
            "roofline": {"bound": "mfma", "kernel": "k_gemm256p_nreg<EpiSwiGLUT<LN>, FULL>: w12 GEMM 1536->8192, LayerNorm folded into the epilogue + fused SiLU gate", "achieved": achieved,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, **measured_traffic(),
                         "launches_timed": n_launch, "avg_launch_ms": k_ms, "flops_per_launch": k_flops},
            "board": board_stats,
        }
        if board_stats.get("energy_j"):  # board energy over the timed region (median power x its wall time), this rank's GPU
            line["board"]["joules_per_tomogram"] = board_stats["energy_j"] / args.steps
            line["board"]["voxels_per_joule"] = args.steps * voxels / board_stats["energy_j"]
        if per_rank is not None:
            line["per_rank_voxels_per_s"] = per_rank
        if not args.no_stages:
            st = stage_breakdown(vit, head, vols[0], labels, feats_cl, feats_f16, sb)
            line["stages_ms"] = st
            # algorithmic TFLOP/s of the five stages that hold 99 % of the FLOPs (same accounting as VitEngine.flops, SURVEY s.8d)
            nt_, C_, Hd_, L_ = hp * wp + 1 + cfg.n_reg, cfg.dim, cfg.ffn_hidden, cfg.depth
            toks = D_ * nt_ * L_
            st_fl = {"gemm_qkv": 2.0 * toks * 3 * C_ * C_, "attention": 4.0 * toks * nt_ * C_, "gemm_proj": 2.0 * toks * C_ * C_,
                     "gemm_w12": 2.0 * toks * 2 * C_ * Hd_, "gemm_w3": 2.0 * toks * Hd_ * C_}
            line["stages_tflops"] = {k: round(v / st[k] / 1e9, 1) for k, v in st_fl.items() if st.get(k)}
            head_bytes = 402.65e6 + 134.2e6 + 33.6e6 + 16.8e6  # SURVEY s.8d: compulsory HBM bytes of configs[2]
            line["configs"] = {
                "configs[1] ViT-g features, one tomogram": {"ms": st["vit_ms"], "voxels_per_s": voxels / st["vit_ms"] * 1e3,
                                                            "tflops": vit.flops(D_, H_, W_) / st["vit_ms"] / 1e9},
                "configs[2] head fwd + Dice": {"ms": st["head_ms"], "voxels_per_s": voxels / st["head_ms"] * 1e3,
                                               "tflops": head.flops(D_, hp, wp) / st["head_ms"] / 1e9,
                                               "compulsory_GBps": head_bytes / st["head_ms"] / 1e6, "hbm_peak_GBps": PEAK_HBM_TBS * 1e3,
                                               "layers": head_layer_breakdown(head, feats_cl, labels, D_, hp, wp)},
            }
            if world == 1:
                try:
                    line["configs"].update(other_configs(dev, vols[0]))
                except Exception as e:  # the extra configurations must never cost the headline line
                    line["configs"]["other_configs_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_dry(args, rank: int, world: int, dist) -> None:
    """The multi-rank plumbing without a GPU: gloo rendezvous on 127.0.0.1, per-rank shard of the 32 tomogram seeds, barrier
    on both sides of a timed region, MAX over ranks, one JSON line from rank 0."""
    import torch

    if world > 1:
        dist.init_process_group("gloo")
    K = args.steps
    seeds = [100 + (rank * K + i) % N_TOMOGRAMS for i in range(K)]

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    acc = 0
    for s in seeds:  # stand-in "step": deterministic host work per tomogram seed
        acc += int(torch.randint(0, 256, (64, 64), generator=torch.Generator().manual_seed(s)).sum())
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    all_seeds = [None] * world
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather_object(all_seeds, seeds)
    else:
        all_seeds = [seeds]
    if rank == 0:
        voxels = D_ * H_ * W_
        print(json.dumps({"metric": "tomogram voxels/sec (DINO feats + 3D seg fwd)", "value": world * K * voxels / float(t.item()),
                          "unit": "voxels/s", "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": float(t.item()) / K * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "dry-run (no GPU work)",
                          "config": {"workload": "launcher rehearsal", "seeds_per_rank": all_seeds}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
