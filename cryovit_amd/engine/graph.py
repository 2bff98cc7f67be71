"""HIP-graph replay of the per-tomogram launch sequence (fused resize + ViT + head + Dice).

One tomogram is ~470 kernel launches (40 layers x 11 + the head); every entry point of ``libcryovit_hip.so`` launches on the
caller's stream, allocates nothing and never synchronises, so the whole sequence can be captured once per tomogram shape
and replayed with a single ``hipGraphLaunch``.  At the benchmark size the kernels are long (the launch stream stays ahead of
the GPU and the gain is < 1 %); for small volumes (BASELINE configs[0]: 64x256x256, ViT-S) the path is launch-bound and
replay removes the per-launch gaps.  ``torch.cuda.CUDAGraph`` is the capture / replay plumbing (hipGraph underneath).
"""

from __future__ import annotations

import math

import torch

from cryovit_amd.engine import ops


class GraphedTomogram:
    """Static-shape pipeline for tomograms ``[D,H,W]`` (uint8 or float32): ``run(vol, labels)`` copies the inputs into the
    captured buffers, replays the graph and returns the captured outputs (valid until the next ``run``)."""

    def __init__(self, vit, head, D: int, H: int, W: int, *, dtype=torch.uint8, slice_batch: int = 128, with_labels: bool = True,
                 want_f16: bool = True, mask_threshold: float | None = None):
        self.vit, self.head = vit, head
        dev = vit.device
        self.shape = (D, H, W)
        hp, wp = math.ceil(H / 16), math.ceil(W / 16)
        C = vit.cfg.dim
        self.vol = torch.zeros(D, H, W, dtype=dtype, device=dev)
        up = 2 ** len(head.widths[1])
        self.labels = torch.full((D, hp * up, wp * up), -1, dtype=torch.int8, device=dev) if with_labels else None
        self.f16 = torch.zeros(C, D, hp, wp, dtype=torch.float16, device=dev) if want_f16 else None
        self.cl = torch.zeros(ops.alloc_rows(D * hp * wp), C, dtype=torch.float16, device=dev)
        self._args = (D, hp, wp, slice_batch, mask_threshold)
        self.out = None
        # warm-up on a side stream (allocates every workspace, sets kernel attributes), then capture the same calls
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._launch()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._launch()

    def _launch(self):
        D, hp, wp, sb, thr = self._args
        for d0 in range(0, D, sb):
            b = min(sb, D - d0)
            self.vit.features(self.vol[d0 : d0 + b], feats_f16=self.f16, d_total=D, d0=d0, feats_cl=self.cl[d0 * hp * wp :])
        return self.head.forward(self.cl, D, hp, wp, labels=self.labels, want_probs=True, mask_threshold=thr)

    @torch.inference_mode()
    def run(self, vol: torch.Tensor, labels: torch.Tensor | None = None) -> dict:
        if tuple(vol.shape) != self.shape or vol.dtype != self.vol.dtype:
            raise ValueError(f"graph was captured for {self.vol.dtype} {self.shape}, got {vol.dtype} {tuple(vol.shape)}")
        self.vol.copy_(vol, non_blocking=True)
        if self.labels is not None:
            if labels is None:
                self.labels.fill_(-1)
            else:
                self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        out = dict(self.out)
        out["feats_f16"] = self.f16
        return out
