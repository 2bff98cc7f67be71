"""``UNet3D`` -- the ``_target_`` of ``configs/model/unet3d.yaml`` (mirror of ``/root/reference/src/cryovit/models/unet3d.py:12-216``;
same constructor kwargs as ``BaseModel``, same ``state_dict`` keys, same ``forward`` contract).  A parameter container with
reference-compatible names; the arithmetic runs in ``cryovit_amd.engine.unet3d.UNet3DEngine`` (HIP kernels).  Forward only."""

from __future__ import annotations

import torch
from torch import Tensor, nn

from cryovit_amd.engine.unet3d import UNet3DEngine
from cryovit_amd.models.base import EvalProtocol
from cryovit_amd.models.cryovit import _Params, _Slot

REF_WIDTHS = ((16, 64, 256), 384)


class _Proj(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.proj = _Params((cout, cin), (cout,))


def _k3(cin, cout):
    return _Params((cout, cin, 3, 3, 3), (cout,))


def _norm(c):
    return _Params((c,), (c,))


class AnalysisBlock(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.pool = nn.Sequential(_Params((cout, cout, 2, 2, 2), (cout,)), _norm(cout), _Slot())
        self.layers = nn.Sequential(_k3(cin, cout), _norm(cout), _Slot(), _k3(cout, cout), _norm(cout), _Slot())


class SynthesisBlock(nn.Module):
    def __init__(self, cin: int, cskip: int, cout: int) -> None:
        super().__init__()
        self.upconv = nn.Sequential(_Params((cin, cout, 2, 2, 2), (cout,)), _norm(cout), _Slot())
        self.layers = nn.Sequential(_Proj(cout + cskip, cout), _norm(cout), _Slot(), _k3(cout, cout), _norm(cout), _Slot())


class UNet3D(EvalProtocol, nn.Module):
    def __init__(self, input_key: str = "data", lr: float = 1e-3, weight_decay: float = 1e-3, losses=None, metrics=None, name: str = "UNet3D",
                 custom_kwargs=None, device="cuda:0", widths=REF_WIDTHS, **kwargs) -> None:
        super().__init__()
        self.input_key, self.lr, self.weight_decay, self.name = input_key, lr, weight_decay, name
        self.loss_fns, self.metric_fns = dict(losses or {}), dict(metrics or {})
        (a1, a2, a3), cb = widths
        self.bottom_layer = nn.Sequential(_k3(a3, cb), _norm(cb), _Slot(), _k3(cb, a3), _norm(a3), _Slot())
        self.analysis_layers = nn.ModuleList([AnalysisBlock(1, a1), AnalysisBlock(a1, a2), AnalysisBlock(a2, a3)])
        self.synthesis_layers = nn.ModuleList([SynthesisBlock(a3, a3, a2), SynthesisBlock(a2, a2, a1), SynthesisBlock(a1, a1, a1)])
        self.output_layer = _Params((1, a1, 1, 1, 1), (1,))
        self.PAD = 16
        self._device = torch.device(device)
        self._engine: UNet3DEngine | None = None
        for m in self.modules():  # norm weights start at 1 like torch's affine InstanceNorm
            if isinstance(m, _Params) and m.weight.dim() == 1:
                nn.init.ones_(m.weight)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        own = {k: v for k, v in state_dict.items() if not k.startswith(("metric_fns.", "loss_fns."))}
        r = super().load_state_dict(own, strict=strict, **kw)
        self._engine = None
        return r

    def engine(self) -> UNet3DEngine:
        if self._engine is None:
            self._engine = UNet3DEngine(self.state_dict(), self._device)
        return self._engine

    @torch.inference_mode()
    def forward(self, batch) -> Tensor:
        """batch.tomo_batch [B, D, C=1, H, W] -> probabilities [B, D, H, W] (unet3d.py:73-96)."""
        x = batch.tomo_batch
        if x.shape[2] != 1:
            raise ValueError(f"UNet3D expects the raw single-channel volume ('{self.input_key}'), got {x.shape[2]} channels")
        return torch.stack([self.engine().forward(xb[:, 0]) for xb in x])

    @torch.inference_mode()
    def predict_mask(self, batch, threshold: float = 0.5) -> list[Tensor]:
        """``predict_step`` + ``PredictionWriter``'s threshold (base_model.py:243-273, callbacks.py:100-102): per tomogram the uint8
        segmentation ``forward(batch) >= threshold`` [D, H, W] (device tensors)."""
        return [(self.engine().forward(xb[:, 0]) >= threshold).to(torch.uint8) for xb in batch.tomo_batch]
