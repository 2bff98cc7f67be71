"""``CryoVIT`` -- the ``_target_`` of ``configs/model/cryovit.yaml`` (mirror of
``/root/reference/src/cryovit/models/cryovit.py:10-49``; same constructor kwargs as ``BaseModel``,
``models/base_model.py:20-56``, same ``state_dict`` keys, same ``forward`` / ``forward_volume`` contracts).

The module is a parameter container with reference-compatible names (``layers.0``, ``layers.{2..5}.layers.{0,1,3,5}``,
``output_layer.{0,2}``) so ``load_state_dict(torch.load("weights.pt"))`` works unchanged; the arithmetic runs in
``cryovit_amd.engine.head.HeadEngine`` (HIP kernels).  Inference only: there is no backward.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from cryovit_amd.engine import ops
from cryovit_amd.engine.head import REF_WIDTHS, HeadEngine
from cryovit_amd.models.base import EvalProtocol
from cryovit_amd.models.metrics import dice_from_sums


class _Params(nn.Module):
    """weight/bias holder named like the torch layer it stands for"""

    def __init__(self, w_shape, b_shape):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(*w_shape), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(*b_shape), requires_grad=False)


class _Slot(nn.Module):
    """parameter-less position in a Sequential (GELU in the reference)"""


class SynthesisBlock(nn.Module):
    def __init__(self, c1: int, c2: int, c3: int, d1: int, d2: int) -> None:
        super().__init__()
        self.dilations = (d1, d2)
        self.layers = nn.Sequential(
            _Params((c1,), (c1,)),                       # 0: GroupNorm(max(8, c1//8), c1, eps=1e-3)
            _Params((c2, c1, 3, 3, 3), (c2,)), _Slot(),   # 1: Conv3d dil (d1,1,1); 2: GELU
            _Params((c2, c2, 3, 3, 3), (c2,)), _Slot(),   # 3: Conv3d dil (d2,1,1); 4: GELU
            _Params((c2, c3, 1, 2, 2), (c3,)), _Slot(),   # 5: ConvTranspose3d (1,2,2); 6: GELU
        )


class CryoVIT(EvalProtocol, nn.Module):
    def __init__(self, input_key: str = "dino_features", lr: float = 1e-3, weight_decay: float = 1e-3, losses=None, metrics=None,
                 name: str = "CryoVIT", custom_kwargs=None, device="cuda:0", **kwargs) -> None:
        super().__init__()
        self.input_key, self.lr, self.weight_decay, self.name = input_key, lr, weight_decay, name
        self.loss_fns = dict(losses or {})
        self.metric_fns = dict(metrics or {})
        c_in, blocks, c_tail = REF_WIDTHS
        self.layers = nn.Sequential(_Params((blocks[0][0], c_in, 1, 1, 1), (blocks[0][0],)), _Slot(),
                                    *[SynthesisBlock(*b) for b in blocks])
        self.output_layer = nn.Sequential(_Params((c_tail, c_tail, 3, 3, 3), (c_tail,)), _Slot(), _Params((1, c_tail, 3, 3, 3), (1,)))
        self._device = torch.device(device)
        self._engine: HeadEngine | None = None

    # any weight change invalidates the packed copy
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        own = {k: v for k, v in state_dict.items() if not k.startswith(("metric_fns.", "loss_fns."))}
        r = super().load_state_dict(own, strict=strict, **kw)
        self._engine = None
        return r

    def _check_channels(self, C: int) -> None:
        c_in = self.layers[0].weight.shape[1]
        if C != c_in:
            raise ValueError(f"CryoVIT expects {c_in}-channel input features ('{self.input_key}'), got {C} channels: run the feature "
                             "stage first (cryovit features / training.dino_features) or pass an encoder to run_inference")

    def engine(self) -> HeadEngine:
        if self._engine is None:
            self._engine = HeadEngine(self.state_dict(), self._device)
        return self._engine

    @torch.inference_mode()
    def forward_volume(self, x: Tensor) -> Tensor:
        """[B,1536,D,h,w] -> logits [B,1,D,16h,16w] clipped to [-5,5] (cryovit.py:36-40)."""
        outs = []
        for xb in x:
            C, D, h, w = xb.shape
            self._check_channels(C)
            cl = torch.zeros(ops.alloc_rows(D * h * w), C, dtype=torch.float16, device=self._device)
            src = xb.to(self._device)
            if src.dtype == torch.float16 and src.is_contiguous():
                ops.features_to_channels_last(src, cl)
            else:
                cl[: D * h * w] = src.permute(1, 2, 3, 0).reshape(-1, C).to(torch.float16)
            outs.append(self.engine().forward(cl, D, h, w, want_logits=True, want_probs=False)["logits"])
        return torch.stack(outs).unsqueeze(1)

    @torch.inference_mode()
    def forward(self, batch) -> Tensor:
        """batch.tomo_batch [B,D,C,h,w] -> probabilities [B,D,H,W] (cryovit.py:42-49)."""
        x = batch.tomo_batch.permute(0, 2, 1, 3, 4)
        return torch.sigmoid(self.forward_volume(x).squeeze(1))

    @torch.inference_mode()
    def predict_mask(self, batch, threshold: float = 0.5) -> list[Tensor]:
        """``predict_step`` + ``PredictionWriter``'s threshold (base_model.py:243-273, callbacks.py:100-102): per tomogram
        the uint8 segmentation ``forward(batch) >= threshold`` [D,H,W], compared inside the head's last kernel."""
        outs = []
        for xb in batch.tomo_batch:  # [D,C,h,w]
            D, C, h, w = xb.shape
            self._check_channels(C)
            cl = torch.zeros(ops.alloc_rows(D * h * w), C, dtype=torch.float16, device=self._device)
            cl[: D * h * w] = xb.to(self._device).permute(0, 2, 3, 1).reshape(-1, C).to(torch.float16)
            outs.append(self.engine().forward(cl, D, h, w, want_probs=False, mask_threshold=threshold)["mask"])
        return outs

    @torch.inference_mode()
    def predict_with_dice(self, feats_cl: Tensor, D: int, h: int, w: int, labels: Tensor | None):
        """Fused inference used by the end-to-end runner: channels-last fp16 features straight from the encoder ->
        probabilities and (with labels) the masked Dice of ``_masked_predict`` + ``DiceMetric``."""
        thr = next((m.thresh for m in getattr(self, "metric_fns", {}).values() if hasattr(m, "thresh")), None)  # DiceMetric(threshold)
        out = self.engine().forward(feats_cl, D, h, w, labels=None if labels is None else labels.to(self._device, torch.int8).contiguous(),
                                    dice_threshold=thr)
        dice = None
        if labels is not None:
            i, sy, sp = out["dice_sums"].cpu().tolist()
            dice = dice_from_sums(i, sy, sp)
        return out["probs"], dice
