"""AdamW on the device: the optimizer ``BaseModel.configure_optimizers`` builds
(``/root/reference/src/cryovit/models/base_model.py:57-63``: ``torch.optim.AdamW(params, lr, weight_decay)``), as ONE fused
launch per step over a flat copy-free view of all parameters instead of torch's five element-wise passes per tensor.

Parameters, gradients and both moments live in four flat fp32 buffers; every ``nn.Parameter`` handed in is re-pointed at its
slice of the parameter buffer and its ``.grad`` at the matching slice of the gradient buffer, so autograd accumulates
straight into the buffer the kernel reads (``zero_grad`` is one memset).  28 B of HBM traffic per parameter and step.
"""

from __future__ import annotations

import torch

from cryovit_amd.engine import ops


class AdamW:
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("AdamW: no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise ops._lib.CvxError("AdamW: parameters must live on a HIP device (no CPU fallback)")
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("AdamW: fp32 parameters on one device")
        self.lr, (self.beta1, self.beta2), self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4  # slices stay 16-B aligned
        self.n, self.offsets = n, offs
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat_p[o : o + p.numel()].view(p.shape)
                view.copy_(p)
                p.data = view
                p.grad = self.flat_g[o : o + p.numel()].view(p.shape)

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat_g.zero_()  # (the .grad views stay attached: autograd accumulates in place)

    def _check_attached(self) -> None:
        """Parameters and gradients must still be the slices handed out at construction: ``model.zero_grad()`` (set_to_none),
        ``model.to()`` / ``half()`` or a ``load_state_dict(assign=True)`` replace them, and the kernel would then step stale
        buffers.  A dropped ``.grad`` (None) is re-attached -- the buffer slice is zero after ``zero_grad`` either way;
        anything else is an error."""
        esz = self.flat_p.element_size()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if p.data_ptr() != self.flat_p.data_ptr() + o * esz or p.dtype != torch.float32:
                raise RuntimeError(f"AdamW: parameter {i} no longer lives in the optimizer's flat buffer (moved or cast "
                                   f"after the optimizer was built); rebuild the optimizer")
            if p.grad is None:
                self.flat_g[o : o + p.numel()].zero_()
                p.grad = self.flat_g[o : o + p.numel()].view(p.shape)
            elif p.grad.data_ptr() != self.flat_g.data_ptr() + o * esz:
                raise RuntimeError(f"AdamW: the gradient of parameter {i} was replaced (use this optimizer's zero_grad, or "
                                   f"zero_grad(set_to_none=True) which is re-attached); rebuild the optimizer")

    @torch.no_grad()
    def step(self) -> None:
        """One fused launch.  The optimizer is standalone (there is no HIP backward yet, SURVEY s.8f N4): engines that cache
        packed weights (``model._engine``) are not invalidated here -- drop them after a step (``model._engine = None``)."""
        self._check_attached()
        self.step_count += 1
        ops.adamw_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, lr=self.lr, beta1=self.beta1, beta2=self.beta2,
                       eps=self.eps, weight_decay=self.weight_decay, step=self.step_count)

    def state_dict(self) -> dict:
        return {"n": self.n, "offsets": list(self.offsets), "shapes": [tuple(p.shape) for p in self.params],
                "step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.lr, "betas": (self.beta1, self.beta2), "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd: dict) -> None:
        if "n" in sd and (int(sd["n"]) != self.n or list(sd["offsets"]) != list(self.offsets)
                          or [tuple(x) for x in sd["shapes"]] != [tuple(p.shape) for p in self.params]):
            raise ValueError("AdamW.load_state_dict: the flat parameter layout of the checkpoint differs from this optimizer's")
        if sd["exp_avg"].numel() != self.n or sd["exp_avg_sq"].numel() != self.n:
            raise ValueError("AdamW.load_state_dict: moment buffers of the wrong size")
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
