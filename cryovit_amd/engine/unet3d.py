"""UNet3D baseline on the HIP kernels: weight packing + the launch sequence.

Replaces the ``torch.nn`` calls behind ``UNet3D.forward_volume`` / ``forward``
(``/root/reference/src/cryovit/models/unet3d.py:49-96``), ``AnalysisBlock`` (l.113-148), ``SynthesisBlock`` (l.151-191) and
``LinearProjection`` (l.194-216).  Activations are channels-last FP16 volumes ``[D][H][W][C]`` like the CryoVIT head's (the
reference trains and evaluates under fp16 autocast), the single input channel is padded to 8:

  Conv3d k=3 "same"           -> cvx_conv3d_f16 (dil 1; the few-channel layers run in the z-marching LDS-ring kernel)
  InstanceNorm3d + GELU       -> cvx_groupnorm_act_f16 with G = C (statistics without atomics: reproducible)
  Conv3d k=2 stride 2 (pool)  -> cvx_conv2s2_f16 (implicit GEMM over the 2x2x2 voxels an output voxel covers)
  ConvTranspose3d k=2 s=2     -> cvx_gemm_bf16 with N = 8*C_out and the 3-D pixel-shuffle epilogue
  torch.cat + Linear          -> cvx_concat_channels_f16 + cvx_gemm_bf16 (K = C_up + C_skip)
  Conv3d k=1 + clip + sigmoid -> cvx_pointwise_out_f16
"""

from __future__ import annotations

import math

import torch

from cryovit_amd._lib import EPI_BF16, EPI_CONVT
from cryovit_amd.engine import ops
from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1, _pad2
from cryovit_amd.engine.ops import round_up

PAD = 16  # unet3d.py:46: max(16, 2 ** 3)


class UNet3DEngine:
    def __init__(self, state_dict: dict, device="cuda:0"):
        if not torch.cuda.is_available():
            raise ops._lib.CvxError("UNet3DEngine needs a HIP device (no CPU fallback)")
        ops._lib.load()
        self.device = ops.norm_device(device)
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if not k.startswith(("metric_fns.", "loss_fns."))}
        up = lambda t: t.contiguous().to(self.device)  # noqa: E731
        self.sd_shapes = {k: tuple(v.shape) for k, v in sd.items()}

        def conv3(prefix):  # Conv3d k3: first layer's single input channel padded to 8
            w = sd[prefix + ".weight"]
            if w.shape[1] % 8:
                wp = torch.zeros(w.shape[0], round_up(w.shape[1], 8), 3, 3, 3)
                wp[:, : w.shape[1]] = w
                w = wp
            return {"w": up(_conv3_weight(w)), "b": up(_pad1(sd[prefix + ".bias"], _npad(w.shape[0]))), "cin": w.shape[1], "cout": w.shape[0]}

        def norm(prefix):
            return {"w": up(sd[prefix + ".weight"]), "b": up(sd[prefix + ".bias"])}

        def pool(prefix):  # [O][C][2][2][2] -> [n_pad][8*C], k = ((iz*2+iy)*2+ix)*C + c
            w = sd[prefix + ".weight"]
            O, Cc = w.shape[:2]
            return {"w": up(_pad2(w.permute(0, 2, 3, 4, 1).reshape(O, 8 * Cc), _npad(O), 8 * Cc)), "b": up(_pad1(sd[prefix + ".bias"], _npad(O))),
                    "cin": Cc, "cout": O}

        def convt(prefix):  # [C][O][2][2][2] -> rows n = ((iz*2+i)*2+j)*O + o, cols c
            w = sd[prefix + ".weight"]
            Cc, O = w.shape[:2]
            wg = w.permute(2, 3, 4, 1, 0).reshape(8 * O, Cc)
            return {"w": up(_pad2(wg, _npad(8 * O), round_up(Cc, 64))), "b": up(_pad1(sd[prefix + ".bias"].repeat(8), _npad(8 * O))), "cin": Cc, "cout": O}

        def linear(prefix):
            w = sd[prefix + ".weight"]
            return {"w": up(_pad2(w, _npad(w.shape[0]), round_up(w.shape[1], 64))), "b": up(_pad1(sd[prefix + ".bias"], _npad(w.shape[0]))),
                    "cin": w.shape[1], "cout": w.shape[0]}

        self.analysis = []
        for i in range(3):
            p = f"analysis_layers.{i}."
            self.analysis.append({"c1": conv3(p + "layers.0"), "n1": norm(p + "layers.1"), "c2": conv3(p + "layers.3"), "n2": norm(p + "layers.4"),
                                  "pool": pool(p + "pool.0"), "np": norm(p + "pool.1")})
        self.bottom = {"c1": conv3("bottom_layer.0"), "n1": norm("bottom_layer.1"), "c2": conv3("bottom_layer.3"), "n2": norm("bottom_layer.4")}
        self.synthesis = []
        for i in range(3):
            p = f"synthesis_layers.{i}."
            self.synthesis.append({"up": convt(p + "upconv.0"), "nu": norm(p + "upconv.1"), "lin": linear(p + "layers.0.proj"),
                                   "n1": norm(p + "layers.1"), "c": conv3(p + "layers.3"), "n2": norm(p + "layers.4")})
        ow = sd["output_layer.weight"]
        self.out_w, self.out_b, self.out_c = up(ow.reshape(-1)), float(sd["output_layer.bias"][0]), ow.shape[1]
        if self.out_c not in (8, 16, 32, 64):
            raise ValueError("UNet3D: the output layer takes 8, 16, 32 or 64 channels")
        gmax = max(max(b["c2"]["cout"] for b in self.analysis), self.bottom["c1"]["cout"])
        if gmax > ops._lib.GN_MAX_GROUPS:
            raise ValueError(f"UNet3D: InstanceNorm over {gmax} channels exceeds CVX_GN_MAX_GROUPS")
        self.stats = torch.zeros(ops.gn_stats_size(gmax), dtype=torch.float32, device=self.device)
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self._bufs: dict = {}
        # torch.cat of a synthesis block: True = dense tensors + cvx_concat_channels_f16 (default); False = both InstanceNorm passes write
        # straight into their column block of the concatenated buffer (cvx_groupnorm_act_strided_f16).  Measured on one board, A/B in one
        # process (tools/bench_unet.py --ab): 15.1 ms with the copy kernel, 15.9 ms in place -- half-row writes at a doubled row pitch and
        # the second (dense) copy of the skip tensor cost more than the 1-ms copy kernel they remove.
        self.concat_copy = True

    def _buf(self, name: str, rows: int, ch: int) -> torch.Tensor:
        key = (name, rows, ch)
        if key not in self._bufs:
            self._bufs[key] = torch.zeros(ops.alloc_rows(rows) * ch + 4096, dtype=torch.float16, device=self.device)
        return self._bufs[key]

    # ---- layer helpers -------------------------------------------------------------------------------------------------------
    def _conv(self, x, L, name, D, H, W):
        out = self._buf(name, D * H * W, L["cout"])
        ops.conv3d(x, L["w"], L["b"], out, self.zero_page, Cin=L["cin"], D=D, H=H, W=W, dil=1, cout=L["cout"], act=0)
        return out

    def _norm_gelu(self, x, N, name, nv, Cc):
        out = self._buf(name, nv, Cc)
        ops.groupnorm(x, N["w"], N["b"], out, self.stats, nvox=nv, Cdim=Cc, G=Cc, eps=1e-3, act=1)
        return out

    @torch.inference_mode()
    def forward_volume(self, vol: torch.Tensor, want_logits: bool = False):
        """vol: [D, H, W] (any float dtype or uint8 already scaled by the loader) on the engine's device, every axis a multiple of
        16 -> probabilities fp32 [D, H, W] (and the clipped logits)."""
        D, H, W = vol.shape
        if D % PAD or H % PAD or W % PAD:
            raise ValueError("UNet3DEngine.forward_volume: pad the volume to multiples of 16 first (UNet3D.forward does)")
        x = self._buf("in", D * H * W, 8)
        xin = x[: D * H * W * 8].view(D, H, W, 8)
        xin.zero_()
        xin[..., 0] = vol.to(torch.float16)
        skips = []
        d, h, w = D, H, W
        nlev = len(self.analysis)
        for i, B in enumerate(self.analysis):
            nv = d * h * w
            t = self._conv(x, B["c1"], f"a{i}c1", d, h, w)
            t = self._norm_gelu(t, B["n1"], f"a{i}n1", nv, B["c1"]["cout"])
            t = self._conv(t, B["c2"], f"a{i}c2", d, h, w)
            # the skip tensor is written TWICE by its InstanceNorm + GELU pass: dense (input of the pooling convolution) and straight
            # into its column block of the synthesis block's concatenated input (torch.cat of unet3d.py:64 without a copy kernel)
            cs, cu = B["c2"]["cout"], self.synthesis[nlev - 1 - i]["up"]["cout"]
            cat = self._buf(f"s{nlev - 1 - i}cat", nv, cu + cs)
            skip = self._buf(f"a{i}skip", nv, cs)
            if self.concat_copy:  # round 2's form (A/B: UNet3DEngine.concat_copy = True): dense skip, cvx_concat_channels_f16 later
                ops.groupnorm(t, B["n2"]["w"], B["n2"]["b"], skip, self.stats, nvox=nv, Cdim=cs, G=cs, eps=1e-3, act=1)
            else:
                ops.groupnorm_into(t, B["n2"]["w"], B["n2"]["b"], cat, cu, cu + cs, self.stats, nvox=nv, Cdim=cs, G=cs, eps=1e-3, act=1, out2=skip)
            skips.append((cat, cs, skip))
            p = self._buf(f"a{i}pool", nv // 8, B["pool"]["cout"])
            ops.conv2s2(skip, B["pool"]["w"], B["pool"]["b"], p, self.zero_page, Cin=B["pool"]["cin"], D=d, H=h, W=w, cout=B["pool"]["cout"], act=0)
            d, h, w = d // 2, h // 2, w // 2
            x = self._norm_gelu(p, B["np"], f"a{i}pn", d * h * w, B["pool"]["cout"])
        nv = d * h * w
        t = self._conv(x, self.bottom["c1"], "b1", d, h, w)
        t = self._norm_gelu(t, self.bottom["n1"], "b1n", nv, self.bottom["c1"]["cout"])
        t = self._conv(t, self.bottom["c2"], "b2", d, h, w)
        x = self._norm_gelu(t, self.bottom["n2"], "b2n", nv, self.bottom["c2"]["cout"])
        for i, S in enumerate(self.synthesis):
            U = S["up"]
            nv = d * h * w
            upb = self._buf(f"s{i}up", nv * 8, U["cout"])
            a2 = torch.as_strided(x, (ops.alloc_rows(nv), U["cin"]), (U["cin"], 1))
            ops.gemm(EPI_CONVT, a2, U["w"], upb, U["b"], m=nv, n=8 * U["cout"], H=h, W=w, cout=U["cout"], act=0, ldc=U["cout"], convt_up_z=1)
            d, h, w = 2 * d, 2 * h, 2 * w
            nv = d * h * w
            cat, cs, skip = skips.pop()  # (its skip half was filled by the analysis path)
            if self.concat_copy:
                un = self._norm_gelu(upb, S["nu"], f"s{i}upn", nv, U["cout"])
                ops.concat_channels(un, skip, cat, nvox=nv, Ca=U["cout"], Cb=cs)
            else:
                ops.groupnorm_into(upb, S["nu"]["w"], S["nu"]["b"], cat, 0, U["cout"] + cs, self.stats, nvox=nv, Cdim=U["cout"], G=U["cout"],
                                   eps=1e-3, act=1)
            Lp = S["lin"]
            lin = self._buf(f"s{i}lin", nv, Lp["cout"])
            a2 = torch.as_strided(cat, (ops.alloc_rows(nv), Lp["cin"]), (Lp["cin"], 1))
            ops.gemm(EPI_BF16, a2, Lp["w"], lin, Lp["b"], m=nv, n=Lp["cout"], ldc=Lp["cout"])
            t = self._norm_gelu(lin, S["n1"], f"s{i}ln", nv, Lp["cout"])
            t = self._conv(t, S["c"], f"s{i}c", d, h, w)
            x = self._norm_gelu(t, S["n2"], f"s{i}cn", nv, S["c"]["cout"])
        nv = d * h * w
        probs = torch.empty(D, H, W, dtype=torch.float32, device=self.device)
        logits = torch.empty(D, H, W, dtype=torch.float32, device=self.device) if want_logits else None
        ops.pointwise_out(x, self.out_w, self.out_b, logits, probs, nvox=nv, Cdim=self.out_c)
        return (probs, logits) if want_logits else probs

    def flops(self, D: int, H: int, W: int) -> float:
        """Algorithmic FLOPs (2 * MACs of the convolutions, transposed convolutions and 1x1x1 layers; norms and GELUs excluded -- the
        accounting of SURVEY.md s.8d) of one forward over a [D, H, W] volume (axes padded to multiples of 16)."""
        D, H, W = (PAD * math.ceil(v / PAD) for v in (D, H, W))
        nv, f = D * H * W, 0.0
        conv = lambda L, n: 2.0 * 27 * L["cin"] * L["cout"] * n  # noqa: E731
        for B in self.analysis:
            f += conv(B["c1"], nv) + conv(B["c2"], nv) + 2.0 * 8 * B["pool"]["cin"] * B["pool"]["cout"] * (nv // 8)
            nv //= 8
        f += conv(self.bottom["c1"], nv) + conv(self.bottom["c2"], nv)
        for S in self.synthesis:
            f += 2.0 * S["up"]["cin"] * 8 * S["up"]["cout"] * nv
            nv *= 8
            f += 2.0 * S["lin"]["cin"] * S["lin"]["cout"] * nv + conv(S["c"], nv)
        return f + 2.0 * self.out_c * nv

    @torch.inference_mode()
    def forward(self, vol: torch.Tensor) -> torch.Tensor:
        """unet3d.py:73-96 for one tomogram [D, H, W]: zero-pad every axis to a multiple of 16, run, crop."""
        D, H, W = vol.shape
        Dp, Hp, Wp = (PAD * math.ceil(v / PAD) for v in (D, H, W))
        v = vol.to(self.device)
        if (Dp, Hp, Wp) != (D, H, W):
            vp = torch.zeros(Dp, Hp, Wp, dtype=v.dtype, device=self.device)
            vp[:D, :H, :W] = v
            v = vp
        return self.forward_volume(v)[:D, :H, :W]
