"""CryoVIT segmentation head on the HIP kernels: weight packing + the launch sequence.

Replaces the ``torch.nn`` calls behind ``CryoVIT.forward_volume`` / ``forward``
(``/root/reference/src/cryovit/models/cryovit.py:13-49``) and ``SynthesisBlock`` (``:52-83``), plus the masked
Dice reductions (``models/base_model.py:99-110``, ``models/metrics.py:30-43``).  Layer algebra: SURVEY.md App. B.

Activations are channels-last FP16 volumes ``[D][h][w][C]`` (the ViT's token-major output IS this layout; fp16 because the
reference runs the head under fp16 autocast and bf16 storage costs 10x the logit error -- DESIGN.md s.2), so
every convolution is a GEMM over contiguous channel vectors:
  Conv3d k=1            -> cvx_gemm_bf16 (GELU epilogue)
  GroupNorm             -> cvx_groupnorm_f16 (stats + apply; zero padding must see normalised zeros)
  Conv3d 3x3x3 dilated  -> cvx_conv3d_f16 (implicit GEMM, LDS-DMA gather of the 27 taps, GELU epilogue)
  ConvTranspose (1,2,2) -> cvx_gemm_bf16 with N = 4*C_out and a pixel-shuffle epilogue (GELU)
  Conv3d 8->1 + clip + sigmoid + Dice -> cvx_conv3_out_fused
"""

from __future__ import annotations

import torch

import ctypes as C

from cryovit_amd._lib import EPI_BF16_GELU, EPI_CONVT, HeadBlock, HeadDesc, HeadWs, check
from cryovit_amd.engine import ops
from cryovit_amd.engine.ops import round_up

# (c_in, ((c1, c2, c3, d1, d2) x4), c_tail) -- cryovit.py:18-34
REF_WIDTHS = (1536, ((1024, 192, 128, 32, 24), (128, 64, 32, 16, 12), (32, 32, 32, 8, 4), (32, 16, 8, 2, 1)), 8)


def widths_from_state_dict(sd: dict):
    """Recover (c_in, blocks, c_tail) from reference-layout keys (dilations are fixed by the architecture)."""
    dil = ((32, 24), (16, 12), (8, 4), (2, 1))
    c_in = sd["layers.0.weight"].shape[1]
    blocks = []
    for bi in range(4):
        p = f"layers.{bi + 2}.layers."
        c2, c1 = sd[p + "1.weight"].shape[:2]
        c3 = sd[p + "5.weight"].shape[1]
        blocks.append((c1, c2, c3, *dil[bi]))
    return c_in, tuple(blocks), sd["output_layer.0.weight"].shape[0]


def _pad2(w: torch.Tensor, n_pad: int, k_pad: int) -> torch.Tensor:
    out = torch.zeros(n_pad, k_pad, dtype=torch.float16)  # GEMM-side weights of the head are stored in fp16
    out[: w.shape[0], : w.shape[1]] = w.clamp(-65504.0, 65504.0).to(torch.float16)
    return out


def _pad1(v: torch.Tensor, n_pad: int) -> torch.Tensor:
    out = torch.zeros(n_pad, dtype=torch.float32)
    out[: v.numel()] = v.float().reshape(-1)
    return out


def _npad(n: int) -> int:
    """Smallest tile-compatible padding of an output width (tile N sizes are 16/32/64/128)."""
    for t in (16, 32, 64):
        if n <= t:
            return t
    return round_up(n, 64) if n <= 192 else round_up(n, 128)


def _conv3_weight(w: torch.Tensor) -> torch.Tensor:
    """[O][C][3][3][3] -> fp16 [n_pad][k_pad], k = ((kz*3+ky)*3+kx)*C + c."""
    O, Cin = w.shape[:2]
    return _pad2(w.permute(0, 2, 3, 4, 1).reshape(O, 27 * Cin), _npad(O), round_up(27 * Cin, 64))


class HeadEngine:
    def __init__(self, state_dict: dict, device="cuda:0"):
        if not torch.cuda.is_available():
            raise ops._lib.CvxError("HeadEngine needs a HIP device (no CPU fallback)")
        ops._lib.load()
        self.device = ops.norm_device(device)
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.widths = widths_from_state_dict(sd)
        c_in, blocks, c_tail = self.widths
        if c_tail != 8:
            raise ValueError("the fused output kernel is specialised for 8 tail channels (cryovit.py:30-33)")
        up = lambda t: t.contiguous().to(self.device)  # noqa: E731
        w = {}
        c0 = blocks[0][0]
        w["proj_w"] = up(_pad2(sd["layers.0.weight"].reshape(c0, c_in), _npad(c0), round_up(c_in, 64)))
        w["proj_b"] = up(_pad1(sd["layers.0.bias"], _npad(c0)))
        self.blocks = []
        for bi, (c1, c2, c3, d1, d2) in enumerate(blocks):
            p = f"layers.{bi + 2}.layers."
            wt = sd[p + "5.weight"]  # [c2][c3][1][2][2] -> rows n = (i*2+j)*c3 + o, cols c
            wt_g = wt[:, :, 0].permute(2, 3, 1, 0).reshape(4 * c3, c2)
            self.blocks.append({
                "c": (c1, c2, c3, d1, d2),
                "G": max(8, c1 // 8),
                "gn_w": up(sd[p + "0.weight"]), "gn_b": up(sd[p + "0.bias"]),
                "c1_w": up(_conv3_weight(sd[p + "1.weight"])), "c1_b": up(_pad1(sd[p + "1.bias"], _npad(c2))),
                "c2_w": up(_conv3_weight(sd[p + "3.weight"])), "c2_b": up(_pad1(sd[p + "3.bias"], _npad(c2))),
                "ct_w": up(_pad2(wt_g, _npad(4 * c3), round_up(c2, 64))),
                "ct_b": up(_pad1(sd[p + "5.bias"].repeat(4), _npad(4 * c3))),
            })
        w["o0_w"] = up(_conv3_weight(sd["output_layer.0.weight"]))
        w["o0_b"] = up(_pad1(sd["output_layer.0.bias"], _npad(c_tail)))
        w["o2_w"] = up(sd["output_layer.2.weight"][0].permute(1, 2, 3, 0).reshape(27, c_tail).contiguous())
        self.o2_b = float(sd["output_layer.2.bias"][0])
        self.w = w
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self.stats = torch.zeros(ops.gn_stats_size(128), dtype=torch.float32, device=self.device)
        self._ws = {}
        self._desc = self._make_desc()

    def _make_desc(self) -> HeadDesc:
        """Weight table for ``cvx_head_forward`` (host struct of device pointers; the tensors stay owned by ``self``)."""
        c_in, blocks, c_tail = self.widths
        W = self.w
        d = HeadDesc()
        d.c_in, d.c0, d.c_tail, d.n_blocks = c_in, blocks[0][0], c_tail, len(blocks)
        d.proj_w, d.proj_b, d.proj_npad, d.proj_kpad = W["proj_w"].data_ptr(), W["proj_b"].data_ptr(), *W["proj_w"].shape
        self._cblocks = (HeadBlock * len(blocks))()
        for cb, blk in zip(self._cblocks, self.blocks):
            cb.c1, cb.c2, cb.c3, cb.d1, cb.d2 = blk["c"]
            cb.groups = blk["G"]
            cb.gn_w, cb.gn_b = blk["gn_w"].data_ptr(), blk["gn_b"].data_ptr()
            cb.conv1_w, cb.conv1_b, cb.conv1_npad, cb.conv1_kpad = blk["c1_w"].data_ptr(), blk["c1_b"].data_ptr(), *blk["c1_w"].shape
            cb.conv2_w, cb.conv2_b, cb.conv2_npad, cb.conv2_kpad = blk["c2_w"].data_ptr(), blk["c2_b"].data_ptr(), *blk["c2_w"].shape
            cb.convt_w, cb.convt_b, cb.convt_npad, cb.convt_kpad = blk["ct_w"].data_ptr(), blk["ct_b"].data_ptr(), *blk["ct_w"].shape
        d.blocks = C.cast(self._cblocks, C.POINTER(HeadBlock))
        d.out0_w, d.out0_b, d.out0_npad, d.out0_kpad = W["o0_w"].data_ptr(), W["o0_b"].data_ptr(), *W["o0_w"].shape
        d.out2_w, d.out2_b, d.zero_page = W["o2_w"].data_ptr(), self.o2_b, self.zero_page.data_ptr()
        return d

    def _make_ws(self, D: int, h: int, w_: int) -> HeadWs:
        key = ("ws", D, h, w_)
        if key not in self._ws:
            _, blocks, c_tail = self.widths
            ws = HeadWs()
            ws.act0 = self._buf("a", D * h * w_, blocks[0][0]).data_ptr()
            H_, W_ = h, w_
            for bi, (c1, c2, c3, _, _) in enumerate(blocks):
                nv = D * H_ * W_
                ws.gn[bi] = self._buf(f"g{bi}", nv, c1).data_ptr()
                ws.t1[bi] = self._buf(f"t1_{bi}", nv, c2).data_ptr()
                ws.t2[bi] = self._buf(f"t2_{bi}", nv, c2).data_ptr()
                ws.up[bi] = self._buf(f"u{bi}", nv * 4, c3).data_ptr()
                H_, W_ = 2 * H_, 2 * W_
            ws.mid = self._buf("mid", D * H_ * W_, c_tail).data_ptr()
            ws.gn_stats = self.stats.data_ptr()
            ws.dice_scratch = ops.dice_scratch(self.device).data_ptr()
            self._ws[key] = ws
        return self._ws[key]

    def clone_for_stream(self) -> "HeadEngine":
        """Second launch context over the same packed weights with private activation buffers (one per HIP stream)."""
        other = object.__new__(HeadEngine)
        other.__dict__.update(self.__dict__)
        other._ws = {}
        other.stats = torch.zeros_like(self.stats)
        other._desc = other._make_desc()
        return other

    def _buf(self, name: str, rows: int, ch: int) -> torch.Tensor:
        """fp16 [rows + slack][ch] channels-last buffer (slack rows let K-padded GEMM loads run past the end)."""
        key = (name, rows, ch)
        if key not in self._ws:
            self._ws[key] = torch.zeros(ops.alloc_rows(rows) * ch + 4096, dtype=torch.float16, device=self.device)
        return self._ws[key]

    @staticmethod
    def _threshold(mask_threshold, dice_threshold) -> float:
        """The fused output kernel compares against ONE threshold: DiceMetric's (metrics.py:38) and the uint8 mask's (callbacks.py:100)."""
        if mask_threshold is not None and dice_threshold is not None and float(mask_threshold) != float(dice_threshold):
            raise ops._lib.CvxError("head: the mask threshold and the Dice threshold of one call must agree (one fused comparison)")
        t = mask_threshold if mask_threshold is not None else dice_threshold
        return 0.5 if t is None else float(t)

    def forward(self, feats_cl: torch.Tensor, D: int, h: int, w_: int, labels=None, want_logits=False, want_probs=True,
                mask_threshold: float | None = None, dice_threshold: float | None = None):
        """feats_cl: fp16 channels-last features [D*h*w (+slack rows), C_in].  Returns dict with
        ``probs`` / ``logits`` fp32 [D, 16h, 16w], ``dice_sums`` (fp32[3] device tensor) when labels are given and
        ``mask`` uint8 [D, 16h, 16w] = (probs >= mask_threshold) when a threshold is given.  ONE C call: cvx_head_forward."""
        c_in, blocks, _ = self.widths
        if ops._dev_check(feats_cl, labels) != self.device:
            raise ops._lib.CvxError(f"head: inputs live on {feats_cl.device}, the engine on {self.device}")
        if feats_cl.dtype != torch.float16 or feats_cl.numel() < ops.alloc_rows(D * h * w_) * c_in:
            raise ops._lib.CvxError("head: features must be fp16 [rup(D*h*w,256)+256 rows][c_in]")
        up = 2 ** len(blocks)
        shape = (D, h * up, w_ * up)
        dev = self.device
        logits = torch.empty(shape, dtype=torch.float32, device=dev) if want_logits else None
        probs = torch.empty(shape, dtype=torch.float32, device=dev) if want_probs else None
        mask = torch.empty(shape, dtype=torch.uint8, device=dev) if mask_threshold is not None else None
        dice = torch.zeros(3, dtype=torch.float32, device=dev) if labels is not None else None
        if labels is not None and (labels.dtype != torch.int8 or tuple(labels.shape) != shape):
            raise ops._lib.CvxError(f"head: labels must be int8 {shape}")
        ops.call(dev, "cvx_head_forward", ops._lib.load().cvx_head_forward, C.byref(self._desc), C.byref(self._make_ws(D, h, w_)),
                 feats_cl.data_ptr(), D, h, w_, ops._p(logits), ops._p(probs), ops._p(labels), ops._p(dice), ops._p(mask),
                 self._threshold(mask_threshold, dice_threshold))
        return {"logits": logits, "probs": probs, "dice_sums": dice, "mask": mask}

    def _forward_py(self, feats_cl: torch.Tensor, D: int, h: int, w_: int, labels=None, want_logits=False, want_probs=True,
                    mask_threshold: float | None = None):
        """The same launch sequence spelled out with the op-level entry points (what cvx_head_forward does; kept for A/B tests)."""
        c_in, blocks, c_tail = self.widths
        W = self.w
        nvox = D * h * w_
        c0 = blocks[0][0]
        act = self._buf("a", nvox, c0)
        ops.gemm(EPI_BF16_GELU, feats_cl.reshape(-1, c_in), W["proj_w"], act, W["proj_b"], m=nvox, n=c0, ldc=c0)
        H_, W_ = h, w_
        for bi, blk in enumerate(self.blocks):
            c1, c2, c3, d1, d2 = blk["c"]
            nv = D * H_ * W_
            gn = self._buf(f"g{bi}", nv, c1)
            ops.groupnorm(act, blk["gn_w"], blk["gn_b"], gn, self.stats, nvox=nv, Cdim=c1, G=blk["G"], eps=1e-3)
            t1 = self._buf(f"t1_{bi}", nv, c2)
            ops.conv3d(gn, blk["c1_w"], blk["c1_b"], t1, self.zero_page, Cin=c1, D=D, H=H_, W=W_, dil=d1, cout=c2, act=1)
            t2 = self._buf(f"t2_{bi}", nv, c2)
            ops.conv3d(t1, blk["c2_w"], blk["c2_b"], t2, self.zero_page, Cin=c2, D=D, H=H_, W=W_, dil=d2, cout=c2, act=1)
            up_ = self._buf(f"u{bi}", nv * 4, c3)
            a2 = torch.as_strided(t2, (ops.alloc_rows(nv), c2), (c2, 1))
            ops.gemm(EPI_CONVT, a2, blk["ct_w"], up_, blk["ct_b"], m=nv, n=4 * c3, H=H_, W=W_, cout=c3, act=1, ldc=c3)
            act, H_, W_ = up_, 2 * H_, 2 * W_
        nv = D * H_ * W_
        mid = self._buf("mid", nv, c_tail)
        ops.conv3d(act, W["o0_w"], W["o0_b"], mid, self.zero_page, Cin=c_tail, D=D, H=H_, W=W_, dil=1, cout=c_tail, act=1)
        out = {}
        logits = torch.empty(D, H_, W_, dtype=torch.float32, device=self.device) if want_logits else None
        probs = torch.empty(D, H_, W_, dtype=torch.float32, device=self.device) if want_probs else None
        dice = None
        if labels is not None:
            dice = torch.zeros(3, dtype=torch.float32, device=self.device)
        mask = torch.empty(D, H_, W_, dtype=torch.uint8, device=self.device) if mask_threshold is not None else None
        ops.conv3_out_fused(mid, W["o2_w"], self.o2_b, logits, probs, labels, dice, D=D, H=H_, W=W_, mask=mask,
                            mask_threshold=0.5 if mask_threshold is None else mask_threshold)
        out["logits"], out["probs"], out["dice_sums"], out["mask"] = logits, probs, dice, mask
        return out

    def flops(self, D: int, h: int, w_: int) -> float:
        c_in, blocks, c_tail = self.widths
        f = 2.0 * D * h * w_ * c_in * blocks[0][0]
        H_, W_ = h, w_
        for c1, c2, c3, _, _ in blocks:
            nv = D * H_ * W_
            f += 2.0 * nv * 27 * (c1 * c2 + c2 * c2) + 2.0 * nv * 4 * c2 * c3
            H_, W_ = 2 * H_, 2 * W_
        nv = D * H_ * W_
        return f + 2.0 * nv * 27 * (c_tail * c_tail + c_tail)
