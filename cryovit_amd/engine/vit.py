"""DINOv2-with-registers encoder on the HIP kernels: weight packing + the per-batch launch sequence.

Replaces what the reference delegates to the third-party hub model at
``/root/reference/src/cryovit/run/dino_features.py:58`` (``forward_features``) together with the host-side
resize of ``/root/reference/src/cryovit/datasets/vit_dataset.py:117-123`` and the reshape/permute/half of
``run/dino_features.py:59-61``.  Algorithm: SURVEY.md App. A.

Data layout in HBM (b slices, C channels, NT = hp*wp+1+n_reg tokens, NTP = NT rounded up to 8):
  xh, xl bf16 [b*NTP (+pad)][C]   residual stream as a bf16 PAIR x = hi + lo (row = slice*NTP + token; rows NT..NTP-1 are
                                  finite padding).  hi = bf16(x) IS the A operand of the qk / V^T / FFN-in GEMMs: the LayerNorm
                                  gain is folded into their weights, its normalisation into their epilogues
  part   fp32 [C/64][..][2]       partial row sums written by the residual GEMMs' epilogues
  rowst  fp32 [..][2]             (rstd, -mean*rstd) per row
  (fold_ln=False keeps the round-2 plan: x fp32 [..][C] + xn bf16 [..][C] = LayerNorm output, separate LayerNorm launches)
  qk     bf16 [..][2C]            Q (pre-scaled by head_dim^-0.5 * log2 e) | K, token-major
  vt     bf16 [b][heads][64][KP]  V transposed per head (KP = NT rounded up to 64), written by the V GEMM epilogue
  ao     bf16 [..][C]             attention output
  hid    bf16 [..][Hd_pad]        gated / activated FFN hidden
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from cryovit_amd._lib import EPI_BF16, EPI_BF16_GELU, EPI_PATCH, EPI_RESID, EPI_RESID_HL, EPI_SWIGLU, EPI_VT
from cryovit_amd.engine import ops
from cryovit_amd.engine.ops import alloc_rows, round_up


@dataclass(frozen=True)
class VitConfig:
    dim: int
    depth: int
    heads: int
    ffn: str  # "swiglu" | "mlp"
    ffn_hidden: int
    n_reg: int = 4
    patch: int = 14
    pos_grid: int = 37
    ln_eps: float = 1e-6

    def __post_init__(self):
        if self.dim // self.heads != 64 or self.dim % self.heads:
            raise ValueError("the HIP attention kernel is specialised for head_dim 64 (every DINOv2 variant)")
        if self.patch != 14:
            raise ValueError("patch size must be 14")


def _swiglu_hidden(dim: int) -> int:
    return (int(dim * 4 * 2 / 3) + 7) // 8 * 8


VIT_CONFIGS = {
    # the model the reference hard-codes (run/dino_features.py:25-28)
    "dinov2_vitg14_reg": VitConfig(1536, 40, 24, "swiglu", _swiglu_hidden(1536)),
    "dinov2_vitl14_reg": VitConfig(1024, 24, 16, "mlp", 4096),
    "dinov2_vitb14_reg": VitConfig(768, 12, 12, "mlp", 3072),
    "dinov2_vits14_reg": VitConfig(384, 12, 6, "mlp", 1536),
}


def _bf16_padded(w: torch.Tensor, n_pad: int, k_pad: int) -> torch.Tensor:
    out = torch.zeros(n_pad, k_pad, dtype=torch.bfloat16, device=w.device)
    out[: w.shape[0], : w.shape[1]] = w.to(torch.bfloat16)
    return out


def _f32_padded(v: torch.Tensor, n_pad: int) -> torch.Tensor:
    out = torch.zeros(n_pad, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.float().reshape(-1)
    return out


def random_state_dict(cfg: "VitConfig", seed: int, device="cpu", std: float = 0.02) -> dict:
    """Synthetic weights in the upstream key layout, generated on `device` (no checkpoint is
    reachable offline).  N(0, std) linears, LayerNorm / LayerScale ~ 1 + N(0, std)."""
    g = torch.Generator(device=device).manual_seed(seed)
    C = cfg.dim

    def n(*shape, mean=0.0):
        return torch.empty(*shape, device=device).normal_(mean, std, generator=g)

    sd = {
        "cls_token": n(1, 1, C), "pos_embed": n(1, 1 + cfg.pos_grid**2, C), "register_tokens": n(1, cfg.n_reg, C),
        "mask_token": torch.zeros(1, C, device=device),
        "patch_embed.proj.weight": n(C, 3, 14, 14), "patch_embed.proj.bias": n(C),
        "norm.weight": n(C, mean=1.0), "norm.bias": n(C),
    }
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = n(C, mean=1.0), n(C)
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = n(3 * C, C), n(3 * C)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = n(C, C), n(C)
        sd[p + "ls1.gamma"], sd[p + "ls2.gamma"] = n(C, mean=1.0), n(C, mean=1.0)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = n(C, mean=1.0), n(C)
        if cfg.ffn == "swiglu":
            sd[p + "mlp.w12.weight"], sd[p + "mlp.w12.bias"] = n(2 * cfg.ffn_hidden, C), n(2 * cfg.ffn_hidden)
            sd[p + "mlp.w3.weight"], sd[p + "mlp.w3.bias"] = n(C, cfg.ffn_hidden), n(C)
        else:
            sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = n(cfg.ffn_hidden, C), n(cfg.ffn_hidden)
            sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = n(C, cfg.ffn_hidden), n(C)
    return sd


def interpolate_pos_embed(cfg: VitConfig, pos_embed: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
    """[1,1+G*G,C] -> fp32 [1+hp*wp, C]; bicubic + antialias (App. A-2).  Weight-only, once per (hp, wp)."""
    G = cfg.pos_grid
    pe = pos_embed.float()
    if hp == G and wp == G:
        return pe[0].contiguous()
    patch = pe[:, 1:].reshape(1, G, G, cfg.dim).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(hp, wp), mode="bicubic", align_corners=False, antialias=True)
    patch = patch.permute(0, 2, 3, 1).reshape(hp * wp, cfg.dim)
    return torch.cat([pe[0, :1], patch], dim=0).contiguous()


class VitEngine:
    """Holds packed device weights and per-shape workspaces; ``features()`` runs one slice batch."""

    def __init__(self, cfg: VitConfig, state_dict: dict, device="cuda:0", fold_ln: bool = True, merge_qkv: bool | None = None):
        if not torch.cuda.is_available():
            raise ops._lib.CvxError("VitEngine needs a HIP device (no CPU fallback)")
        ops._lib.load()
        self.cfg, self.device = cfg, ops.norm_device(device)
        self.fold_ln = bool(fold_ln)  # False: the round-2 plan (fp32 stream + LayerNorm launches), kept for A/B runs
        # one qkv GEMM (V row-major beside Q and K, transposed by the attention kernel's LDS reads) instead of a qk GEMM + a V^T GEMM;
        # needs the folded path and a row pitch of 3C that is a multiple of 64 elements
        self.merge_qkv = self.fold_ln and (3 * cfg.dim) % 64 == 0 if merge_qkv is None else bool(merge_qkv and self.fold_ln)
        self._pos_src = state_dict["pos_embed"].detach().float().cpu()
        self._cls = state_dict["cls_token"].detach().float().cpu().reshape(-1)
        self._ws = {}
        self._pos = {}
        self._pack(state_dict)

    # ---- weight packing (host, once) -----------------------------------------------------------------------------
    def _pack(self, sd: dict) -> None:
        cfg, dev = self.cfg, self.device
        C = cfg.dim
        n128 = round_up(C, 128)

        def up(t):
            return t.contiguous().to(dev)

        g = lambda k: sd[k].detach().float()  # noqa: E731  (packing runs on whatever device the checkpoint is on)
        w = {}
        # patch embed: the three input channels are identical copies (vit_dataset.py:117-118) -> sum the kernel
        # over channels (exact in real arithmetic), K = 196 padded to 256
        pe = g("patch_embed.proj.weight").sum(dim=1).reshape(C, 196)
        w["pe_w"] = up(_bf16_padded(pe, n128, 256))
        w["pe_b"] = up(_f32_padded(g("patch_embed.proj.bias"), n128))
        # general 3-channel kernel for the protocol entry point forward_features(x): K = 588 padded to 640
        w["pe3_w"] = up(_bf16_padded(g("patch_embed.proj.weight").reshape(C, 588), n128, 640))
        w["reg"] = up(g("register_tokens").reshape(cfg.n_reg, C))
        w["norm_w"], w["norm_b"] = up(g("norm.weight")), up(g("norm.bias"))
        self.hid_pad = round_up(cfg.ffn_hidden, 64)
        blocks = []
        # head_dim^-0.5 * log2(e) folded into the Q rows in fp32 BEFORE the bf16 rounding: the attention kernel works in log2
        # units (p = exp2(s - m), no per-score multiply; oracle/dinov2.py::forward_features_bf16_storage mirrors the rounding point)
        scale = 64**-0.5 * math.log2(math.e)
        fold = self.fold_ln

        def ln_linear(wm, bias, gamma, beta, n_pad):
            """(packed bf16 weight, fp32 bias tensor) of a linear layer that consumes LayerNorm(gamma, beta).  fold_ln: the gain goes
            into the weight, W' = bf16(W * gamma) (ONE rounding), and the bias tensor becomes [2, n_pad] = b' = b + W beta (fp64
            matvec) | cs[n] = sum_k W'[n][k] (of the ROUNDED weight: it multiplies -mean*rstd against the same products the MFMA
            accumulates)."""
            if not fold:
                return _bf16_padded(wm, n_pad, wm.shape[1]), _f32_padded(bias, n_pad)
            wq = _bf16_padded(wm * gamma[None, :], n_pad, wm.shape[1])
            bc = torch.zeros(2, n_pad, dtype=torch.float32, device=wm.device)
            bc[0, : wm.shape[0]] = (bias.double() + wm.double() @ beta.double()).float()
            bc[1] = wq.double().sum(dim=1).float()
            return wq, bc

        for i in range(cfg.depth):
            p = f"blocks.{i}."
            qkv_w, qkv_b = g(p + "attn.qkv.weight").clone(), g(p + "attn.qkv.bias").clone()
            qkv_w[:C] *= scale
            qkv_b[:C] *= scale
            g1, b1, g2, b2 = g(p + "norm1.weight"), g(p + "norm1.bias"), g(p + "norm2.weight"), g(p + "norm2.bias")
            if self.merge_qkv:
                qk_w, qk_b = ln_linear(qkv_w, qkv_b, g1, b1, round_up(3 * C, 128))  # all 3C rows: q (scaled) | k | v
                v_w, v_b = qk_w[:0], qk_b[..., :0]  # (not read)
            else:
                qk_w, qk_b = ln_linear(qkv_w[: 2 * C], qkv_b[: 2 * C], g1, b1, round_up(2 * C, 128))
                v_w, v_b = ln_linear(qkv_w[2 * C :], qkv_b[2 * C :], g1, b1, n128)
            blk = {
                "ln1_w": up(g1), "ln1_b": up(b1),
                "qk_w": up(qk_w), "qk_b": up(qk_b),
                "v_w": up(v_w), "v_b": up(v_b),
                "proj_w": up(_bf16_padded(g(p + "attn.proj.weight"), n128, C)),
                "proj_b": up(_f32_padded(g(p + "attn.proj.bias"), n128)),
                "ls1": up(_f32_padded(g(p + "ls1.gamma"), n128)),
                "ln2_w": up(g2), "ln2_b": up(b2),
                "ls2": up(_f32_padded(g(p + "ls2.gamma"), n128)),
            }
            Hd, Hp = cfg.ffn_hidden, self.hid_pad
            if cfg.ffn == "swiglu":
                w12, b12 = g(p + "mlp.w12.weight"), g(p + "mlp.w12.bias")
                a_w, b_w = torch.zeros(Hp, C, device=w12.device), torch.zeros(Hp, C, device=w12.device)
                a_b, b_b = torch.zeros(Hp, device=w12.device), torch.zeros(Hp, device=w12.device)
                a_w[:Hd], b_w[:Hd], a_b[:Hd], b_b[:Hd] = w12[:Hd], w12[Hd:], b12[:Hd], b12[Hd:]
                # interleave in blocks of 8 so one lane's 16 accumulators are 8 gates + their 8 values (EpiSwiGLU)
                inter_w = torch.stack([a_w.reshape(-1, 8, C), b_w.reshape(-1, 8, C)], dim=1).reshape(2 * Hp, C)
                inter_b = torch.stack([a_b.reshape(-1, 8), b_b.reshape(-1, 8)], dim=1).reshape(2 * Hp)
                f1_w, f1_b = ln_linear(inter_w, inter_b, g2, b2, 2 * Hp)
                blk["ffn1_w"], blk["ffn1_b"] = up(f1_w), up(f1_b)
                blk["ffn2_w"] = up(_bf16_padded(g(p + "mlp.w3.weight"), n128, Hp))
                blk["ffn2_b"] = up(_f32_padded(g(p + "mlp.w3.bias"), n128))
            else:
                f1_w, f1_b = ln_linear(g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"), g2, b2, Hp)
                blk["ffn1_w"], blk["ffn1_b"] = up(f1_w), up(f1_b)
                blk["ffn2_w"] = up(_bf16_padded(g(p + "mlp.fc2.weight"), n128, Hp))
                blk["ffn2_b"] = up(_f32_padded(g(p + "mlp.fc2.bias"), n128))
            blocks.append(blk)
        self.w, self.blocks = w, blocks

    def clone_for_stream(self) -> "VitEngine":
        """A second launch context over the SAME packed weights with its own workspaces -- one per HIP stream, so two slice
        batches (or two volumes) can be in flight at once and fill each other's tails / epilogue bubbles."""
        other = object.__new__(VitEngine)
        other.__dict__.update(self.__dict__)
        other._ws = {}
        return other

    def weight_bytes(self) -> int:
        n = sum(t.numel() * t.element_size() for t in self.w.values())
        return n + sum(t.numel() * t.element_size() for b in self.blocks for t in b.values())

    # ---- per-shape state ------------------------------------------------------------------------------------------
    def geometry(self, H: int, W: int):
        """raw slice size -> (hp, wp, tokens, padded tokens, padded keys)"""
        return self._geometry(math.ceil(H / 16), math.ceil(W / 16))  # (ceil16(H)*14/16)/14

    def _geometry(self, hp: int, wp: int):
        nt = hp * wp + 1 + self.cfg.n_reg
        return hp, wp, nt, round_up(nt, 8), round_up(nt, 64)

    def _pos_for(self, hp: int, wp: int):
        key = (hp, wp)
        if key not in self._pos:
            pos = interpolate_pos_embed(self.cfg, self._pos_src, hp, wp)
            cls_pos0 = self._cls + pos[0]
            self._pos[key] = (pos.to(self.device), cls_pos0.to(self.device))
        return self._pos[key]

    def _workspace(self, b: int, hp: int, wp: int):
        key = (b, hp, wp)
        if key in self._ws:
            self._ws[key] = self._ws.pop(key)  # most recently used last
            return self._ws[key]
        cfg, dev = self.cfg, self.device
        _, _, nt, ntp, kp = self._geometry(hp, wp)
        C = cfg.dim
        rows = alloc_rows(b * ntp)
        z = lambda *s, dt=torch.bfloat16: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        ws = {
            "ape": z(alloc_rows(b * hp * wp), 640),
            "qk": z(rows, 3 * C if self.merge_qkv else 2 * C),
            "vt": z(b, cfg.heads, 64, kp) if not self.merge_qkv else z(8),
            "ao": z(rows, C),
            "hid": z(rows, self.hid_pad),
        }
        if self.fold_ln:
            ws["xh"], ws["xl"] = z(rows, C), z(rows, C)
            ws["part"] = z(C // 64, rows, 2, dt=torch.float32)
            ws["rowstat"] = z(rows, 2, dt=torch.float32)
            # fp32 staging of the embedded tokens (init + patch-embed GEMM -> split): dead before the first FFN, so it lives in
            # the hidden-activation buffer when that is large enough (every DINOv2 variant: hid_pad >= 2.6 C)
            ws["x"] = (ws["hid"].view(-1)[: rows * C * 2].view(torch.float32).view(rows, C) if self.hid_pad >= 2 * C
                       else z(rows, C, dt=torch.float32))
        else:
            ws["x"], ws["xn"] = z(rows, C, dt=torch.float32), z(rows, C)
        # keep the two most recent shapes resident: a tomogram whose depth is not a multiple of the slice batch alternates
        # between the full batch and the remainder, and must not re-allocate and re-zero ~4 GB twice per tomogram
        while len(self._ws) >= 2:
            self._ws.pop(next(iter(self._ws)))
        self._ws[key] = ws
        return ws

    # ---- forward ----------------------------------------------------------------------------------------------------
    def features(self, slices: torch.Tensor, feats_f16=None, d_total: int = 0, d0: int = 0, feats_cl=None) -> None:
        """slices: device uint8/float32 [b,H,W] (raw tomogram slices).  Writes
        feats_f16 fp16 [C, d_total, hp, wp] at depth d0..d0+b-1 and/or feats_cl fp16 [b, hp, wp, C]."""
        b, H, W = slices.shape
        hp, wp = math.ceil(H / 16), math.ceil(W / 16)
        ws = self._workspace(b, hp, wp)
        ape = ws["ape"].view(-1)[: ws["ape"].shape[0] * 256].view(-1, 256)  # K = 196 -> 256 view of the patch matrix
        ops.preprocess_patches(slices, ape)
        self._encode(b, hp, wp, ape, self.w["pe_w"], feats_f16, d_total, d0, feats_cl, None)

    @torch.inference_mode()
    def forward_features(self, x: torch.Tensor) -> dict:
        """The reference's encoder protocol (run/dino_features.py:58): x fp32 [b,3,H',W'] already resized,
        H', W' multiples of 14 -> {"x_norm_patchtokens": fp32 [b, (H'/14)*(W'/14), C]} on the device."""
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 14 or x.shape[3] % 14:
            raise ValueError(f"forward_features expects [b,3,H,W] with H,W multiples of 14, got {tuple(x.shape)}")
        x = x.to(self.device, torch.float32).contiguous()
        b, hp, wp = x.shape[0], x.shape[2] // 14, x.shape[3] // 14
        ws = self._workspace(b, hp, wp)
        ops.im2col_patches(x, ws["ape"])
        tokens = torch.empty(b, hp * wp, self.cfg.dim, dtype=torch.float32, device=self.device)
        self._encode(b, hp, wp, ws["ape"], self.w["pe3_w"], None, 0, 0, None, tokens)
        return {"x_norm_patchtokens": tokens}

    def _c_desc(self):
        """ctypes view of the packed weights for cvx_vit_encode (built once; keeps the arrays alive)."""
        if getattr(self, "_cdesc", None) is None:
            from cryovit_amd._lib import VitDesc, VitLayer

            layers = (VitLayer * self.cfg.depth)()
            for i, blk in enumerate(self.blocks):
                for name in ("ln1_w", "ln1_b", "qk_w", "qk_b", "v_w", "v_b", "proj_w", "proj_b", "ls1", "ln2_w", "ln2_b", "ffn1_w",
                             "ffn1_b", "ffn2_w", "ffn2_b", "ls2"):
                    setattr(layers[i], name, blk[name].data_ptr())
            d = VitDesc(dim=self.cfg.dim, depth=self.cfg.depth, heads=self.cfg.heads, n_reg=self.cfg.n_reg,
                        ffn_swiglu=int(self.cfg.ffn == "swiglu"), hid_pad=self.hid_pad, ln_eps=self.cfg.ln_eps, ln_fold=int(self.fold_ln),
                        qkv_merged=int(self.merge_qkv),
                        pe_b=self.w["pe_b"].data_ptr(), reg=self.w["reg"].data_ptr(), norm_w=self.w["norm_w"].data_ptr(),
                        norm_b=self.w["norm_b"].data_ptr(), layers=layers)
            self._cdesc = (d, layers)
        return self._cdesc[0]

    def _encode(self, b, hp, wp, ape, pe_w, feats_f16, d_total, d0, feats_cl, tokens_f32) -> None:
        """One C-ABI call for the whole encoder (cvx_vit_encode); `_encode_py` below is the same launch list in Python."""
        import ctypes as C

        from cryovit_amd import _lib
        from cryovit_amd._lib import VitWs

        ws = self._workspace(b, hp, wp)
        pos, cls_pos0 = self._pos_for(hp, wp)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        cws = VitWs(x=ws["x"].data_ptr(), xn=p(ws.get("xn")), qk=ws["qk"].data_ptr(), vt=ws["vt"].data_ptr(),
                    ao=ws["ao"].data_ptr(), hid=ws["hid"].data_ptr(), xh=p(ws.get("xh")), xl=p(ws.get("xl")),
                    stat_part=p(ws.get("part")), rowstat=p(ws.get("rowstat")))
        if ops._dev_check(ape, pe_w, feats_f16, feats_cl, tokens_f32, ws["x"]) != self.device:
            raise _lib.CvxError(f"encoder: operands live on {ape.device}, the engine on {self.device}")
        ops.call(self.device, "cvx_vit_encode", _lib.load().cvx_vit_encode, C.byref(self._c_desc()), C.byref(cws), b, hp, wp,
                 ape.data_ptr(), ape.stride(0), pe_w.data_ptr(), pos.data_ptr(), cls_pos0.data_ptr(), p(feats_f16), d_total, d0,
                 p(feats_cl), p(tokens_f32))

    def _encode_py(self, b, hp, wp, ape, pe_w, feats_f16, d_total, d0, feats_cl, tokens_f32) -> None:
        cfg = self.cfg
        _, _, nt, ntp, kp = self._geometry(hp, wp)
        C, npatch, tok0 = cfg.dim, hp * wp, 1 + cfg.n_reg
        ws, w = self._workspace(b, hp, wp), self.w
        pos, cls_pos0 = self._pos_for(hp, wp)
        M = b * ntp

        ops.init_tokens(ws["x"], cls_pos0, w["reg"], n_reg=cfg.n_reg, slices=b, ntok=nt, ntp=ntp, Cdim=C)
        ops.gemm(EPI_PATCH, ape, pe_w, ws["x"], w["pe_b"], m=b * npatch, n=C, pos=pos, npatch=npatch, ntp=ntp, tok0=tok0)
        if self.fold_ln:
            rs = ws["rowstat"]
            ops.split_stream(ws["x"], ws["xh"], ws["xl"], rs, rows=M, Cdim=C, eps=cfg.ln_eps)
            for i, blk in enumerate(self.blocks):
                if self.merge_qkv:
                    ops.gemm(EPI_BF16, ws["xh"], blk["qk_w"], ws["qk"], blk["qk_b"], m=M, n=3 * C, ln_rowstat=rs)
                    ops.attention_qkv(ws["qk"], ws["ao"], slices=b, heads=cfg.heads, ntok=nt, ntp=ntp)
                else:
                    ops.gemm(EPI_BF16, ws["xh"], blk["qk_w"], ws["qk"], blk["qk_b"], m=M, n=2 * C, ln_rowstat=rs)
                    ops.gemm(EPI_VT, ws["xh"], blk["v_w"], ws["vt"], blk["v_b"], m=M, n=C, heads=cfg.heads, ntp=ntp, kp=kp, ldc=0, ln_rowstat=rs)
                    ops.attention(ws["qk"], ws["vt"], ws["ao"], slices=b, heads=cfg.heads, ntok=nt, ntp=ntp, kp=kp)
                ops.gemm(EPI_RESID_HL, ws["ao"], blk["proj_w"], ws["xh"], blk["proj_b"], m=M, n=C, gamma=blk["ls1"], out2=ws["xl"],
                         stat_part=ws["part"])
                ops.rowstat_finalize(ws["part"], rs, rows=M, Cdim=C, eps=cfg.ln_eps)
                if cfg.ffn == "swiglu":
                    ops.gemm(EPI_SWIGLU, ws["xh"], blk["ffn1_w"], ws["hid"], blk["ffn1_b"], m=M, n=2 * self.hid_pad, ln_rowstat=rs)
                else:
                    ops.gemm(EPI_BF16_GELU, ws["xh"], blk["ffn1_w"], ws["hid"], blk["ffn1_b"], m=M, n=self.hid_pad, ln_rowstat=rs)
                ops.gemm(EPI_RESID_HL, ws["hid"], blk["ffn2_w"], ws["xh"], blk["ffn2_b"], m=M, n=C, gamma=blk["ls2"], out2=ws["xl"],
                         stat_part=ws["part"])
                if i + 1 < len(self.blocks):
                    ops.rowstat_finalize(ws["part"], rs, rows=M, Cdim=C, eps=cfg.ln_eps)
            ops.final_norm_features_hl(ws["xh"], ws["xl"], w["norm_w"], w["norm_b"], cfg.ln_eps, slices=b, ntp=ntp, tok0=tok0, hp=hp,
                                       wp=wp, Cdim=C, feats_f16=feats_f16, d_total=d_total, d0=d0, feats_cl=feats_cl, tokens_f32=tokens_f32)
            return
        for blk in self.blocks:
            ops.layernorm(ws["x"], blk["ln1_w"], blk["ln1_b"], ws["xn"], M, C, cfg.ln_eps)
            ops.gemm(EPI_BF16, ws["xn"], blk["qk_w"], ws["qk"], blk["qk_b"], m=M, n=2 * C)
            ops.gemm(EPI_VT, ws["xn"], blk["v_w"], ws["vt"], blk["v_b"], m=M, n=C, heads=cfg.heads, ntp=ntp, kp=kp, ldc=0)
            ops.attention(ws["qk"], ws["vt"], ws["ao"], slices=b, heads=cfg.heads, ntok=nt, ntp=ntp, kp=kp)
            ops.gemm(EPI_RESID, ws["ao"], blk["proj_w"], ws["x"], blk["proj_b"], m=M, n=C, gamma=blk["ls1"])
            ops.layernorm(ws["x"], blk["ln2_w"], blk["ln2_b"], ws["xn"], M, C, cfg.ln_eps)
            if cfg.ffn == "swiglu":
                ops.gemm(EPI_SWIGLU, ws["xn"], blk["ffn1_w"], ws["hid"], blk["ffn1_b"], m=M, n=2 * self.hid_pad)
            else:
                ops.gemm(EPI_BF16_GELU, ws["xn"], blk["ffn1_w"], ws["hid"], blk["ffn1_b"], m=M, n=self.hid_pad)
            ops.gemm(EPI_RESID, ws["hid"], blk["ffn2_w"], ws["x"], blk["ffn2_b"], m=M, n=C, gamma=blk["ls2"])
        ops.final_norm_features(ws["x"], w["norm_w"], w["norm_b"], cfg.ln_eps, slices=b, ntp=ntp, tok0=tok0, hp=hp, wp=wp,
                                Cdim=C, feats_f16=feats_f16, d_total=d_total, d0=d0, feats_cl=feats_cl, tokens_f32=tokens_f32)

    def flops(self, b: int, H: int, W: int) -> float:
        """Algorithmic FLOPs (2*MACs of linears + attention matmuls + patch embed, SURVEY s.8d) for b slices."""
        cfg = self.cfg
        hp, wp, nt, _, _ = self.geometry(H, W)
        C, Hd = cfg.dim, cfg.ffn_hidden
        lin = 2 * (3 * C * C + C * C + (3 if cfg.ffn == "swiglu" else 2) * C * Hd)
        per_tok_layer = lin + 4 * nt * C
        return b * (cfg.depth * nt * per_tok_layer + 2 * hp * wp * 588 * C)
