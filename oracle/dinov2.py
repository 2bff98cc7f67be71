"""Oracle: DINOv2-with-registers ``forward_features`` (TEST INFRASTRUCTURE).

The reference calls the third-party hub model at
``/root/reference/src/cryovit/run/dino_features.py:58``
(``model.forward_features(vec)["x_norm_patchtokens"]``; identity of the model
at ``:25-28``: ``facebookresearch/dinov2`` / ``dinov2_vitg14_reg``).  That
repo is NOT under /root/reference and cannot be fetched, so this file restates
its published algorithm on a ``state_dict`` in the upstream
key layout (App. A-5).  It is cross-checked against the locally installed HF
port (``oracle/make_golden.py::check_vs_hf``).  Weight-level parity with the
real checkpoint: **parity unpinned**.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VitCfg:
    dim: int
    depth: int
    heads: int
    ffn: str  # "swiglu" (ViT-g) or "mlp" (ViT-S/B/L)
    ffn_hidden: int  # swiglu: hidden (w12 is 2*hidden); mlp: fc1 out
    n_reg: int = 4
    patch: int = 14
    pos_grid: int = 37  # 518 / 14
    ln_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads


def swiglu_hidden(dim: int, mlp_ratio: float = 4.0) -> int:
    """``(int(4*dim*2/3)+7)//8*8`` -- App. A-1 (4096 for dim 1536)."""
    return (int(int(dim * mlp_ratio) * 2 / 3) + 7) // 8 * 8


VITG14_REG = VitCfg(dim=1536, depth=40, heads=24, ffn="swiglu", ffn_hidden=swiglu_hidden(1536))
VITS14_REG = VitCfg(dim=384, depth=12, heads=6, ffn="mlp", ffn_hidden=1536)
# test-sized members of the same family (head_dim 64 like every DINOv2 variant)
VIT_TINY_SWIGLU = VitCfg(dim=128, depth=2, heads=2, ffn="swiglu", ffn_hidden=swiglu_hidden(128))
VIT_TINY_MLP = VitCfg(dim=128, depth=2, heads=2, ffn="mlp", ffn_hidden=512)


def init_state_dict(cfg: VitCfg, seed: int, std: float = 0.02, ls_gamma: float = 1.0) -> dict[str, torch.Tensor]:
    """Seeded synthetic weights in the upstream key layout.

    Cheap per-tensor ``normal_(0, std)`` (HF's default init takes minutes for
    ViT-g -- App. E).  LayerNorm weights ~ 1 + N(0, std), LayerScale = ls_gamma
    (+ N(0,std) jitter so a dropped gamma is caught).
    """
    g = torch.Generator().manual_seed(seed)
    C = cfg.dim

    def n(*shape, mean=0.0, s=std):
        return torch.empty(*shape).normal_(mean, s, generator=g)

    sd: dict[str, torch.Tensor] = {
        "cls_token": n(1, 1, C),
        "pos_embed": n(1, 1 + cfg.pos_grid * cfg.pos_grid, C),
        "register_tokens": n(1, cfg.n_reg, C),
        "mask_token": torch.zeros(1, C),
        "patch_embed.proj.weight": n(C, 3, cfg.patch, cfg.patch),
        "patch_embed.proj.bias": n(C),
        "norm.weight": n(C, mean=1.0),
        "norm.bias": n(C),
    }
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = n(C, mean=1.0)
        sd[p + "norm1.bias"] = n(C)
        sd[p + "attn.qkv.weight"] = n(3 * C, C)
        sd[p + "attn.qkv.bias"] = n(3 * C)
        sd[p + "attn.proj.weight"] = n(C, C)
        sd[p + "attn.proj.bias"] = n(C)
        sd[p + "ls1.gamma"] = n(C, mean=ls_gamma)
        sd[p + "norm2.weight"] = n(C, mean=1.0)
        sd[p + "norm2.bias"] = n(C)
        if cfg.ffn == "swiglu":
            sd[p + "mlp.w12.weight"] = n(2 * cfg.ffn_hidden, C)
            sd[p + "mlp.w12.bias"] = n(2 * cfg.ffn_hidden)
            sd[p + "mlp.w3.weight"] = n(C, cfg.ffn_hidden)
            sd[p + "mlp.w3.bias"] = n(C)
        else:
            sd[p + "mlp.fc1.weight"] = n(cfg.ffn_hidden, C)
            sd[p + "mlp.fc1.bias"] = n(cfg.ffn_hidden)
            sd[p + "mlp.fc2.weight"] = n(C, cfg.ffn_hidden)
            sd[p + "mlp.fc2.bias"] = n(C)
        sd[p + "ls2.gamma"] = n(C, mean=ls_gamma)
    return sd


def interpolate_pos_embed(cfg: VitCfg, pos_embed: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
    """App. A-2: ``[1,1+G*G,C] -> [1,1+hp*wp,C]``; bicubic, antialias, fp32.

    Depends only on the weights and (hp, wp): computed once per shape.
    """
    G = cfg.pos_grid
    if hp == G and wp == G:
        return pos_embed
    cls_pos = pos_embed[:, :1]
    patch_pos = pos_embed[:, 1:].reshape(1, G, G, cfg.dim).permute(0, 3, 1, 2)
    patch_pos = F.interpolate(
        patch_pos.float(), size=(hp, wp), mode="bicubic", align_corners=False, antialias=True
    )
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, hp * wp, cfg.dim)
    return torch.cat([cls_pos, patch_pos], dim=1)


def embed_tokens(cfg: VitCfg, sd: dict, x: torch.Tensor) -> torch.Tensor:
    """App. A-2: patch conv, [cls|patch] + pos, registers inserted after cls."""
    b, _, H, W = x.shape
    hp, wp = H // cfg.patch, W // cfg.patch
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg.patch)
    t = t.flatten(2).transpose(1, 2)  # [b, hp*wp, C], token = row*wp + col
    t = torch.cat([sd["cls_token"].expand(b, -1, -1), t], dim=1)
    t = t + interpolate_pos_embed(cfg, sd["pos_embed"], hp, wp)
    t = torch.cat([t[:, :1], sd["register_tokens"].expand(b, -1, -1), t[:, 1:]], dim=1)
    return t


def block_forward(cfg: VitCfg, sd: dict, i: int, x: torch.Tensor) -> torch.Tensor:
    """App. A-3: one pre-norm block with LayerScale."""
    p = f"blocks.{i}."
    C, nh, hd = cfg.dim, cfg.heads, cfg.head_dim
    b, N, _ = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(b, N, 3, nh, hd).permute(2, 0, 3, 1, 4)  # [3,b,nh,N,hd]
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = torch.softmax((q * hd**-0.5) @ k.transpose(-2, -1), dim=-1) @ v
    a = a.transpose(1, 2).reshape(b, N, C)
    a = F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    x = x + sd[p + "ls1.gamma"] * a
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
    if cfg.ffn == "swiglu":
        h12 = F.linear(h, sd[p + "mlp.w12.weight"], sd[p + "mlp.w12.bias"])
        x1, x2 = h12.chunk(2, dim=-1)
        m = F.linear(F.silu(x1) * x2, sd[p + "mlp.w3.weight"], sd[p + "mlp.w3.bias"])
    else:
        m = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        m = F.linear(F.gelu(m), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + sd[p + "ls2.gamma"] * m


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


@torch.inference_mode()
def forward_features_bf16_storage(cfg: VitCfg, sd: dict, x: torch.Tensor) -> dict[str, torch.Tensor]:
    """The same network with exact (fp32) arithmetic but every tensor the HIP path STORES in bf16 rounded to bf16: GEMM
    operands (LayerNorm outputs, weights, patch pixels), q/k/v, the softmax probabilities fed to the PV product, the
    attention output and the FFN hidden activations; the residual stream, accumulators, softmax statistics and norms stay
    fp32.  It separates what bf16 storage costs (any implementation with this storage plan) from what the kernels add."""
    b, _, H, W = x.shape
    hp, wp = H // cfg.patch, W // cfg.patch
    C, nh, hd = cfg.dim, cfg.heads, cfg.head_dim
    t = F.conv2d(_bf(x), _bf(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=cfg.patch).flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(b, -1, -1), t], dim=1) + interpolate_pos_embed(cfg, sd["pos_embed"], hp, wp)
    t = torch.cat([t[:, :1], sd["register_tokens"].expand(b, -1, -1), t[:, 1:]], dim=1)
    N = t.shape[1]
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        h = _bf(F.layer_norm(t, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps))
        w = sd[p + "attn.qkv.weight"].clone()
        bias = sd[p + "attn.qkv.bias"].clone()
        # the engine folds head_dim^-0.5 * log2(e) into the q rows before rounding them to bf16 and its attention kernel
        # works in log2 units (exp2): same mathematics, these rounding points
        w[:C] *= hd**-0.5 * math.log2(math.e)
        bias[:C] *= hd**-0.5 * math.log2(math.e)
        qkv = _bf(F.linear(h, _bf(w), bias)).reshape(b, N, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        sc = q @ k.transpose(-2, -1)
        m = sc.amax(-1, keepdim=True)
        e = torch.exp2(sc - m)
        a = (_bf(e) @ v) / e.sum(-1, keepdim=True)  # probabilities are rounded after exp(s - max), the sum stays fp32
        a = _bf(a.transpose(1, 2).reshape(b, N, C))
        t = t + sd[p + "ls1.gamma"] * F.linear(a, _bf(sd[p + "attn.proj.weight"]), sd[p + "attn.proj.bias"])
        h = _bf(F.layer_norm(t, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps))
        if cfg.ffn == "swiglu":
            x1, x2 = F.linear(h, _bf(sd[p + "mlp.w12.weight"]), sd[p + "mlp.w12.bias"]).chunk(2, dim=-1)
            mm = F.linear(_bf(F.silu(x1) * x2), _bf(sd[p + "mlp.w3.weight"]), sd[p + "mlp.w3.bias"])
        else:
            mm = F.linear(_bf(F.gelu(F.linear(h, _bf(sd[p + "mlp.fc1.weight"]), sd[p + "mlp.fc1.bias"]))), _bf(sd[p + "mlp.fc2.weight"]),
                          sd[p + "mlp.fc2.bias"])
        t = t + sd[p + "ls2.gamma"] * mm
    pre = t
    t = F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    return {"x_prenorm": pre, "x_norm_clstoken": t[:, 0], "x_norm_regtokens": t[:, 1 : 1 + cfg.n_reg],
            "x_norm_patchtokens": t[:, 1 + cfg.n_reg :]}


def split_hi_lo(t: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """The HIP path's residual-stream storage: hi = bf16(x), lo = bf16(x - hi); x is read back as hi + lo (exact in fp32)."""
    hi = _bf(t)
    return hi, _bf(t - hi)


@torch.inference_mode()
def forward_features_folded_storage(cfg: VitCfg, sd: dict, x: torch.Tensor) -> dict[str, torch.Tensor]:
    """Second storage-plan emulation: the plan the HIP path SHIPS (round 3).  Exact (fp32) arithmetic, with
      * the residual stream stored as a bf16 pair (hi, lo) after the patch embedding and after every residual update;
      * every LayerNorm inside a block FOLDED into the linear layer that consumes it: the GEMM's A operand is hi = bf16(x)
        un-normalised, its weight is W' = bf16(W * ln_gamma), and the normalisation is applied to the accumulator,
        y = rstd * (hi W'^T) - mean * rstd * cs + b',  cs[n] = sum_k W'[n][k],  b' = b + W ln_beta;
      * row statistics (mean, E[x^2] - mean^2) taken from the fp32 value of the stream BEFORE it is split;
      * everything else as in ``forward_features_bf16_storage`` (q scale folded in log2 units, bf16 q/k/v, probabilities, attention
        output and hidden activations).
    The final LayerNorm reads hi + lo."""
    b, _, H, W = x.shape
    hp, wp = H // cfg.patch, W // cfg.patch
    C, nh, hd = cfg.dim, cfg.heads, cfg.head_dim
    t = F.conv2d(_bf(x), _bf(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=cfg.patch).flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(b, -1, -1), t], dim=1) + interpolate_pos_embed(cfg, sd["pos_embed"], hp, wp)
    t = torch.cat([t[:, :1], sd["register_tokens"].expand(b, -1, -1), t[:, 1:]], dim=1)
    N = t.shape[1]

    def stats(v):
        mu = v.mean(-1, keepdim=True)
        var = ((v * v).mean(-1, keepdim=True) - mu * mu).clamp_min(0.0)
        rstd = torch.rsqrt(var + cfg.ln_eps)
        return rstd, -mu * rstd

    def ln_linear(hi, rs, gamma, beta, wm, bias):
        wg = _bf(wm * gamma[None, :])
        bp = (bias.double() + wm.double() @ beta.double()).float()
        return rs[0] * (hi @ wg.t()) + rs[1] * wg.sum(1) + bp

    rs = stats(t)
    hi, lo = split_hi_lo(t)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        w = sd[p + "attn.qkv.weight"].clone()
        bias = sd[p + "attn.qkv.bias"].clone()
        w[:C] *= hd**-0.5 * math.log2(math.e)
        bias[:C] *= hd**-0.5 * math.log2(math.e)
        qkv = _bf(ln_linear(hi, rs, sd[p + "norm1.weight"], sd[p + "norm1.bias"], w, bias)).reshape(b, N, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        sc = q @ k.transpose(-2, -1)
        e = torch.exp2(sc - sc.amax(-1, keepdim=True))
        a = (_bf(e) @ v) / e.sum(-1, keepdim=True)
        a = _bf(a.transpose(1, 2).reshape(b, N, C))
        t = (hi + lo) + sd[p + "ls1.gamma"] * F.linear(a, _bf(sd[p + "attn.proj.weight"]), sd[p + "attn.proj.bias"])
        rs = stats(t)
        hi, lo = split_hi_lo(t)
        if cfg.ffn == "swiglu":
            x1, x2 = ln_linear(hi, rs, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.w12.weight"], sd[p + "mlp.w12.bias"]).chunk(2, dim=-1)
            mm = F.linear(_bf(F.silu(x1) * x2), _bf(sd[p + "mlp.w3.weight"]), sd[p + "mlp.w3.bias"])
        else:
            h1 = ln_linear(hi, rs, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
            mm = F.linear(_bf(F.gelu(h1)), _bf(sd[p + "mlp.fc2.weight"]), sd[p + "mlp.fc2.bias"])
        t = (hi + lo) + sd[p + "ls2.gamma"] * mm
        rs = stats(t)
        hi, lo = split_hi_lo(t)
    pre = hi + lo
    t = F.layer_norm(pre, (C,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    return {"x_prenorm": pre, "x_norm_clstoken": t[:, 0], "x_norm_regtokens": t[:, 1 : 1 + cfg.n_reg],
            "x_norm_patchtokens": t[:, 1 + cfg.n_reg :]}


@torch.inference_mode()
def forward_features(cfg: VitCfg, sd: dict, x: torch.Tensor) -> dict[str, torch.Tensor]:
    """``[b,3,H',W'] fp32 -> {"x_norm_patchtokens": [b, hp*wp, C], ...}`` (App. A-4)."""
    t = embed_tokens(cfg, sd, x)
    for i in range(cfg.depth):
        t = block_forward(cfg, sd, i, t)
    pre = t
    t = F.layer_norm(t, (cfg.dim,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    return {
        "x_prenorm": pre,
        "x_norm_clstoken": t[:, 0],
        "x_norm_regtokens": t[:, 1 : 1 + cfg.n_reg],
        "x_norm_patchtokens": t[:, 1 + cfg.n_reg :],
    }


class OracleDino:
    """Duck-typed encoder (``forward_features``) for the reference's protocol."""

    def __init__(self, cfg: VitCfg, sd: dict):
        self.cfg, self.sd = cfg, sd

    def forward_features(self, x: torch.Tensor) -> dict[str, torch.Tensor]:
        return forward_features(self.cfg, self.sd, x)
