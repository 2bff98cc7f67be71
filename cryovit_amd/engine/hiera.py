"""SAM2.1 Hiera image encoder + FPN neck on the HIP kernels (the alternate encoder of BASELINE configs[4]).

Replaces ``self.model.image_encoder(flat_data)`` and the resize in front of it -- ``SAM2.forward_features``,
``/root/reference/src/cryovit/models/sam2.py:190-209`` -- whose ``backbone_fpn`` / ``vision_pos_enc`` lists ``_sam_features``
(``run/dino_features.py:67-106``) stores as float16.  The architecture is the third-party ``sam2`` package's
``ImageEncoder(trunk=Hiera, neck=FpnNeck, scalp=1)`` (``sam2.1_hiera_l.yaml``); weights are consumed in the upstream
checkpoint key layout (``image_encoder.trunk.*`` / ``image_encoder.neck.*``, with or without that prefix).

Data layout in HBM for a batch of b slices (token rows are channels-last, row = (slice * G + y) * G + x):
  per stage s (grid G_s = S/4 / 2^s, C_s channels, K_s = C_s rounded up to 64):
    x_s    fp32 [b*G_s^2 (+pad)][C_s]     residual stream (kept until the neck has read it)
    xn_s   bf16 [..][K_s]                 LayerNorm output = GEMM A operand (pad columns stay zero)
    qkv_s  bf16 [..][3*C_s]               q | k | v, head h at columns h*head_dim
    ao_s   bf16 [..][K_s]                 attention output
    hid_s  bf16 [..][4*C_s]               GELU(fc1)
  stage transitions (first block of stages 2..4): qkv of the NEW width on the OLD grid, queries and the projected
  shortcut 2x2-max-pooled onto the new grid (``cvx_pool2x2``), attention with pooled queries inside the old windows.
Launch list per block: cvx_layernorm_bf16, cvx_gemm_bf16(qkv), cvx_window_attention_bf16, cvx_gemm_bf16(proj, +residual),
cvx_layernorm_bf16, cvx_gemm_bf16(fc1, GELU), cvx_gemm_bf16(fc2, +residual).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from cryovit_amd._lib import EPI_RESID_HL, EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_PATCH, EPI_RESID
from cryovit_amd.engine import ops
from cryovit_amd.engine.ops import alloc_rows, round_up


@dataclass(frozen=True)
class HieraConfig:
    embed_dim: int
    num_heads: int
    stages: tuple
    global_att_blocks: tuple
    window_spec: tuple
    window_pos_embed_bkg_spatial_size: tuple = (7, 7)
    q_pool: int = 3
    d_model: int = 256
    fpn_top_down_levels: tuple = (2, 3)
    scalp: int = 1
    image_size: int = 512  # SAM_IMAGE_SIZE (reference config.py:18)

    @property
    def dims(self):
        return tuple(self.embed_dim * 2**i for i in range(len(self.stages)))

    @property
    def depth(self):
        return sum(self.stages)

    def block_plan(self):
        """Per block (dim_in, dim_out, heads, window (0 = global), q_stride) and the last block index of every stage
        (Hiera.__init__ of the sam2 package: the window size lags one block behind the stage change)."""
        ends = [sum(self.stages[: i + 1]) - 1 for i in range(len(self.stages))]
        q_pool_blocks = [e + 1 for e in ends[:-1]][: self.q_pool]
        plan, cur, dim, heads = [], 1, self.embed_dim, self.num_heads
        for i in range(self.depth):
            dim_out, window = dim, self.window_spec[cur - 1]
            if i in self.global_att_blocks:
                window = 0
            if i - 1 in ends:
                dim_out, heads, cur = dim * 2, heads * 2, cur + 1
            plan.append((dim, dim_out, heads, window, 2 if i in q_pool_blocks else 0))
            dim = dim_out
        return plan, ends


HIERA_CONFIGS = {
    # facebook/sam2.1-hiera-large: the model the reference loads (models/sam2.py:32-35)
    "sam2.1_hiera_l": HieraConfig(144, 2, (2, 6, 36, 4), (23, 33, 43), (8, 4, 16, 8)),
    "sam2.1_hiera_b+": HieraConfig(112, 2, (2, 3, 16, 3), (12, 16, 20), (8, 4, 14, 7), (14, 14)),
    "sam2.1_hiera_s": HieraConfig(96, 1, (1, 2, 11, 2), (7, 10, 13), (8, 4, 14, 7)),
    # wanglab/MedSAM2 (models/sam2.py:36-39)
    "sam2.1_hiera_t": HieraConfig(96, 1, (1, 2, 7, 2), (5, 7, 9), (8, 4, 14, 7)),
}


def random_state_dict(cfg: HieraConfig, seed: int, device="cpu", std: float = 0.02) -> dict:
    """Synthetic weights in the upstream key layout, generated on ``device`` (no checkpoint is reachable offline)."""
    g = torch.Generator(device=device).manual_seed(seed)

    def n(*shape, sd=std, mean=0.0):
        return torch.empty(*shape, device=device).normal_(mean, sd, generator=g)

    E = cfg.embed_dim
    sd = {"trunk.patch_embed.proj.weight": n(E, 3, 7, 7, sd=0.05), "trunk.patch_embed.proj.bias": n(E),
          "trunk.pos_embed": n(1, E, *cfg.window_pos_embed_bkg_spatial_size, sd=0.2),
          "trunk.pos_embed_window": n(1, E, cfg.window_spec[0], cfg.window_spec[0], sd=0.2)}
    plan, _ = cfg.block_plan()
    for i, (dim, dout, _, _, _) in enumerate(plan):
        p = f"trunk.blocks.{i}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = n(dim, sd=0.1, mean=1.0), n(dim, sd=0.1)
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = n(3 * dout, dim, sd=dim**-0.5), n(3 * dout, sd=0.1)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = n(dout, dout, sd=dout**-0.5), n(dout, sd=0.1)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = n(dout, sd=0.1, mean=1.0), n(dout, sd=0.1)
        sd[p + "mlp.layers.0.weight"], sd[p + "mlp.layers.0.bias"] = n(4 * dout, dout, sd=dout**-0.5), n(4 * dout, sd=0.1)
        sd[p + "mlp.layers.1.weight"], sd[p + "mlp.layers.1.bias"] = n(dout, 4 * dout, sd=(4 * dout) ** -0.5), n(dout, sd=0.1)
        if dim != dout:
            sd[p + "proj.weight"], sd[p + "proj.bias"] = n(dout, dim, sd=dim**-0.5), n(dout, sd=0.1)
    for j, c in enumerate(reversed(cfg.dims)):
        sd[f"neck.convs.{j}.conv.weight"], sd[f"neck.convs.{j}.conv.bias"] = n(cfg.d_model, c, 1, 1, sd=c**-0.5), n(cfg.d_model, sd=0.1)
    return sd


def sine_position_encoding(d_model: int, h: int, w: int, temperature: float = 10000.0) -> torch.Tensor:
    """[d_model,h,w] fp32: ``PositionEmbeddingSine(num_pos_feats=d_model, normalize=True)`` (input independent: computed once
    per grid on the host)."""
    npf = d_model // 2
    y = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    y = y / (y[-1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()


def _npad(n: int, k: int = 0) -> int:
    """Output-width padding = GEMM tile choice.  The 256x256 pipeline (K >= 256) runs at about twice the rate of the
    128-wide tile kernel, so it is worth up to a third of padded columns (N = 576 -> 768); otherwise the widest tile whose
    padding stays below 15 %."""
    if k >= 256 and round_up(n, 256) - n <= 0.34 * n:
        return round_up(n, 256)
    for t in (256, 128):
        if round_up(n, t) - n <= 0.15 * n:
            return round_up(n, t)
    return round_up(n, 64)


class HieraEngine:
    """Packed device weights + per-batch workspaces; ``encode()`` runs one batch of slices."""

    def __init__(self, cfg: HieraConfig, state_dict: dict, device="cuda:0", fold_ln: bool = True):
        if not torch.cuda.is_available():
            raise ops._lib.CvxError("HieraEngine needs a HIP device (no CPU fallback)")
        ops._lib.load()
        self.cfg, self.device = cfg, ops.norm_device(device)
        # round 3: in the stages whose width is a multiple of 64 (Hiera-L: stages 3 and 4, 40 of its 48 blocks) the residual stream is
        # kept as a bf16 (hi, lo) pair and the LayerNorms are folded into the qkv / fc1 GEMMs, exactly as in the ViT path
        # (engine/vit.py); stage transitions and the narrow stages run the round-2 plan (fp32 stream + LayerNorm kernels)
        self.fold_ln = bool(fold_ln)
        S = cfg.image_size
        self.grids = tuple(S // 4 // 2**i for i in range(len(cfg.stages)))
        self.plan, self.stage_ends = cfg.block_plan()
        if cfg.embed_dim % cfg.num_heads or not 8 <= cfg.embed_dim // cfg.num_heads <= 96 or (cfg.embed_dim // cfg.num_heads) % 4:
            raise ValueError("head_dim must be a multiple of 4 in [8, 96]")
        n = len(cfg.stages) - 1
        if any(i not in (n - 1, n) for i in cfg.fpn_top_down_levels):
            raise NotImplementedError("top-down FPN levels other than the two coarsest are not built (no SAM2 config uses them)")
        g = self.grids[0]
        for i, (_, _, _, window, q_stride) in enumerate(self.plan):  # every window must tile its grid: no padding path
            w = window or g
            if g % w or (w * w) % 16 or (q_stride and (w % 2 or g % 2)):
                raise ValueError(f"block {i}: window {w} does not tile the {g}x{g} token grid in whole windows of 16k tokens "
                                 f"(image_size {S}); pick an image size that does")
            if q_stride:
                g //= 2
        self._ws: dict = {}
        self._pos_enc: dict = {}
        self._pack({k.removeprefix("image_encoder."): v for k, v in state_dict.items()})

    # ---- weight packing (once) ---------------------------------------------------------------------------------
    def _pack(self, sd: dict) -> None:
        cfg, dev = self.cfg, self.device
        g = lambda k: sd[k].detach().float()  # noqa: E731

        def lin(wk, bk):
            w, b = g(wk), g(bk)
            w = w.reshape(w.shape[0], -1)
            n_pad, k_pad = _npad(w.shape[0], w.shape[1]), round_up(w.shape[1], 64)
            wp = torch.zeros(n_pad, k_pad, dtype=torch.bfloat16, device=dev)
            wp[: w.shape[0], : w.shape[1]] = w.to(dev).to(torch.bfloat16)
            bp = torch.zeros(n_pad, dtype=torch.float32, device=dev)
            bp[: b.numel()] = b.to(dev)
            return wp, bp

        def ln_lin(wk, bk, gk, betak):
            """A linear layer behind LayerNorm(gamma, beta), folded: W' = bf16(W * gamma), bias table [2, n_pad] = b + W beta | column
            sums of W' (engine/vit.py::_pack)."""
            w, b, gamma, beta = g(wk), g(bk), g(gk), g(betak)
            n_pad, k_pad = _npad(w.shape[0], w.shape[1]), round_up(w.shape[1], 64)
            wp = torch.zeros(n_pad, k_pad, dtype=torch.bfloat16, device=dev)
            wp[: w.shape[0], : w.shape[1]] = (w * gamma[None, :]).to(dev).to(torch.bfloat16)
            bc = torch.zeros(2, n_pad, dtype=torch.float32, device=dev)
            bc[0, : w.shape[0]] = (b.double() + w.double() @ beta.double()).float().to(dev)
            bc[1] = wp.double().sum(dim=1).float()
            return wp, bc

        E, G0 = cfg.embed_dim, self.grids[0]
        self.pe_w, self.pe_b = lin("trunk.patch_embed.proj.weight", "trunk.patch_embed.proj.bias")  # K = c*49 + ky*7 + kx
        # position table: bicubic(pos_embed) + tiled window embedding (hieradet.py _get_pos_embed); weight-only, once.
        win = g("trunk.pos_embed_window")
        pe = F.interpolate(g("trunk.pos_embed"), size=(G0, G0), mode="bicubic")
        pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
        pos = torch.zeros(1 + G0 * G0, E)  # row 0 unused: the patch epilogue addresses row 1 + patch
        pos[1:] = pe[0].permute(1, 2, 0).reshape(G0 * G0, E)
        self.pos = pos.contiguous().to(dev)
        self.blocks = []
        for i, (dim, dout, heads, window, q_stride) in enumerate(self.plan):
            p = f"trunk.blocks.{i}."
            blk = {"n1": (g(p + "norm1.weight").to(dev), g(p + "norm1.bias").to(dev)),
                   "n2": (g(p + "norm2.weight").to(dev), g(p + "norm2.bias").to(dev)),
                   "qkv": lin(p + "attn.qkv.weight", p + "attn.qkv.bias"), "proj": lin(p + "attn.proj.weight", p + "attn.proj.bias"),
                   "fc1": lin(p + "mlp.layers.0.weight", p + "mlp.layers.0.bias"),
                   "fc2": lin(p + "mlp.layers.1.weight", p + "mlp.layers.1.bias")}
            if dim != dout:
                blk["short"] = lin(p + "proj.weight", p + "proj.bias")
            if self.fold_ln and dout % 64 == 0:  # second half of the block (norm2 -> fc1) on the folded stream
                blk["fc1_ln"] = ln_lin(p + "mlp.layers.0.weight", p + "mlp.layers.0.bias", p + "norm2.weight", p + "norm2.bias")
                if dim == dout:                  # ... and the first half (norm1 -> qkv) unless this is a stage transition
                    blk["qkv_ln"] = ln_lin(p + "attn.qkv.weight", p + "attn.qkv.bias", p + "norm1.weight", p + "norm1.bias")
            self.blocks.append(blk)
        n = len(cfg.stages) - 1
        self.neck = [lin(f"neck.convs.{n - s}.conv.weight", f"neck.convs.{n - s}.conv.bias") for s in range(n + 1)]  # by stage
        self.ones = torch.ones(round_up(max(max(cfg.dims), cfg.d_model), 256) + 256, dtype=torch.float32, device=dev)

    # ---- workspaces ---------------------------------------------------------------------------------------------
    def _buf(self, name: str, rows: int, cols: int, dtype) -> torch.Tensor:
        key = (name, rows, cols, dtype)
        if key not in self._ws:
            self._ws[key] = torch.zeros(alloc_rows(rows), cols, dtype=dtype, device=self.device)
        return self._ws[key]

    def _part(self, stage: int, rows: int, dout: int) -> torch.Tensor:
        """fp32 [dout / 64, alloc_rows(rows), 2]: the 64-column partial row sums a hi/lo residual GEMM leaves for cvx_rowstat_finalize."""
        key = ("part", stage, rows, dout)
        if key not in self._ws:
            self._ws[key] = torch.zeros(dout // 64, alloc_rows(rows), 2, dtype=torch.float32, device=self.device)
        return self._ws[key]

    def pos_enc(self, level: int) -> torch.Tensor:
        """fp16 [d_model, g, g] of FPN level ``level`` (host tensor)."""
        if level not in self._pos_enc:
            g = self.grids[level]
            self._pos_enc[level] = sine_position_encoding(self.cfg.d_model, g, g).half()
        return self._pos_enc[level]

    # ---- one batch of slices -------------------------------------------------------------------------------------
    @torch.inference_mode()
    def encode(self, src: torch.Tensor, outs: list[torch.Tensor], d0: int = 0) -> None:
        """src: device uint8 / float32 [b,H,W] or float32 [b,3,H,W].  outs: per kept FPN level a float16 device tensor
        [D_total, d_model, g, g]; slices d0 .. d0+b are written."""
        cfg = self.cfg
        b = src.shape[0]
        bf, f32 = torch.bfloat16, torch.float32
        G = self.grids[0]
        rows = b * G * G
        patches = self._buf("patches", rows, 192, bf)
        ops.sam_patches(src, patches, S=cfg.image_size)
        x = self._buf("x0", rows, cfg.embed_dim, f32)
        ops.gemm(EPI_PATCH, patches, self.pe_w, x, self.pe_b, m=rows, n=cfg.embed_dim, pos=self.pos, npatch=G * G, ntp=G * G, tok0=0)
        stage, stage_out = 0, []
        folded = None  # (xh, xl, part, rowstat) while the stream of the current stage lives as a bf16 pair
        for i, (dim, dout, heads, window, q_stride) in enumerate(self.plan):
            blk = self.blocks[i]
            hd = dout // heads
            w = window or G
            if folded is not None and "qkv_ln" in blk:
                # ---- a whole block on the folded stream: no LayerNorm kernel, no fp32 x ----
                xh, xl, part, rs = folded
                qkv = self._buf(f"qkv{stage}", rows, 3 * dout, bf)
                ops.gemm(EPI_BF16, xh, blk["qkv_ln"][0], qkv, blk["qkv_ln"][1], m=rows, n=3 * dout, ln_rowstat=rs)
                ao = self._buf(f"ao{stage}", rows, round_up(dout, 64), bf)
                ops.window_attention(qkv, 0, qkv, dout, 2 * dout, ao, slices=b, heads=heads, head_dim=hd, grid=G, window=w, q_grid=G,
                                     q_window=w)
                ops.gemm(EPI_RESID_HL, ao, blk["proj"][0], xh, blk["proj"][1], gamma=self.ones, m=rows, n=dout, out2=xl, stat_part=part)
                ops.rowstat_finalize(part, rs, rows=rows, Cdim=dout, eps=1e-6)
                hid = self._buf(f"hid{stage}", rows, round_up(4 * dout, 64), bf)
                ops.gemm(EPI_BF16_GELU, xh, blk["fc1_ln"][0], hid, blk["fc1_ln"][1], m=rows, n=4 * dout, ln_rowstat=rs)
                ops.gemm(EPI_RESID_HL, hid, blk["fc2"][0], xh, blk["fc2"][1], gamma=self.ones, m=rows, n=dout, out2=xl, stat_part=part)
                ops.rowstat_finalize(part, rs, rows=rows, Cdim=dout, eps=1e-6)
                if i in self.stage_ends:
                    stage_out.append((None, G, rows, dout, xh))  # the neck reads bf16(x) = hi
                continue
            if folded is not None:  # a stage transition behind a folded stage: back to fp32 for the round-2 first half
                xh, xl, _, _ = folded
                x = self._buf(f"x{stage}", rows, dim, f32)
                ops.merge_stream(xh, xl, x, rows=rows, Cdim=dim)
                folded = None
            xn = self._buf(f"xn{stage}", rows, round_up(dim, 64), bf)
            ops.layernorm(x, *blk["n1"], xn, rows, dim, 1e-6)
            if dim != dout:  # stage transition: new width on the old grid, then pool queries and shortcut
                qkv = self._buf(f"qkv_t{stage}", rows, 3 * dout, bf)
                ops.gemm(EPI_BF16, xn, blk["qkv"][0], qkv, blk["qkv"][1], m=rows, n=3 * dout)
                sc = self._buf(f"short{stage}", rows, dout, f32)
                ops.gemm(EPI_F32, xn, blk["short"][0], sc, blk["short"][1], gamma=self.ones, m=rows, n=dout)
                if q_stride:
                    Gn = G // 2
                    nrows = b * Gn * Gn
                    xnew = self._buf(f"x{stage + 1}", nrows, dout, f32)
                    ops.pool2x2(sc, xnew, slices=b, grid=G, C=dout)
                    qp = self._buf(f"qp{stage}", nrows, dout, bf)
                    ops.pool2x2(qkv, qp, slices=b, grid=G, C=dout)
                    ao = self._buf(f"ao{stage + 1}", nrows, round_up(dout, 64), bf)
                    ops.window_attention(qp, 0, qkv, dout, 2 * dout, ao, slices=b, heads=heads, head_dim=hd, grid=G, window=w,
                                         q_grid=Gn, q_window=w // 2)
                    x, G, rows = xnew, Gn, nrows
                else:  # widening without pooling (not used by the SAM2 configs, kept for completeness of the plan)
                    ao = self._buf(f"ao{stage + 1}", rows, round_up(dout, 64), bf)
                    ops.window_attention(qkv, 0, qkv, dout, 2 * dout, ao, slices=b, heads=heads, head_dim=hd, grid=G, window=w,
                                         q_grid=G, q_window=w)
                    x = sc
                stage += 1
            else:
                qkv = self._buf(f"qkv{stage}", rows, 3 * dout, bf)
                ops.gemm(EPI_BF16, xn, blk["qkv"][0], qkv, blk["qkv"][1], m=rows, n=3 * dout)
                ao = self._buf(f"ao{stage}", rows, round_up(dout, 64), bf)
                ops.window_attention(qkv, 0, qkv, dout, 2 * dout, ao, slices=b, heads=heads, head_dim=hd, grid=G, window=w, q_grid=G,
                                     q_window=w)
            ops.gemm(EPI_RESID, ao, blk["proj"][0], x, blk["proj"][1], gamma=self.ones, m=rows, n=dout)
            hid = self._buf(f"hid{stage}", rows, round_up(4 * dout, 64), bf)
            if "fc1_ln" in blk:
                # from here the stage's stream is a bf16 pair: split x (and take the row constants of norm2 from it), then the second
                # half of this block in the folded form
                xh, xl = self._buf(f"xh{stage}", rows, dout, bf), self._buf(f"xl{stage}", rows, dout, bf)
                part = self._part(stage, rows, dout)
                rs = self._buf(f"rowstat{stage}", rows, 2, f32)
                ops.split_stream(x, xh, xl, rs, rows=rows, Cdim=dout, eps=1e-6)
                ops.gemm(EPI_BF16_GELU, xh, blk["fc1_ln"][0], hid, blk["fc1_ln"][1], m=rows, n=4 * dout, ln_rowstat=rs)
                ops.gemm(EPI_RESID_HL, hid, blk["fc2"][0], xh, blk["fc2"][1], gamma=self.ones, m=rows, n=dout, out2=xl, stat_part=part)
                ops.rowstat_finalize(part, rs, rows=rows, Cdim=dout, eps=1e-6)
                folded = (xh, xl, part, rs)
                if i in self.stage_ends:
                    stage_out.append((None, G, rows, dout, xh))
                continue
            xn2 = self._buf(f"xn{stage}", rows, round_up(dout, 64), bf)
            ops.layernorm(x, *blk["n2"], xn2, rows, dout, 1e-6)
            ops.gemm(EPI_BF16_GELU, xn2, blk["fc1"][0], hid, blk["fc1"][1], m=rows, n=4 * dout)
            ops.gemm(EPI_RESID, hid, blk["fc2"][0], x, blk["fc2"][1], gamma=self.ones, m=rows, n=dout)
            if i in self.stage_ends:
                stage_out.append((x, G, rows, dout, None))
        # FPN neck: lateral 1x1 convs in fp32, top-down nearest upsampling for the listed levels, float16 [b,256,g,g] out
        n = len(stage_out) - 1
        lats = []
        for s, (xs, g, r, c, xhi) in enumerate(stage_out):
            if s > n - cfg.scalp and not (s == n and (n - 1) in cfg.fpn_top_down_levels):
                lats.append(None)  # dropped by scalp and not needed by a finer level
                continue
            if xhi is not None:
                xb = xhi  # a folded stage ends as a bf16 pair: its hi half IS bf16(x), the lateral conv's operand
            else:
                xb = self._buf(f"neck_in{s}", r, round_up(c, 64), bf)
                ops.cast_bf16(xs, xb, rows=r, C=c)
            lat = self._buf(f"lat{s}", r, cfg.d_model, f32)
            ops.gemm(EPI_F32, xb, self.neck[s][0], lat, self.neck[s][1], gamma=self.ones, m=r, n=cfg.d_model)
            lats.append(lat)
        for s in range(n + 1 - cfg.scalp):
            _, g, _, _, _ = stage_out[s]
            coarse = lats[s + 1] if (s in cfg.fpn_top_down_levels and s < n) else None
            ops.fpn_level_out(lats[s], coarse, outs[s][d0 : d0 + b], slices=b, C=cfg.d_model, grid=g)

    def n_levels(self) -> int:
        return len(self.cfg.stages) - self.cfg.scalp

    def flops(self, slices: int) -> float:
        """2 * MACs of the linear / attention products (same accounting as the ViT path)."""
        cfg, f = self.cfg, 0.0
        G = self.grids[0]
        f += 2.0 * G * G * 147 * cfg.embed_dim
        stage = 0
        for i, (dim, dout, heads, window, q_stride) in enumerate(self.plan):
            t = G * G
            w = window or G
            f += 2.0 * t * dim * 3 * dout
            if dim != dout:
                f += 2.0 * t * dim * dout
            tq = t // 4 if q_stride else t
            f += 4.0 * tq * (w * w) * dout  # QK^T and PV
            if q_stride:
                G //= 2
            t = G * G
            f += 2.0 * t * dout * dout + 16.0 * t * dout * dout
            if i in self.stage_ends:
                f += 2.0 * t * dout * cfg.d_model
        return f * slices
