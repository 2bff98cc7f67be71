"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the alternate encoder of BASELINE configs[4]: the SAM2.1 Hiera image encoder +
FPN neck that ``SAM2.forward_features`` runs (``/root/reference/src/cryovit/models/sam2.py:190-209``:
trilinear resize of ``[1,D,3,H,W]`` to ``SAM_IMAGE_SIZE`` = 512 (``config.py:18``), flatten to ``[D,3,512,512]``,
``self.model.image_encoder(flat)``) and whose ``backbone_fpn`` / ``vision_pos_enc`` lists ``_sam_features`` stores as float16
(``run/dino_features.py:67-106,139-146``).

The arithmetic lives in the third-party ``sam2`` package (``pyproject.toml: sam2>=1.0``; absent here): ``ImageEncoder(trunk=
Hiera, neck=FpnNeck, scalp=1)`` configured by ``sam2.1_hiera_l.yaml``.  This file restates its published algorithm in torch
fp32 with the upstream checkpoint key names (``image_encoder.trunk.*`` / ``image_encoder.neck.*`` without the prefix), and
``hf_cross_check`` pins it against the locally installed HF port ``transformers.models.sam2`` (random init from a config
object, no download).  Weight-level parity with ``sam2.1_hiera_large.pt`` is **parity unpinned** (no checkpoint offline).

Hiera (Ryali et al. 2023; SAM2 variant "hieradet"):
  patch embed Conv2d(3, E, k=7, s=4, p=3) -> [B,H/4,W/4,E];  + bicubic(pos_embed [1,E,7,7]) + tile(pos_embed_window [1,E,8,8])
  blocks: x = shortcut + unpartition(attn(partition(norm1(x))));  x = x + mlp(norm2(x))      (MLP = Linear, GELU, Linear)
    first block of stages 2..4: dim doubles, heads double; shortcut = maxpool2x2(proj(norm1(x))); q = maxpool2x2(q) inside each
    window (window size of the PREVIOUS stage); blocks listed in global_att_blocks use one window = the whole grid
  neck: 1x1 conv of each stage output to 256 channels; levels in fpn_top_down_levels (other than the coarsest) add the 2x
    nearest-upsampled previous level; sine position encoding (normalised, temperature 1e4) per level; scalp drops the coarsest.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class HieraCfg:
    embed_dim: int = 144
    num_heads: int = 2
    stages: tuple = (2, 6, 36, 4)
    global_att_blocks: tuple = (23, 33, 43)
    window_pos_embed_bkg_spatial_size: tuple = (7, 7)
    window_spec: tuple = (8, 4, 16, 8)
    q_pool: int = 3
    d_model: int = 256
    fpn_top_down_levels: tuple = (2, 3)
    scalp: int = 1
    image_size: int = 512  # SAM_IMAGE_SIZE of the reference (config.py:18)

    @property
    def dims(self):
        return tuple(self.embed_dim * 2**i for i in range(len(self.stages)))

    @property
    def heads(self):
        return tuple(self.num_heads * 2**i for i in range(len(self.stages)))

    @property
    def depth(self):
        return sum(self.stages)

    def block_plan(self):
        """Per block: (dim_in, dim_out, heads, window (0 = global), q_stride (0 / 2))."""
        plan, stage_ends = [], [sum(self.stages[: i + 1]) - 1 for i in range(len(self.stages))]
        q_pool_blocks = [e + 1 for e in stage_ends[:-1]][: self.q_pool]
        cur, dim, heads = 1, self.embed_dim, self.num_heads
        for i in range(self.depth):
            dim_out = dim
            window = self.window_spec[cur - 1]  # lags by one block: the first block of a stage keeps the previous window
            if i in self.global_att_blocks:
                window = 0
            if i - 1 in stage_ends:
                dim_out, heads, cur = dim * 2, heads * 2, cur + 1
            plan.append((dim, dim_out, heads, window, 2 if i in q_pool_blocks else 0))
            dim = dim_out
        return plan, stage_ends


HIERA_L = HieraCfg()
# small enough for the CPU at 128x128 (token grids 32, 16, 8, 4), same head dim (72) and the same structural cases as Hiera-L:
# windowed + q-pooled transitions, a window that spans the whole grid, global blocks, every neck level
HIERA_TEST = HieraCfg(embed_dim=72, num_heads=1, stages=(1, 2, 4, 1), global_att_blocks=(4, 6), window_spec=(8, 4, 8, 4),
                      image_size=128)


def init_state_dict(cfg: HieraCfg, seed: int, std: float = 0.02) -> dict[str, torch.Tensor]:
    """Seeded random weights in the upstream key layout (``trunk.*`` / ``neck.*``).  Norm gains ~ N(1, 0.1) and ``std`` =
    0.02 like the upstream init; biases non-zero so that every term is exercised."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sd=std: torch.randn(*s, generator=g) * sd  # noqa: E731
    E = cfg.embed_dim
    sd = {
        "trunk.patch_embed.proj.weight": rn(E, 3, 7, 7, sd=0.05), "trunk.patch_embed.proj.bias": rn(E),
        "trunk.pos_embed": rn(1, E, *cfg.window_pos_embed_bkg_spatial_size, sd=0.2),
        "trunk.pos_embed_window": rn(1, E, cfg.window_spec[0], cfg.window_spec[0], sd=0.2),
    }
    plan, _ = cfg.block_plan()
    for i, (dim, dim_out, _, _, _) in enumerate(plan):
        p = f"trunk.blocks.{i}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = 1 + rn(dim, sd=0.1), rn(dim, sd=0.1)
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = rn(3 * dim_out, dim, sd=dim**-0.5), rn(3 * dim_out, sd=0.1)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = rn(dim_out, dim_out, sd=dim_out**-0.5), rn(dim_out, sd=0.1)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = 1 + rn(dim_out, sd=0.1), rn(dim_out, sd=0.1)
        sd[p + "mlp.layers.0.weight"], sd[p + "mlp.layers.0.bias"] = rn(4 * dim_out, dim_out, sd=dim_out**-0.5), rn(4 * dim_out, sd=0.1)
        sd[p + "mlp.layers.1.weight"], sd[p + "mlp.layers.1.bias"] = rn(dim_out, 4 * dim_out, sd=(4 * dim_out) ** -0.5), rn(dim_out, sd=0.1)
        if dim != dim_out:
            sd[p + "proj.weight"], sd[p + "proj.bias"] = rn(dim_out, dim, sd=dim**-0.5), rn(dim_out, sd=0.1)
    for j, c in enumerate(reversed(cfg.dims)):  # convs[0] serves the coarsest level
        sd[f"neck.convs.{j}.conv.weight"], sd[f"neck.convs.{j}.conv.bias"] = rn(cfg.d_model, c, 1, 1, sd=c**-0.5), rn(cfg.d_model, sd=0.1)
    return sd


def resize_input(data: torch.Tensor, size: int) -> torch.Tensor:
    """``SAM2.forward_features`` l.194-207: [b,d,c,h,w] -> [b*d,c,size,size]; trilinear over (c,h,w) with the channel extent
    unchanged is bilinear (align_corners=False) in the plane; nothing happens when h == w == size."""
    b, d, c, h, w = data.shape
    if h != size or w != size:
        data = F.interpolate(data, size=(c, size, size), mode="trilinear", align_corners=False)
    return data.reshape(-1, c, size, size)


def window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (-H) % ws, (-W) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def do_pool(x, stride):
    return F.max_pool2d(x.permute(0, 3, 1, 2), kernel_size=stride, stride=stride).permute(0, 2, 3, 1)


def _id(t):
    return t


def _bf(t):
    return t.to(torch.bfloat16).float()


def attention(sd, p, x, heads, q_stride, r=_id):
    """``r`` rounds what the HIP path STORES in bf16 (identity: the fp32 oracle): the q/k/v rows, the probabilities fed to the
    second product (the normaliser keeps the fp32 sum) and the attention output; weights are rounded where they are used."""
    B, H, W, _ = x.shape
    qkv = r(F.linear(x, r(sd[p + "qkv.weight"]), sd[p + "qkv.bias"])).reshape(B, H * W, 3, heads, -1)
    q, k, v = qkv.unbind(2)
    if q_stride:
        q = do_pool(q.reshape(B, H, W, -1), q_stride)
        H, W = q.shape[1:3]
        q = q.reshape(B, H * W, heads, -1)
    if r is _id:
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    else:
        qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        sc = (qt @ kt.transpose(-2, -1)) * (qt.shape[-1] ** -0.5)
        e = torch.exp(sc - sc.amax(-1, keepdim=True))
        o = (r(e) @ vt) / e.sum(-1, keepdim=True)
    o = r(o.transpose(1, 2).reshape(B, H, W, -1))
    return F.linear(o, r(sd[p + "proj.weight"]), sd[p + "proj.bias"])


def block_forward(sd, i, spec, x, r=_id):
    dim, dim_out, heads, window, q_stride = spec
    p = f"trunk.blocks.{i}."
    shortcut = x
    x = r(F.layer_norm(x, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6))
    if dim != dim_out:
        shortcut = do_pool(F.linear(x, r(sd[p + "proj.weight"]), sd[p + "proj.bias"]), q_stride or None) if q_stride else \
            F.linear(x, r(sd[p + "proj.weight"]), sd[p + "proj.bias"])
    ws = window
    if window > 0:
        H, W = x.shape[1:3]
        x, pad_hw = window_partition(x, window)
    x = attention(sd, p + "attn.", x, heads, q_stride, r)
    if q_stride:
        ws = window // q_stride
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (-H) % ws, W + (-W) % ws) if ws > 0 else (H, W)
    if window > 0:
        x = window_unpartition(x, ws, pad_hw, (H, W))
    x = shortcut + x
    h = r(F.layer_norm(x, (dim_out,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6))
    h = F.linear(r(F.gelu(F.linear(h, r(sd[p + "mlp.layers.0.weight"]), sd[p + "mlp.layers.0.bias"]))), r(sd[p + "mlp.layers.1.weight"]),
                 sd[p + "mlp.layers.1.bias"])
    return x + h


def pos_embed(cfg: HieraCfg, sd, h: int, w: int) -> torch.Tensor:
    """[1,h,w,E]: bicubic(pos_embed) + tiled window embedding (hieradet.py ``_get_pos_embed``)."""
    win = sd["trunk.pos_embed_window"]
    pe = F.interpolate(sd["trunk.pos_embed"], size=(h, w), mode="bicubic")
    pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
    return pe.permute(0, 2, 3, 1)


def trunk_forward(cfg: HieraCfg, sd, img: torch.Tensor, r=_id) -> list[torch.Tensor]:
    """[B,3,S,S] -> stage outputs, channels-last [B,h,w,C_stage], finest first."""
    x = F.conv2d(r(img), r(sd["trunk.patch_embed.proj.weight"]), sd["trunk.patch_embed.proj.bias"], stride=4, padding=3).permute(0, 2, 3, 1)
    x = x + pos_embed(cfg, sd, x.shape[1], x.shape[2])
    plan, stage_ends = cfg.block_plan()
    outs = []
    for i, spec in enumerate(plan):
        x = block_forward(sd, i, spec, x, r)
        if i in stage_ends:
            outs.append(x)
    return outs


def trunk_forward_folded(cfg: HieraCfg, sd, img: torch.Tensor) -> list[torch.Tensor]:
    """``trunk_forward`` with the storage plan the HIP path ships in round 3 (exact arithmetic, ``_bf`` where it stores bf16): in the
    stages whose width is a multiple of 64 the residual stream is a bf16 pair (hi = bf16(x), lo = bf16(x - hi)) and norm1 / norm2 are
    FOLDED into the qkv / fc1 layers -- the GEMM's A operand is hi, its weight bf16(W * gamma), and the accumulator is normalised with
    row statistics of the fp32 value before the split: y = rstd (hi W'^T) - mean rstd sum_k W' + (b + W beta).  A stage transition
    (and the narrow stages) run the round-2 plan on hi + lo; the first half of a transition INTO a wide stage is un-folded, its second
    half folded.  A folded stage hands bf16(x) = hi to the neck."""
    r = _bf
    x = F.conv2d(r(img), r(sd["trunk.patch_embed.proj.weight"]), sd["trunk.patch_embed.proj.bias"], stride=4, padding=3).permute(0, 2, 3, 1)
    x = x + pos_embed(cfg, sd, x.shape[1], x.shape[2])
    plan, stage_ends = cfg.block_plan()

    def stats(v):
        mu = v.mean(-1, keepdim=True)
        var = ((v * v).mean(-1, keepdim=True) - mu * mu).clamp_min(0.0)
        rstd = torch.rsqrt(var + 1e-6)
        return rstd, -mu * rstd

    def ln_linear(hi, rs, gamma, beta, wm, bias):
        wg = r(wm * gamma[None, :])
        return rs[0] * (hi @ wg.t()) + rs[1] * wg.sum(1) + (bias.double() + wm.double() @ beta.double()).float()

    def split(v):
        hi = r(v)
        return hi, r(v - hi)

    outs, pair = [], None  # pair = (hi, lo, row statistics) while the stage's stream is folded
    for i, spec in enumerate(plan):
        dim, dim_out, heads, window, q_stride = spec
        p = f"trunk.blocks.{i}."
        if pair is not None and dim == dim_out:
            hi, lo, rs = pair
            B, H, W, _ = hi.shape
            h1 = ln_linear(hi, rs, sd[p + "norm1.weight"], sd[p + "norm1.bias"], sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
            if window > 0:
                h1, pad_hw = window_partition(h1, window)
            # attention on the already-computed qkv rows (the q/k/v rounding, probabilities and output as in ``attention``)
            Bw, Hw, Ww, _ = h1.shape
            qkv = r(h1).reshape(Bw, Hw * Ww, 3, heads, -1)
            q, k, v = qkv.unbind(2)
            qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
            sc = (qt @ kt.transpose(-2, -1)) * (qt.shape[-1] ** -0.5)
            e = torch.exp(sc - sc.amax(-1, keepdim=True))
            o = r(((r(e) @ vt) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(Bw, Hw, Ww, -1))
            a = F.linear(o, r(sd[p + "attn.proj.weight"]), sd[p + "attn.proj.bias"])
            if window > 0:
                a = window_unpartition(a, window, pad_hw, (H, W))
            t = (hi + lo) + a
            rs = stats(t)
            hi, lo = split(t)
            h2 = ln_linear(hi, rs, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.layers.0.weight"], sd[p + "mlp.layers.0.bias"])
            t = (hi + lo) + F.linear(r(F.gelu(h2)), r(sd[p + "mlp.layers.1.weight"]), sd[p + "mlp.layers.1.bias"])
            rs = stats(t)
            hi, lo = split(t)
            pair = (hi, lo, rs)
            if i in stage_ends:
                outs.append(hi)
            continue
        if pair is not None:  # transition out of a folded stage: the round-2 first half on hi + lo
            x = pair[0] + pair[1]
            pair = None
        if dim_out % 64 == 0:
            # first half as in ``block_forward`` (un-folded), second half folded
            shortcut = x
            xn = r(F.layer_norm(x, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6))
            if dim != dim_out:
                shortcut = F.linear(xn, r(sd[p + "proj.weight"]), sd[p + "proj.bias"])
                if q_stride:
                    shortcut = do_pool(shortcut, q_stride)
            ws = window
            if window > 0:
                H, W = xn.shape[1:3]
                xn, pad_hw = window_partition(xn, window)
            a = attention(sd, p + "attn.", xn, heads, q_stride, r)
            if q_stride:
                ws = window // q_stride
                H, W = shortcut.shape[1:3]
                pad_hw = (H + (-H) % ws, W + (-W) % ws) if ws > 0 else (H, W)
            if window > 0:
                a = window_unpartition(a, ws, pad_hw, (H, W))
            t = shortcut + a
            rs = stats(t)
            hi, lo = split(t)
            h2 = ln_linear(hi, rs, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.layers.0.weight"], sd[p + "mlp.layers.0.bias"])
            t = (hi + lo) + F.linear(r(F.gelu(h2)), r(sd[p + "mlp.layers.1.weight"]), sd[p + "mlp.layers.1.bias"])
            rs = stats(t)
            hi, lo = split(t)
            pair = (hi, lo, rs)
            if i in stage_ends:
                outs.append(hi)
            continue
        x = block_forward(sd, i, spec, x, r)
        if i in stage_ends:
            outs.append(x)
    return outs


@torch.no_grad()
def forward_features_folded_storage(cfg: HieraCfg, sd, data: torch.Tensor) -> dict:
    """The round-3 storage plan of the HIP Hiera path (``trunk_forward_folded``) + the neck of ``forward_features_bf16_storage``."""
    feats, pos = neck_forward(cfg, sd, trunk_forward_folded(cfg, sd, resize_input(data.float(), cfg.image_size)), _bf)
    if cfg.scalp > 0:
        feats, pos = feats[: -cfg.scalp], pos[: -cfg.scalp]
    return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}


def sine_position_encoding(d_model: int, h: int, w: int, temperature: float = 10000.0) -> torch.Tensor:
    """[d_model,h,w] ``PositionEmbeddingSine(num_pos_feats=d_model, normalize=True)`` of the sam2 package (scale 2*pi, eps 1e-6)."""
    npf = d_model // 2
    y = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    y = y / (y[-1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)


def neck_forward(cfg: HieraCfg, sd, xs: list[torch.Tensor], r=_id):
    """FpnNeck: (features, pos) lists, finest first, each [B,256,h,w]."""
    n = len(xs) - 1
    out, pos = [None] * (n + 1), [None] * (n + 1)
    prev = None
    for i in range(n, -1, -1):
        lat = F.conv2d(r(xs[i]).permute(0, 3, 1, 2), r(sd[f"neck.convs.{n - i}.conv.weight"]), sd[f"neck.convs.{n - i}.conv.bias"])
        if i in cfg.fpn_top_down_levels and prev is not None:
            prev = lat + F.interpolate(prev.float(), scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
        pos[i] = sine_position_encoding(cfg.d_model, prev.shape[-2], prev.shape[-1])[None].expand(prev.shape[0], -1, -1, -1)
    return out, pos


def image_encoder(cfg: HieraCfg, sd, img: torch.Tensor, r=_id) -> dict:
    """``ImageEncoder.forward`` of the sam2 package: the dict ``SAM2.forward_features`` returns."""
    feats, pos = neck_forward(cfg, sd, trunk_forward(cfg, sd, img, r), r)
    if cfg.scalp > 0:
        feats, pos = feats[: -cfg.scalp], pos[: -cfg.scalp]
    return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}


@torch.no_grad()
def forward_features(cfg: HieraCfg, sd, data: torch.Tensor) -> dict:
    """``SAM2.forward_features`` (sam2.py:190-209) on ``[b,d,3,h,w]`` float data."""
    return image_encoder(cfg, sd, resize_input(data.float(), cfg.image_size))


@torch.no_grad()
def forward_features_bf16_storage(cfg: HieraCfg, sd, data: torch.Tensor) -> dict:
    """The same encoder with exact (fp32) arithmetic but every tensor the HIP path STORES in bf16 rounded to bf16: the gathered
    patch pixels, all GEMM weights, LayerNorm outputs, q / k / v, the softmax probabilities fed to the second product, attention
    outputs, GELU outputs and the stage outputs cast for the neck; the residual stream, the projected shortcut, accumulators,
    softmax statistics and the FPN sums stay fp32.  It separates what bf16 storage costs from what the kernels add
    (tests/test_gpu_sam.py), like ``oracle.dinov2.forward_features_bf16_storage`` does for the ViT."""
    return image_encoder(cfg, sd, resize_input(data.float(), cfg.image_size), _bf)


@torch.no_grad()
def sam_features(cfg: HieraCfg, sd, data: torch.Tensor) -> dict:
    """``_sam_features`` (run/dino_features.py:67-106) for the dataset's ``[1,D,3,H,W]`` item: every key except
    ``vision_features`` as a list of float16 arrays ``[D,256,h_l,w_l]``."""
    out = forward_features(cfg, sd, data)
    return {k: [t.half().numpy() for t in v] for k, v in out.items() if k != "vision_features"}


def to_hf_state_dict(cfg: HieraCfg, sd: dict) -> dict:
    """Upstream key layout -> ``transformers.models.sam2.Sam2VisionModel`` key layout."""
    out = {}
    for k, v in sd.items():
        k = k.replace("trunk.patch_embed.proj.", "backbone.patch_embed.projection.").replace("trunk.", "backbone.")
        k = k.replace(".norm1.", ".layer_norm1.").replace(".norm2.", ".layer_norm2.")
        k = k.replace(".mlp.layers.0.", ".mlp.proj_in.").replace(".mlp.layers.1.", ".mlp.proj_out.")
        k = k.replace(".conv.weight", ".weight").replace(".conv.bias", ".bias") if k.startswith("neck.") else k
        out[k] = v
    return out


def hf_model(cfg: HieraCfg, sd: dict):
    from transformers.models.sam2.configuration_sam2 import Sam2HieraDetConfig, Sam2VisionConfig
    from transformers.models.sam2.modeling_sam2 import Sam2VisionModel

    bc = Sam2HieraDetConfig(hidden_size=cfg.embed_dim, num_attention_heads=cfg.num_heads, image_size=[cfg.image_size] * 2,
                            blocks_per_stage=list(cfg.stages), embed_dim_per_stage=list(cfg.dims),
                            num_attention_heads_per_stage=list(cfg.heads), window_size_per_stage=list(cfg.window_spec),
                            global_attention_blocks=list(cfg.global_att_blocks), num_query_pool_stages=cfg.q_pool,
                            window_positional_embedding_background_size=list(cfg.window_pos_embed_bkg_spatial_size))
    vc = Sam2VisionConfig(backbone_config=bc, backbone_channel_list=list(reversed(cfg.dims)), fpn_hidden_size=cfg.d_model,
                          fpn_top_down_levels=list(cfg.fpn_top_down_levels), num_feature_levels=len(cfg.stages) - cfg.scalp)
    vc._attn_implementation = "eager"
    bc._attn_implementation = "eager"
    m = Sam2VisionModel(vc).eval()
    missing, unexpected = m.load_state_dict(to_hf_state_dict(cfg, sd), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


@torch.no_grad()
def hf_cross_check(cfg: HieraCfg, sd: dict, img: torch.Tensor) -> dict[str, float]:
    """max |restatement - HF port| per output level (features and position encodings)."""
    mine = image_encoder(cfg, sd, img)
    out = hf_model(cfg, sd)(pixel_values=img)
    errs = {}
    for i, (a, b) in enumerate(zip(mine["backbone_fpn"], out.fpn_hidden_states)):
        errs[f"fpn{i}"] = float((a - b).abs().max())
    for i, (a, b) in enumerate(zip(mine["vision_pos_enc"], out.fpn_position_encoding)):
        errs[f"pos{i}"] = float((a - b).abs().max())
    return errs
