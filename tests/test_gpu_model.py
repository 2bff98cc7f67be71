"""Model-level parity on the GPU: HIP path (through the C ABI) vs the CPU oracle / committed golden fixtures.

Stated tolerances of the HIP path (bf16 ViT, fp16 head) against the fp32 CPU path:
  x_norm_patchtokens (unit-scale LayerNorm outputs): max abs <= 1e-1, mean abs <= 1e-2; fp16 store adds <= 2e-3
  head logits (range [-5,5], std ~1) on identical input features: max abs <= 5e-2, mean abs <= 5e-3 (the bound SURVEY
    App. E proposes).  The head stores its 14 inter-layer activations in FP16 like the reference's fp16 autocast: a CPU
    emulation with exact arithmetic (oracle.head.forward_volume_16bit_storage) puts fp16 storage at ~0.02 max / 0.002 mean
    on the narrow fixture where bf16 storage costs 0.21 / 0.014, and the HIP path must match THAT emulation to max 3e-2 /
    mean 2e-3 (kernel-correctness bar; measured 0.019 / 0.0012)
  end to end (head fed with the GPU's own ViT features): logits max abs <= 2.5e-1, mean abs <= 2e-2
  |dDice| <= 1e-3
"""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _vit_cfgs():
    from oracle import dinov2 as o

    return {"vit_tiny_swiglu": o.VIT_TINY_SWIGLU, "vit_tiny_mlp": o.VIT_TINY_MLP}


def _engine_cfg(ocfg):
    from cryovit_amd.engine.vit import VitConfig

    return VitConfig(ocfg.dim, ocfg.depth, ocfg.heads, ocfg.ffn, ocfg.ffn_hidden, ocfg.n_reg, ocfg.patch, ocfg.pos_grid, ocfg.ln_eps)


def _check_tokens(got, ref, max_tol=1e-1, mean_tol=1e-2):
    err = (got - ref).abs()
    assert float(err.max()) <= max_tol, f"max abs {float(err.max())}"
    assert float(err.mean()) <= mean_tol, f"mean abs {float(err.mean())}"


def _emulation(o, fold_ln):
    """The exact-arithmetic CPU emulation of the storage plan the engine runs: round 3's shipped plan (bf16 hi/lo residual stream,
    LayerNorms folded into the consuming GEMMs) or round 2's (fp32 stream, bf16 LayerNorm outputs; ``fold_ln=False``)."""
    return o.forward_features_folded_storage if fold_ln else o.forward_features_bf16_storage


@pytest.mark.parametrize("fold_ln", [True, "split_qkv", False])
@pytest.mark.parametrize("name", ["vit_tiny_swiglu", "vit_tiny_mlp"])
def test_vit_tiny_golden(gpu, gold, name, fold_ln):
    """Fixture tokens were produced by the oracle that is cross-checked against the HF port (make_golden.py).  Both storage plans
    of the engine (the shipped folded one and the round-2 plan kept for A/B runs) against the fixture and their own emulation."""
    from cryovit_amd.engine.vit import VitEngine
    from oracle import dinov2 as o
    from oracle.make_golden import sd_checksum

    g = gold(f"{name}.npz")
    ocfg = _vit_cfgs()[name]
    sd = o.init_state_dict(ocfg, int(g["seed"]))
    assert sd_checksum(sd) == str(g["sd_sha256"]), "seeded weights differ from the ones the fixture was made with"
    # True: the shipped plan (folded LayerNorms, ONE qkv GEMM, attention reads V row-major); "split_qkv": folded, qk + V^T GEMMs
    eng = VitEngine(_engine_cfg(ocfg), sd, gpu, fold_ln=bool(fold_ln), merge_qkv=None if fold_ln is True else False)
    fold_ln = bool(fold_ln)
    # the fixture input is an already-resized [b,56,84] image: feed it through the protocol entry point
    x = torch.from_numpy(g["x"])  # [2,56,84] one channel (3 identical)
    x3 = x.unsqueeze(1).expand(-1, 3, -1, -1).contiguous()
    tok = eng.forward_features(x3.to(gpu))["x_norm_patchtokens"]
    _check_tokens(tok.float().cpu(), torch.from_numpy(g["tokens"]))
    # vs the oracle with the same bf16 storage points (exact arithmetic): the kernels add almost nothing on top of storage
    emu = _emulation(o, fold_ln)(ocfg, sd, x3)["x_norm_patchtokens"]
    _check_tokens(tok.float().cpu(), emu, max_tol=2e-2, mean_tol=2e-3)
    # the C-ABI call (cvx_vit_encode) and the op-by-op launch list are the same kernels in the same order: bit-identical
    b, hp, wp = x3.shape[0], x3.shape[2] // 14, x3.shape[3] // 14
    ws = eng._workspace(b, hp, wp)
    from cryovit_amd.engine import ops

    ops.im2col_patches(x3.to(gpu), ws["ape"])
    tok_py = torch.empty_like(tok)
    eng._encode_py(b, hp, wp, ws["ape"], eng.w["pe3_w"], None, 0, 0, None, tok_py)
    assert torch.equal(tok_py, tok)


def test_vit_s_raw_slices_vs_oracle(gpu):
    """ViT-S/14-reg (config 1's encoder) on raw uint8 slices incl. the fused resize, 3 slices of 64x96."""
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine
    from oracle import dinov2 as o
    from oracle import features as ofe
    from oracle import preprocess as opre

    sd = o.init_state_dict(o.VITS14_REG, 51)
    vol = np.random.default_rng(52).integers(0, 256, size=(3, 64, 96), dtype=np.uint8)
    ref = ofe.dino_features(opre.dino_transform(opre.load_scale(vol)), o.OracleDino(o.VITS14_REG, sd), 2)  # fp16 [C,D,h,w]
    eng = VitEngine(VIT_CONFIGS["dinov2_vits14_reg"], sd, gpu)
    C, D, hp, wp = ref.shape
    f16 = torch.zeros(C, D, hp, wp, dtype=torch.float16, device=gpu)
    sl = torch.from_numpy(vol).to(gpu)
    eng.features(sl[:2], feats_f16=f16, d_total=D, d0=0)  # two slice batches, like the reference's batch loop
    eng.features(sl[2:], feats_f16=f16, d_total=D, d0=2)
    _check_tokens(f16.float().cpu(), torch.from_numpy(ref.astype(np.float32)))


def test_vit_s_outlier_channels_vs_oracle(gpu):
    """Pretrained DINOv2 weights are not N(0, 0.02): the residual stream carries a few massive channels, LayerScale gains
    span orders of magnitude and the softmax is sharper.  No checkpoint is reachable offline, so that stress is built
    synthetically on ViT-S: 6 residual channels driven 50x (proj / fc2 rows), LayerScale over ~3 decades, LayerNorm gains in
    [0.1, 5], q/k projections x2 each.  The fp32 oracle's pre-norm stream then peaks at ~150x its median.
    Two references: the fp32 oracle, and the oracle with exact arithmetic but the HIP path's bf16 STORAGE points
    (``forward_features_bf16_storage``).  Storage alone moves single outputs by ~0.55 here (measured: emulation vs fp32), so
    the bound vs fp32 is "no worse than storage costs" (on the mean; the maximum only has a ceiling), and the kernels must
    stay close to the storage emulation (measured on MI355X: max 0.24 / mean 0.0022 -- what remains is fp32 summation order,
    the deferred softmax maximum (P up to 2^3 before a row is re-anchored) and the exp2 / erf approximations, amplified by
    the sharp softmax)."""
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine
    from oracle import dinov2 as o
    from oracle import preprocess as opre

    sd = o.init_state_dict(o.VITS14_REG, 61)
    g = torch.Generator().manual_seed(62)
    C = 384
    hot = torch.randperm(C, generator=g)[:6]
    sd["patch_embed.proj.bias"][hot] += 3.0
    for i in range(12):
        p = f"blocks.{i}."
        sd[p + "attn.proj.weight"][hot] *= 50.0
        sd[p + "mlp.fc2.weight"][hot] *= 50.0
        sd[p + "ls1.gamma"] = torch.exp(torch.randn(C, generator=g) * 1.5) * 0.3
        sd[p + "ls2.gamma"] = torch.exp(torch.randn(C, generator=g) * 1.5) * 0.3
        sd[p + "norm1.weight"] = torch.exp(torch.rand(C, generator=g) * 3.9 - 2.3)
        sd[p + "norm2.weight"] = torch.exp(torch.rand(C, generator=g) * 3.9 - 2.3)
        sd[p + "attn.qkv.weight"][: 2 * C] *= 2.0
    vol = np.random.default_rng(63).integers(0, 256, size=(2, 64, 96), dtype=np.uint8)
    x = opre.dino_transform(opre.load_scale(vol))
    f32 = o.forward_features(o.VITS14_REG, sd, x)
    emu = o.forward_features_folded_storage(o.VITS14_REG, sd, x)["x_norm_patchtokens"]  # the shipped storage plan
    assert float(f32["x_prenorm"].abs().max()) > 100 * float(f32["x_prenorm"].abs().median())  # the stress is real
    ref = f32["x_norm_patchtokens"]
    eng = VitEngine(VIT_CONFIGS["dinov2_vits14_reg"], sd, gpu)
    got = eng.forward_features(x.to(gpu))["x_norm_patchtokens"].float().cpu()
    assert torch.isfinite(got).all()
    e_emu, e_f32 = (got - emu).abs(), (got - ref).abs()
    e_store = (emu - ref).abs()
    stats = (float(e_emu.max()), float(e_emu.mean()), float(e_f32.max()), float(e_f32.mean()), float(e_store.max()), float(e_store.mean()))
    # Single-element maxima are chaotic under this stress (sharp softmax behind a 150x outlier stream: the emulation's own
    # maximum vs fp32 moved between 0.21 and 0.55 with nothing but the rounding point of the q scale), so the comparison with
    # fp32 is on the MEAN (no worse than 1.5x what storage alone costs) plus an absolute ceiling on the maximum; closeness
    # to the exact-arithmetic emulation of the same storage plan is the kernel-correctness bar.
    assert float(e_f32.mean()) <= 1.5 * stats[5] + 1e-3 and float(e_f32.mean()) <= 1e-2 and float(e_f32.max()) <= 0.75, stats
    assert float(e_emu.max()) <= 3e-1 and float(e_emu.mean()) <= 5e-3, stats


def test_head_narrow_golden(gpu, gold):
    """Fixture logits come from the AST-extracted reference SynthesisBlock (bit-equal to the oracle)."""
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine
    from oracle import head as oh
    from oracle.make_golden import sd_checksum

    g = gold("head_narrow.npz")
    head = oh.CryoVITHead(oh.NARROW_WIDTHS)
    oh.rescaled_init_(head, seed=int(g["seed"]))
    assert sd_checksum(head.state_dict()) == str(g["sd_sha256"])
    eng = HeadEngine(head.state_dict(), gpu)
    feats = torch.from_numpy(g["feats"])  # [C,D,h,w]
    C, D, h, w = feats.shape
    cl = torch.zeros(ops.alloc_rows(D * h * w), C, dtype=torch.float16, device=gpu)
    cl[: D * h * w] = feats.permute(1, 2, 3, 0).reshape(-1, C).to(torch.float16).to(gpu)
    labels = torch.from_numpy(g["labels"]).to(gpu)
    out = eng.forward(cl, D, h, w, labels=labels, want_logits=True)
    ref = torch.from_numpy(g["logits"])
    got = out["logits"].cpu()
    emu = oh.forward_volume_16bit_storage(head, feats.unsqueeze(0), torch.float16)[0, 0]
    e_emu = (got - emu).abs()  # kernels vs exact-arithmetic fp16 storage: 1-ulp rounding flips of single activations
    assert float(e_emu.max()) <= 3e-2 and float(e_emu.mean()) <= 2e-3, (float(e_emu.max()), float(e_emu.mean()))
    err = (got - ref).abs()
    assert float(err.max()) <= 5e-2 and float(err.mean()) <= 5e-3, (float(err.max()), float(err.mean()))  # vs fp32 CPU
    e_bf = (oh.forward_volume_bf16_storage(head, feats.unsqueeze(0))[0, 0] - ref).abs()
    assert float(e_bf.max()) > 3 * float(err.max())  # why the head is not bf16: that storage alone is several times worse
    i, sy, sp = out["dice_sums"].cpu().tolist()
    dice = 2 * i / (sy + sp + 1e-3)
    near = int(((ref.abs() < 5e-2) & (torch.from_numpy(g["labels"]) > -1)).sum())
    assert abs(dice - float(g["dice"])) <= 1e-3, (dice, float(g["dice"]), f"{near} voxels within tolerance of the threshold")
    # the single C call (cvx_head_forward) and the op-by-op launch list are the same kernels in the same order: bit-identical
    py = eng._forward_py(cl, D, h, w, labels=labels, want_logits=True, mask_threshold=0.5)
    one = eng.forward(cl, D, h, w, labels=labels, want_logits=True, mask_threshold=0.5)
    for k in ("logits", "probs", "dice_sums", "mask"):
        assert torch.equal(py[k], one[k]), k
    assert torch.equal(one["mask"], (one["probs"] >= 0.5).to(torch.uint8))


def test_synthesis_block_golden(gpu, gold):
    from cryovit_amd._lib import EPI_CONVT
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1, _pad2

    g = gold("synthesis_block.npz")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])  # [32,6,8,8] -> [8,6,16,16]
    C, D, H, W = x.shape
    nv = D * H * W
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    xin = x.permute(1, 2, 3, 0).reshape(nv, C).contiguous().to(torch.float16).to(gpu)
    zero = torch.zeros(256, dtype=torch.uint8, device=gpu)
    stats = torch.zeros(ops.gn_stats_size(8), device=gpu)
    gn = torch.zeros_like(xin)
    ops.groupnorm(xin, t("layers_0_weight").to(gpu), t("layers_0_bias").to(gpu), gn, stats, nvox=nv, Cdim=32, G=8, eps=1e-3)
    t1 = torch.zeros(nv, 16, dtype=torch.float16, device=gpu)
    ops.conv3d(gn, _conv3_weight(t("layers_1_weight")).to(gpu), _pad1(t("layers_1_bias"), 16).to(gpu), t1, zero, Cin=32, D=D, H=H, W=W,
               dil=2, cout=16, act=1)
    t2 = torch.zeros(ops.alloc_rows(nv) * 16 + 4096, dtype=torch.float16, device=gpu)
    ops.conv3d(t1, _conv3_weight(t("layers_3_weight")).to(gpu), _pad1(t("layers_3_bias"), 16).to(gpu), t2, zero, Cin=16, D=D, H=H, W=W,
               dil=1, cout=16, act=1)
    wt = t("layers_5_weight")
    wg = wt[:, :, 0].permute(2, 3, 1, 0).reshape(32, 16)
    out = torch.zeros(D, 2 * H, 2 * W, 8, dtype=torch.float16, device=gpu)
    ops.gemm(EPI_CONVT, torch.as_strided(t2, (ops.alloc_rows(nv), 16), (16, 1)), _pad2(wg, _npad(32), 64).to(gpu), out,
             _pad1(t("layers_5_bias").repeat(4), _npad(32)).to(gpu), m=nv, n=32, H=H, W=W, cout=8, act=1, ldc=8)
    got = out.float().cpu().permute(3, 0, 1, 2)
    # activations here reach |y| ~ 27 (weights N(0,0.2)): fp16 storage of 3 intermediate layers -> rtol 3e-3 of scale
    assert torch.allclose(got, y, atol=3e-2, rtol=3e-3), float((got - y).abs().max())


def test_e2e_tiny_golden(gpu, gold):
    """Raw uint8 volume -> fused resize + ViT -> head -> probabilities + Dice, against the committed fixture."""
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine
    from cryovit_amd.engine.vit import VitEngine
    from oracle import dinov2 as o
    from oracle import head as oh

    g = gold("e2e_tiny.npz")
    ocfg = o.VIT_TINY_SWIGLU
    vit = VitEngine(_engine_cfg(ocfg), o.init_state_dict(ocfg, int(g["vit_seed"])), gpu)
    head = oh.CryoVITHead(oh.NARROW_WIDTHS)
    oh.rescaled_init_(head, seed=int(g["head_seed"]))
    heng = HeadEngine(head.state_dict(), gpu)
    vol = torch.from_numpy(g["vol"]).to(gpu)
    D, H, W = vol.shape
    hp, wp = math.ceil(H / 16), math.ceil(W / 16)
    C = ocfg.dim
    f16 = torch.zeros(C, D, hp, wp, dtype=torch.float16, device=gpu)
    cl = torch.zeros(ops.alloc_rows(D * hp * wp), C, dtype=torch.float16, device=gpu)
    for d0 in range(0, D, 3):
        b = min(3, D - d0)
        vit.features(vol[d0 : d0 + b], feats_f16=f16, d_total=D, d0=d0, feats_cl=cl[d0 * hp * wp :])
    ref_f = torch.from_numpy(g["feats"].astype(np.float32))
    err = (f16.float().cpu() - ref_f).abs()
    assert float(err.max()) <= 1e-1 and float(err.mean()) <= 1e-2, (float(err.max()), float(err.mean()))
    out = heng.forward(cl, D, hp, wp, labels=torch.from_numpy(g["labels"]).to(gpu), want_logits=True)
    ref_p = torch.from_numpy(g["probs"])
    ref_logit = torch.logit(ref_p.double()).float()
    got = out["logits"].cpu()
    err = (got - ref_logit).abs()
    assert float(err.max()) <= 2.5e-1 and float(err.mean()) <= 2e-2, (float(err.max()), float(err.mean()))
    i, sy, sp = out["dice_sums"].cpu().tolist()
    dice = 2 * i / (sy + sp + 1e-3)
    flips = int(((got > 0) != (ref_logit > 0)).sum())
    assert abs(dice - float(g["dice"])) <= 1e-3, (dice, float(g["dice"]), f"{flips} of {got.numel()} voxels changed side of the threshold")


def test_vit_g_full_depth_one_slice_vs_oracle(gpu):
    """The REAL encoder configuration (ViT-g/14-reg: dim 1536, 40 layers, 24 heads, SwiGLU 4096) end to end on two raw
    128x128 uint8 slices against the CPU oracle with the same seeded weights: 40 layers of bf16 GEMM / attention error
    accumulation must stay inside the stated tolerance."""
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict
    from oracle import dinov2 as o
    from oracle import features as ofe
    from oracle import preprocess as opre

    cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
    sd_dev = random_state_dict(cfg, seed=2, device=gpu)
    eng = VitEngine(cfg, sd_dev, gpu)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    vol = np.random.default_rng(77).integers(0, 256, size=(2, 128, 128), dtype=np.uint8)
    ref = ofe.dino_features(opre.dino_transform(opre.load_scale(vol)), o.OracleDino(o.VITG14_REG, sd), 2)  # [1536,2,8,8] fp16
    f16 = torch.zeros(1536, 2, 8, 8, dtype=torch.float16, device=gpu)
    eng.features(torch.from_numpy(vol).to(gpu), feats_f16=f16, d_total=2, d0=0)
    _check_tokens(f16.float().cpu(), torch.from_numpy(ref.astype(np.float32)))


def test_vit_g_headline_slice_vs_oracle(gpu):
    """Oracle parity AT THE BENCHMARKED GEOMETRY (BASELINE configs[1]): ViT-g/14-reg, all 40 layers, one raw 512x512 uint8
    slice -> fused resize to 448x448 -> N = 1029 tokens per slice (9 query blocks, 17 key tiles, the padded 1032/1088 row
    and key counts of the real run), through ``VitEngine.features`` against
      * the fp32 CPU oracle (reference call site /root/reference/src/cryovit/run/dino_features.py:53-61): max abs <= 1e-1,
        mean abs <= 1e-2 on the fp16 ``dino_features`` values;
      * the exact-arithmetic bf16-STORAGE emulation of the same network: what remains is the kernels' own contribution.
    The slice sits at batch row 1 of 2 so its tokens straddle a 256-row GEMM tile boundary like the rows of a full batch."""
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict
    from oracle import dinov2 as o
    from oracle import preprocess as opre

    cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
    sd_dev = random_state_dict(cfg, seed=2, device=gpu)
    eng = VitEngine(cfg, sd_dev, gpu)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    vol = np.random.default_rng(1).integers(0, 256, size=(2, 512, 512), dtype=np.uint8)
    f16 = torch.zeros(1536, 2, 32, 32, dtype=torch.float16, device=gpu)
    eng.features(torch.from_numpy(vol).to(gpu), feats_f16=f16, d_total=2, d0=0)
    got = f16[:, 1].float().cpu().reshape(1536, 1024).t()  # [tokens, C] of slice 1
    x = opre.dino_transform(opre.load_scale(vol[1:2]))  # [1,3,448,448]
    assert tuple(x.shape) == (1, 3, 448, 448)
    ref = o.forward_features(o.VITG14_REG, sd, x)["x_norm_patchtokens"][0]
    assert ref.shape == (1024, 1536)
    _check_tokens(got, ref.half().float())  # K9: the reference stores fp16 (run/dino_features.py:61)
    emu = o.forward_features_folded_storage(o.VITG14_REG, sd, x)["x_norm_patchtokens"][0]  # the shipped storage plan (round 3)
    e_emu, e_store = (got - emu).abs(), (emu - ref).abs()
    # Over 40 layers two implementations of the SAME storage plan drift apart by about what either drifts from fp32 (every
    # 1-ulp bf16 rounding flip is a perturbation of the storage plan's own size), so the bar for the kernels is: no further
    # from the exact-arithmetic emulation than the emulation is from fp32.  Measured on MI355X: GPU vs emulation
    # 0.035 max / 0.0062 mean, emulation vs fp32 0.044 / 0.0080.
    assert float(e_emu.max()) <= 1.25 * float(e_store.max()) + 1e-2 and float(e_emu.mean()) <= float(e_store.mean()), (
        float(e_emu.max()), float(e_emu.mean()), float(e_store.max()), float(e_store.mean()))
    # ... and an ABSOLUTE ceiling beside the relative bar, so that a regression of both implementations cannot pass together
    assert float(e_emu.mean()) <= 8e-3 and float(e_emu.max()) <= 6e-2, (float(e_emu.max()), float(e_emu.mean()))
    print(f"ViT-g N=1029: vs fp32 max {float((got - ref).abs().max()):.3e} mean {float((got - ref).abs().mean()):.3e}; "
          f"vs bf16-storage emulation max {float(e_emu.max()):.3e} mean {float(e_emu.mean()):.3e}; "
          f"emulation vs fp32 max {float(e_store.max()):.3e} mean {float(e_store.mean()):.3e}")


def vit_g_outlier_state_dict(seed=91):
    """ViT-g/14-reg weights with the outlier structure pretrained DINOv2 checkpoints show (no checkpoint is reachable offline):
    the construction of ``test_vit_s_outlier_channels_vs_oracle`` at dim 1536 and depth 40 -- 24 residual channels (1.6 %) driven 50x
    through the proj / w3 rows, LayerScale log-normal over ~2 decades, LayerNorm gains in [0.5, 2].  Calibrated on the CPU so that
    40 layers stay in the regime where the fp32 network itself is well conditioned (the ViT-S test's x2 on q / k and gains in
    [0.1, 5] turn chaotic at this depth: there the exact-arithmetic bf16-storage emulation ALONE moves outputs by 0.1 mean / 20 max
    from fp32, which says nothing about kernels); the fp32 pre-norm stream peaks at ~250x its median."""
    from oracle import dinov2 as o

    cfg = o.VITG14_REG
    sd = o.init_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1)
    C = cfg.dim
    hot = torch.randperm(C, generator=g)[:24]
    sd["patch_embed.proj.bias"][hot] += 3.0
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        sd[p + "attn.proj.weight"][hot] *= 50.0
        sd[p + "mlp.w3.weight"][hot] *= 50.0
        sd[p + "ls1.gamma"] = torch.exp(torch.randn(C, generator=g)) * 0.03
        sd[p + "ls2.gamma"] = torch.exp(torch.randn(C, generator=g)) * 0.03
        sd[p + "norm1.weight"] = torch.exp(torch.rand(C, generator=g) * 1.4 - 0.7)
        sd[p + "norm2.weight"] = torch.exp(torch.rand(C, generator=g) * 1.4 - 0.7)
    return sd


def test_vit_g_headline_outlier_channels_vs_oracle(gpu):
    """ViT-g/14-reg, all 40 layers, one raw 512x512 slice (N = 1029: the benchmarked geometry) with OUTLIER-CHANNEL weights
    (``vit_g_outlier_state_dict``): the full-depth check on benign N(0, 0.02) weights sits at 80 % of the mean budget, this one
    shows the budget under a realistic residual stream.  Against the fp32 oracle with the bounds of the ViT-S stress test
    (mean <= 1e-2 and <= 1.5x what the storage plan alone costs, maximum <= 0.75) and against the exact-arithmetic emulation of
    the shipped storage plan (bf16 hi/lo stream, folded LayerNorms: ``forward_features_folded_storage``)."""
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine
    from oracle import dinov2 as o
    from oracle import preprocess as opre

    sd = vit_g_outlier_state_dict()
    eng = VitEngine(VIT_CONFIGS["dinov2_vitg14_reg"], sd, gpu)
    vol = np.random.default_rng(1).integers(0, 256, size=(2, 512, 512), dtype=np.uint8)
    f16 = torch.zeros(1536, 2, 32, 32, dtype=torch.float16, device=gpu)
    eng.features(torch.from_numpy(vol).to(gpu), feats_f16=f16, d_total=2, d0=0)
    got = f16[:, 1].float().cpu().reshape(1536, 1024).t()
    assert torch.isfinite(got).all()
    x = opre.dino_transform(opre.load_scale(vol[1:2]))
    f32 = o.forward_features(o.VITG14_REG, sd, x)
    pre, ref = f32["x_prenorm"], f32["x_norm_patchtokens"][0]
    assert float(pre.abs().max()) > 100 * float(pre.abs().median())  # the stress is real
    emu = o.forward_features_folded_storage(o.VITG14_REG, sd, x)["x_norm_patchtokens"][0]
    e_f32, e_emu, e_store = (got - ref).abs(), (got - emu).abs(), (emu - ref).abs()
    stats = tuple(float(v) for v in (e_f32.max(), e_f32.mean(), e_emu.max(), e_emu.mean(), e_store.max(), e_store.mean()))
    print("ViT-g N=1029 outlier channels: vs fp32 max %.3e mean %.3e; vs folded emulation max %.3e mean %.3e; emulation vs fp32 max %.3e mean %.3e" % stats)
    assert stats[1] <= 1.5 * stats[5] + 1e-3 and stats[1] <= 1e-2 and stats[0] <= 0.75, stats
    assert stats[3] <= 5e-3 and stats[2] <= 0.5, stats


def test_head_full_width_depth128_vs_oracle(gpu):
    """Oracle parity at the benchmarked DEPTH (BASELINE configs[2]): the FULL-WIDTH head on features [1536, 128, 4, 4] ->
    128 x 64 x 64 voxels.  At D = 128 every one of the eight depth dilations (32, 24, 16, 12, 8, 4, 2, 1 --
    /root/reference/src/cryovit/models/cryovit.py:24-28) has voxels with BOTH z +- d taps inside the volume (D = 40 had none
    for d = 32), and GroupNorm statistics span the real depth.  Logits vs the fp32 CPU oracle (max <= 5e-2, mean <= 5e-3),
    masked Dice within 1e-3, and the number of labelled voxels within the logit tolerance of the threshold reported."""
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine
    from oracle import dice as od
    from oracle import head as oh
    from oracle.make_golden import synth_labels

    head = oh.CryoVITHead()
    oh.rescaled_init_(head, seed=5)
    C, D, h, w = 1536, 128, 4, 4
    feats = torch.randn(C, D, h, w, generator=torch.Generator().manual_seed(3)).half()
    with torch.inference_mode():
        ref = head.forward_volume(feats.float().unsqueeze(0))[0, 0]
    eng = HeadEngine(head.state_dict(), gpu)
    cl = torch.zeros(ops.alloc_rows(D * h * w), C, dtype=torch.float16, device=gpu)
    ops.features_to_channels_last(feats.to(gpu), cl)
    labels = torch.from_numpy(synth_labels(D, 16 * h, 16 * w, seed=4))
    out = eng.forward(cl, D, h, w, labels=labels.to(gpu), want_logits=True)
    got = out["logits"].cpu()
    assert tuple(got.shape) == (D, 16 * h, 16 * w)
    err = (got - ref).abs()
    assert float(err.max()) <= 5e-2 and float(err.mean()) <= 5e-3, (float(err.max()), float(err.mean()))
    i, sy, sp = out["dice_sums"].cpu().tolist()
    dice = 2 * i / (sy + sp + 1e-3)
    want = od.dice_metric(torch.sigmoid(ref), labels.float())
    near = int(((ref.abs() < 5e-2) & (labels > -1)).sum())
    flips = int((((got > 0) != (ref > 0)) & (labels > -1)).sum())
    assert abs(dice - want) <= 1e-3, (dice, want, f"{flips} flipped of {near} labelled voxels within 5e-2 of the threshold")
    fg = float((ref > 0).float().mean())
    assert 0.05 < fg < 0.95
    print(f"head D=128: logits max err {float(err.max()):.3e} mean {float(err.mean()):.3e}; dice {dice:.5f} vs {want:.5f}; "
          f"{flips} flips / {near} near-threshold labelled voxels")


def test_full_size_properties(gpu):
    """BASELINE size (128x512x512, ViT-g + full-width head): properties that need no CPU oracle.
      * determinism: the ViT path has no atomics -> two runs give bit-identical fp16 features
      * slice equivariance: reversing the slice order reverses the features (slices are independent in the ViT)
      * head output range: probabilities inside [sigmoid(-5), sigmoid(5)] (the clip of cryovit.py:39)
      * Dice sums: recomputed from the GPU's own probabilities and labels, bit-exact (integer-valued fp32 sums)"""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine
    from cryovit_amd.engine.vit import VIT_CONFIGS, VitEngine, random_state_dict

    cfg = VIT_CONFIGS["dinov2_vitg14_reg"]
    vit = VitEngine(cfg, random_state_dict(cfg, seed=2, device=gpu), gpu)
    head = HeadEngine(bench.synthetic_head_state_dict(5, gpu), gpu)
    D, H, W = 128, 512, 512
    vol = (torch.rand(D, H, W, generator=torch.Generator().manual_seed(100)) * 255).to(torch.uint8).to(gpu)
    hp = wp = 32
    f_a = torch.zeros(1536, D, hp, wp, dtype=torch.float16, device=gpu)
    f_b = torch.zeros_like(f_a)
    cl = torch.zeros(ops.alloc_rows(D * hp * wp), 1536, dtype=torch.float16, device=gpu)
    vit.features(vol, feats_f16=f_a, d_total=D, d0=0, feats_cl=cl)
    vit.features(vol, feats_f16=f_b, d_total=D, d0=0)
    assert torch.equal(f_a, f_b), "ViT path is not deterministic"
    assert torch.isfinite(f_a.float()).all()
    s = f_a.float()
    assert abs(float(s.mean())) < 0.2 and 0.5 < float(s.std()) < 2.0  # LayerNorm outputs: ~zero mean, ~unit scale
    vit.features(vol.flip(0).contiguous(), feats_f16=f_b, d_total=D, d0=0)
    err = (f_b.flip(1).float() - f_a.float()).abs()  # same slices, different rows of the batch -> different tiles
    assert float(err.max()) <= 5e-2 and float(err.mean()) <= 2e-3, (float(err.max()), float(err.mean()))
    labels = bench.synthetic_labels(gpu, 4)
    out = head.forward(cl, D, hp, wp, labels=labels, want_logits=True)
    p, lg = out["probs"], out["logits"]
    assert tuple(p.shape) == (D, H, W)
    assert float(lg.min()) >= -5.0 and float(lg.max()) <= 5.0
    assert float(p.min()) >= 0.00669 and float(p.max()) <= 0.99331
    mask = labels > -1
    ph = (p >= 0.5) & mask
    lab1 = (labels == 1) & mask
    want = torch.tensor([float((ph & lab1).sum()), float(lab1.sum()), float(ph.sum())])
    assert torch.equal(out["dice_sums"].cpu(), want), (out["dice_sums"].cpu(), want)
    fg = float((p >= 0.5).float().mean())
    assert 0.05 < fg < 0.95, f"degenerate synthetic head (fg fraction {fg})"


def test_head_full_width_midsize_vs_oracle(gpu):
    """BASELINE configs[2] at reduced extent: the FULL-WIDTH head (1536 -> 1024 -> ... -> 1, all four dilation pairs)
    on features [1536, 40, 8, 6] -> 40 x 128 x 96 voxels against the fp32 CPU oracle: logits and masked Dice.
    Depth 40 > the largest dilation (32), so taps at z +- 32 are exercised both inside and outside the volume."""
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine
    from oracle import dice as od
    from oracle import head as oh
    from oracle.make_golden import synth_labels

    head = oh.CryoVITHead()
    oh.rescaled_init_(head, seed=5)
    C, D, h, w = 1536, 40, 8, 6
    feats = torch.randn(C, D, h, w, generator=torch.Generator().manual_seed(3)).half()
    with torch.inference_mode():
        ref = head.forward_volume(feats.float().unsqueeze(0))[0, 0]
    eng = HeadEngine(head.state_dict(), gpu)
    cl = torch.zeros(ops.alloc_rows(D * h * w), C, dtype=torch.float16, device=gpu)
    ops.features_to_channels_last(feats.to(gpu), cl)
    labels = torch.from_numpy(synth_labels(D, 16 * h, 16 * w, seed=4))
    out = eng.forward(cl, D, h, w, labels=labels.to(gpu), want_logits=True)
    got = out["logits"].cpu()
    err = (got - ref).abs()
    assert float(err.max()) <= 5e-2 and float(err.mean()) <= 5e-3, (float(err.max()), float(err.mean()))  # fp16 head vs fp32 CPU
    i, sy, sp = out["dice_sums"].cpu().tolist()
    dice = 2 * i / (sy + sp + 1e-3)
    want = od.dice_metric(torch.sigmoid(ref), labels.float())
    flips = int((((got > 0) != (ref > 0)) & (labels > -1)).sum())
    assert abs(dice - want) <= 1e-3, (dice, want, f"{flips} labelled voxels changed side of the threshold")
    fg = float((ref > 0).float().mean())
    assert 0.05 < fg < 0.95
