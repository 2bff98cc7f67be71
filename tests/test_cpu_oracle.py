"""CPU suite, part 1: the oracle against the committed golden vectors (which were produced from the reference's own
code by AST extraction -- oracle/make_golden.py) and against its own closed forms."""

import numpy as np
import torch

from oracle import dice as o_dice
from oracle import dinov2 as o_vit
from oracle import features as o_feat
from oracle import head as o_head
from oracle import preprocess as o_pre


def test_preprocess_matches_reference_fixture(gold):
    g = gold("preprocess.npz")
    out = o_pre.dino_transform(o_pre.load_scale(g["vol_u8"]))
    assert out.shape[1] == 3 and torch.equal(out[:, 0], out[:, 2])
    assert np.array_equal(out[:, 0].numpy(), g["out_u8"])
    assert np.array_equal(o_pre.dino_transform(g["vol_f32"])[:, 0].numpy(), g["out_f32"])


def test_preprocess_closed_form(gold):
    g = gold("preprocess.npz")
    x = o_pre.pad_to_16(o_pre.load_scale(g["vol_u8"]))
    cf = o_pre.bicubic_14_16_closed_form(x[1])
    assert np.abs(cf - g["out_u8"][1]).max() < 2e-5


def test_vit_fixtures(gold):
    for name, cfg in (("vit_tiny_swiglu", o_vit.VIT_TINY_SWIGLU), ("vit_tiny_mlp", o_vit.VIT_TINY_MLP)):
        g = gold(f"{name}.npz")
        sd = o_vit.init_state_dict(cfg, int(g["seed"]))
        x = torch.from_numpy(g["x"]).unsqueeze(1).expand(-1, 3, -1, -1).contiguous()
        out = o_vit.forward_features(cfg, sd, x)["x_norm_patchtokens"]
        assert torch.allclose(out, torch.from_numpy(g["tokens"]), atol=1e-5)


def test_vit_param_counts():
    sd = o_vit.init_state_dict(o_vit.VITS14_REG, 0)
    assert sum(v.numel() for v in sd.values()) == 22_058_112  # ViT-S/14-reg "22.06 M" (SURVEY s.8c)
    cfg = o_vit.VITG14_REG
    assert cfg.ffn_hidden == 4096 and cfg.head_dim == 64


def test_head_state_dict_layout_and_fixture(gold):
    head = o_head.CryoVITHead()
    keys = list(head.state_dict())
    assert keys[:2] == ["layers.0.weight", "layers.0.bias"]
    assert "layers.2.layers.0.weight" in keys and "layers.5.layers.5.bias" in keys and "output_layer.2.bias" in keys
    assert sum(p.numel() for p in head.parameters()) == 8_401_737
    g = gold("head_narrow.npz")
    narrow = o_head.CryoVITHead(o_head.NARROW_WIDTHS)
    o_head.rescaled_init_(narrow, seed=int(g["seed"]))
    with torch.inference_mode():
        logits = narrow.forward_volume(torch.from_numpy(g["feats"]).unsqueeze(0))[0, 0]
    assert torch.allclose(logits, torch.from_numpy(g["logits"]), atol=1e-5)
    d = o_dice.dice_metric(torch.sigmoid(logits), torch.from_numpy(g["labels"]).float())
    assert abs(d - float(g["dice"])) < 1e-6


def test_dice_fixture(gold):
    g = gold("dice.npz")
    d = o_dice.dice_metric(torch.from_numpy(g["preds"]), torch.from_numpy(g["labels"]).float())
    assert abs(d - float(g["dice"])) < 1e-6


def test_k9_layout_fixture(gold):
    g = gold("k9_layout.npz")

    class Ramp:
        def forward_features(self, vec):
            b, _, H, W = vec.shape
            hp, wp, C = H // 14, W // 14, 6
            s = vec[:, 0, 0, 0].reshape(b, 1, 1)
            return {"x_norm_patchtokens": s * 1000.0 + torch.arange(hp * wp).reshape(1, -1, 1) * 10.0 + torch.arange(C).reshape(1, 1, -1) * 0.125}

    data = torch.zeros(5, 3, 28, 42)
    data[:, :, 0, 0] = torch.arange(5).reshape(5, 1).float()
    assert np.array_equal(o_feat.dino_features(data, Ramp(), 2), g["feats"])


def test_e2e_fixture_is_nondegenerate(gold):
    g = gold("e2e_tiny.npz")
    fg = float((g["probs"] > 0.5).mean())
    assert 0.05 < fg < 0.95 and 0.0 < float(g["dice"]) < 1.0


def test_sam2_hiera_oracle_matches_hf_port():
    """The SAM2 image-encoder restatement (oracle/sam2_hiera.py; the sam2 package is absent) against the installed HF port
    ``transformers.models.sam2`` with identical seeded weights: FPN features to fp32 round-off, position encodings exactly.
    Also pins the block plan of Hiera-L (window lag, q-pool blocks, global blocks) to the HF config's per-block attributes."""
    import torch

    from oracle import sam2_hiera as oh

    cfg = oh.HIERA_TEST
    sd = oh.init_state_dict(cfg, seed=3)
    img = torch.rand(2, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(0))
    errs = oh.hf_cross_check(cfg, sd, img)
    assert max(v for k, v in errs.items() if k.startswith("fpn")) <= 5e-5, errs
    assert max(v for k, v in errs.items() if k.startswith("pos")) == 0.0, errs
    # resize convention of SAM2.forward_features: trilinear over (c,h,w) with c unchanged == bilinear in the plane
    d = torch.rand(1, 2, 3, 40, 56, generator=torch.Generator().manual_seed(1))
    r = oh.resize_input(d, 64)
    assert r.shape == (2, 3, 64, 64)
    assert torch.allclose(r, torch.nn.functional.interpolate(d[0], size=(64, 64), mode="bilinear", align_corners=False), atol=1e-6)
    assert oh.resize_input(d[..., :40, :40], 40).shape == (2, 3, 40, 40)
    # Hiera-L plan against the HF blocks built from the same hyper-parameters (structure only: meta device, no weights)
    from transformers.models.sam2.configuration_sam2 import Sam2HieraDetConfig
    from transformers.models.sam2.modeling_sam2 import Sam2MultiScaleBlock

    L = oh.HIERA_L
    hc = Sam2HieraDetConfig(hidden_size=L.embed_dim, num_attention_heads=L.num_heads, blocks_per_stage=list(L.stages),
                            embed_dim_per_stage=list(L.dims), num_attention_heads_per_stage=list(L.heads),
                            window_size_per_stage=list(L.window_spec), global_attention_blocks=list(L.global_att_blocks),
                            num_query_pool_stages=L.q_pool)
    plan, ends = L.block_plan()
    assert ends == [1, 7, 43, 47] and len(plan) == 48
    t = 0
    with torch.device("meta"):
        for s, nb in enumerate(L.stages):
            for b in range(nb):
                blk = Sam2MultiScaleBlock(hc, s, b, t)
                dim, dout, heads, window, qs = plan[t]
                assert (blk.dim, blk.dim_out, blk.attn.num_attention_heads, blk.window_size, 2 if blk.query_stride else 0) == (dim, dout, heads, window, qs), t
                t += 1


def test_vit_bf16_storage_emulation_is_close_to_fp32_on_benign_weights():
    """``forward_features_bf16_storage`` (exact arithmetic, bf16 at the HIP path's storage points) against the fp32 oracle
    on the golden tiny ViT: it bounds what ANY implementation with this storage plan can reach -- the GPU tolerances of
    tests/test_gpu_model.py (max 1e-1 / mean 1e-2) leave ~6x headroom over it."""
    import torch

    from oracle import dinov2 as o

    for cfg in (o.VIT_TINY_SWIGLU, o.VIT_TINY_MLP):
        sd = o.init_state_dict(cfg, 7)
        x = torch.rand(2, 3, 56, 84, generator=torch.Generator().manual_seed(1))
        a = o.forward_features(cfg, sd, x)["x_norm_patchtokens"]
        b = o.forward_features_bf16_storage(cfg, sd, x)["x_norm_patchtokens"]
        err = (a - b).abs()
        assert 1e-4 < float(err.max()) <= 2e-2 and float(err.mean()) <= 3e-3, (float(err.max()), float(err.mean()))
