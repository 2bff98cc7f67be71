"""Generate tests/golden/*.npz and pin the oracle (TEST INFRASTRUCTURE).

Run in the BUILD container only (needs /root/reference; the GPU box has
neither that tree nor any need for this script):

    python -m oracle.make_golden

What "pin" means here.  The reference package cannot be imported offline
(``cryovit._version`` is generated, and hydra/tensordict/lightning/h5py/
torchvision/torchmetrics are absent -- ordinary ImportErrors), so
the hot-path functions are executed *where they lie* by AST extraction: the
function/class definition is parsed out of the reference file, compiled, and
run with only ``torch``/``numpy`` (plus trivial stubs for names that appear in
base-class lists / annotations) in its namespace.  Outputs -- never source -- are
written as fixtures.  Every oracle restatement is asserted equal to the
extracted reference code on the same inputs before a fixture is saved.

The ViT has no reference source in the tree (third-party hub repo); the oracle
is cross-checked against the installed HF port instead (``check_vs_hf``).
"""

from __future__ import annotations

import ast
import hashlib
import logging
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from oracle import dice as o_dice
from oracle import dinov2 as o_vit
from oracle import features as o_feat
from oracle import head as o_head
from oracle import preprocess as o_pre

REF = Path("/root/reference/src/cryovit")
GOLD = Path(__file__).resolve().parent.parent / "tests" / "golden"


# ----------------------------------------------------------------------------------------------
# AST extraction helpers
# ----------------------------------------------------------------------------------------------
def _parse(rel: str) -> ast.Module:
    return ast.parse((REF / rel).read_text())


def _find(tree: ast.AST, name: str, kind=(ast.FunctionDef, ast.ClassDef)):
    for node in ast.walk(tree):
        if isinstance(node, kind) and node.name == name:
            return node
    raise KeyError(name)


def _exec_nodes(nodes, ns: dict, filename: str) -> dict:
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, filename, "exec"), ns)
    return ns


def ref_head_classes():
    """``SynthesisBlock`` and ``CryoVIT`` from models/cryovit.py:10-83.

    ``BaseModel`` (a LightningModule) is replaced by a bare ``nn.Module`` that
    swallows kwargs; ``BatchedTomogramData`` only appears as an annotation.
    """
    tree = _parse("models/cryovit.py")

    class BaseModel(nn.Module):
        def __init__(self, **kwargs):
            super().__init__()

    ns = {"torch": torch, "nn": nn, "Tensor": Tensor, "BaseModel": BaseModel, "BatchedTomogramData": object}
    _exec_nodes([_find(tree, "SynthesisBlock"), _find(tree, "CryoVIT")], ns, "ref:models/cryovit.py")
    return ns["CryoVIT"], ns["SynthesisBlock"]


def ref_dino_transform():
    tree = _parse("datasets/vit_dataset.py")
    fn = _find(_find(tree, "VITDataset"), "_dino_transform")
    ns = {"np": np, "torch": torch, "F": F, "logging": logging, "DINO_PATCH_SIZE": 14, "NDArray": np.ndarray}
    _exec_nodes([fn], ns, "ref:datasets/vit_dataset.py")
    fake_self = types.SimpleNamespace(_printed_resize_warning=True)
    return lambda data: ns["_dino_transform"](fake_self, data)


def ref_dino_features():
    """``_dino_features`` from run/dino_features.py:31-64; ``.cuda()`` is made a
    no-op through a Tensor subclass (there is no GPU in the build container)."""
    tree = _parse("run/dino_features.py")
    fn = _find(tree, "_dino_features")
    ns = {"np": np, "torch": torch, "NDArray": np.ndarray}
    _exec_nodes([fn], ns, "ref:run/dino_features.py")

    class HostTensor(torch.Tensor):
        def cuda(self, *a, **k):
            return self

    def run(data: torch.Tensor, model, batch_size: int):
        return ns["_dino_features"](data.as_subclass(HostTensor), model, batch_size)

    return run


def ref_dice_update():
    tree = _parse("models/metrics.py")
    fn = _find(_find(tree, "DiceMetric"), "update")
    ns = {"torch": torch, "Tensor": Tensor}
    _exec_nodes([fn], ns, "ref:models/metrics.py")

    def run(y_pred, y_true, thresh=0.5):
        st = types.SimpleNamespace(thresh=thresh, dice_score=torch.tensor(0.0), total=torch.tensor(0.0))
        ns["update"](st, y_pred, y_true)
        return float(st.dice_score / st.total)

    return run


def ref_masked_predict():
    tree = _parse("models/base_model.py")
    fn = _find(_find(tree, "BaseModel"), "_masked_predict")
    ns = {"torch": torch, "Tensor": Tensor, "BatchedTomogramData": object}
    _exec_nodes([fn], ns, "ref:models/base_model.py")

    def run(preds_full: Tensor, labels: Tensor):
        class FakeSelf:
            def __call__(self, batch):
                return preds_full

        batch = types.SimpleNamespace(labels=labels, aux_data=None)
        return ns["_masked_predict"](FakeSelf(), batch)

    return run


# ----------------------------------------------------------------------------------------------
# HF cross-check for the ViT restatement
# ----------------------------------------------------------------------------------------------
def to_hf_state_dict(cfg: o_vit.VitCfg, sd: dict) -> dict:
    C = cfg.dim
    out = {
        "embeddings.cls_token": sd["cls_token"],
        "embeddings.mask_token": sd["mask_token"],
        "embeddings.register_tokens": sd["register_tokens"],
        "embeddings.position_embeddings": sd["pos_embed"],
        "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
        "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
        "layernorm.weight": sd["norm.weight"],
        "layernorm.bias": sd["norm.bias"],
    }
    for i in range(cfg.depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        w, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nm in enumerate(("query", "key", "value")):
            out[q + f"attention.attention.{nm}.weight"] = w[j * C : (j + 1) * C]
            out[q + f"attention.attention.{nm}.bias"] = b[j * C : (j + 1) * C]
        out[q + "attention.output.dense.weight"] = sd[p + "attn.proj.weight"]
        out[q + "attention.output.dense.bias"] = sd[p + "attn.proj.bias"]
        out[q + "layer_scale1.lambda1"] = sd[p + "ls1.gamma"]
        out[q + "layer_scale2.lambda1"] = sd[p + "ls2.gamma"]
        for nm in ("norm1", "norm2"):
            out[q + nm + ".weight"] = sd[p + nm + ".weight"]
            out[q + nm + ".bias"] = sd[p + nm + ".bias"]
        if cfg.ffn == "swiglu":
            out[q + "mlp.weights_in.weight"] = sd[p + "mlp.w12.weight"]
            out[q + "mlp.weights_in.bias"] = sd[p + "mlp.w12.bias"]
            out[q + "mlp.weights_out.weight"] = sd[p + "mlp.w3.weight"]
            out[q + "mlp.weights_out.bias"] = sd[p + "mlp.w3.bias"]
        else:
            for a, b_ in (("fc1", "fc1"), ("fc2", "fc2")):
                out[q + f"mlp.{a}.weight"] = sd[p + f"mlp.{b_}.weight"]
                out[q + f"mlp.{a}.bias"] = sd[p + f"mlp.{b_}.bias"]
    return out


def check_vs_hf(cfg: o_vit.VitCfg, sd: dict, x: torch.Tensor) -> float:
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel

    hf_cfg = Dinov2WithRegistersConfig(
        hidden_size=cfg.dim,
        num_hidden_layers=cfg.depth,
        num_attention_heads=cfg.heads,
        mlp_ratio=4 if cfg.ffn == "swiglu" else cfg.ffn_hidden // cfg.dim,
        use_swiglu_ffn=cfg.ffn == "swiglu",
        num_register_tokens=cfg.n_reg,
        image_size=cfg.pos_grid * cfg.patch,
        patch_size=cfg.patch,
        layer_norm_eps=cfg.ln_eps,
        hidden_act="gelu",
    )
    with torch.device("meta"):
        model = Dinov2WithRegistersModel(hf_cfg)
    model = model.to_empty(device="cpu")
    missing, unexpected = model.load_state_dict(to_hf_state_dict(cfg, sd), strict=False)
    assert not unexpected, unexpected
    assert all("pooler" in k for k in missing), missing
    model.eval()
    with torch.inference_mode():
        hf = model(pixel_values=x).last_hidden_state[:, 1 + cfg.n_reg :]
        mine = o_vit.forward_features(cfg, sd, x)["x_norm_patchtokens"]
    return float((hf - mine).abs().max())


# ----------------------------------------------------------------------------------------------
def sd_checksum(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def synth_labels(D: int, H: int, W: int, seed: int) -> np.ndarray:
    """int8 {-1,0,1}: top/bottom eighth of z unlabeled, one ellipsoid blob foreground."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    c = np.array([D / 2, H / 2, W / 2]) + rng.uniform(-0.1, 0.1, 3) * np.array([D, H, W])
    r = np.array([D * 0.3, H * 0.3, W * 0.3])
    blob = ((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2 < 1.0
    lab = blob.astype(np.int8)
    k = max(1, D // 8)
    lab[:k] = -1
    lab[D - k :] = -1
    return lab


def main() -> None:
    assert REF.exists(), "run in the build container (needs /root/reference)"
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    report = {}

    # ---- (i) pre-processing: vit_dataset.py:90-123 ------------------------------------------
    rng = np.random.default_rng(10)
    vol_u8 = rng.integers(0, 256, size=(3, 40, 52), dtype=np.uint8)  # needs padding -> 48x64 -> 42x56
    ref_tf = ref_dino_transform()
    x_f = o_pre.load_scale(vol_u8)
    ref_out = ref_tf(x_f)
    mine = o_pre.dino_transform(x_f)
    assert torch.equal(ref_out, mine), "oracle.preprocess != reference _dino_transform"
    assert torch.equal(ref_out[:, 0], ref_out[:, 1]) and torch.equal(ref_out[:, 0], ref_out[:, 2])
    cf = o_pre.bicubic_14_16_closed_form(o_pre.pad_to_16(x_f)[0])
    err = float(np.abs(cf - ref_out[0, 0].numpy()).max())
    assert err < 2e-5, err  # fp32 source-coordinate rounding in F.interpolate
    report["preprocess closed-form vs F.interpolate max abs"] = err
    vol_f32 = rng.random((2, 32, 48), dtype=np.float32)  # no padding path
    ref_out2 = ref_tf(vol_f32)
    assert torch.equal(ref_out2, o_pre.dino_transform(vol_f32))
    np.savez_compressed(
        GOLD / "preprocess.npz",
        vol_u8=vol_u8, out_u8=ref_out[:, 0].numpy(), vol_f32=vol_f32, out_f32=ref_out2[:, 0].numpy(),
    )

    # ---- (ii) ViT restatement vs HF port; fixtures for tiny configs ---------------------------
    for name, cfg, seed in (("vit_tiny_swiglu", o_vit.VIT_TINY_SWIGLU, 21), ("vit_tiny_mlp", o_vit.VIT_TINY_MLP, 22)):
        sd = o_vit.init_state_dict(cfg, seed)
        x = torch.rand(2, 1, 56, 84, generator=torch.Generator().manual_seed(seed + 100)).expand(-1, 3, -1, -1).contiguous()
        e = check_vs_hf(cfg, sd, x)
        assert e < 5e-5, (name, e)
        report[f"{name} oracle vs HF max abs"] = e
        out = o_vit.forward_features(cfg, sd, x)["x_norm_patchtokens"]
        np.savez_compressed(
            GOLD / f"{name}.npz", seed=seed, x=x[:, 0].numpy(), tokens=out.numpy(), sd_sha256=sd_checksum(sd)
        )
    # ViT-S/14-reg (config 1's encoder), one 224x224 slice, vs HF
    sd = o_vit.init_state_dict(o_vit.VITS14_REG, 23)
    x = torch.rand(1, 1, 224, 224, generator=torch.Generator().manual_seed(123)).expand(-1, 3, -1, -1).contiguous()
    e = check_vs_hf(o_vit.VITS14_REG, sd, x)
    assert e < 2e-4, e
    report["vit_s14_reg oracle vs HF max abs"] = e

    # ---- (iii) head: cryovit.py:10-83 ---------------------------------------------------------
    RefCryoVIT, RefSB = ref_head_classes()
    ref_full = RefCryoVIT()
    mine_full = o_head.CryoVITHead()
    assert [(k, tuple(v.shape)) for k, v in ref_full.state_dict().items()] == [
        (k, tuple(v.shape)) for k, v in mine_full.state_dict().items()
    ], "state_dict layout differs from the reference"
    n_params = sum(p.numel() for p in ref_full.parameters())
    assert n_params == 8_401_737, n_params
    o_head.rescaled_init_(mine_full, seed=5)
    ref_full.load_state_dict(mine_full.state_dict())
    xin = torch.randn(1, 1536, 4, 2, 2, generator=torch.Generator().manual_seed(3))
    with torch.inference_mode():
        a, b = ref_full.forward_volume(xin), mine_full.forward_volume(xin)
    assert torch.equal(a, b), "oracle.head full-width != reference CryoVIT.forward_volume"
    report["head full-width logits range"] = (float(a.min()), float(a.max()))
    # forward(): permute + squeeze + sigmoid (cryovit.py:42-49)
    with torch.inference_mode():
        pr = ref_full.forward(types.SimpleNamespace(tomo_batch=xin.permute(0, 2, 1, 3, 4)))
        pm = mine_full.forward_tomo_batch(xin.permute(0, 2, 1, 3, 4))
    assert torch.equal(pr, pm)

    narrow = o_head.CryoVITHead(o_head.NARROW_WIDTHS)
    narrow_ref = o_head.CryoVITHead(o_head.NARROW_WIDTHS, block_cls=RefSB)
    o_head.rescaled_init_(narrow, seed=6)
    narrow_ref.load_state_dict(narrow.state_dict())
    xin = torch.randn(1, 128, 8, 4, 4, generator=torch.Generator().manual_seed(7))
    with torch.inference_mode():
        ln, lr = narrow.forward_volume(xin), narrow_ref.forward_volume(xin)
    assert torch.equal(ln, lr)
    fg = float((ln > 0).float().mean())
    report["head narrow fg fraction"] = fg
    assert 0.05 < fg < 0.95, fg
    labels = synth_labels(8, 64, 64, seed=4)
    probs = torch.sigmoid(ln[0, 0])
    d = o_dice.dice_metric(probs, torch.from_numpy(labels).float())
    np.savez_compressed(
        GOLD / "head_narrow.npz", seed=6, feats=xin[0].numpy(), logits=ln[0, 0].numpy(), labels=labels, dice=d,
        sd_sha256=sd_checksum(narrow.state_dict()),
    )
    sb = RefSB(32, 16, 8, 2, 1)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for p in sb.parameters():
            p.normal_(0.0, 0.2, generator=g)
        sb.layers[0].weight.add_(1.0)
    xs = torch.randn(1, 32, 6, 8, 8, generator=g)
    with torch.inference_mode():
        ys = sb(xs)
    np.savez_compressed(
        GOLD / "synthesis_block.npz", x=xs[0].numpy(), y=ys[0].numpy(),
        **{k.replace(".", "_"): v.numpy() for k, v in sb.state_dict().items()},
    )

    # ---- (iv) masked Dice: base_model.py:91-112 + metrics.py:30-53 ---------------------------
    g = torch.Generator().manual_seed(9)
    preds = torch.rand(1, 6, 16, 16, generator=g)
    labels = torch.from_numpy(synth_labels(6, 16, 16, seed=11)).float().unsqueeze(0)
    mp = ref_masked_predict()(preds, labels)
    ref_d = ref_dice_update()(mp["preds"], mp["labels"])
    mine_d = o_dice.dice_metric(preds, labels)
    assert abs(ref_d - mine_d) < 1e-6, (ref_d, mine_d)
    a, b = o_dice.masked_select_pair(preds, labels)
    assert torch.equal(a, mp["preds"]) and torch.equal(b, mp["labels"])
    np.savez_compressed(GOLD / "dice.npz", preds=preds[0].numpy(), labels=labels[0].numpy().astype(np.int8), dice=ref_d)

    # ---- (v) K9 layout: run/dino_features.py:31-64 -------------------------------------------
    class RampModel:  # token value encodes (slice, row, col, channel) -> catches any transposition
        def forward_features(self, vec):
            b, _, H, W = vec.shape
            hp, wp, C = H // 14, W // 14, 6
            s = vec[:, 0, 0, 0].reshape(b, 1, 1)
            tok = torch.arange(hp * wp).reshape(1, -1, 1) * 10.0
            ch = torch.arange(C).reshape(1, 1, -1) * 0.125
            return {"x_norm_patchtokens": s * 1000.0 + tok + ch}

    data = torch.zeros(5, 3, 28, 42)
    data[:, :, 0, 0] = torch.arange(5).reshape(5, 1).float()
    ref_f = ref_dino_features()(data, RampModel(), 2)
    mine_f = o_feat.dino_features(data, RampModel(), 2)
    assert ref_f.dtype == np.float16 and ref_f.shape == (6, 5, 2, 3)
    assert np.array_equal(ref_f, mine_f)
    np.savez_compressed(GOLD / "k9_layout.npz", feats=ref_f)

    # ---- (vi) tiny end-to-end: raw u8 volume -> features -> head logits -> dice ---------------
    cfg = o_vit.VIT_TINY_SWIGLU
    sd = o_vit.init_state_dict(cfg, 31)
    vol = np.random.default_rng(32).integers(0, 256, size=(8, 64, 64), dtype=np.uint8)
    x = o_pre.dino_transform(o_pre.load_scale(vol))
    feats = o_feat.dino_features(x, o_vit.OracleDino(cfg, sd), 3)
    head = o_head.CryoVITHead(o_head.NARROW_WIDTHS)
    o_head.rescaled_init_(head, seed=33)
    with torch.inference_mode():
        probs = head.forward_tomo_batch(o_feat.collate_features(feats))[0]
    labels = synth_labels(8, 64, 64, seed=34)
    d = o_dice.dice_metric(probs, torch.from_numpy(labels).float())
    report["e2e tiny dice / fg"] = (d, float((probs > 0.5).float().mean()))
    np.savez_compressed(
        GOLD / "e2e_tiny.npz", vol=vol, feats=feats, probs=probs.numpy(), labels=labels, dice=d,
        vit_seed=31, head_seed=33,
    )

    for k, v in report.items():
        print(f"{k}: {v}")
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    sys.exit(main())
