"""UNet3D baseline (SURVEY s.8f row N4) on the HIP kernels against the oracle restatement, which oracle/make_golden_unet.py pinned
bit-for-bit against the reference's own classes (models/unet3d.py:12-216) before writing tests/golden/unet3d_narrow.npz.
fp16 storage + fp32 accumulation against an fp32 CPU run: tolerances as for the CryoVIT head."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def hf(t):
    return t.to(torch.float16)


@pytest.mark.parametrize("C,Cout,D,H,W", [(8, 8, 4, 6, 10), (16, 64, 6, 8, 8), (64, 256, 2, 4, 6), (32, 24, 8, 2, 2)])
def test_conv2s2(gpu, C, Cout, D, H, W):
    """nn.Conv3d(C, Cout, 2, stride=2) as an implicit GEMM over 2x2x2 voxel groups."""
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _npad, _pad1, _pad2

    x = hf(rnd(D, H, W, C, seed=80))
    w, b = rnd(Cout, C, 2, 2, 2, seed=81, scale=(8 * C) ** -0.5), rnd(Cout, seed=82)
    nvo = D * H * W // 8
    out = torch.full((nvo + 8, Cout), 7.0, dtype=torch.float16, device=gpu)
    zero = torch.zeros(256, dtype=torch.uint8, device=gpu)
    wp = _pad2(w.permute(0, 2, 3, 4, 1).reshape(Cout, 8 * C), _npad(Cout), 8 * C)
    ops.conv2s2(x.to(gpu), wp.to(gpu), _pad1(b, _npad(Cout)).to(gpu), out, zero, Cin=C, D=D, H=H, W=W, cout=Cout, act=0)
    ref = F.conv3d(x.float().permute(3, 0, 1, 2).unsqueeze(0), hf(w).float(), b, stride=2)[0].permute(1, 2, 3, 0).reshape(nvo, Cout)
    got = out[:nvo].float().cpu()
    assert torch.allclose(got, ref, atol=3e-3, rtol=2e-3), float((got - ref).abs().max())
    assert torch.all(out[nvo:].float() == 7.0)


@pytest.mark.parametrize("c2,c3", [(32, 16), (16, 8), (256, 64)])
def test_conv_transpose_3d(gpu, c2, c3):
    """nn.ConvTranspose3d(c2, c3, 2, stride=2): GEMM with N = 8*c3 and the 3-D pixel-shuffle epilogue."""
    from cryovit_amd._lib import EPI_CONVT
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _npad, _pad1, _pad2

    D, H, W = 3, 4, 5
    nv = D * H * W
    x = hf(rnd(nv, c2, seed=83))
    wt, b = rnd(c2, c3, 2, 2, 2, seed=84, scale=c2**-0.5), rnd(c3, seed=85)
    A = torch.zeros(ops.alloc_rows(nv) * c2 + 4096, dtype=torch.float16)
    A[: nv * c2] = x.reshape(-1)
    A = A.to(gpu)
    a2 = torch.as_strided(A, (ops.alloc_rows(nv), c2), (c2, 1))
    wg = wt.permute(2, 3, 4, 1, 0).reshape(8 * c3, c2)
    out = torch.zeros(2 * D, 2 * H, 2 * W, c3, dtype=torch.float16, device=gpu)
    ops.gemm(EPI_CONVT, a2, _pad2(wg, _npad(8 * c3), ops.round_up(c2, 64)).to(gpu), out, _pad1(b.repeat(8), _npad(8 * c3)).to(gpu),
             m=nv, n=8 * c3, H=H, W=W, cout=c3, act=0, ldc=c3, convt_up_z=1)
    xin = x.float().reshape(D, H, W, c2).permute(3, 0, 1, 2).unsqueeze(0)
    ref = F.conv_transpose3d(xin, hf(wt).float(), b, stride=2)[0].permute(1, 2, 3, 0)
    assert torch.allclose(out.float().cpu(), ref, atol=3e-3, rtol=2e-3), float((out.float().cpu() - ref).abs().max())


@pytest.mark.parametrize("C", [16, 48, 384])
def test_instance_norm_gelu(gpu, C):
    """G = C (InstanceNorm3d, affine, eps 1e-3) with the GELU fused; C / 8 not a power of two (48 -> 6, 384 -> 48 chunks)."""
    from cryovit_amd.engine import ops

    D, H, W = 4, 6, 5
    x = hf(rnd(D, H, W, C, seed=86) * 1.5 + 0.3)
    w, b = rnd(C, seed=87) * 0.2 + 1, rnd(C, seed=88) * 0.2
    out = torch.zeros_like(x, device=gpu)
    stats = torch.zeros(ops.gn_stats_size(C), device=gpu)
    ops.groupnorm(x.to(gpu), w.to(gpu), b.to(gpu), out, stats, nvox=D * H * W, Cdim=C, G=C, eps=1e-3, act=1)
    ref = F.gelu(F.instance_norm(x.float().permute(3, 0, 1, 2).unsqueeze(0), weight=w, bias=b, eps=1e-3))[0].permute(1, 2, 3, 0)
    assert torch.allclose(out.float().cpu(), ref, atol=4e-3, rtol=2e-3), float((out.float().cpu() - ref).abs().max())


def test_concat_and_pointwise_out(gpu):
    from cryovit_amd.engine import ops

    nv, Ca, Cb = 1000, 16, 24
    a, b = hf(rnd(nv, Ca, seed=89)), hf(rnd(nv, Cb, seed=90))
    out = torch.zeros(nv, Ca + Cb, dtype=torch.float16, device=gpu)
    ops.concat_channels(a.to(gpu), b.to(gpu), out, nvox=nv, Ca=Ca, Cb=Cb)
    assert torch.equal(out.cpu(), torch.cat([a, b], 1))
    w = rnd(16, seed=91)
    logits, probs = torch.zeros(nv, device=gpu), torch.zeros(nv, device=gpu)
    ops.pointwise_out((a * 3).to(gpu), w.to(gpu), 0.25, logits, probs, nvox=nv, Cdim=16)
    ref = ((a * 3).float() @ w + 0.25).clip(-5, 5)
    assert torch.allclose(logits.cpu(), ref, atol=1e-4, rtol=1e-4) and torch.allclose(probs.cpu(), torch.sigmoid(ref), atol=1e-5)
    assert float(ref.abs().max()) == 5.0  # the clip is exercised


def test_unet3d_narrow_against_reference_fixture(gpu, gold):
    """The whole model (narrow family, reference-layout state_dict from the oracle's seeded init) on a [20, 40, 37] volume that
    needs padding on every axis: probabilities against the fixture the reference's own classes produced."""
    from cryovit_amd.models import UNet3D
    from oracle import unet3d as ou

    g = gold("unet3d_narrow.npz")
    orc = ou.UNet3D(ou.NARROW_WIDTHS)
    ou.rescaled_init_(orc, seed=int(g["seed"]))
    model = UNet3D(device=gpu, widths=ou.NARROW_WIDTHS)
    model.load_state_dict(orc.state_dict(), strict=True)
    vol = torch.from_numpy(g["vol"])  # [1, D, 1, H, W]

    class Batch:
        tomo_batch = vol

    probs = model(Batch()).cpu()
    want = torch.from_numpy(g["probs"])
    assert tuple(probs.shape) == tuple(want.shape) == (1, 20, 40, 37)
    err = (probs - want).abs()
    # logits: the clipped fp32 logits of the padded volume; fp16 storage through 23 layers with 15 InstanceNorms
    _, lg = model.engine().forward_volume(F.pad(vol[0, :, 0], (0, 48 - 37, 0, 48 - 40, 0, 32 - 20)).to(gpu), want_logits=True)
    lerr = (lg.cpu() - torch.from_numpy(g["logits_padded"])[0, 0]).abs()
    print(f"unet3d narrow: prob err max {float(err.max()):.2e} mean {float(err.mean()):.2e}; logit err max {float(lerr.max()):.2e} mean {float(lerr.mean()):.2e}")
    # measured 1.2e-1 / 3.1e-3 (logits, |logit| up to 5) and 8.2e-3 / 5.0e-4 (probabilities): fp16 storage through 23 layers, 15 of
    # them InstanceNorms that rescale the round-off of 8-channel activations
    assert float(lerr.max()) <= 2e-1 and float(lerr.mean()) <= 6e-3
    assert float(err.max()) <= 1.5e-2 and float(err.mean()) <= 1.5e-3
    agree = ((probs >= 0.5) == (want >= 0.5)).float().mean()
    assert float(agree) >= 0.999
    # torch.cat written in place by the two InstanceNorm passes (cvx_groupnorm_act_strided_f16) == dense tensors + copy kernel, bit for bit
    model.engine().concat_copy = False
    assert torch.equal(model(Batch()).cpu(), probs)
    model.engine().concat_copy = True


def test_unet3d_reference_widths_runs_and_matches_oracle(gpu):
    """Reference widths (16 / 64 / 256 | 384) on a [16, 32, 32] volume against the oracle on the CPU."""
    from cryovit_amd.engine.unet3d import UNet3DEngine
    from oracle import unet3d as ou

    orc = ou.UNet3D(ou.REF_WIDTHS)
    ou.rescaled_init_(orc, seed=17)
    vol = torch.rand(16, 32, 32, generator=torch.Generator().manual_seed(18))
    with torch.no_grad():
        want = torch.clip(orc.forward_volume(vol[None, None]), -5, 5)[0, 0]
    eng = UNet3DEngine(orc.state_dict(), gpu)
    probs, lg = eng.forward_volume(vol.to(gpu), want_logits=True)
    lerr = (lg.cpu() - want).abs()
    print(f"unet3d ref widths: logit err max {float(lerr.max()):.2e} mean {float(lerr.mean()):.2e}")
    assert float(lerr.max()) <= 2e-1 and float(lerr.mean()) <= 8e-3
    assert torch.allclose(probs.cpu(), torch.sigmoid(lg.cpu()), atol=1e-6)


def test_unet3d_evaluation_protocol(gpu, gold):
    """``test_step`` / ``validation_step`` of the shared evaluation protocol (base_model.py:91-112, 153-165, 176-241) on the UNet3D
    mirror: TomogramData -> collate_fn -> masked prediction -> DiceLoss + DiceMetric, against the same quantities computed from the
    reference-pinned fixture's probabilities on the CPU."""
    from cryovit_amd.datasets import collate_fn
    from cryovit_amd.models import DiceLoss, DiceMetric, UNet3D
    from cryovit_amd.models.metrics import dice_from_sums
    from cryovit_amd.types import TomogramData
    from oracle import train_pieces as tp
    from oracle import unet3d as ou

    g = gold("unet3d_narrow.npz")
    orc = ou.UNet3D(ou.NARROW_WIDTHS)
    ou.rescaled_init_(orc, seed=int(g["seed"]))
    model = UNet3D(device=gpu, widths=ou.NARROW_WIDTHS, losses={"dice_loss": DiceLoss()}, metrics={"dice_metric": DiceMetric()})
    model.load_state_dict(orc.state_dict(), strict=True)
    vol = torch.from_numpy(g["vol"])[0, :, 0]  # [D, H, W]
    D, H, W = vol.shape
    labels = (torch.rand(D, H, W, generator=torch.Generator().manual_seed(3)) < 0.4).to(torch.int8)
    labels[:3] = -1
    item = TomogramData(sample="S", tomo_name="t.hdf", split_id=0, data=vol[None].clone(), label=labels, aux_data={"data": vol.numpy()})
    batch = collate_fn([item])
    assert tuple(batch.tomo_batch.shape) == (1, D, 1, H, W)
    res = model.test_step(batch)
    want = torch.from_numpy(g["probs"])[0]
    assert float((torch.from_numpy(res.preds[0]) - want).abs().max()) <= 1.5e-2
    m = labels > -1
    ph = (want >= 0.5).float()
    dice_ref = dice_from_sums(float((labels.float() * ph)[m].sum()), float(labels.float()[m].sum()), float(ph[m].sum()))
    loss_ref = float(tp.masked_dice_loss(want, labels.float()))
    assert abs(res.metrics["dice_metric"] - dice_ref) <= 5e-3 and abs(res.losses["dice_loss"] - loss_ref) <= 2e-3
    assert abs(res.losses["total"] - res.losses["dice_loss"]) < 1e-12
    assert abs(model.validation_step(batch) - res.losses["dice_loss"]) < 1e-7


def test_eval_model_entry_point_unet3d(gpu, tmp_path):
    """``python -m cryovit_amd.training.eval_model model=unet3d`` from files on disk (weights.pt in the reference's state_dict
    layout, raw uint8 ``data`` volumes + ``labels/mito``): the metrics CSV and the prediction files against the fp32 CPU oracle
    (restatement pinned bit-for-bit to the reference's UNet3D classes) on the same weights."""
    import csv

    from cryovit_amd import io
    from cryovit_amd.training import eval_model as entry
    from oracle import dice as od
    from oracle import unet3d as ou

    ref = ou.UNet3D(ou.REF_WIDTHS)
    ou.rescaled_init_(ref, seed=23)
    data_dir, exp_dir = tmp_path / "data", tmp_path / "exp"
    name = "single_any_unet3d_mito"
    (exp_dir / name / "Q109" / "split_1").mkdir(parents=True)
    torch.save(ref.state_dict(), exp_dir / name / "Q109" / "split_1" / "weights.pt")
    (data_dir / "csv").mkdir(parents=True)
    rng = np.random.default_rng(41)
    rows, truth = [], {}
    for i, (D, split) in enumerate(((12, 1), (16, 0))):
        vol = rng.integers(0, 256, size=(D, 24, 20), dtype=np.uint8)
        lab = rng.integers(-1, 2, size=(D, 24, 20)).astype(np.int8)
        p = data_dir / "tomograms" / "Q109" / f"u{i}.hdf"
        p.parent.mkdir(parents=True, exist_ok=True)
        with io.FileWriter(p) as f:
            f.create_dataset("data", vol, compression="gzip")
            f.create_dataset("labels/mito", lab, compression="gzip")
        rows.append({"sample": "Q109", "tomo_name": f"u{i}.hdf", "split_id": split})
        truth[f"u{i}.hdf"] = (vol, lab)
    with open(data_dir / "csv" / "splits.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample", "tomo_name", "split_id"])
        w.writeheader()
        w.writerows(rows)
    entry.main(["model=unet3d", "datamodule=single", "datamodule.sample=Q109", "datamodule.split_id=1", "label_key=mito",
                f"paths.model_dir={tmp_path}", f"paths.data_dir={data_dir}", f"paths.exp_dir={exp_dir}", f"paths.results_dir={tmp_path / 'results'}"])
    got = list(csv.DictReader(open(tmp_path / "results" / "results" / name / "Q109_1.csv")))
    assert [r["tomo_name"] for r in got] == ["u0.hdf"]
    vol, lab = truth["u0.hdf"]
    with torch.no_grad():
        probs = ref.forward_tomo_batch(torch.from_numpy(vol.astype(np.float32) / 255.0)[None, :, None])[0]  # uint8 -> /255 (tomo_dataset.py:118)
    labt = torch.from_numpy(lab).float()
    pred = tmp_path / "results" / "predictions" / name / "Q109" / "u0.hdf"
    pp = io.read_dataset(pred, "mito_preds")
    assert pp.dtype == np.float32 and pp.shape == lab.shape
    err = np.abs(pp - probs.numpy())
    print(f"unet3d eval_model: prob err max {err.max():.2e} mean {err.mean():.2e}")
    assert err.max() <= 3e-2 and err.mean() <= 2e-3
    # Dice of thresholded predictions: exact given the GPU's own probabilities, and close to the oracle's where no voxel sits on the threshold
    assert abs(float(got[0]["dice_metric"]) - od.dice_metric(torch.from_numpy(pp), labt)) <= 1e-6
    assert abs(float(got[0]["dice_metric"]) - od.dice_metric(probs, labt)) <= 2e-2


def test_run_inference_unet3d(gpu, tmp_path):
    """``run_inference`` (``cryovit infer``) with a UNet3D ``.model`` container on a raw tomogram file: uint8 segmentation against the
    oracle's probabilities (disagreement only where the oracle sits within 3e-2 of the threshold)."""
    from cryovit_amd import io
    from cryovit_amd.run.infer_model import run_inference
    from cryovit_amd.types import ModelType
    from cryovit_amd.utils import load_data, save_model_from_weights
    from oracle import unet3d as ou

    ref = ou.UNet3D(ou.REF_WIDTHS)
    ou.rescaled_init_(ref, seed=29)
    torch.save(ref.state_dict(), tmp_path / "weights.pt")
    save_model_from_weights("unet_demo", "mito", ModelType.UNET3D, tmp_path / "weights.pt", tmp_path / "unet.model")
    vol = np.random.default_rng(12).integers(0, 256, size=(10, 20, 24), dtype=np.uint8)
    (tmp_path / "in").mkdir()
    with io.FileWriter(tmp_path / "in" / "raw0.hdf") as f:
        f.create_dataset("data", vol, compression="gzip")
    thr = 0.6
    paths = run_inference([tmp_path / "in" / "raw0.hdf"], tmp_path / "unet.model", tmp_path / "out", threshold=thr)
    assert paths == [tmp_path / "out" / "raw0.hdf"] and sorted(io.list_keys(paths[0])) == ["data", "mito_preds"]
    seg = io.read_dataset(paths[0], "mito_preds")
    assert seg.dtype == np.uint8 and seg.shape == vol.shape
    x = torch.from_numpy(np.ascontiguousarray(load_data(tmp_path / "in" / "raw0.hdf", key="data")[0].squeeze(0), dtype=np.float32))  # the loader's normalisation
    with torch.no_grad():
        probs = ref.forward_tomo_batch(x[None, :, None])[0].numpy()
    want = (probs >= thr).astype(np.uint8)
    bad = seg != want
    assert bad.mean() <= 0.02 and np.all(np.abs(probs[bad] - thr) < 3e-2), (bad.mean(), np.abs(probs[bad] - thr).max() if bad.any() else 0)
