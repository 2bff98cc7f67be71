"""Training-side pieces (SURVEY s.8f row N4) on the GPU, through the C ABI: Dice loss value + gradient against the fixture made
from the reference's own DiceLoss, the fused sigmoid/clip chain against torch autograd, AdamW against torch.optim.AdamW, and
a short optimisation loop built from the two against the same loop on the CPU oracle."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dice_loss_matches_reference_fixture(gpu, gold):
    from cryovit_amd.models import DiceLoss

    g = gold("train_pieces.npz")
    probs = torch.from_numpy(g["dice_probs"]).to(gpu).requires_grad_(True)
    labels = torch.from_numpy(g["dice_labels"]).to(gpu)  # int8, -1 = ignore: masked inside the kernel
    loss = DiceLoss()(probs, labels)
    assert abs(float(loss.detach()) - float(g["dice_loss"])) <= 2e-6  # fp32 sums in a different (fixed) order
    (3.0 * loss).backward()
    want = 3.0 * torch.from_numpy(g["dice_grad"])
    assert torch.allclose(probs.grad.cpu(), want, rtol=2e-5, atol=1e-9)
    assert torch.all(probs.grad[labels < 0] == 0)
    # the reference's call shape: masked vectors [n, 1] with float labels
    m = labels > -1
    yp = torch.masked_select(probs.detach(), m).view(-1, 1).requires_grad_(True)
    loss2 = DiceLoss()(yp, torch.masked_select(labels, m).view(-1, 1).float())
    assert abs(float(loss2) - float(g["dice_loss"])) <= 2e-6
    loss2.backward()
    assert torch.allclose(yp.grad.view(-1).cpu(), torch.from_numpy(g["dice_grad"])[m.cpu()], rtol=2e-5, atol=1e-9)
    # bitwise reproducible (no atomics)
    outs = [float(DiceLoss()(probs.detach(), labels)) for _ in range(3)]
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("n", [1, 3, 1026, 1 << 20, 33554432 // 8 + 5])
def test_dice_loss_sizes_and_logit_chain(gpu, n):
    """Ragged sizes (n % 4 != 0, fewer elements than one block, more than CVX_DICE_BLOCKS blocks' worth) and the fused
    d loss / d logit through p = sigmoid(clip(x, -5, 5)) against torch autograd of the oracle expression."""
    from cryovit_amd.models.losses import dice_loss_and_logit_grad
    from oracle import train_pieces as tp

    gen = torch.Generator().manual_seed(n % 1000)
    raw = torch.randn(n, generator=gen) * 4.0  # a good share beyond the +-5 clip
    labels = torch.randint(-1, 2, (n,), generator=gen)
    x = raw.clone().requires_grad_(True)
    clipped = torch.clip(x, -5.0, 5.0)
    probs = torch.sigmoid(clipped)
    ref = tp.masked_dice_loss(probs.double(), labels.double())
    ref.backward()
    loss, grad = dice_loss_and_logit_grad(probs.detach().to(gpu), clipped.detach().to(gpu), labels.to(torch.int8).to(gpu), 1.0)
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    inside = raw.abs() < 5.0
    assert torch.allclose(grad.cpu()[inside], x.grad[inside].float(), rtol=1e-4, atol=1e-12 + 1e-4 * float(x.grad.abs().max()))
    assert torch.all(grad.cpu()[~inside] == 0)


@pytest.mark.parametrize("n,kw", [(8401737, dict(lr=1e-4, weight_decay=1e-2)), (1027, dict(lr=3e-3, weight_decay=0.0, betas=(0.8, 0.95), eps=1e-6)),
                                  (2, dict(lr=1e-2, weight_decay=0.1))])
def test_adamw_matches_torch(gpu, n, kw):
    """cvx_adamw_step against torch.optim.AdamW (CPU, single-tensor path) over 6 steps; n = the head's parameter count, a
    ragged size and a tiny one.  Same operations in the same order; hipcc contracts a*b+c into one FMA where torch rounds
    twice, hence a few ulp instead of bit equality."""
    from cryovit_amd.engine import ops

    gen = torch.Generator().manual_seed(9)
    p0 = torch.randn(n, generator=gen)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], foreach=False, **kw)
    b1, b2 = kw.get("betas", (0.9, 0.999))
    p, m, v = p0.to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    for step in range(1, 7):
        g = torch.randn(n, generator=gen) * (0.1 if step % 2 else 3.0)
        ref_p.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g.to(gpu), m, v, lr=kw["lr"], beta1=b1, beta2=b2, eps=kw.get("eps", 1e-8), weight_decay=kw["weight_decay"], step=step)
    assert torch.allclose(p.cpu(), ref_p.detach(), rtol=2e-6, atol=2e-7), float((p.cpu() - ref_p.detach()).abs().max())
    st = opt.state[ref_p]
    # (the moments: values near zero after cancellation between steps carry the absolute round-off of their O(1) terms)
    assert torch.allclose(m.cpu(), st["exp_avg"], rtol=2e-6, atol=1e-6) and torch.allclose(v.cpu(), st["exp_avg_sq"], rtol=4e-6, atol=1e-8)


def test_optimisation_loop_dice_plus_adamw(gpu):
    """The two pieces together: a bias volume added to fixed logits is trained with DiceLoss + the fused AdamW (parameters
    re-pointed into the flat buffers, gradients accumulated in place); the loss trajectory follows the CPU loop built from the
    oracle's Dice loss and torch.optim.AdamW, and goes down."""
    from cryovit_amd.models import DiceLoss
    from cryovit_amd.training.optim import AdamW
    from oracle import train_pieces as tp

    gen = torch.Generator().manual_seed(21)
    base = torch.randn(4, 24, 40, generator=gen)
    labels = (torch.rand(4, 24, 40, generator=gen) < 0.3).float()
    labels[0] = -1.0
    w_gpu = [torch.nn.Parameter(torch.zeros(4, 24, 40, device=gpu)), torch.nn.Parameter(torch.zeros(3, device=gpu))]  # (+ an odd-sized tensor)
    w_cpu = [torch.nn.Parameter(torch.zeros(4, 24, 40)), torch.nn.Parameter(torch.zeros(3))]
    opt_g, opt_c = AdamW(w_gpu, lr=5e-2, weight_decay=1e-2), torch.optim.AdamW(w_cpu, lr=5e-2, weight_decay=1e-2, foreach=False)
    loss_fn = DiceLoss()
    lg, lc = [], []
    for _ in range(12):
        opt_g.zero_grad()
        pg = torch.sigmoid(torch.clip(base.to(gpu) + w_gpu[0] + w_gpu[1].sum(), -5.0, 5.0))
        loss = loss_fn(pg, labels.to(gpu))
        loss.backward()
        opt_g.step()
        lg.append(float(loss))
        opt_c.zero_grad()
        pc = torch.sigmoid(torch.clip(base + w_cpu[0] + w_cpu[1].sum(), -5.0, 5.0))
        loss_c = tp.masked_dice_loss(pc, labels)
        loss_c.backward()
        opt_c.step()
        lc.append(float(loss_c))
    assert np.allclose(lg, lc, rtol=2e-4, atol=2e-5), (lg, lc)
    assert lg[-1] < lg[0] - 0.05
    assert torch.allclose(w_gpu[0].detach().cpu(), w_cpu[0].detach(), atol=2e-3)
    assert w_gpu[0].data_ptr() == opt_g.flat_p.data_ptr()  # the parameter IS a view of the flat buffer


def test_adamw_stays_attached_or_raises(gpu):
    """Gradients dropped by ``zero_grad(set_to_none=True)`` are re-attached to the flat buffer (the step still sees what
    autograd then accumulates); a parameter or gradient that moved away from its slice is an error, not a silent no-op; the
    state dict carries the flat layout and a foreign one is refused."""
    from cryovit_amd.training.optim import AdamW

    lin = torch.nn.Linear(5, 3).to(gpu)
    opt = AdamW(lin.parameters(), lr=1e-1, weight_decay=0.0)
    x = torch.randn(7, 5, device=gpu)
    lin.zero_grad(set_to_none=True)          # torch's default: .grad = None
    opt.step()                               # re-attaches, steps with zero gradients: parameters unchanged
    assert all(p.grad is not None and p.grad.data_ptr() == opt.flat_g.data_ptr() + o * 4 for p, o in zip(opt.params, opt.offsets))
    before = opt.flat_p.clone()
    lin(x).square().sum().backward()         # accumulates into the re-attached slices
    assert float(opt.flat_g.abs().sum()) > 0
    opt.step()
    assert not torch.equal(opt.flat_p, before)
    sd = opt.state_dict()
    assert sd["n"] == opt.n and sd["offsets"] == opt.offsets
    opt.load_state_dict(sd)
    other = AdamW(torch.nn.Linear(5, 4).to(gpu).parameters())
    with pytest.raises(ValueError):
        other.load_state_dict(sd)
    lin.weight.grad = torch.zeros_like(lin.weight)  # a replaced gradient tensor
    with pytest.raises(RuntimeError):
        opt.step()
    lin.weight.grad = None
    lin.weight.data = lin.weight.data.clone()       # a moved parameter (model.to(), assign=True load ...)
    with pytest.raises(RuntimeError):
        opt.step()


@pytest.mark.parametrize("gamma", [2, 1.5])
def test_focal_loss_against_oracle(gpu, gamma):
    """FocalLoss (losses.py:35-64) against the oracle restatement of torchvision's sigmoid_focal_loss (float64): value and autograd
    gradient, masked in the kernel; alpha = (n - sum y) / n is a constant of the gradient like the reference's weight.item()."""
    from cryovit_amd.models.losses import FocalLoss
    from oracle import train_pieces as tp

    gen = torch.Generator().manual_seed(31)
    n = 50021
    x = torch.rand(n, generator=gen)            # the reference feeds probabilities
    x[:100] = torch.randn(100, generator=gen) * 4  # ... and the formula must hold for any real input
    labels = torch.randint(-1, 2, (n,), generator=gen)
    m = labels > -1
    xr = x.double().clone().requires_grad_(True)
    ref = tp.focal_loss(xr[m], labels[m].double(), gamma=gamma)
    ref.backward()
    xg = x.to(gpu).requires_grad_(True)
    loss = FocalLoss(gamma=gamma)(xg, labels.to(gpu))
    assert abs(float(loss.detach()) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    (2.0 * loss).backward()
    want = 2.0 * xr.grad.float()
    assert torch.allclose(xg.grad.cpu(), want, rtol=2e-4, atol=1e-9 + 2e-6 * float(want.abs().max()))
    assert torch.all(xg.grad[labels.to(gpu) < 0] == 0)
