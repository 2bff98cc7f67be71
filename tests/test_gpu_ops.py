"""Kernel-level parity: every HIP op (through the C ABI) against a torch-CPU fp32 computation of the same op on the
same bf16-rounded operands.  Tolerances are stated per test; bf16 has 8 significand bits (rel. 2^-9 per rounding)."""

import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import os

# The product library holds the SHIPPED variant of every kernel (plus the general fall-backs); the measured-and-rejected schedules
# live in the -DCVX_ABLATION build only.  Their parity tests run when this module is imported with CVX_ABLATION_LIB=1 -- which
# test_variants_on_the_ablation_build (at the end of this file) does in a child process.
ABLATION = os.environ.get("CVX_ABLATION_LIB") == "1"


def set_option_or_skip(name, value):
    from cryovit_amd import _lib

    try:
        _lib.set_option(name, value)
    except _lib.CvxError as e:
        if ABLATION:
            raise
        pytest.skip(f"{name}={value} is an ablation-build variant: {e}")


def bf(t):
    return t.to(torch.bfloat16)


def hf(t):  # the segmentation head stores fp16 (11 significand bits: rel. 2^-12 per rounding)
    return t.to(torch.float16)


LOG2E = 1.4426950408889634


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def padded_bf16(t, rows, cols, dev):
    out = torch.zeros(rows, cols, dtype=torch.bfloat16)
    out[: t.shape[0], : t.shape[1]] = bf(t)
    return out.to(dev)


def padded_f32(v, n, dev):
    out = torch.zeros(n)
    out[: v.numel()] = v
    return out.to(dev)


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (1000, 192, 128), (257, 40, 64), (130, 16, 320), (4096, 1536, 1536)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_bf16(gpu, M, N, K, gelu):
    from cryovit_amd._lib import EPI_BF16, EPI_BF16_GELU
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _npad

    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K**-0.5), rnd(N, seed=3)
    n_pad, k_pad = _npad(N), ops.round_up(K, 64)
    A = padded_bf16(a, ops.alloc_rows(M), k_pad, gpu)
    Wd = padded_bf16(w, n_pad, k_pad, gpu)
    out = torch.full((ops.alloc_rows(M), N), 7.0, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_BF16_GELU if gelu else EPI_BF16, A, Wd, out, padded_f32(b, n_pad, gpu), m=M, n=N)
    ref = bf(a).float() @ bf(w).float().T + b
    ref = F.gelu(ref) if gelu else ref
    got = out[:M].float().cpu()
    # bf16 output rounding (2^-9 relative) + fp32 accumulation order
    assert torch.allclose(got, ref, atol=2e-2, rtol=1e-2), float((got - ref).abs().max())
    assert torch.all(out[M:].float() == 7.0), "rows beyond M were written"


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (257, 40, 64), (4096, 1024, 1536)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_fp16_operands(gpu, M, N, K, gelu):
    """The head's GEMMs: fp16 operands on v_mfma_f32_16x16x32_f16, fp16 output (the last shape takes the 256x256 pipeline).
    fp16 output rounding is 2^-12 relative: the tolerance is 8x tighter than the bf16 test above."""
    from cryovit_amd._lib import EPI_BF16, EPI_BF16_GELU
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _npad

    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K**-0.5), rnd(N, seed=3)
    n_pad, k_pad = _npad(N), ops.round_up(K, 64)
    A = torch.zeros(ops.alloc_rows(M), k_pad, dtype=torch.float16)
    A[:M, :K] = hf(a)
    Wd = torch.zeros(n_pad, k_pad, dtype=torch.float16)
    Wd[:N, :K] = hf(w)
    out = torch.full((ops.alloc_rows(M), N), 7.0, dtype=torch.float16, device=gpu)
    ops.gemm(EPI_BF16_GELU if gelu else EPI_BF16, A.to(gpu), Wd.to(gpu), out, padded_f32(b, n_pad, gpu), m=M, n=N)
    ref = hf(a).float() @ hf(w).float().T + b
    ref = F.gelu(ref) if gelu else ref
    got = out[:M].float().cpu()
    assert torch.allclose(got, ref, atol=3e-3, rtol=2e-3), float((got - ref).abs().max())
    assert torch.all(out[M:].float() == 7.0), "rows beyond M were written"


def test_gemm_fp16_saturates_instead_of_inf(gpu):
    """fp16 stores clamp to the largest finite half (65504): an activation spike cannot turn into inf / NaN downstream."""
    from cryovit_amd._lib import EPI_BF16
    from cryovit_amd.engine import ops

    M, N, K = 64, 64, 64
    A = torch.full((ops.alloc_rows(M), K), 200.0, dtype=torch.float16, device=gpu)
    Wd = torch.full((N, K), 100.0, dtype=torch.float16, device=gpu)
    Wd[1::2] *= -1
    out = torch.zeros(ops.alloc_rows(M), N, dtype=torch.float16, device=gpu)
    ops.gemm(EPI_BF16, A, Wd, out, torch.zeros(N, device=gpu), m=M, n=N)  # +-1.28e6 before rounding
    got = out[:M].float().cpu()
    assert torch.all(got[:, 0::2] == 65504.0) and torch.all(got[:, 1::2] == -65504.0)
    with pytest.raises(Exception, match="fp16 operands are built for"):
        from cryovit_amd._lib import EPI_RESID

        ops.gemm(EPI_RESID, A, Wd, torch.zeros(ops.alloc_rows(M), N, device=gpu), torch.zeros(N, device=gpu), gamma=torch.ones(N, device=gpu), m=M, n=N)


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (2048, 1536, 576), (130, 144, 320)])
def test_gemm_f32_output(gpu, M, N, K):
    """CVX_EPI_F32: out fp32 = gamma * (acc + bias), WRITTEN (stale contents of the output must not leak in)."""
    from cryovit_amd._lib import EPI_F32
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.hiera import _npad

    a, w, b, gam = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=K**-0.5), rnd(N, seed=73), rnd(N, seed=74) + 1
    n_pad, k_pad = _npad(N, K), ops.round_up(K, 64)
    out = torch.full((ops.alloc_rows(M), N), float("nan"), device=gpu)  # NaN: any read-modify-write would show
    ops.gemm(EPI_F32, padded_bf16(a, ops.alloc_rows(M), k_pad, gpu), padded_bf16(w, n_pad, k_pad, gpu), out, padded_f32(b, n_pad, gpu),
             gamma=padded_f32(gam, n_pad + 256, gpu), m=M, n=N)
    ref = gam * (bf(a).float() @ bf(w).float().T + b)
    got = out[:M].cpu()
    assert torch.allclose(got, ref, atol=2e-3, rtol=1e-3), float((got - ref).abs().max())
    assert torch.isnan(out[M:]).all(), "rows beyond M were written"


def test_gemm_swiglu(gpu):
    from cryovit_amd._lib import EPI_SWIGLU
    from cryovit_amd.engine import ops

    M, K, Hd = 500, 128, 344  # hidden not a multiple of 64 -> padded to 384
    Hp = ops.round_up(Hd, 64)
    a, w12, b12 = rnd(M, K, seed=4), rnd(2 * Hd, K, seed=5, scale=K**-0.5), rnd(2 * Hd, seed=6)
    aw, bw, ab, bb = torch.zeros(Hp, K), torch.zeros(Hp, K), torch.zeros(Hp), torch.zeros(Hp)
    aw[:Hd], bw[:Hd], ab[:Hd], bb[:Hd] = w12[:Hd], w12[Hd:], b12[:Hd], b12[Hd:]
    iw = torch.stack([aw.reshape(-1, 8, K), bw.reshape(-1, 8, K)], 1).reshape(2 * Hp, K)
    ib = torch.stack([ab.reshape(-1, 8), bb.reshape(-1, 8)], 1).reshape(2 * Hp)
    A = padded_bf16(a, ops.alloc_rows(M), K, gpu)
    out = torch.zeros(ops.alloc_rows(M), Hp, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_SWIGLU, A, bf(iw).to(gpu), out, ib.to(gpu), m=M, n=2 * Hp)
    h = bf(a).float() @ bf(w12).float().T + b12
    ref = F.silu(h[:, :Hd]) * h[:, Hd:]
    got = out[:M].float().cpu()
    assert torch.allclose(got[:, :Hd], ref, atol=2e-2, rtol=1e-2), float((got[:, :Hd] - ref).abs().max())
    assert torch.all(got[:, Hd:] == 0)


def test_gemm_resid(gpu):
    from cryovit_amd._lib import EPI_RESID
    from cryovit_amd.engine import ops

    M, N, K = 700, 384, 256
    a, w, b, g, x0 = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=K**-0.5), rnd(N, seed=9), rnd(N, seed=10), rnd(M, N, seed=11)
    A = padded_bf16(a, ops.alloc_rows(M), K, gpu)
    x = torch.zeros(ops.alloc_rows(M), N, device=gpu)
    x[:M] = x0.to(gpu)
    ops.gemm(EPI_RESID, A, padded_bf16(w, N, K, gpu), x, b.to(gpu), m=M, n=N, gamma=g.to(gpu))
    ref = x0 + g * (bf(a).float() @ bf(w).float().T + b)
    assert torch.allclose(x[:M].cpu(), ref, atol=1e-4, rtol=1e-4), float((x[:M].cpu() - ref).abs().max())  # fp32 out
    assert torch.all(x[M:] == 0)


def test_gemm_patch_and_tokens(gpu):
    from cryovit_amd._lib import EPI_PATCH
    from cryovit_amd.engine import ops

    b, npatch, C, K, n_reg = 3, 24, 128, 196, 4
    nt = npatch + 1 + n_reg
    ntp = ops.round_up(nt, 8)
    a, w, bias = rnd(b * npatch, K, seed=12), rnd(C, K, seed=13, scale=K**-0.5), rnd(C, seed=14)
    pos, cls, reg = rnd(1 + npatch, C, seed=15), rnd(C, seed=16), rnd(n_reg, C, seed=17)
    A = padded_bf16(a, ops.alloc_rows(b * npatch), 256, gpu)
    x = torch.full((ops.alloc_rows(b * ntp), C), 5.0, device=gpu)
    ops.init_tokens(x, (cls + pos[0]).to(gpu), reg.to(gpu), n_reg=n_reg, slices=b, ntok=nt, ntp=ntp, Cdim=C)
    ops.gemm(EPI_PATCH, A, padded_bf16(w, C, 256, gpu), x, bias.to(gpu), m=b * npatch, n=C, pos=pos.to(gpu), npatch=npatch,
             ntp=ntp, tok0=1 + n_reg)
    pe = (bf(a).float() @ bf(w).float().T + bias).reshape(b, npatch, C) + pos[1:]
    got = x[: b * ntp].cpu().reshape(b, ntp, C)
    assert torch.allclose(got[:, 1 + n_reg : nt], pe, atol=1e-4, rtol=1e-4)
    assert torch.equal(got[:, 0], (cls + pos[0]).expand(b, C))
    assert torch.equal(got[:, 1 : 1 + n_reg], reg.expand(b, n_reg, C))
    assert torch.all(got[:, nt:] == 0)


def test_gemm_vt(gpu):
    from cryovit_amd._lib import EPI_VT
    from cryovit_amd.engine import ops

    b, heads, nt, K = 3, 2, 29, 128
    ntp, kp, C = ops.round_up(nt, 8), 64, heads * 64
    M = b * ntp
    a, w, bias = rnd(M, K, seed=18), rnd(C, K, seed=19, scale=K**-0.5), rnd(C, seed=20)
    A = padded_bf16(a, ops.alloc_rows(M), K, gpu)
    vt = torch.zeros(b, heads, 64, kp, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_VT, A, padded_bf16(w, C, K, gpu), vt, bias.to(gpu), m=M, n=C, heads=heads, ntp=ntp, kp=kp, ldc=0)
    v = (bf(a).float() @ bf(w).float().T + bias).reshape(b, ntp, heads, 64).permute(0, 2, 3, 1)  # [b,h,d,t]
    got = vt.float().cpu()
    assert torch.allclose(got[..., :ntp], v, atol=2e-2, rtol=1e-2), float((got[..., :ntp] - v).abs().max())
    assert torch.all(got[..., ntp:] == 0)


@pytest.mark.parametrize("C", [128, 144, 288, 384, 576, 1152, 1536])
def test_layernorm(gpu, C):
    from cryovit_amd.engine import ops

    rows = 1030
    x = rnd(rows, C, seed=21) * 3 + 0.5
    w, b = rnd(C, seed=22) + 1, rnd(C, seed=23)
    out = torch.zeros(rows, C, dtype=torch.bfloat16, device=gpu)
    ops.layernorm(x.to(gpu), w.to(gpu), b.to(gpu), out, rows, C, 1e-6)
    ref = F.layer_norm(x, (C,), w, b, 1e-6)
    assert torch.allclose(out.float().cpu(), ref, atol=2e-2, rtol=1e-2)
    # the cache-policy / row-order variants of the wide-row kernel are the same arithmetic: bit-identical output
    from cryovit_amd import _lib

    try:
        for pol in (0, 1, 2):
            _lib.set_option("ln_policy", pol)
            o2 = torch.zeros_like(out)
            ops.layernorm(x.to(gpu), w.to(gpu), b.to(gpu), o2, rows, C, 1e-6)
            assert torch.equal(o2, out), pol
    finally:
        _lib.set_option("ln_policy", 3)


@pytest.fixture
def attn_variant(request):
    """All attention kernels kept in attention.hip are parity-tested: 0 = 32 query rows per wave (4-wave blocks),
    3 = three K/V buffers, 4 / 5 = 64 query rows per wave in 3- / 4-wave blocks, 6 (default) = 0 with the running maximum
    subtracted inside the QK^T product (augmented k-step) and raised only when a row outgrows it by 2^3, 7 (default) = 6 with the
    re-anchoring triggered by the tile's probability SUMS (the maximum is only computed in tile 0; a jump beyond the range of one
    exp2 recomputes the tile against raised anchors)."""
    from cryovit_amd import _lib

    set_option_or_skip("attn_variant", request.param)
    yield request.param
    _lib.set_option("attn_variant", 7)  # the default


@pytest.mark.parametrize("attn_variant", [0, 3, 4, 5, 6, 7, 8, 9], indirect=True)
@pytest.mark.parametrize("nt,slices,heads", [(29, 3, 2), (261, 2, 6), (1029, 2, 3), (1029, 8, 1), (96, 2, 1), (100, 2, 1), (128, 1, 2)])
def test_attention(gpu, nt, slices, heads, attn_variant):
    from cryovit_amd.engine import ops

    C = heads * 64
    ntp, kp = ops.round_up(nt, 8), ops.round_up(nt, 64)
    M = slices * ntp
    q, k, v = rnd(slices, nt, heads, 64, seed=24), rnd(slices, nt, heads, 64, seed=25), rnd(slices, nt, heads, 64, seed=26)
    q = q * 1.5  # spread the logits so the online-softmax rescale path is exercised
    qk = torch.randn(ops.alloc_rows(M), 2 * C, generator=torch.Generator().manual_seed(27)).to(torch.bfloat16)  # junk pad
    qkv = qk[:M].reshape(slices, ntp, 2 * C)
    qkv[:, :nt, :C] = bf(q * 0.125 * LOG2E).reshape(slices, nt, C)  # the kernel takes Q in log2 units (header)
    qkv[:, :nt, C:] = bf(k).reshape(slices, nt, C)
    vt = torch.zeros(slices, heads, 64, kp, dtype=torch.bfloat16)
    vt[..., :nt] = bf(v).permute(0, 2, 3, 1)
    vt[..., nt:ntp] = 3.0  # finite garbage in the pad tokens must be masked out
    out = torch.zeros(ops.alloc_rows(M), C, dtype=torch.bfloat16, device=gpu)
    ops.attention(qk.to(gpu), vt.to(gpu), out, slices=slices, heads=heads, ntok=nt, ntp=ntp, kp=kp)
    qf, kf, vf = bf(q * 0.125 * LOG2E).float() / LOG2E, bf(k).float(), bf(v).float()
    att = torch.softmax(torch.einsum("snhd,smhd->shnm", qf, kf), dim=-1)
    ref = torch.einsum("shnm,smhd->snhd", att, vf).reshape(slices, nt, C)
    got = out[:M].float().cpu().reshape(slices, ntp, C)
    # P is rounded to bf16 before PV (rel 2^-9) and the output is bf16
    assert torch.allclose(got[:, :nt], ref, atol=2e-2, rtol=2e-2), float((got[:, :nt] - ref).abs().max())
    assert torch.all(got[:, nt:] == 0), "padding rows must not be written"


@pytest.mark.parametrize("nt,slices,heads", [(29, 3, 2), (261, 2, 6), (1029, 2, 3), (1029, 8, 1), (133, 2, 1),
                                             (96, 2, 1), (100, 2, 1), (128, 1, 2)])  # (last key tile: exactly half / more than half / all valid)
def test_attention_qkv_row_major_v(gpu, nt, slices, heads):
    """cvx_attention_qkv_bf16: Q | K | V row-major in ONE buffer (the output of a single qkv GEMM); V fragments are transposed by the
    LDS reads (ds_read_b64_tr_b16).  Same reference and tolerance as test_attention; junk in the pad rows must not leak."""
    from cryovit_amd.engine import ops

    C = heads * 64
    ntp = ops.round_up(nt, 8)
    M = slices * ntp
    q, k, v = rnd(slices, nt, heads, 64, seed=24) * 1.5, rnd(slices, nt, heads, 64, seed=25), rnd(slices, nt, heads, 64, seed=26)
    buf = torch.randn(ops.alloc_rows(M), 3 * C, generator=torch.Generator().manual_seed(27)).to(torch.bfloat16)  # junk pad rows
    rows = buf[:M].reshape(slices, ntp, 3 * C)
    rows[:, :nt, :C] = bf(q * 0.125 * LOG2E).reshape(slices, nt, C)
    rows[:, :nt, C : 2 * C] = bf(k).reshape(slices, nt, C)
    rows[:, :nt, 2 * C :] = bf(v).reshape(slices, nt, C)
    out = torch.zeros(ops.alloc_rows(M), C, dtype=torch.bfloat16, device=gpu)
    ops.attention_qkv(buf.to(gpu), out, slices=slices, heads=heads, ntok=nt, ntp=ntp)
    qf, kf, vf = bf(q * 0.125 * LOG2E).float() / LOG2E, bf(k).float(), bf(v).float()
    att = torch.softmax(torch.einsum("snhd,smhd->shnm", qf, kf), dim=-1)
    ref = torch.einsum("shnm,smhd->snhd", att, vf).reshape(slices, nt, C)
    got = out[:M].float().cpu().reshape(slices, ntp, C)
    assert torch.allclose(got[:, :nt], ref, atol=2e-2, rtol=2e-2), float((got[:, :nt] - ref).abs().max())
    assert torch.all(got[:, nt:] == 0), "padding rows must not be written"
    # ... and the same numbers as the V^T form on the same operands (same products, same order: bit-identical)
    kp = ops.round_up(nt, 64)
    vt = torch.zeros(slices, heads, 64, kp, dtype=torch.bfloat16)
    vt[..., :ntp] = rows[..., 2 * C :].reshape(slices, ntp, heads, 64).permute(0, 2, 3, 1)
    qk2 = buf[:, : 2 * C].contiguous()
    out2 = torch.zeros_like(out)
    ops.attention(qk2.to(gpu), vt.to(gpu), out2, slices=slices, heads=heads, ntok=nt, ntp=ntp, kp=kp)
    assert torch.equal(out2[:M].reshape(slices, ntp, C)[:, :nt], out[:M].reshape(slices, ntp, C)[:, :nt])


@pytest.mark.parametrize("pad", [8, 24, 40])
def test_attention_odd_leading_dimension(gpu, pad):
    """ldqk NOT a multiple of 64 with a single-wave last query block (ntok % 128 in (0, 32]): the default kernel's lone-wave DMA
    offsets assume a 128-byte-multiple K row pitch; such calls must take the general kernel and stay correct."""
    from cryovit_amd.engine import ops

    nt, slices, heads = 128 + 5, 2, 1
    C = heads * 64
    ntp, kp = ops.round_up(nt, 8), ops.round_up(nt, 64)
    M, ld = slices * ntp, 2 * C + pad
    q, k, v = rnd(slices, nt, heads, 64, seed=124) * 1.5, rnd(slices, nt, heads, 64, seed=125), rnd(slices, nt, heads, 64, seed=126)
    qk = torch.randn(ops.alloc_rows(M), ld, generator=torch.Generator().manual_seed(127)).to(torch.bfloat16)
    qkv = qk[:M].reshape(slices, ntp, ld)
    qkv[:, :nt, :C] = bf(q * 0.125 * LOG2E).reshape(slices, nt, C)
    qkv[:, :nt, C : 2 * C] = bf(k).reshape(slices, nt, C)
    vt = torch.zeros(slices, heads, 64, kp, dtype=torch.bfloat16)
    vt[..., :nt] = bf(v).permute(0, 2, 3, 1)
    out = torch.zeros(ops.alloc_rows(M), C, dtype=torch.bfloat16, device=gpu)
    ops.attention(qk.to(gpu), vt.to(gpu), out, slices=slices, heads=heads, ntok=nt, ntp=ntp, kp=kp)
    qf, kf, vf = bf(q * 0.125 * LOG2E).float() / LOG2E, bf(k).float(), bf(v).float()
    att = torch.softmax(torch.einsum("snhd,smhd->shnm", qf, kf), dim=-1)
    ref = torch.einsum("shnm,smhd->snhd", att, vf).reshape(slices, nt, C)
    got = out[:M].float().cpu().reshape(slices, ntp, C)
    assert torch.allclose(got[:, :nt], ref, atol=2e-2, rtol=2e-2), float((got[:, :nt] - ref).abs().max())


@pytest.mark.parametrize("spike", [4.0, 9.0, 60.0, -60.0])
@pytest.mark.parametrize("attn_variant", [0, 4, 6, 7, 8, 9], indirect=True)
def test_attention_forced_rescale(gpu, attn_variant, spike):
    """One key row spiked against one query so the running max jumps in a late tile (rare-branch test).  spike 60: the jump
    is far above the deferred-maximum threshold of variant 6 (its raise-and-rescale branch) and beyond what one exp2 can hold
    (variant 7: the tile is recomputed against raised anchors, several times); spike 9: a jump of ~2^100, variant 7's in-place
    rescale from the probabilities; spike -60 with the shift below:
    every score of the tomogram is far below zero (rows must be anchored at their own maximum, not at 0)."""
    from cryovit_amd.engine import ops

    nt, slices, heads, C = 200, 1, 1, 64
    ntp, kp = ops.round_up(nt, 8), ops.round_up(nt, 64)
    q, k, v = rnd(1, nt, 1, 64, seed=28), rnd(1, nt, 1, 64, seed=29), rnd(1, nt, 1, 64, seed=30)
    if spike < 0:  # all logits ~ -300 (natural units): q . k = -8 * 300 / 0.125 ... through one shared coordinate
        q[0, :, 0, 1], k[0, :, 0, 1] = 50.0, -48.0
    k[0, 150, 0] = q[0, 7, 0] * abs(spike)
    qk = torch.zeros(ops.alloc_rows(ntp), 2 * C, dtype=torch.bfloat16)
    qk[:nt, :C], qk[:nt, C:] = bf(q * 0.125 * LOG2E).reshape(nt, C), bf(k).reshape(nt, C)
    vt = torch.zeros(1, 1, 64, kp, dtype=torch.bfloat16)
    vt[0, 0, :, :nt] = bf(v).reshape(nt, 64).T
    out = torch.zeros(ops.alloc_rows(ntp), C, dtype=torch.bfloat16, device=gpu)
    ops.attention(qk.to(gpu), vt.to(gpu), out, slices=1, heads=1, ntok=nt, ntp=ntp, kp=kp)
    qf, kf, vf = bf(q * 0.125 * LOG2E).float().reshape(nt, 64) / LOG2E, bf(k).float().reshape(nt, 64), bf(v).float().reshape(nt, 64)
    ref = torch.softmax(qf @ kf.T, -1) @ vf
    got = out[:nt].float().cpu()
    assert torch.allclose(got, ref, atol=2e-2, rtol=2e-2), float((got - ref).abs().max())


@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_preprocess_patches(gpu, gold, dtype):
    from cryovit_amd.engine import ops

    g = gold("preprocess.npz")
    vol, ref = (g["vol_u8"], g["out_u8"]) if dtype == "u8" else (g["vol_f32"], g["out_f32"])
    b, H, W = vol.shape
    hp, wp = math.ceil(H / 16), math.ceil(W / 16)
    out = torch.zeros(ops.alloc_rows(b * hp * wp), 256, dtype=torch.bfloat16, device=gpu)
    ops.preprocess_patches(torch.from_numpy(vol).to(gpu), out)
    got = out[: b * hp * wp].float().cpu().reshape(b, hp, wp, 256)
    assert torch.all(got[..., 196:] == 0)
    img = got[..., :196].reshape(b, hp, wp, 14, 14).permute(0, 1, 3, 2, 4).reshape(b, hp * 14, wp * 14)
    reft = torch.from_numpy(ref)
    # fixture = the reference's own _dino_transform output; bf16 storage of values in [-0.1, 1.1]
    assert torch.allclose(img, bf(reft).float(), atol=4e-3, rtol=0), float((img - reft).abs().max())


def test_final_norm_and_k9_layout(gpu):
    from cryovit_amd.engine import ops

    b, hp, wp, C, n_reg = 3, 4, 6, 128, 4
    npatch, nt = hp * wp, hp * wp + 5
    ntp = ops.round_up(nt, 8)
    x = rnd(b, ntp, C, seed=31) * 2 + 0.3
    w, bb = rnd(C, seed=32) + 1, rnd(C, seed=33)
    d_total, d0 = 5, 1
    f16 = torch.full((C, d_total, hp, wp), -9.0, dtype=torch.float16, device=gpu)
    cl = torch.zeros(b, hp, wp, C, dtype=torch.float16, device=gpu)
    ops.final_norm_features(x.reshape(-1, C).to(gpu), w.to(gpu), bb.to(gpu), 1e-6, slices=b, ntp=ntp, tok0=1 + n_reg, hp=hp, wp=wp,
                            Cdim=C, feats_f16=f16, d_total=d_total, d0=d0, feats_cl=cl)
    ref = F.layer_norm(x[:, 1 + n_reg : nt], (C,), w, bb, 1e-6)  # [b, npatch, C]
    ref_k9 = ref.reshape(b, hp, wp, C).permute(3, 0, 1, 2)  # run/dino_features.py:59-60
    got = f16.float().cpu()
    assert torch.allclose(got[:, d0 : d0 + b], ref_k9, atol=3e-3, rtol=2e-3)  # fp16 store
    assert torch.all(got[:, :d0] == -9.0) and torch.all(got[:, d0 + b :] == -9.0)
    assert torch.allclose(cl.float().cpu().reshape(b, npatch, C), ref, atol=3e-3, rtol=2e-3)  # fp16 channels-last copy
    # the two fp16 copies (file layout / head layout) hold bit-identical values: the head sees the same input either way
    assert torch.equal(cl.cpu().reshape(b, hp, wp, C).permute(3, 0, 1, 2), f16[:, d0 : d0 + b].cpu())


def test_features_to_channels_last(gpu):
    from cryovit_amd.engine import ops

    C, D, h, w = 136, 3, 5, 7
    f = rnd(C, D, h, w, seed=34).half()
    out = torch.zeros(D * h * w, C, dtype=torch.float16, device=gpu)
    ops.features_to_channels_last(f.to(gpu), out)
    assert torch.equal(out.cpu(), f.reshape(C, -1).T)  # a transpose of the fp16 file contents: exact


@pytest.mark.parametrize("C,G", [(128, 16), (32, 8), (8, 8), (1024, 128)])
def test_groupnorm(gpu, C, G):
    from cryovit_amd.engine import ops

    D, H, W = 5, 6, 7
    x = hf(rnd(D, H, W, C, seed=35) * 1.5 + 0.4)
    w, b = rnd(C, seed=36) + 1, rnd(C, seed=37)
    out = torch.zeros_like(x, device=gpu)
    stats = torch.zeros(ops.gn_stats_size(G), device=gpu)
    ops.groupnorm(x.to(gpu), w.to(gpu), b.to(gpu), out, stats, nvox=D * H * W, Cdim=C, G=G, eps=1e-3)
    ref = F.group_norm(x.float().permute(3, 0, 1, 2).unsqueeze(0), G, w, b, 1e-3)[0].permute(1, 2, 3, 0)
    assert torch.allclose(out.float().cpu(), ref, atol=4e-3, rtol=2e-3), float((out.float().cpu() - ref).abs().max())


def test_groupnorm_many_blocks_reproducible(gpu):
    """More voxels than CVX_GN_BLOCKS * 2048 (blocks take several chunks) and a fixed-order reduction: the group sums match
    float64 sums of the fp16 inputs to fp32 round-off and repeated calls are bit-identical (no atomics)."""
    from cryovit_amd.engine import ops

    C, G, nvox = 16, 8, 1024 * 2048 + 12345
    g = torch.Generator(device=gpu).manual_seed(5)
    x = (torch.randn(nvox, C, device=gpu, generator=g) * 2 + 0.3).to(torch.float16)
    w, b = torch.ones(C, device=gpu), torch.zeros(C, device=gpu)
    stats = torch.zeros(ops.gn_stats_size(G), device=gpu)
    outs = []
    for _ in range(3):
        out = torch.empty_like(x)
        ops.groupnorm(x, w, b, out, stats, nvox=nvox, Cdim=C, G=G, eps=1e-3)
        outs.append((out.clone(), stats[: 2 * G].clone()))
    assert all(torch.equal(outs[0][0], o) and torch.equal(outs[0][1], s) for o, s in outs[1:])
    xd = x.double().reshape(nvox, G, C // G)
    want = torch.cat([xd.sum((0, 2)), (xd * xd).sum((0, 2))]).cpu()
    got = outs[0][1].double().cpu()
    assert torch.allclose(got, want, rtol=1e-6), (got, want)
    mean = want[:G] / (nvox * 2)
    var = want[G:] / (nvox * 2) - mean**2
    ref = ((x[:4096].float().cpu().reshape(-1, G, 2) - mean.float()[None, :, None]) / torch.sqrt(var.float() + 1e-3)[None, :, None]).reshape(-1, C)
    assert torch.allclose(outs[0][0][:4096].float().cpu(), ref, atol=4e-3, rtol=2e-3)


@pytest.mark.parametrize("Cin,Cout,dil,D,H,W", [(128, 24, 32, 8, 4, 4), (32, 16, 2, 6, 8, 8), (8, 8, 1, 5, 9, 11), (192, 192, 3, 7, 5, 6),
                                               (64, 64, 12, 16, 8, 8)])
def test_conv3d(gpu, Cin, Cout, dil, D, H, W):
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1

    x = hf(rnd(D, H, W, Cin, seed=38))
    w, b = rnd(Cout, Cin, 3, 3, 3, seed=39, scale=(27 * Cin) ** -0.5), rnd(Cout, seed=40)
    nv = D * H * W
    out = torch.full((nv + 8, Cout), 7.0, dtype=torch.float16, device=gpu)
    zero = torch.zeros(256, dtype=torch.uint8, device=gpu)
    ops.conv3d(x.to(gpu), _conv3_weight(w).to(gpu), _pad1(b, _npad(Cout)).to(gpu), out, zero, Cin=Cin, D=D, H=H, W=W, dil=dil,
               cout=Cout, act=1)
    ref = F.gelu(F.conv3d(x.float().permute(3, 0, 1, 2).unsqueeze(0), hf(w).float(), b, padding="same", dilation=(dil, 1, 1)))
    ref = ref[0].permute(1, 2, 3, 0).reshape(nv, Cout)
    got = out[:nv].float().cpu()
    assert torch.allclose(got, ref, atol=3e-3, rtol=2e-3), float((got - ref).abs().max())  # fp16 operands, fp32 accumulation
    assert torch.all(out[nv:].float() == 7.0)


def test_conv3d_wide_tile_matches_narrow(gpu):
    """C_out = 192 (SynthesisBlock 1): the 192-wide implicit-GEMM tile against the three 64-wide tiles of the same call -- same
    fp16 operands, same k order inside a tile, so the results are bit-identical; dilation 32 with D = 40 (taps on both sides)."""
    from cryovit_amd import _lib
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1

    Cin, Cout, dil, D, H, W = 128, 192, 32, 40, 6, 7
    x = hf(rnd(D, H, W, Cin, seed=63))
    w, b = rnd(Cout, Cin, 3, 3, 3, seed=64, scale=(27 * Cin) ** -0.5), rnd(Cout, seed=65)
    nv = D * H * W
    zero = torch.zeros(256, dtype=torch.uint8, device=gpu)
    outs = []
    try:
        for wide in (1, 0):
            _lib.set_option("conv_wide", wide)
            out = torch.full((nv + 8, Cout), 7.0, dtype=torch.float16, device=gpu)
            ops.conv3d(x.to(gpu), _conv3_weight(w).to(gpu), _pad1(b, _npad(Cout)).to(gpu), out, zero, Cin=Cin, D=D, H=H, W=W, dil=dil,
                       cout=Cout, act=1)
            outs.append(out)
    finally:
        _lib.set_option("conv_wide", 1)
    ref = F.gelu(F.conv3d(x.float().permute(3, 0, 1, 2).unsqueeze(0), hf(w).float(), b, padding="same", dilation=(dil, 1, 1)))
    ref = ref[0].permute(1, 2, 3, 0).reshape(nv, Cout)
    assert torch.allclose(outs[0][:nv].float().cpu(), ref, atol=3e-3, rtol=2e-3)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Cin,Cout,dil,D,H,W", [(32, 16, 2, 7, 9, 70), (16, 16, 1, 5, 12, 131), (8, 8, 1, 4, 7, 66), (16, 8, 3, 6, 4, 64),
                                               (32, 12, 1, 3, 5, 33), (32, 32, 8, 19, 6, 40), (32, 32, 4, 9, 10, 65), (8, 4, 1, 1, 3, 5),
                                               (16, 32, 5, 4, 5, 17), (8, 8, 2, 11, 13, 129)])
def test_conv3d_halo_kernel(gpu, Cin, Cout, dil, D, H, W):
    """The LDS kernels of the few-channel layers (C_in 8/16/32): the z-marching ring kernel (option 2: C_out <= 32, one residue
    class of z per workgroup, one new plane per step by LDS-DMA) and the per-tile halo kernel (option 1: C_out <= 16) --
    ragged tiles in x and y, dilation in z reaching outside the volume (also dil >= D: every tap but the centre plane is
    padding), D not a multiple of the dilation, C_out 4 / 8 (fragments finished by the upper lanes) / 12 / 16 / 32 -- against
    torch, and against the implicit-GEMM kernel the same call uses when the option is off (same fp16 operands, fp32
    accumulation in a different order)."""
    from cryovit_amd import _lib
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1

    x = hf(rnd(D, H, W, Cin, seed=58))
    w, b = rnd(Cout, Cin, 3, 3, 3, seed=59, scale=(27 * Cin) ** -0.5), rnd(Cout, seed=60)
    nv = D * H * W
    zero = torch.zeros(256, dtype=torch.uint8, device=gpu)
    outs = []
    try:
        for halo in ((2, 1, 0) if ABLATION else (2, 0, 0)):  # (1 = the round-1 tile-halo kernel: ablation build)
            _lib.set_option("conv_halo", halo)
            out = torch.full((nv + 8, Cout), 7.0, dtype=torch.float16, device=gpu)
            ops.conv3d(x.to(gpu), _conv3_weight(w).to(gpu), _pad1(b, _npad(Cout)).to(gpu), out, zero, Cin=Cin, D=D, H=H, W=W, dil=dil,
                       cout=Cout, act=1)
            outs.append(out)
    finally:
        _lib.set_option("conv_halo", 2)
    ref = F.gelu(F.conv3d(x.float().permute(3, 0, 1, 2).unsqueeze(0), hf(w).float(), b, padding="same", dilation=(dil, 1, 1)))
    ref = ref[0].permute(1, 2, 3, 0).reshape(nv, Cout)
    for out in outs:
        got = out[:nv].float().cpu()
        assert torch.allclose(got, ref, atol=3e-3, rtol=2e-3), float((got - ref).abs().max())
        assert torch.all(out[nv:].float() == 7.0)
    assert float((outs[0][:nv].float() - outs[2][:nv].float()).abs().max()) <= 4e-3
    assert float((outs[1][:nv].float() - outs[2][:nv].float()).abs().max()) <= 4e-3


def test_conv3d_march_many_columns_reproducible(gpu):
    """More columns x residues than resident workgroups (each workgroup marches several columns: the ring is re-filled
    between them), no activation: bit-identical on repetition and equal to the implicit-GEMM kernel within fp32 re-ordering."""
    from cryovit_amd import _lib
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _conv3_weight, _npad, _pad1

    Cin, Cout, dil, D, H, W = 8, 8, 3, 7, 96, 640
    g = torch.Generator(device=gpu).manual_seed(7)
    x = torch.randn(D * H * W, Cin, device=gpu, generator=g).to(torch.float16)
    w, b = rnd(Cout, Cin, 3, 3, 3, seed=61, scale=(27 * Cin) ** -0.5), rnd(Cout, seed=62)
    zero = torch.zeros(256, dtype=torch.uint8, device=gpu)
    outs = []
    try:
        for halo in (2, 2, 0):
            _lib.set_option("conv_halo", halo)
            out = torch.zeros(D * H * W, Cout, dtype=torch.float16, device=gpu)
            ops.conv3d(x, _conv3_weight(w).to(gpu), _pad1(b, _npad(Cout)).to(gpu), out, zero, Cin=Cin, D=D, H=H, W=W, dil=dil, cout=Cout, act=0)
            outs.append(out)
    finally:
        _lib.set_option("conv_halo", 2)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0].float() - outs[2].float()).abs().max()) <= 4e-3


@pytest.mark.parametrize("c2,c3", [(24, 16), (192, 128), (16, 8)])
def test_conv_transpose(gpu, c2, c3):
    from cryovit_amd._lib import EPI_CONVT
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _npad, _pad1, _pad2

    D, H, W = 3, 5, 6
    nv = D * H * W
    x = hf(rnd(nv, c2, seed=41))
    wt, b = rnd(c2, c3, 1, 2, 2, seed=42, scale=c2**-0.5), rnd(c3, seed=43)
    A = torch.zeros(ops.alloc_rows(nv) * c2 + 4096, dtype=torch.float16)
    A[: nv * c2] = x.reshape(-1)
    A = A.to(gpu)
    a2 = torch.as_strided(A, (ops.alloc_rows(nv), c2), (c2, 1))
    wg = wt[:, :, 0].permute(2, 3, 1, 0).reshape(4 * c3, c2)
    out = torch.zeros(D, 2 * H, 2 * W, c3, dtype=torch.float16, device=gpu)
    ops.gemm(EPI_CONVT, a2, _pad2(wg, _npad(4 * c3), ops.round_up(c2, 64)).to(gpu), out, _pad1(b.repeat(4), _npad(4 * c3)).to(gpu),
             m=nv, n=4 * c3, H=H, W=W, cout=c3, act=1, ldc=c3)
    xin = x.float().reshape(D, H, W, c2).permute(3, 0, 1, 2).unsqueeze(0)
    ref = F.gelu(F.conv_transpose3d(xin, hf(wt).float(), b, stride=(1, 2, 2)))[0].permute(1, 2, 3, 0)
    assert torch.allclose(out.float().cpu(), ref, atol=3e-3, rtol=2e-3), float((out.float().cpu() - ref).abs().max())


@pytest.mark.parametrize("c2,c3,D,H,W", [(16, 8, 3, 5, 32), (32, 32, 2, 3, 16), (16, 8, 1, 1, 16), (32, 32, 5, 7, 48)])
def test_conv_transpose_small_kernel(gpu, c2, c3, D, H, W):
    """The dedicated (1,2,2) transposed-convolution kernel of the few-channel blocks (16 -> 8, 32 -> 32; W % 16 == 0) against torch
    and against the GEMM-tile path of the same call (option off)."""
    from cryovit_amd import _lib
    from cryovit_amd._lib import EPI_CONVT
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import _npad, _pad1, _pad2

    nv = D * H * W
    x = hf(rnd(nv, c2, seed=66))
    wt, b = rnd(c2, c3, 1, 2, 2, seed=67, scale=c2**-0.5), rnd(c3, seed=68)
    A = torch.zeros(ops.alloc_rows(nv) * c2 + 4096, dtype=torch.float16)
    A[: nv * c2] = x.reshape(-1)
    A = A.to(gpu)
    a2 = torch.as_strided(A, (ops.alloc_rows(nv), c2), (c2, 1))
    wg = wt[:, :, 0].permute(2, 3, 1, 0).reshape(4 * c3, c2)
    wp, bp = _pad2(wg, _npad(4 * c3), ops.round_up(c2, 64)).to(gpu), _pad1(b.repeat(4), _npad(4 * c3)).to(gpu)
    outs = []
    try:
        for small in (1, 0):
            _lib.set_option("convt_small", small)
            out = torch.full((D, 2 * H, 2 * W, c3), 7.0, dtype=torch.float16, device=gpu)
            ops.gemm(EPI_CONVT, a2, wp, out, bp, m=nv, n=4 * c3, H=H, W=W, cout=c3, act=1, ldc=c3)
            outs.append(out)
    finally:
        _lib.set_option("convt_small", 1)
    xin = x.float().reshape(D, H, W, c2).permute(3, 0, 1, 2).unsqueeze(0)
    ref = F.gelu(F.conv_transpose3d(xin, hf(wt).float(), b, stride=(1, 2, 2)))[0].permute(1, 2, 3, 0)
    for out in outs:
        assert torch.allclose(out.float().cpu(), ref, atol=3e-3, rtol=2e-3), float((out.float().cpu() - ref).abs().max())
    assert float((outs[0].float() - outs[1].float()).abs().max()) <= 2e-3


def test_conv3_out_fused_and_dice(gpu, gold):
    from cryovit_amd.engine import ops

    D, H, W = 6, 16, 70
    x = hf(rnd(D, H, W, 8, seed=44))
    w, b = rnd(1, 8, 3, 3, 3, seed=45, scale=0.3), 0.1
    labels = torch.from_numpy(np.random.default_rng(46).integers(-1, 2, size=(D, H, W)).astype(np.int8))
    logits = torch.zeros(D, H, W, device=gpu)
    probs = torch.zeros(D, H, W, device=gpu)
    dice = torch.zeros(3, device=gpu)
    seg = torch.full((D, H, W), 7, dtype=torch.uint8, device=gpu)
    ops.conv3_out_fused(x.to(gpu), w[0].permute(1, 2, 3, 0).reshape(27, 8).contiguous().to(gpu), b, logits, probs, labels.to(gpu),
                        dice, D=D, H=H, W=W, mask=seg, mask_threshold=0.3)
    # the uint8 segmentation PredictionWriter stores: bit-exact (preds >= threshold) of the GPU's own probabilities
    assert torch.equal(seg.cpu(), (probs.cpu() >= 0.3).to(torch.uint8))
    # (the weights are rounded to fp16 inside the kernel -- fp16 autocast of the reference -- and a tap is four v_dot2_f32_f16)
    ref = F.conv3d(x.float().permute(3, 0, 1, 2).unsqueeze(0), hf(w).float(), torch.tensor([b]), padding="same")[0, 0].clip(-5, 5)
    assert torch.allclose(logits.cpu(), ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(probs.cpu(), torch.sigmoid(ref), atol=1e-5)
    # Dice sums must be exact integers given the GPU's own probabilities; DiceMetric's threshold and the mask's are ONE parameter
    # (a DiceMetric(threshold != 0.5) must see the same p_hat in the fused kernel as in the stand-alone reduction)
    p = probs.cpu()
    mask = labels > -1
    for thr in (0.3, 0.5):
        if thr != 0.3:
            dice.zero_()
            ops.conv3_out_fused(x.to(gpu), w[0].permute(1, 2, 3, 0).reshape(27, 8).contiguous().to(gpu), b, logits, probs, labels.to(gpu),
                                dice, D=D, H=H, W=W, mask=seg, mask_threshold=thr)
        ph = (p >= thr).float()
        exp = torch.tensor([float((labels.float() * ph)[mask].sum()), float(labels.float()[mask].sum()), float(ph[mask].sum())])
        assert torch.equal(dice.cpu(), exp), thr
        d2 = torch.zeros(3, device=gpu)
        ops.dice_sums(probs, labels.to(gpu), d2, thr)
        assert torch.equal(d2.cpu(), exp), thr
    # stand-alone reduction against the reference-pinned fixture
    g = gold("dice.npz")
    d2 = torch.zeros(3, device=gpu)
    ops.dice_sums(torch.from_numpy(g["preds"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu), d2, 0.5)
    i, sy, sp = d2.cpu().tolist()
    assert abs(2 * i / (sy + sp + 1e-3) - float(g["dice"])) < 1e-6


# ---- the 256x256 phase-pipelined tile (gemm256.h): every epilogue at shapes that dispatch to it ---------------------------
@pytest.fixture
def gemm256_variant(request):
    """Every pipeline schedule kept in gemm256.h is parity-tested, not only the default one."""
    from cryovit_amd import _lib

    kernel, variant = request.param
    set_option_or_skip("use_gemm256", kernel)
    try:
        set_option_or_skip("gemm256_variant", variant)
    except BaseException:
        _lib.set_option("use_gemm256", DEFAULT_GEMM[0])
        raise
    yield request.param
    _lib.set_option("use_gemm256", DEFAULT_GEMM[0])
    _lib.set_option("gemm256_variant", DEFAULT_GEMM[1])


DEFAULT_GEMM = (1, 9)  # (tile kernel: 1 = 8-wave gemm256.h, 2 = 4-wave gemm4w.h; schedule variant, 9 = persistent gemm256p.h)


@pytest.mark.parametrize("gemm256_variant", [(1, 0), (1, 1), (1, 5), (1, 6), (1, 9), (2, 0)], indirect=True)
@pytest.mark.parametrize("M,N,K", [(1029 * 2 + 7, 512, 256), (3000, 256, 1536), (2048, 1536, 4096), (66500, 512, 256),
                                   (17920, 1536, 1536), (18000, 1536, 320)])
def test_gemm256_bf16_and_resid(gpu, M, N, K, gemm256_variant):
    from cryovit_amd._lib import EPI_BF16, EPI_RESID
    from cryovit_amd.engine import ops

    a, w, b, gm = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=K**-0.5), rnd(N, seed=63), rnd(N, seed=64)
    A, Wd = padded_bf16(a, ops.alloc_rows(M), K, gpu), padded_bf16(w, N, K, gpu)
    ref = bf(a).float() @ bf(w).float().T + b
    out = torch.full((ops.alloc_rows(M), N), 7.0, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_BF16, A, Wd, out, b.to(gpu), m=M, n=N)
    got = out[:M].float().cpu()
    assert torch.allclose(got, ref, atol=2e-2, rtol=1e-2), float((got - ref).abs().max())
    assert torch.all(out[M:].float() == 7.0)
    x0 = rnd(M, N, seed=65)
    x = torch.zeros(ops.alloc_rows(M), N, device=gpu)
    x[:M] = x0.to(gpu)
    ops.gemm(EPI_RESID, A, Wd, x, b.to(gpu), m=M, n=N, gamma=gm.to(gpu))
    assert torch.allclose(x[:M].cpu(), x0 + gm * ref, atol=2e-4, rtol=1e-4)
    # (M = 66500: 260 x 2 = 520 tiles -> the launch is cut into 2 whole rounds of 256x256 tiles + a 128x128-tile tail)
    # (M = 17920 = 70 x 256, N = 1536: 420 interior tiles -> the persistent kernel's workgroups walk 1-2 tiles each with exact
    #  store counts in the waits; M = 18000: the same walk with predicated stores and the conservative waits, odd K-tile count)
    # race screen: the pipeline's waits/barriers are hand-counted -- repeated launches must be bit-identical
    outs = []
    for _ in range(8 if M < 10000 else 2):
        o = torch.zeros(ops.alloc_rows(M), N, dtype=torch.bfloat16, device=gpu)
        ops.gemm(EPI_BF16, A, Wd, o, b.to(gpu), m=M, n=N)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("gemm256_variant", [(1, 5), (1, 9)], indirect=True)
def test_gemm256_padded_columns(gpu, gemm256_variant):
    """N not a multiple of the tile: W / bias / gamma padded to 512 rows, 400 valid output columns, the output buffers exactly
    400 (bf16) / 400 (fp32) wide -- the 256-tile kernels must not touch a column >= N (a stray store would land in the next
    row), in the persistent kernel's predicated edge variant and in the one-shot kernel; 9 x 2 tiles, M not a tile multiple."""
    from cryovit_amd._lib import EPI_BF16, EPI_RESID
    from cryovit_amd.engine import ops

    M, N, NP, K = 2100, 400, 512, 320
    a, w, b, gm = rnd(M, K, seed=91), rnd(N, K, seed=92, scale=K**-0.5), rnd(N, seed=93), rnd(N, seed=94)
    A, Wd = padded_bf16(a, ops.alloc_rows(M), K, gpu), padded_bf16(w, NP, K, gpu)
    ref = bf(a).float() @ bf(w).float().T + b
    out = torch.full((ops.alloc_rows(M), N), 7.0, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_BF16, A, Wd, out, padded_f32(b, NP, gpu), m=M, n=N)
    assert torch.allclose(out[:M].float().cpu(), ref, atol=2e-2, rtol=1e-2)
    assert torch.all(out[M:].float() == 7.0)
    x0 = rnd(M, N, seed=95)
    x = torch.full((ops.alloc_rows(M), N), 3.0, device=gpu)
    x[:M] = x0.to(gpu)
    ops.gemm(EPI_RESID, A, Wd, x, padded_f32(b, NP, gpu), m=M, n=N, gamma=padded_f32(gm, NP, gpu))
    assert torch.allclose(x[:M].cpu(), x0 + gm * ref, atol=2e-4, rtol=1e-4)
    assert torch.all(x[M:] == 3.0)


@pytest.mark.parametrize("gemm256_variant", [(1, 5), (1, 9), (2, 0)], indirect=True)
def test_gemm256_swiglu_vt_patch(gpu, gemm256_variant):
    from cryovit_amd._lib import EPI_PATCH, EPI_SWIGLU, EPI_VT
    from cryovit_amd.engine import ops

    # SwiGLU: N = 2*Hp = 1024
    M, K, Hd = 1500, 256, 512
    a, w12, b12 = rnd(M, K, seed=66), rnd(2 * Hd, K, seed=67, scale=K**-0.5), rnd(2 * Hd, seed=68)
    iw = torch.stack([w12[:Hd].reshape(-1, 8, K), w12[Hd:].reshape(-1, 8, K)], 1).reshape(2 * Hd, K)
    ib = torch.stack([b12[:Hd].reshape(-1, 8), b12[Hd:].reshape(-1, 8)], 1).reshape(2 * Hd)
    A = padded_bf16(a, ops.alloc_rows(M), K, gpu)
    out = torch.zeros(ops.alloc_rows(M), Hd, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_SWIGLU, A, bf(iw).to(gpu), out, ib.to(gpu), m=M, n=2 * Hd)
    h = bf(a).float() @ bf(w12).float().T + b12
    ref = F.silu(h[:, :Hd]) * h[:, Hd:]
    assert torch.allclose(out[:M].float().cpu(), ref, atol=2e-2, rtol=1e-2)
    # SwiGLU over 40 x 8 = 320 interior tiles (persistent kernel: several tiles per workgroup, unpredicated stores)
    M2, N2 = 10240, 2048
    a2 = rnd(M2, K, seed=76)
    w2, b2 = rnd(N2, K, seed=77, scale=K**-0.5), rnd(N2, seed=78)
    iw2 = torch.stack([w2[: N2 // 2].reshape(-1, 8, K), w2[N2 // 2 :].reshape(-1, 8, K)], 1).reshape(N2, K)
    ib2 = torch.stack([b2[: N2 // 2].reshape(-1, 8), b2[N2 // 2 :].reshape(-1, 8)], 1).reshape(N2)
    out2 = torch.zeros(ops.alloc_rows(M2), N2 // 2, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_SWIGLU, padded_bf16(a2, ops.alloc_rows(M2), K, gpu), bf(iw2).to(gpu), out2, ib2.to(gpu), m=M2, n=N2)
    h2 = bf(a2).float() @ bf(w2).float().T + b2
    assert torch.allclose(out2[:M2].float().cpu(), F.silu(h2[:, : N2 // 2]) * h2[:, N2 // 2 :], atol=2e-2, rtol=1e-2)
    assert torch.all(out2[M2:] == 0)
    # V^T (MREG orientation): 4 heads -> N = 256
    b_, heads, nt = 5, 4, 261
    ntp, kp, C = ops.round_up(nt, 8), 320, heads * 64
    Mv = b_ * ntp
    av, wv, bv = rnd(Mv, K, seed=69), rnd(C, K, seed=70, scale=K**-0.5), rnd(C, seed=71)
    vt = torch.zeros(b_, heads, 64, kp, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_VT, padded_bf16(av, ops.alloc_rows(Mv), K, gpu), padded_bf16(wv, C, K, gpu), vt, bv.to(gpu), m=Mv, n=C, heads=heads,
             ntp=ntp, kp=kp, ldc=0)
    v = (bf(av).float() @ bf(wv).float().T + bv).reshape(b_, ntp, heads, 64).permute(0, 2, 3, 1)
    got = vt.float().cpu()
    assert torch.allclose(got[..., :ntp], v, atol=2e-2, rtol=1e-2) and torch.all(got[..., ntp:] == 0)
    # patch embed into the token stream: C = 256
    bp, npatch, Cp, n_reg = 5, 256, 256, 4
    ntk = npatch + 5
    ntpp = ops.round_up(ntk, 8)
    ap, wp_, bias = rnd(bp * npatch, 196, seed=72), rnd(Cp, 196, seed=73, scale=196**-0.5), rnd(Cp, seed=74)
    pos = rnd(1 + npatch, Cp, seed=75)
    x = torch.zeros(ops.alloc_rows(bp * ntpp), Cp, device=gpu)
    ops.gemm(EPI_PATCH, padded_bf16(ap, ops.alloc_rows(bp * npatch), 256, gpu), padded_bf16(wp_, Cp, 256, gpu), x, bias.to(gpu),
             m=bp * npatch, n=Cp, pos=pos.to(gpu), npatch=npatch, ntp=ntpp, tok0=1 + n_reg)
    pe = (bf(ap).float() @ bf(wp_).float().T + bias).reshape(bp, npatch, Cp) + pos[1:]
    got = x[: bp * ntpp].cpu().reshape(bp, ntpp, Cp)
    assert torch.allclose(got[:, 1 + n_reg : ntk], pe, atol=1e-4, rtol=1e-4)
    assert torch.all(got[:, :1 + n_reg] == 0) and torch.all(got[:, ntk:] == 0)


def test_gemm256_tail_split_vt_swiglu(gpu):
    """Tail split (whole rounds of 256-tiles + a 128-tile tail over the last rows) for the V^T and SwiGLU epilogues."""
    from cryovit_amd._lib import EPI_SWIGLU, EPI_VT
    from cryovit_amd.engine import ops

    K = 256
    b_, heads, nt = 260, 4, 250  # ntp = 256 -> M = 66560 rows = 260 M-tiles; N = 256 -> 260 tiles ... use N = 512 via 8 heads
    heads = 8
    ntp, kp, C = 256, 256, heads * 64
    Mv = b_ * ntp
    av, wv, bv = rnd(Mv, K, seed=81), rnd(C, K, seed=82, scale=K**-0.5), rnd(C, seed=83)
    vt = torch.zeros(b_, heads, 64, kp, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_VT, padded_bf16(av, ops.alloc_rows(Mv), K, gpu), padded_bf16(wv, C, K, gpu), vt, bv.to(gpu), m=Mv, n=C, heads=heads,
             ntp=ntp, kp=kp, ldc=0)
    v = (bf(av).float() @ bf(wv).float().T + bv).reshape(b_, ntp, heads, 64).permute(0, 2, 3, 1)
    assert torch.allclose(vt.float().cpu(), v, atol=2e-2, rtol=1e-2)
    Hd = 256  # N = 2*Hd = 512
    w12, b12 = rnd(2 * Hd, K, seed=84, scale=K**-0.5), rnd(2 * Hd, seed=85)
    iw = torch.stack([w12[:Hd].reshape(-1, 8, K), w12[Hd:].reshape(-1, 8, K)], 1).reshape(2 * Hd, K)
    ib = torch.stack([b12[:Hd].reshape(-1, 8), b12[Hd:].reshape(-1, 8)], 1).reshape(2 * Hd)
    M = 66500
    out = torch.zeros(ops.alloc_rows(M), Hd, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_SWIGLU, padded_bf16(av[:M], ops.alloc_rows(M), K, gpu), bf(iw).to(gpu), out, ib.to(gpu), m=M, n=2 * Hd)
    h = bf(av[:M]).float() @ bf(w12).float().T + b12
    ref = F.silu(h[:, :Hd]) * h[:, Hd:]
    assert torch.allclose(out[:M].float().cpu(), ref, atol=2e-2, rtol=1e-2)
    assert torch.all(out[M:] == 0)


# ---- round 3: bf16 (hi, lo) residual stream + LayerNorm folded into the consuming GEMMs ------------------------------------------


def _split(t):
    hi = bf(t)
    return hi, bf(t - hi.float())


@pytest.mark.parametrize("rows,C", [(37, 128), (300, 384), (1032, 1536)])
def test_split_stream(gpu, rows, C):
    """fp32 rows -> hi = bf16(x), lo = bf16(x - hi) BIT-EXACT, row constants (rstd, -mean*rstd) of LayerNorm(eps 1e-6)."""
    from cryovit_amd.engine import ops

    x = rnd(rows, C, seed=301, scale=3.0) + 0.7
    x[:, 5] *= 150.0  # an outlier channel
    R = ops.alloc_rows(rows)
    xh = torch.full((R, C), 7.0, dtype=torch.bfloat16, device=gpu)
    xl = torch.full((R, C), 7.0, dtype=torch.bfloat16, device=gpu)
    rs = torch.zeros(R, 2, device=gpu)
    xd = torch.zeros(R, C, device=gpu)
    xd[:rows] = x.to(gpu)
    ops.split_stream(xd, xh, xl, rs, rows=rows, Cdim=C, eps=1e-6)
    hi, lo = _split(x)
    assert torch.equal(xh[:rows].cpu(), hi) and torch.equal(xl[:rows].cpu(), lo)
    assert torch.all(xh[rows:].float() == 7.0) and torch.all(xl[rows:].float() == 7.0)
    xd64 = x.double()
    mu, var = xd64.mean(1), xd64.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    got = rs[:rows].cpu().double()
    assert torch.allclose(got[:, 0], rstd, rtol=2e-6, atol=0) and torch.allclose(got[:, 1], -mu * rstd, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,K", [(700, 384, 256), (300, 128, 192), (2048, 1536, 512), (2311, 512, 256), (5000, 1536, 1536),
                                   (66500, 512, 256), (66300, 512, 128)])
def test_gemm_resid_hl(gpu, M, N, K):
    """CVX_EPI_RESID_HL: x = hi + lo; x += gamma * (acc + bias); hi', lo' = split(x); stat_part[n/64][m] = row sums of the new x
    over 64-column slots.  Shapes cover the 128-wide tiles (M < 1024), the persistent 256 tile with interior tiles only (2048),
    with a ragged last M tile (2311), 120 tiles on the persistent kernel (5000 x 1536) and the main + tail split (66500 / 66300 x 512:
    520 tiles = 2 whole rounds of 256 x 256 tiles + the last 964 / 764 rows on the 64 x 128 tile's 4-deep ring, with a ragged last
    tile; K = 128 is fewer K tiles than the ring has stages)."""
    from cryovit_amd._lib import EPI_RESID_HL
    from cryovit_amd.engine import ops

    a, w, b, g = rnd(M, K, seed=311), rnd(N, K, seed=312, scale=K**-0.5), rnd(N, seed=313), rnd(N, seed=314)
    x0 = rnd(M, N, seed=315, scale=2.0)
    x0[:, 3] *= 100.0
    hi0, lo0 = _split(x0)
    R = ops.alloc_rows(M)
    xh = torch.full((R, N), 7.0, dtype=torch.bfloat16, device=gpu)
    xl = torch.full((R, N), 7.0, dtype=torch.bfloat16, device=gpu)
    xh[:M], xl[:M] = hi0.to(gpu), lo0.to(gpu)
    part = torch.full((N // 64, R, 2), float("nan"), device=gpu)
    ops.gemm(EPI_RESID_HL, padded_bf16(a, R, K, gpu), padded_bf16(w, N, K, gpu), xh, b.to(gpu), m=M, n=N, gamma=g.to(gpu), out2=xl, stat_part=part)
    ref = (hi0.float() + lo0.float()).double() + g.double() * (bf(a).double() @ bf(w).double().T + b.double())
    got_h, got_l = xh[:M].float().cpu(), xl[:M].float().cpu()
    got = (got_h + got_l).double()
    # the pair carries 16+ significant bits; the fp32 accumulation order of the MFMA differs from the CPU's
    assert torch.allclose(got, ref, atol=2e-4, rtol=3e-5), float((got - ref).abs().max())
    # hi = bf16(x); re-rounding hi + lo can only differ where lo itself was rounded up to exactly half an ulp of hi (a tie)
    assert float((got_h != bf(got_h + got_l).float()).float().mean()) <= 5e-3, "hi is not bf16(hi + lo)"  # (measured: 1e-3)
    assert float((got_l.abs() - got_h.abs() * 2.0**-8).max()) <= 0, "lo is not the rounding remainder of hi"
    assert torch.all(xh[M:].float() == 7.0) and torch.all(xl[M:].float() == 7.0), "rows beyond M were written"
    p = part[:, :M].cpu().double()
    assert torch.isfinite(p).all()
    gs = got.reshape(M, N // 64, 64)
    s_ref, q_ref = gs.sum(-1).t(), (gs * gs).sum(-1).t()
    scale = gs.abs().sum(-1).t()
    assert float(((p[..., 0] - s_ref).abs() / (scale + 1e-3)).max()) <= 2e-5  # (sums of x before the split vs of hi + lo)
    assert torch.allclose(p[..., 1], q_ref, rtol=5e-5, atol=1e-4)
    # the statistics kernel on those partials
    rs = torch.zeros(R, 2, device=gpu)
    ops.rowstat_finalize(part, rs, rows=M, Cdim=N, eps=1e-6)
    mu, var = got.mean(1), got.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    r = rs[:M].cpu().double()
    assert torch.allclose(r[:, 0], rstd, rtol=2e-4) and torch.allclose(r[:, 1], -mu * rstd, rtol=2e-3, atol=2e-4)


def _ln_fold_pack(w, bias, gamma, beta, n_pad, dev):
    """bf16(W * gamma) and [2, n_pad] = (b + W beta | column sums of the rounded weight): VitEngine._pack's ln_linear."""
    wq = torch.zeros(n_pad, w.shape[1], dtype=torch.bfloat16)
    wq[: w.shape[0]] = bf(w * gamma[None, :])
    bc = torch.zeros(2, n_pad)
    bc[0, : w.shape[0]] = (bias.double() + w.double() @ beta.double()).float()
    bc[1] = wq.double().sum(1).float()
    return wq.to(dev), bc.to(dev)


def _ln_ref(a, w, bias, gamma, beta, rs_dev, M):
    """The folded form on the CPU from the SAME row constants the GPU read."""
    rs = rs_dev[:M].cpu()
    wq = bf(w * gamma[None, :]).float()
    bp = (bias.double() + w.double() @ beta.double()).float()
    return rs[:, :1] * (bf(a).float() @ wq.T) + rs[:, 1:] * wq.sum(1) + bp


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (1000, 192, 128), (2048, 1024, 256), (2311, 768, 384), (5000, 3072, 1536)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_bf16_ln_fold(gpu, M, N, K, gelu):
    """BF16 / BF16_GELU epilogue with the LayerNorm folded in: A = raw rows (outlier channel, non-zero mean), row constants from
    cvx_split_stream, against (i) the folded arithmetic on the CPU and (ii) LayerNorm -> Linear in fp32."""
    from cryovit_amd._lib import EPI_BF16, EPI_BF16_GELU
    from cryovit_amd.engine import ops

    x = rnd(M, K, seed=321, scale=1.5) + 0.3
    x[:, 7] *= 40.0
    w, bias = rnd(N, K, seed=322, scale=K**-0.5), rnd(N, seed=323)
    gamma, beta = torch.exp(rnd(K, seed=324) * 0.5), rnd(K, seed=325, scale=0.2)
    R = ops.alloc_rows(M)
    xd = torch.zeros(R, K, device=gpu)
    xd[:M] = x.to(gpu)
    xh, xl, rs = torch.zeros(R, K, dtype=torch.bfloat16, device=gpu), torch.zeros(R, K, dtype=torch.bfloat16, device=gpu), torch.zeros(R, 2, device=gpu)
    ops.split_stream(xd, xh, xl, rs, rows=M, Cdim=K, eps=1e-6)
    n_pad = ops.round_up(N, 128)
    wq, bc = _ln_fold_pack(w, bias, gamma, beta, n_pad, gpu)
    out = torch.full((R, N), 7.0, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_BF16_GELU if gelu else EPI_BF16, xh, wq, out, bc, m=M, n=N, ln_rowstat=rs)
    ref = _ln_ref(x, w, bias, gamma, beta, rs, M)
    full = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-6), w, bias)
    if gelu:
        ref, full = F.gelu(ref), F.gelu(full)
    got = out[:M].float().cpu()
    assert torch.allclose(got, ref, atol=3e-2, rtol=1e-2), float((got - ref).abs().max())
    # vs the un-folded fp32 computation: the A operand is bf16(x) with a 40x outlier channel (error ~ 2^-9 * 60 * |w| per product)
    assert float((got - full).abs().mean()) <= 2e-2 and float((got - full).abs().max()) <= 0.35, (float((got - full).abs().mean()), float((got - full).abs().max()))
    assert torch.all(out[M:].float() == 7.0), "rows beyond M were written"


@pytest.mark.parametrize("M", [500, 2560, 2311])
def test_gemm_swiglu_ln_fold(gpu, M):
    from cryovit_amd._lib import EPI_SWIGLU
    from cryovit_amd.engine import ops

    K, Hd = 128, 344
    Hp = ops.round_up(Hd, 64)
    x = rnd(M, K, seed=331, scale=1.5) - 0.2
    w12, b12 = rnd(2 * Hd, K, seed=332, scale=K**-0.5), rnd(2 * Hd, seed=333)
    gamma, beta = torch.exp(rnd(K, seed=334) * 0.5), rnd(K, seed=335, scale=0.2)
    aw, bw, ab, bb = torch.zeros(Hp, K), torch.zeros(Hp, K), torch.zeros(Hp), torch.zeros(Hp)
    aw[:Hd], bw[:Hd], ab[:Hd], bb[:Hd] = w12[:Hd], w12[Hd:], b12[:Hd], b12[Hd:]
    iw = torch.stack([aw.reshape(-1, 8, K), bw.reshape(-1, 8, K)], 1).reshape(2 * Hp, K)
    ib = torch.stack([ab.reshape(-1, 8), bb.reshape(-1, 8)], 1).reshape(2 * Hp)
    R = ops.alloc_rows(M)
    xd = torch.zeros(R, K, device=gpu)
    xd[:M] = x.to(gpu)
    xh, xl, rs = torch.zeros(R, K, dtype=torch.bfloat16, device=gpu), torch.zeros(R, K, dtype=torch.bfloat16, device=gpu), torch.zeros(R, 2, device=gpu)
    ops.split_stream(xd, xh, xl, rs, rows=M, Cdim=K, eps=1e-6)
    wq, bc = _ln_fold_pack(iw, ib, gamma, beta, 2 * Hp, gpu)
    out = torch.zeros(R, Hp, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_SWIGLU, xh, wq, out, bc, m=M, n=2 * Hp, ln_rowstat=rs)
    h = _ln_ref(x, w12, b12, gamma, beta, rs, M)
    ref = F.silu(h[:, :Hd]) * h[:, Hd:]
    got = out[:M].float().cpu()
    assert torch.allclose(got[:, :Hd], ref, atol=2e-2, rtol=1e-2), float((got[:, :Hd] - ref).abs().max())
    assert torch.all(got[:, Hd:] == 0)


@pytest.mark.parametrize("b,heads,nt,K", [(3, 2, 29, 128), (4, 4, 1029, 256), (5, 4, 1029, 256)])
def test_gemm_vt_ln_fold(gpu, b, heads, nt, K):
    """V^T epilogue (MREG orientation) with the fold: small tiles, the persistent tile alone (4 x 1032 rows = 16 M tiles + a ragged
    one) and the main + tail split."""
    from cryovit_amd._lib import EPI_VT
    from cryovit_amd.engine import ops

    ntp, kp, C = ops.round_up(nt, 8), ops.round_up(nt, 64), heads * 64
    M = b * ntp
    x = rnd(M, K, seed=341, scale=1.5) + 0.4
    w, bias = rnd(C, K, seed=342, scale=K**-0.5), rnd(C, seed=343)
    gamma, beta = torch.exp(rnd(K, seed=344) * 0.5), rnd(K, seed=345, scale=0.2)
    R = ops.alloc_rows(M)
    xd = torch.zeros(R, K, device=gpu)
    xd[:M] = x.to(gpu)
    xh, xl, rs = torch.zeros(R, K, dtype=torch.bfloat16, device=gpu), torch.zeros(R, K, dtype=torch.bfloat16, device=gpu), torch.zeros(R, 2, device=gpu)
    ops.split_stream(xd, xh, xl, rs, rows=M, Cdim=K, eps=1e-6)
    wq, bc = _ln_fold_pack(w, bias, gamma, beta, ops.round_up(C, 128), gpu)
    vt = torch.zeros(b, heads, 64, kp, dtype=torch.bfloat16, device=gpu)
    ops.gemm(EPI_VT, xh, wq, vt, bc, m=M, n=C, heads=heads, ntp=ntp, kp=kp, ldc=0, ln_rowstat=rs)
    v = _ln_ref(x, w, bias, gamma, beta, rs, M).reshape(b, ntp, heads, 64).permute(0, 2, 3, 1)
    got = vt.float().cpu()
    assert torch.allclose(got[..., :ntp], v, atol=2e-2, rtol=1e-2), float((got[..., :ntp] - v).abs().max())
    assert torch.all(got[..., ntp:] == 0)


def test_final_norm_hl_equals_fp32_form(gpu):
    """cvx_final_norm_features_hl on (hi, lo) == cvx_final_norm_features on the fp32 array hi + lo, bit for bit, all three outputs."""
    from cryovit_amd.engine import ops

    slices, hp, wp, C, n_reg = 3, 5, 7, 384, 4
    npatch, tok0 = hp * wp, 1 + n_reg
    ntp = ops.round_up(npatch + tok0, 8)
    R = ops.alloc_rows(slices * ntp)
    x = rnd(R, C, seed=351, scale=4.0) + 1.0
    hi, lo = _split(x)
    xs = (hi.float() + lo.float()).to(gpu)
    w, b = (rnd(C, seed=352) * 0.1 + 1).to(gpu), rnd(C, seed=353).to(gpu)
    outs = []
    for form in (0, 1):
        f16 = torch.zeros(C, slices, hp, wp, dtype=torch.float16, device=gpu)
        cl = torch.zeros(slices * npatch, C, dtype=torch.float16, device=gpu)
        tk = torch.zeros(slices, npatch, C, device=gpu)
        kw = dict(slices=slices, ntp=ntp, tok0=tok0, hp=hp, wp=wp, Cdim=C, feats_f16=f16, d_total=slices, d0=0, feats_cl=cl, tokens_f32=tk)
        if form:
            ops.final_norm_features_hl(hi.to(gpu), lo.to(gpu), w, b, 1e-6, **kw)
        else:
            ops.final_norm_features(xs, w, b, 1e-6, **kw)
        outs.append((f16, cl, tk))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    ref = F.layer_norm(xs.cpu().reshape(-1, C)[: slices * ntp].reshape(slices, ntp, C)[:, tok0 : tok0 + npatch], (C,), w.cpu(), b.cpu(), 1e-6)
    assert torch.allclose(outs[1][2].cpu(), ref, atol=1e-4, rtol=1e-4)


def test_variants_on_the_ablation_build(gpu):
    """The kernel variants that were measured and rejected (GEMM schedules 1 / 5 / 6 / 7 / 8 and the 4-wave tile, attention 0-6
    incl. the 64-rows-per-wave forms, the tile-halo convolution) are compiled into ``libcryovit_hip_ablation.so`` only; their
    parity tests -- the parametrised cases this module skips on the product library -- run here, in a child process that loads
    that build (``CVX_ABLATION_LIB=1``)."""
    import subprocess
    import sys
    from pathlib import Path

    if ABLATION:
        pytest.skip("already running on the ablation build")
    root = Path(__file__).resolve().parent.parent
    if not (root / "cryovit_amd" / "libcryovit_hip_ablation.so").exists():
        pytest.skip("libcryovit_hip_ablation.so not built (python -m cryovit_amd.build --ablation)")
    from cryovit_amd.build import _fingerprint

    stamp = root / "cryovit_amd" / "build_ablation" / "fingerprint"
    if not stamp.exists() or stamp.read_text() != _fingerprint() + "+ablation":
        pytest.skip("libcryovit_hip_ablation.so is older than the sources (python -m cryovit_amd.build --ablation; __graft_entry__.build() does)")
    env = dict(os.environ, CVX_ABLATION_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", str(Path(__file__)), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "test_attention or test_gemm256 or test_conv3d"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail
