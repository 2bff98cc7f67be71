"""Thin torch-tensor wrappers over the C ABI (``include/cryovit_hip.h``).

torch is plumbing here: it owns device memory and the HIP stream; every arithmetic op below is a call into
``libcryovit_hip.so``.  Every wrapper launches on the current stream OF THE DEVICE ITS TENSORS LIVE ON and makes that
device the active HIP device for the duration of the C call (the library's ``hipFuncSetAttribute`` / launch calls act on
the active device), so a rank of a multi-GPU launch never touches GPU 0 by accident.  Tensors on different devices in
one call are rejected.
"""

from __future__ import annotations

import ctypes as C

import torch

from cryovit_amd import _lib
from cryovit_amd._lib import Conv3dDesc, GemmDesc, check

ROW_PAD = 256  # activation matrices are allocated to a multiple of this many rows (+ one spare tile)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def alloc_rows(m: int) -> int:
    return round_up(m, ROW_PAD) + ROW_PAD


def norm_device(device) -> torch.device:
    """torch.device with an explicit index ("cuda" -> the active device), so it compares equal to ``tensor.device``."""
    d = torch.device(device)
    if d.type != "cuda":
        raise _lib.CvxError(f"cryovit_amd runs on HIP devices only, got {d}")
    return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())


def _stream(device=None) -> int:
    """Raw hipStream_t of torch's current stream on `device` (default: the active device)."""
    return torch.cuda.current_stream(device).cuda_stream


def _p(t) -> int | None:
    return None if t is None else t.data_ptr()


def _dev_check(*ts) -> torch.device:
    """Validates the operands of one C call and returns the device they all live on."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.CvxError("cryovit_amd ops need device (HIP) tensors; there is no CPU path")
        if not t.is_contiguous():  # the kernels index raw pointers: a strided view would be read as garbage
            raise _lib.CvxError(f"non-contiguous tensor {tuple(t.shape)} strides {t.stride()} passed to a HIP op")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.CvxError(f"operands of one HIP op live on different devices ({dev} and {t.device})")
    if dev is None:
        raise _lib.CvxError("HIP op called without any device tensor")
    return dev


def call(dev: torch.device, what: str, fn, *args) -> None:
    """fn(*args, stream) with `dev` active and on `dev`'s current stream; raises CvxError on a non-zero status."""
    with torch.cuda.device(dev):
        check(fn(*args, _stream(dev)), what)


def gemm(epilogue: int, a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias: torch.Tensor, *, m: int, n: int,
         gamma=None, pos=None, npatch=0, ntp=0, tok0=0, heads=0, kp=0, H=0, W=0, cout=0, act=0, ldc=None, convt_up_z=0,
         ln_rowstat=None, out2=None, stat_part=None) -> None:
    """C = A W^T with a fused epilogue.  a: bf16 [M_alloc, lda]; w: bf16 [n_pad, k_pad] (packed).  fp16 operands (both a and
    w) select the fp16 MFMA and fp16 outputs (plain / GELU / ConvT epilogues: the segmentation head).
    ln_rowstat (fp32 [rows, 2]): LayerNorm folded into the GEMM -- bias is then fp32 [2, n_pad] (b' | column sums of the packed
    weight).  EPI_RESID_HL: out / out2 = the bf16 hi / lo halves of the residual stream, stat_part fp32 [n_pad/64, rows, 2]."""
    dev = _dev_check(a, w, out, bias, gamma, pos, ln_rowstat, out2, stat_part)
    assert a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float16) and bias.dtype == torch.float32
    assert a.stride(-1) == 1 and w.is_contiguous() and bias.numel() >= w.shape[0] * (2 if ln_rowstat is not None else 1)
    if ln_rowstat is not None and (ln_rowstat.dtype != torch.float32 or ln_rowstat.numel() < 2 * round_up(m, ROW_PAD)):
        raise _lib.CvxError("gemm: ln_rowstat must be fp32 [rows, 2] with rows >= m rounded up to 256 (whole tiles are fetched)")
    if epilogue == _lib.EPI_RESID_HL:
        if out2 is None or stat_part is None or out.dtype != torch.bfloat16 or out2.dtype != torch.bfloat16 or stat_part.dim() != 3:
            raise _lib.CvxError("gemm: EPI_RESID_HL needs bf16 out / out2 and stat_part fp32 [n_pad/64, rows, 2]")
        if stat_part.shape[0] * 64 < n or stat_part.shape[1] < round_up(m, 256) or out2.stride(0) != out.stride(0):
            raise _lib.CvxError("gemm: stat_part too small or hi / lo leading dimensions differ")
    d = GemmDesc()
    d.epilogue = epilogue
    d.a, d.lda = a.data_ptr(), a.stride(0)
    d.w, d.ldw = w.data_ptr(), w.stride(0)
    d.m, d.n, d.n_pad, d.k_pad = m, n, w.shape[0], w.shape[1]
    d.out = out.data_ptr()
    d.ldc = ldc if ldc is not None else (out.stride(0) if out.dim() >= 2 else 0)
    d.bias, d.gamma = bias.data_ptr(), _p(gamma)
    d.pos, d.ldpos = _p(pos), (pos.stride(0) if pos is not None else 0)
    d.npatch, d.ntp, d.tok0, d.heads, d.kp = npatch, ntp, tok0, heads, kp
    d.H, d.W, d.cout, d.act = H, W, cout, act
    d.convt_up_z = convt_up_z
    d.ln_rowstat, d.out2, d.stat_part = _p(ln_rowstat), _p(out2), _p(stat_part)
    d.stat_rows = stat_part.shape[1] if stat_part is not None else 0
    d.dtype = _lib.DTYPE_F16 if a.dtype == torch.float16 else _lib.DTYPE_BF16
    call(dev, "cvx_gemm_bf16", _lib.load().cvx_gemm_bf16, C.byref(d))


def conv3d(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, zero_page: torch.Tensor, *, Cin: int,
           D: int, H: int, W: int, dil: int, cout: int, act: int) -> None:
    dev = _dev_check(x, w, bias, out, zero_page)
    d = Conv3dDesc()
    d.in_, d.w, d.bias, d.zero_page, d.out = x.data_ptr(), w.data_ptr(), bias.data_ptr(), zero_page.data_ptr(), out.data_ptr()
    d.C, d.D, d.H, d.W, d.dil, d.cout = Cin, D, H, W, dil, cout
    d.n_pad, d.k_pad, d.act = w.shape[0], w.shape[1], act
    call(dev, "cvx_conv3d_f16", _lib.load().cvx_conv3d_f16, C.byref(d))


def conv2s2(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, zero_page: torch.Tensor, *, Cin: int, D: int, H: int,
            W: int, cout: int, act: int) -> None:
    """nn.Conv3d(Cin, cout, 2, stride=2) on a channels-last fp16 volume [D,H,W,Cin] -> [D/2,H/2,W/2,cout]; w fp16 [n_pad, 8*Cin]."""
    dev = _dev_check(x, w, bias, out, zero_page)
    d = Conv3dDesc()
    d.in_, d.w, d.bias, d.zero_page, d.out = x.data_ptr(), w.data_ptr(), bias.data_ptr(), zero_page.data_ptr(), out.data_ptr()
    d.C, d.D, d.H, d.W, d.dil, d.cout = Cin, D, H, W, 1, cout
    d.n_pad, d.k_pad, d.act = w.shape[0], w.shape[1], act
    call(dev, "cvx_conv2s2_f16", _lib.load().cvx_conv2s2_f16, C.byref(d))


def concat_channels(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, nvox: int, Ca: int, Cb: int) -> None:
    dev = _dev_check(a, b, out)
    if a.dtype != torch.float16 or b.dtype != torch.float16 or out.dtype != torch.float16 or out.numel() < nvox * (Ca + Cb):
        raise _lib.CvxError("concat_channels: fp16 tensors, out >= nvox*(Ca+Cb) elements")
    call(dev, "cvx_concat_channels_f16", _lib.load().cvx_concat_channels_f16, a.data_ptr(), Ca, b.data_ptr(), Cb, out.data_ptr(), nvox)


def pointwise_out(x: torch.Tensor, w: torch.Tensor, bias: float, logits, probs, *, nvox: int, Cdim: int) -> None:
    dev = _dev_check(x, w, logits, probs)
    if x.dtype != torch.float16 or w.dtype != torch.float32 or w.numel() < Cdim:
        raise _lib.CvxError("pointwise_out: x fp16 [nvox, C], w fp32 [C]")
    call(dev, "cvx_pointwise_out_f16", _lib.load().cvx_pointwise_out_f16, x.data_ptr(), w.data_ptr(), float(bias), _p(logits), _p(probs), nvox,
         Cdim)


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, rows: int, Cdim: int, eps: float) -> None:
    dev = _dev_check(x, w, b, out)
    call(dev, "cvx_layernorm_bf16", _lib.load().cvx_layernorm_bf16, x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(),
         out.data_ptr(), out.stride(0), rows, Cdim, eps)


def attention(qk: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, *, slices: int, heads: int, ntok: int, ntp: int,
              kp: int) -> None:
    dev = _dev_check(qk, vt, out)
    call(dev, "cvx_attention_bf16", _lib.load().cvx_attention_bf16, qk.data_ptr(), qk.stride(0), vt.data_ptr(), out.data_ptr(),
         out.stride(0), slices, heads, ntok, ntp, kp)


def attention_qkv(qkv: torch.Tensor, out: torch.Tensor, *, slices: int, heads: int, ntok: int, ntp: int) -> None:
    """Attention over one [rows, >= 3C] bf16 buffer holding Q (log2 units) | K | V row-major (the output of one qkv GEMM)."""
    dev = _dev_check(qkv, out)
    assert qkv.dtype == out.dtype == torch.bfloat16 and qkv.shape[1] >= 3 * heads * 64 and qkv.shape[0] >= slices * ntp + 64
    call(dev, "cvx_attention_qkv_bf16", _lib.load().cvx_attention_qkv_bf16, qkv.data_ptr(), qkv.stride(0), out.data_ptr(), out.stride(0), slices,
         heads, ntok, ntp)


def preprocess_patches(slices: torch.Tensor, out: torch.Tensor) -> None:
    dev = _dev_check(slices, out)
    assert slices.dim() == 3 and slices.is_contiguous() and slices.dtype in (torch.uint8, torch.float32)
    b, H, W = slices.shape
    call(dev, "cvx_preprocess_patches", _lib.load().cvx_preprocess_patches, slices.data_ptr(), int(slices.dtype == torch.uint8),
         b, H, W, out.data_ptr(), out.stride(0))


def init_tokens(x: torch.Tensor, cls_pos0: torch.Tensor, reg: torch.Tensor, *, n_reg: int, slices: int, ntok: int, ntp: int,
                Cdim: int) -> None:
    dev = _dev_check(x, cls_pos0, reg)
    call(dev, "cvx_init_tokens", _lib.load().cvx_init_tokens, x.data_ptr(), x.stride(0), cls_pos0.data_ptr(), reg.data_ptr(),
         n_reg, slices, ntok, ntp, Cdim)


def final_norm_features(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, *, slices: int, ntp: int, tok0: int,
                        hp: int, wp: int, Cdim: int, feats_f16, d_total: int, d0: int, feats_cl, tokens_f32=None) -> None:
    dev = _dev_check(x, w, b, feats_f16, feats_cl, tokens_f32)
    call(dev, "cvx_final_norm_features", _lib.load().cvx_final_norm_features, x.data_ptr(), x.stride(0), w.data_ptr(),
         b.data_ptr(), eps, slices, ntp, tok0, hp, wp, Cdim, _p(feats_f16), d_total, d0, _p(feats_cl), _p(tokens_f32))


def final_norm_features_hl(xh: torch.Tensor, xl: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, *, slices: int, ntp: int,
                           tok0: int, hp: int, wp: int, Cdim: int, feats_f16, d_total: int, d0: int, feats_cl, tokens_f32=None) -> None:
    dev = _dev_check(xh, xl, w, b, feats_f16, feats_cl, tokens_f32)
    assert xh.dtype == xl.dtype == torch.bfloat16 and xh.stride(0) == xl.stride(0)
    call(dev, "cvx_final_norm_features_hl", _lib.load().cvx_final_norm_features_hl, xh.data_ptr(), xl.data_ptr(), xh.stride(0),
         w.data_ptr(), b.data_ptr(), eps, slices, ntp, tok0, hp, wp, Cdim, _p(feats_f16), d_total, d0, _p(feats_cl), _p(tokens_f32))


def split_stream(x: torch.Tensor, xh: torch.Tensor, xl: torch.Tensor, rowstat: torch.Tensor, *, rows: int, Cdim: int, eps: float) -> None:
    """fp32 rows -> bf16 (hi, lo) pair + LayerNorm row constants (rstd, -mean*rstd)."""
    dev = _dev_check(x, xh, xl, rowstat)
    assert x.dtype == torch.float32 and xh.dtype == xl.dtype == torch.bfloat16 and rowstat.dtype == torch.float32
    assert xh.stride(0) == xl.stride(0) and min(x.shape[0], xh.shape[0], xl.shape[0]) >= rows and rowstat.numel() >= 2 * rows
    call(dev, "cvx_split_stream", _lib.load().cvx_split_stream, x.data_ptr(), x.stride(0), xh.data_ptr(), xl.data_ptr(), xh.stride(0),
         rowstat.data_ptr(), rows, Cdim, eps)


def merge_stream(xh: torch.Tensor, xl: torch.Tensor, x: torch.Tensor, *, rows: int, Cdim: int) -> None:
    """bf16 (hi, lo) pair -> fp32 rows x = hi + lo."""
    dev = _dev_check(xh, xl, x)
    assert xh.dtype == xl.dtype == torch.bfloat16 and x.dtype == torch.float32 and xh.stride(0) == xl.stride(0)
    call(dev, "cvx_merge_stream", _lib.load().cvx_merge_stream, xh.data_ptr(), xl.data_ptr(), xh.stride(0), x.data_ptr(), x.stride(0), rows, Cdim)


def rowstat_finalize(stat_part: torch.Tensor, rowstat: torch.Tensor, *, rows: int, Cdim: int, eps: float) -> None:
    dev = _dev_check(stat_part, rowstat)
    assert stat_part.dtype == rowstat.dtype == torch.float32 and stat_part.dim() == 3 and rowstat.numel() >= 2 * rows
    call(dev, "cvx_rowstat_finalize", _lib.load().cvx_rowstat_finalize, stat_part.data_ptr(), Cdim // 64, stat_part.shape[1],
         rowstat.data_ptr(), rows, Cdim, eps)


def im2col_patches(x: torch.Tensor, out: torch.Tensor) -> None:
    dev = _dev_check(x, out)
    assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32 and x.is_contiguous()
    b, _, Hi, Wi = x.shape
    call(dev, "cvx_im2col_patches", _lib.load().cvx_im2col_patches, x.data_ptr(), b, Hi, Wi, out.data_ptr(), out.stride(0))


def features_to_channels_last(feats_f16: torch.Tensor, out_cl: torch.Tensor) -> None:
    dev = _dev_check(feats_f16, out_cl)
    Cdim = feats_f16.shape[0]
    nvox = feats_f16.numel() // Cdim
    call(dev, "cvx_features_to_channels_last", _lib.load().cvx_features_to_channels_last, feats_f16.data_ptr(),
         out_cl.data_ptr(), Cdim, nvox)


def gn_stats_size(G: int) -> int:
    return 2 * G * (1 + _lib.GN_BLOCKS)


def groupnorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, stats: torch.Tensor, *, nvox: int,
              Cdim: int, G: int, eps: float, act: int = 0) -> None:
    """GroupNorm (G = C: InstanceNorm3d with affine) over a channels-last fp16 volume, optionally with GELU fused (act=1)."""
    dev = _dev_check(x, w, b, out, stats)
    if stats.dtype != torch.float32 or stats.numel() < gn_stats_size(G):
        raise _lib.CvxError(f"groupnorm: stats must be fp32 with >= {gn_stats_size(G)} elements (2*G*(1+CVX_GN_BLOCKS))")
    if act:
        call(dev, "cvx_groupnorm_act_f16", _lib.load().cvx_groupnorm_act_f16, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(),
             stats.data_ptr(), nvox, Cdim, G, eps, act)
        return
    call(dev, "cvx_groupnorm_f16", _lib.load().cvx_groupnorm_f16, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(),
         stats.data_ptr(), nvox, Cdim, G, eps)


def groupnorm_into(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, col0: int, ldo: int, stats: torch.Tensor, *, nvox: int,
                   Cdim: int, G: int, eps: float, act: int = 0, out2=None) -> None:
    """GroupNorm (+ GELU) whose result lands in columns [col0, col0 + C) of a wider channels-last fp16 buffer with rows `ldo` elements
    apart (flat tensor `out`); `out2`: optional second dense [nvox, C] copy."""
    dev = _dev_check(x, w, b, out, stats, out2)
    if stats.dtype != torch.float32 or stats.numel() < gn_stats_size(G):
        raise _lib.CvxError(f"groupnorm: stats must be fp32 with >= {gn_stats_size(G)} elements (2*G*(1+CVX_GN_BLOCKS))")
    if out.dtype != torch.float16 or col0 % 8 or col0 + Cdim > ldo or out.numel() < nvox * ldo:
        raise _lib.CvxError("groupnorm_into: out fp16 with >= nvox*ldo elements, col0 a multiple of 8, col0 + C <= ldo")
    call(dev, "cvx_groupnorm_act_strided_f16", _lib.load().cvx_groupnorm_act_strided_f16, x.data_ptr(), w.data_ptr(), b.data_ptr(),
         out.data_ptr() + 2 * col0, ldo, _p(out2), stats.data_ptr(), nvox, Cdim, G, eps, act)


_dice_scratch = {}


def dice_scratch(device) -> torch.Tensor:
    """Per (device, stream) partial-sum buffer of the fused output kernel (two volumes may be in flight on two streams)."""
    key = (torch.device(device), _stream(device))
    if key not in _dice_scratch:
        _dice_scratch[key] = torch.zeros(3 * _lib.DICE_BLOCKS, dtype=torch.float32, device=device)
    return _dice_scratch[key]


def conv3_out_fused(x: torch.Tensor, w: torch.Tensor, bias: float, logits, probs, labels, dice, *, D: int, H: int, W: int,
                    mask=None, mask_threshold: float = 0.5) -> None:
    dev = _dev_check(x, w, logits, probs, labels, dice, mask)
    scratch = dice_scratch(x.device) if labels is not None else None
    call(dev, "cvx_conv3_out_fused", _lib.load().cvx_conv3_out_fused, x.data_ptr(), w.data_ptr(), float(bias), _p(logits),
         _p(probs), _p(labels), _p(dice), _p(scratch), _p(mask), float(mask_threshold), D, H, W)


def dice_sums(probs: torch.Tensor, labels: torch.Tensor, dice: torch.Tensor, thr: float = 0.5) -> None:
    dev = _dev_check(probs, labels, dice)
    call(dev, "cvx_dice_sums", _lib.load().cvx_dice_sums, probs.data_ptr(), labels.data_ptr(), dice.data_ptr(), probs.numel(),
         thr)


# ---- training-side pieces (SURVEY s.8f N4) ----


def dice_loss_forward(probs: torch.Tensor, labels: torch.Tensor, out4: torch.Tensor) -> None:
    """out4 (fp32[4], device) = I, Sy, Sp, loss over labels > -1; probs fp32 contiguous, labels int8 of the same numel."""
    dev = _dev_check(probs, labels, out4)
    if probs.dtype != torch.float32 or labels.dtype != torch.int8 or probs.numel() != labels.numel() or out4.numel() < 4:
        raise _lib.CvxError("dice_loss_forward: probs fp32, labels int8 of the same size, out4 fp32[4]")
    if not (probs.is_contiguous() and labels.is_contiguous()):
        raise _lib.CvxError("dice_loss_forward: contiguous tensors required")
    call(dev, "cvx_dice_loss_forward", _lib.load().cvx_dice_loss_forward, probs.data_ptr(), labels.data_ptr(), probs.numel(),
         dice_scratch(probs.device).data_ptr(), out4.data_ptr())


def dice_loss_backward(probs, logits, labels: torch.Tensor, sums4: torch.Tensor, grad_out: float, grad: torch.Tensor,
                       through_sigmoid: bool = False) -> None:
    dev = _dev_check(probs, logits, labels, sums4, grad)
    if labels.dtype != torch.int8 or grad.dtype != torch.float32 or grad.numel() != labels.numel():
        raise _lib.CvxError("dice_loss_backward: labels int8, grad fp32 of the same size")
    call(dev, "cvx_dice_loss_backward", _lib.load().cvx_dice_loss_backward, _p(probs), _p(logits), labels.data_ptr(), labels.numel(),
         sums4.data_ptr(), float(grad_out), int(through_sigmoid), grad.data_ptr())


def focal_loss_forward(x: torch.Tensor, labels: torch.Tensor, gamma: float, out4: torch.Tensor) -> None:
    dev = _dev_check(x, labels, out4)
    if x.dtype != torch.float32 or labels.dtype != torch.int8 or x.numel() != labels.numel() or out4.numel() < 4:
        raise _lib.CvxError("focal_loss_forward: x fp32, labels int8 of the same size, out4 fp32[4]")
    call(dev, "cvx_focal_loss_forward", _lib.load().cvx_focal_loss_forward, x.data_ptr(), labels.data_ptr(), x.numel(), float(gamma),
         dice_scratch(x.device).data_ptr(), out4.data_ptr())


def focal_loss_backward(x: torch.Tensor, labels: torch.Tensor, gamma: float, stats4: torch.Tensor, grad_out: float, grad: torch.Tensor) -> None:
    dev = _dev_check(x, labels, stats4, grad)
    call(dev, "cvx_focal_loss_backward", _lib.load().cvx_focal_loss_backward, x.data_ptr(), labels.data_ptr(), x.numel(), float(gamma),
         stats4.data_ptr(), float(grad_out), grad.data_ptr())


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float, step: int) -> None:
    dev = _dev_check(p, g, m, v)
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise _lib.CvxError("adamw_step: four contiguous fp32 tensors of one size")
    call(dev, "cvx_adamw_step", _lib.load().cvx_adamw_step, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
         float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step))


# ---- alternate encoder (SAM2 Hiera) ----


def sam_patches(src: torch.Tensor, out: torch.Tensor, *, S: int) -> None:
    """src: uint8 / float32 [D,H,W] (replicated to 3 channels) or float32 [D,3,H,W]; out bf16 [>= D*(S/4)^2, ld >= 147]."""
    dev = _dev_check(src, out)
    if src.dim() == 4:
        assert src.shape[1] == 3 and src.dtype == torch.float32
        mode, (D, _, H, W) = 2, src.shape
    else:
        assert src.dtype in (torch.uint8, torch.float32)
        mode, (D, H, W) = (0 if src.dtype == torch.uint8 else 1), src.shape
    assert out.dtype == torch.bfloat16 and out.shape[0] >= D * (S // 4) ** 2
    call(dev, "cvx_sam_patches", _lib.load().cvx_sam_patches, src.data_ptr(), mode, D, H, W, S, out.data_ptr(), out.stride(0))


def window_attention(q: torch.Tensor, q_col: int, kv: torch.Tensor, k_col: int, v_col: int, out: torch.Tensor, *, slices: int,
                     heads: int, head_dim: int, grid: int, window: int, q_grid: int, q_window: int) -> None:
    """q / kv: bf16 row buffers; the q, k, v blocks start at the given columns (qkv GEMM output: 0, C, 2C)."""
    dev = _dev_check(q, kv, out)
    assert q.dtype == kv.dtype == out.dtype == torch.bfloat16
    assert q.shape[0] >= slices * q_grid * q_grid and kv.shape[0] >= slices * grid * grid and out.shape[0] >= slices * q_grid * q_grid
    assert q_col + heads * head_dim <= q.shape[1] and max(k_col, v_col) + heads * head_dim <= kv.shape[1]
    assert heads * head_dim <= out.shape[1]
    call(dev, "cvx_window_attention_bf16", _lib.load().cvx_window_attention_bf16, q.data_ptr() + 2 * q_col, q.stride(0),
         kv.data_ptr() + 2 * k_col, kv.data_ptr() + 2 * v_col, kv.stride(0), out.data_ptr(), out.stride(0), slices, heads,
         head_dim, grid, window, q_grid, q_window)


def pool2x2(x: torch.Tensor, out: torch.Tensor, *, slices: int, grid: int, C: int) -> None:
    dev = _dev_check(x, out)
    assert x.dtype == out.dtype and x.dtype in (torch.float32, torch.bfloat16)
    assert x.shape[0] >= slices * grid * grid and out.shape[0] >= slices * (grid // 2) ** 2 and C <= min(x.shape[1], out.shape[1])
    call(dev, "cvx_pool2x2", _lib.load().cvx_pool2x2, x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), slices, grid, C,
         int(x.dtype == torch.bfloat16))


def cast_bf16(x: torch.Tensor, out: torch.Tensor, *, rows: int, C: int) -> None:
    dev = _dev_check(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.bfloat16 and min(x.shape[0], out.shape[0]) >= rows
    call(dev, "cvx_cast_bf16", _lib.load().cvx_cast_bf16, x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, C)


def fpn_level_out(lateral: torch.Tensor, coarse, out: torch.Tensor, *, slices: int, C: int, grid: int) -> None:
    dev = _dev_check(lateral, coarse, out)
    assert lateral.dtype == torch.float32 and lateral.shape[1] == C and out.dtype == torch.float16
    assert out.numel() == slices * C * grid * grid and lateral.shape[0] >= slices * grid * grid
    call(dev, "cvx_fpn_level_out", _lib.load().cvx_fpn_level_out, lateral.data_ptr(), _p(coarse), slices, C, grid,
         out.data_ptr())
