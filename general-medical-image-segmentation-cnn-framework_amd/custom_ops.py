"""``torch.ops.mi355seg.*`` -- the hot-path kernels registered as PyTorch custom operators.

north_star asks for the HIP kernels to be "exposed as torch custom ops".  The boundary stays the C-ABI
(include/mi355seg.h, bound with ctypes in ``_lib.py``); this module registers the same entry points with the
dispatcher through ``torch.library.custom_op`` (schemas inferred from the annotations, ``register_fake`` for shape
propagation, ``register_autograd`` wiring forward to the dgrad / wgrad entry points), so the kernels are reachable as
``torch.ops.mi355seg.conv3d(x, w, b, stride, pad)`` etc. from code that speaks the dispatcher (``torch.compile`` graphs,
``opcheck``, export).  Tensors are channel-last [N, D, H, W, C] (fp32 or bf16) exactly as in ``functional``; the model
mirrors keep using the ``torch.autograd.Function`` wrappers of ``functional.py``, which additionally handle strided
channel slices of concat buffers (a custom op's outputs may not alias its inputs).

Reference calls replaced: nn.Conv3d (unet3d.py:80-98), nn.ConvTranspose3d k2 s2 (unet3d.py:29-43), nn.MaxPool3d(2,2)
(unet3d.py:19-25), nn.Upsample(2, nearest) (residual_unet3d.py:19), nn.BCEWithLogitsLoss + argmax + metric
(train.py:204-221)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import functional as F
from ._lib import lib

_p, _stream, _sfx = F._p, F._stream, F._sfx


def _geom(x, w, stride, pad):
    N, D, H, W, Cin = x.shape
    Cout, k = w.shape[0], w.shape[2]
    Do, Ho, Wo = [(e + 2 * pad - k) // stride + 1 for e in (D, H, W)]
    return N, D, H, W, Cin, Cout, k, Do, Ho, Wo


# ------------------------------------------------------------------------------------------------ Conv3d
@torch.library.custom_op("mi355seg::conv3d", mutates_args=())
def conv3d(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, padding: int) -> Tensor:
    xv, ldx = F.cl_view(x, "conv3d input")
    w = F._w32(weight, "conv3d weight")
    N, D, H, W, Cin, Cout, k, Do, Ho, Wo = _geom(xv, w, stride, padding)
    y = torch.empty((N, Do, Ho, Wo, Cout), dtype=xv.dtype, device=xv.device)
    L = lib()
    ws = F.workspace(F._conv_ws(L, xv, N, D, H, W, Cin, Cout, k, stride, padding), xv.device)
    L.call("mi355seg_conv3d_fwd_" + _sfx(xv), _p(xv), ldx, _p(w), _p(bias), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, padding,
           None, None, _p(ws), ws.numel(), _stream())
    return y


@conv3d.register_fake
def _(x, weight, bias, stride, padding):
    N, D, H, W, Cin, Cout, k, Do, Ho, Wo = _geom(x, weight, stride, padding)
    return x.new_empty((N, Do, Ho, Wo, Cout))


@torch.library.custom_op("mi355seg::conv3d_dgrad", mutates_args=())
def conv3d_dgrad(dy: Tensor, weight: Tensor, D: int, H: int, W: int, stride: int, padding: int) -> Tensor:
    dyv, lddy = F.cl_view(dy, "conv3d grad")
    w = F._w32(weight, "conv3d weight")
    N, Cout, Cin, k = dyv.shape[0], w.shape[0], w.shape[1], w.shape[2]
    dx = torch.empty((N, D, H, W, Cin), dtype=dyv.dtype, device=dyv.device)
    L = lib()
    ws = F.workspace(F._conv_ws(L, dyv, N, D, H, W, Cin, Cout, k, stride, padding), dyv.device)
    L.call("mi355seg_conv3d_dgrad_" + _sfx(dyv), _p(dyv), lddy, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, padding,
           _p(ws), ws.numel(), _stream())
    return dx


@conv3d_dgrad.register_fake
def _(dy, weight, D, H, W, stride, padding):
    return dy.new_empty((dy.shape[0], D, H, W, weight.shape[1]))


@torch.library.custom_op("mi355seg::conv3d_wgrad", mutates_args=())
def conv3d_wgrad(dy: Tensor, x: Tensor, kernel_size: int, stride: int, padding: int, with_bias: bool) -> Tuple[Tensor, Tensor]:
    """(dw fp32 (Cout, Cin, k, k, k), db fp32 (Cout) -- zeros(0) when with_bias is False)."""
    dyv, lddy = F.cl_view(dy, "conv3d grad")
    xv, ldx = F.cl_view(F._like(x, dyv), "conv3d input")
    N, D, H, W, Cin = xv.shape
    Cout, k = dyv.shape[-1], kernel_size
    dw = torch.empty((Cout, Cin, k, k, k), dtype=torch.float32, device=xv.device)
    db = torch.empty(Cout if with_bias else 0, dtype=torch.float32, device=xv.device)
    L = lib()
    ws = F.workspace(F._conv_ws(L, xv, N, D, H, W, Cin, Cout, k, stride, padding), xv.device)
    L.call("mi355seg_conv3d_wgrad_" + _sfx(xv), _p(dyv), lddy, _p(xv), ldx, _p(dw), _p(db) if with_bias else None, N, D, H, W, Cin, Cout,
           k, stride, padding, 0, _p(ws), ws.numel(), _stream())
    return dw, db


@conv3d_wgrad.register_fake
def _(dy, x, kernel_size, stride, padding, with_bias):
    Cout, Cin, k = dy.shape[-1], x.shape[-1], kernel_size
    return (torch.empty((Cout, Cin, k, k, k), dtype=torch.float32, device=x.device),
            torch.empty(Cout if with_bias else 0, dtype=torch.float32, device=x.device))


def _conv3d_setup(ctx, inputs, output):
    x, w, b, stride, pad = inputs
    ctx.save_for_backward(x, w)
    ctx.cfg = (stride, pad, b is not None)


def _conv3d_backward(ctx, dy):
    x, w = ctx.saved_tensors
    stride, pad, has_b = ctx.cfg
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = torch.ops.mi355seg.conv3d_dgrad(dy, w, x.shape[1], x.shape[2], x.shape[3], stride, pad)
    if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
        dw, dbv = torch.ops.mi355seg.conv3d_wgrad(dy, x, w.shape[2], stride, pad, has_b)
        db = dbv if has_b else None
    return dx, dw, db, None, None


torch.library.register_autograd("mi355seg::conv3d", _conv3d_backward, setup_context=_conv3d_setup)


# ------------------------------------------------------------------------------------------------ ConvTranspose3d k2 s2
@torch.library.custom_op("mi355seg::conv_transpose3d_k2s2", mutates_args=())
def conv_transpose3d_k2s2(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    return F._ConvT3dK2S2.apply(x.detach(), weight.detach(), None if bias is None else bias.detach())


@conv_transpose3d_k2s2.register_fake
def _(x, weight, bias):
    N, D, H, W, _ = x.shape
    return x.new_empty((N, 2 * D, 2 * H, 2 * W, weight.shape[1]))


@torch.library.custom_op("mi355seg::conv_transpose3d_k2s2_backward", mutates_args=())
def conv_transpose3d_k2s2_backward(dy: Tensor, x: Tensor, weight: Tensor, with_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    xv, ldx = F.cl_view(x, "conv_transpose3d input")
    dyv, lddy = F.cl_view(F._like(dy, xv), "conv_transpose3d grad")
    w = F._w32(weight, "conv_transpose3d weight")
    N, D, H, W, Cin = xv.shape
    Cout = w.shape[1]
    L = lib()
    ws = F.workspace(L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout), xv.device)
    dx = torch.empty((N, D, H, W, Cin), dtype=xv.dtype, device=xv.device)
    dw = torch.empty_like(w)
    db = torch.empty(Cout if with_bias else 0, dtype=torch.float32, device=xv.device)
    L.call("mi355seg_convt3d_k2s2_dgrad_" + _sfx(xv), _p(dyv), lddy, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, _p(ws), ws.numel(), _stream())
    L.call("mi355seg_convt3d_k2s2_wgrad_" + _sfx(xv), _p(dyv), lddy, _p(xv), ldx, _p(dw), _p(db) if with_bias else None, N, D, H, W, Cin, Cout,
           _p(ws), ws.numel(), _stream())
    return dx, dw, db


@conv_transpose3d_k2s2_backward.register_fake
def _(dy, x, weight, with_bias):
    return (torch.empty_like(x), torch.empty_like(weight), torch.empty(weight.shape[1] if with_bias else 0, dtype=torch.float32, device=x.device))


def _convt_setup(ctx, inputs, output):
    x, w, b = inputs
    ctx.save_for_backward(x, w)
    ctx.has_b = b is not None


def _convt_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.mi355seg.conv_transpose3d_k2s2_backward(dy, x, w, ctx.has_b)
    return dx, dw, (db if ctx.has_b else None)


torch.library.register_autograd("mi355seg::conv_transpose3d_k2s2", _convt_backward, setup_context=_convt_setup)


# ------------------------------------------------------------------------------------------------ pool / upsample
@torch.library.custom_op("mi355seg::max_pool3d_2x", mutates_args=())
def max_pool3d_2x(x: Tensor) -> Tuple[Tensor, Tensor]:
    """(pooled, uint8 arg-max codes 0..7 of each 2x2x2 window)."""
    xv, ldx = F.cl_view(x, "max_pool3d input")
    N, D, H, W, C = xv.shape
    y = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=xv.dtype, device=xv.device)
    idx = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=torch.uint8, device=xv.device)
    lib().call("mi355seg_maxpool2_fwd_" + _sfx(xv), _p(xv), ldx, _p(y), C, _p(idx), N, D, H, W, C, _stream())
    return y, idx


@max_pool3d_2x.register_fake
def _(x):
    N, D, H, W, C = x.shape
    return x.new_empty((N, D // 2, H // 2, W // 2, C)), torch.empty((N, D // 2, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)


@torch.library.custom_op("mi355seg::max_pool3d_2x_backward", mutates_args=())
def max_pool3d_2x_backward(dy: Tensor, idx: Tensor, D: int, H: int, W: int) -> Tensor:
    dyv, lddy = F.cl_view(dy, "max_pool3d grad")
    N, C = dyv.shape[0], dyv.shape[-1]
    dx = torch.empty((N, D, H, W, C), dtype=dyv.dtype, device=dyv.device)
    lib().call("mi355seg_maxpool2_bwd_" + _sfx(dyv), _p(dyv), lddy, _p(idx), _p(dx), C, N, D, H, W, C, _stream())
    return dx


@max_pool3d_2x_backward.register_fake
def _(dy, idx, D, H, W):
    return dy.new_empty((dy.shape[0], D, H, W, dy.shape[-1]))


def _pool_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.ext = tuple(inputs[0].shape[1:4])


def _pool_backward(ctx, dy, _didx):
    (idx,) = ctx.saved_tensors
    return torch.ops.mi355seg.max_pool3d_2x_backward(dy, idx, *ctx.ext)


torch.library.register_autograd("mi355seg::max_pool3d_2x", _pool_backward, setup_context=_pool_setup)


@torch.library.custom_op("mi355seg::upsample_nearest_2x", mutates_args=())
def upsample_nearest_2x(x: Tensor) -> Tensor:
    return F._Upsample2.apply(x.detach())


@upsample_nearest_2x.register_fake
def _(x):
    N, D, H, W, C = x.shape
    return x.new_empty((N, 2 * D, 2 * H, 2 * W, C))


@torch.library.custom_op("mi355seg::upsample_nearest_2x_backward", mutates_args=())
def upsample_nearest_2x_backward(dy: Tensor) -> Tensor:
    dyv, lddy = F.cl_view(dy, "upsample grad")
    N, D2, H2, W2, C = dyv.shape
    dx = torch.empty((N, D2 // 2, H2 // 2, W2 // 2, C), dtype=dyv.dtype, device=dyv.device)
    lib().call("mi355seg_upsample2_bwd_" + _sfx(dyv), _p(dyv), lddy, _p(dx), C, N, D2 // 2, H2 // 2, W2 // 2, C, _stream())
    return dx


@upsample_nearest_2x_backward.register_fake
def _(dy):
    N, D2, H2, W2, C = dy.shape
    return dy.new_empty((N, D2 // 2, H2 // 2, W2 // 2, C))


torch.library.register_autograd("mi355seg::upsample_nearest_2x", lambda ctx, dy: torch.ops.mi355seg.upsample_nearest_2x_backward(dy))


# ------------------------------------------------------------------------------------------------ loss / metric tail
@torch.library.custom_op("mi355seg::bce_argmax_dice", mutates_args=())
def bce_argmax_dice(logits: Tensor, target: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """train.py:204,209,221 in one pass: (BCE-with-logits mean loss, argmax mask int64 [N,1,...], Dice counters int64[4])."""
    loss, mask, counts = F._BCEArgmaxDice.apply(logits.detach(), target.detach())
    return loss, mask, counts


@bce_argmax_dice.register_fake
def _(logits, target):
    return (logits.new_empty(()), torch.empty((logits.shape[0], 1) + tuple(logits.shape[2:]), dtype=torch.int64, device=logits.device),
            torch.empty(4, dtype=torch.int64, device=logits.device))


@torch.library.custom_op("mi355seg::bce_with_logits_backward", mutates_args=())
def bce_with_logits_backward(logits: Tensor, target: Tensor, grad: Tensor) -> Tensor:
    lg, tg = logits.contiguous(), target.contiguous().to(torch.float32)
    d = torch.empty_like(lg)
    lib().call("mi355seg_bce_logits_bwd_f32", _p(lg), _p(tg), _p(grad.contiguous().to(torch.float32)), lg.numel(), _p(d), _stream())
    return d


@bce_with_logits_backward.register_fake
def _(logits, target, grad):
    return torch.empty_like(logits)


def _tail_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _tail_backward(ctx, g, _gm, _gc):
    logits, target = ctx.saved_tensors
    return torch.ops.mi355seg.bce_with_logits_backward(logits, target, g), None


torch.library.register_autograd("mi355seg::bce_argmax_dice", _tail_backward, setup_context=_tail_setup)


@torch.library.custom_op("mi355seg::dice_counts", mutates_args=())
def dice_counts(gt: Tensor, pred: Tensor) -> Tensor:
    """utils/metric.py:26-66 on the device: int64[4] = (sum gt, sum pred, nnz(gt & pred), nnz(gt | pred))."""
    return F.dice_counts(gt, pred)


@dice_counts.register_fake
def _(gt, pred):
    return torch.empty(4, dtype=torch.int64, device=gt.device)


# ------------------------------------------------------------------------------------------------ layout (module boundary)
@torch.library.custom_op("mi355seg::to_channels_last", mutates_args=())
def to_channels_last(x: Tensor) -> Tensor:
    """fp32 NCDHW -> channel-last [N, D, H, W, C] in the current activation storage type (fp32, or bf16 under autocast)."""
    return F.to_channels_last(x.detach()).clone()


@to_channels_last.register_fake
def _(x):
    N, C = x.shape[0], x.shape[1]
    return torch.empty((N,) + tuple(x.shape[2:]) + (C,), dtype=F.compute_dtype(), device=x.device)


@torch.library.custom_op("mi355seg::to_channels_first", mutates_args=())
def to_channels_first(x: Tensor) -> Tensor:
    """channel-last (fp32 | bf16) -> fp32 NCDHW."""
    return F.to_channels_first(x.detach()).clone()


@to_channels_first.register_fake
def _(x):
    return torch.empty((x.shape[0], x.shape[-1]) + tuple(x.shape[1:-1]), dtype=torch.float32, device=x.device)


def _cl_setup(ctx, inputs, output):
    ctx.in_dtype = inputs[0].dtype


torch.library.register_autograd("mi355seg::to_channels_last", lambda ctx, g: torch.ops.mi355seg.to_channels_first(g).to(ctx.in_dtype), setup_context=_cl_setup)
torch.library.register_autograd("mi355seg::to_channels_first", lambda ctx, g: F.cast(torch.ops.mi355seg.to_channels_last(g.contiguous()), ctx.in_dtype),
                                setup_context=_cl_setup)


# ------------------------------------------------------------------------------------------------ BatchNorm3d / InstanceNorm3d + activation
@torch.library.custom_op("mi355seg::norm_stats", mutates_args=("running_mean", "running_var"))
def norm_stats(x: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], momentum: float, eps: float,
               instance: bool) -> Tuple[Tensor, Tensor]:
    """(mean, 1 / sqrt(var + eps)) per channel (per sample and channel for instance norm); BatchNorm running statistics are
    updated in place exactly as nn.BatchNorm3d does in training mode (unet3d.py:84-96)."""
    xv, ldx = F.cl_view(x, "norm input")
    N, D, H, W, C = xv.shape
    groups = N if instance else 1
    rows = D * H * W * (1 if instance else N)
    L = lib()
    ws = F.workspace(L.query("mi355seg_norm_ws_bytes", rows, groups, C), xv.device)
    mean = torch.empty(groups * C, dtype=torch.float32, device=xv.device)
    rstd = torch.empty(groups * C, dtype=torch.float32, device=xv.device)
    L.call("mi355seg_norm_stats_" + _sfx(xv), _p(xv), ldx, rows, groups, C, eps, _p(mean), _p(rstd), _p(running_mean), _p(running_var), momentum,
           _p(ws), ws.numel(), _stream())
    return mean, rstd


@norm_stats.register_fake
def _(x, running_mean, running_var, momentum, eps, instance):
    n = (x.shape[0] if instance else 1) * x.shape[-1]
    return torch.empty(n, dtype=torch.float32, device=x.device), torch.empty(n, dtype=torch.float32, device=x.device)


@torch.library.custom_op("mi355seg::norm_apply_act", mutates_args=())
def norm_apply_act(x: Tensor, mean: Tensor, rstd: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], residual: Optional[Tensor],
                   act: int, slope: float, instance: bool) -> Tensor:
    """act((x - mean) * rstd * gamma + beta [+ residual]) in one pass."""
    xv, ldx = F.cl_view(x, "norm input")
    N, D, H, W, C = xv.shape
    groups = N if instance else 1
    rows = D * H * W * (1 if instance else N)
    res, ldres = (None, 0) if residual is None else F.cl_view(residual, "norm residual")
    y = torch.empty((N, D, H, W, C), dtype=xv.dtype, device=xv.device)
    lib().call("mi355seg_norm_act_fwd_" + _sfx(xv), _p(xv), ldx, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), ldres, _p(y), C, rows, groups, C,
               act, slope, _stream())
    return y


@norm_apply_act.register_fake
def _(x, mean, rstd, gamma, beta, residual, act, slope, instance):
    return torch.empty_like(x)


@torch.library.custom_op("mi355seg::norm_act_backward", mutates_args=())
def norm_act_backward(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], residual: Optional[Tensor],
                      act: int, slope: float, instance: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Backward of norm_stats + norm_apply_act with batch / instance statistics: (dx, dgamma, dbeta, dresidual); empty tensors
    where the forward had no gamma / residual."""
    xv, ldx = F.cl_view(x, "norm input")
    dyv, lddy = F.cl_view(F._like(dy, xv), "norm grad")
    N, D, H, W, C = xv.shape
    groups = N if instance else 1
    rows = D * H * W * (1 if instance else N)
    res, ldres = (None, 0) if residual is None else F.cl_view(residual, "norm residual")
    L = lib()
    dev = xv.device
    ws = F.workspace(L.query("mi355seg_norm_ws_bytes", rows, groups, C), dev)
    dx = torch.empty(xv.shape, dtype=xv.dtype, device=dev)
    dgamma = torch.empty(C if gamma is not None else 0, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C if gamma is not None else 0, dtype=torch.float32, device=dev)
    dres = torch.empty(xv.shape if res is not None else (0,), dtype=xv.dtype, device=dev)
    L.call("mi355seg_norm_act_bwd_" + _sfx(xv), _p(dyv), lddy, _p(xv), ldx, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), ldres,
           _p(dx), C, _p(dgamma) if gamma is not None else None, _p(dbeta) if gamma is not None else None, _p(dres) if res is not None else None, C,
           rows, groups, C, act, slope, _p(ws), ws.numel(), _stream())
    return dx, dgamma, dbeta, dres


@norm_act_backward.register_fake
def _(dy, x, mean, rstd, gamma, beta, residual, act, slope, instance):
    C = x.shape[-1]
    return (torch.empty_like(x), torch.empty(C if gamma is not None else 0, dtype=torch.float32, device=x.device),
            torch.empty(C if gamma is not None else 0, dtype=torch.float32, device=x.device),
            torch.empty_like(x) if residual is not None else torch.empty(0, dtype=x.dtype, device=x.device))


def _norm_apply_setup(ctx, inputs, output):
    x, mean, rstd, gamma, beta, residual, act, slope, instance = inputs
    ctx.save_for_backward(x, mean, rstd, gamma, beta, residual)
    ctx.cfg = (act, slope, instance)


def _norm_apply_backward(ctx, dy):
    """The statistics are functions of x: the backward kernel differentiates through them (batch / instance statistics), so the
    gradient reaches x here and norm_stats itself is registered as non-differentiable."""
    x, mean, rstd, gamma, beta, residual = ctx.saved_tensors
    act, slope, instance = ctx.cfg
    dx, dgamma, dbeta, dres = torch.ops.mi355seg.norm_act_backward(dy, x, mean, rstd, gamma, beta, residual, act, slope, instance)
    return dx, None, None, (dgamma if gamma is not None else None), (dbeta if beta is not None else None), (dres if residual is not None else None), None, None, None


torch.library.register_autograd("mi355seg::norm_apply_act", _norm_apply_backward, setup_context=_norm_apply_setup)


def batch_norm_act(x, gamma, beta, running_mean, running_var, momentum, eps, act, slope, residual=None):
    """Training-mode act(BatchNorm3d(x) [+ residual]) through the dispatcher: two ops, one fused backward."""
    mean, rstd = torch.ops.mi355seg.norm_stats(x.detach(), running_mean, running_var, momentum, eps, False)
    return torch.ops.mi355seg.norm_apply_act(x, mean, rstd, gamma, beta, residual, act, slope, False)


# ------------------------------------------------------------------------------------------------ conv + BatchNorm + activation (one node)
@torch.library.custom_op("mi355seg::conv_bn_act", mutates_args=())
def conv_bn_act(x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor,
                stride: int, padding: int, momentum: float, eps: float, act: int, slope: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Training-mode act(BatchNorm3d(conv3d(x))) (unet3d.py:80-98): the batch statistics come out of the convolution's epilogue.
    Functional (an operator with an autograd formula may not mutate its inputs): returns (activation, raw convolution output,
    mean, rstd, updated running_mean, updated running_var); ``conv_bn_act_train`` below writes the last two back into the
    module's buffers, as aten's native_batch_norm callers do."""
    xv, ldx = F.cl_view(x, "conv3d input")
    w = F._w32(weight, "conv3d weight")
    N, D, H, W, Cin, Cout, k, Do, Ho, Wo = _geom(xv, w, stride, padding)
    dev = xv.device
    L = lib()
    rows = N * Do * Ho * Wo
    ws = F.workspace(max(F._conv_ws(L, xv, N, D, H, W, Cin, Cout, k, stride, padding), L.query("mi355seg_norm_ws_bytes", rows, 1, Cout)), dev)
    y = torch.empty((N, Do, Ho, Wo, Cout), dtype=xv.dtype, device=dev)
    sums = torch.empty(2 * Cout, dtype=torch.float64, device=dev)
    L.call("mi355seg_conv3d_fwd_" + _sfx(xv), _p(xv), ldx, _p(w), _p(bias), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, padding,
           sums.data_ptr(), sums.data_ptr() + 8 * Cout, _p(ws), ws.numel(), _stream())
    mean = torch.empty(Cout, dtype=torch.float32, device=dev)
    rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
    new_rm, new_rv = running_mean.detach().clone(), running_var.detach().clone()
    L.call("mi355seg_norm_stats_from_sums_f32", sums.data_ptr(), sums.data_ptr() + 8 * Cout, rows, Cout, eps, _p(mean), _p(rstd),
           _p(new_rm), _p(new_rv), momentum, _stream())
    a = torch.empty_like(y)
    L.call("mi355seg_norm_act_fwd_" + _sfx(xv), _p(y), Cout, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0, _p(a), Cout, rows, 1, Cout, act, slope, _stream())
    return a, y, mean, rstd, new_rm, new_rv


@conv_bn_act.register_fake
def _(x, weight, bias, gamma, beta, running_mean, running_var, stride, padding, momentum, eps, act, slope):
    N, D, H, W, Cin, Cout, k, Do, Ho, Wo = _geom(x, weight, stride, padding)
    f = lambda: torch.empty(Cout, dtype=torch.float32, device=x.device)
    return x.new_empty((N, Do, Ho, Wo, Cout)), x.new_empty((N, Do, Ho, Wo, Cout)), f(), f(), f(), f()


@torch.library.custom_op("mi355seg::conv_bn_act_backward", mutates_args=())
def conv_bn_act_backward(da: Tensor, x: Tensor, weight: Tensor, y: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, beta: Tensor,
                         stride: int, padding: int, act: int, slope: float, with_bias: bool, need_dx: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(dx, dw, dbias, dgamma, dbeta): BatchNorm + activation backward with the conv-bias column sums in the same pass, then the
    input and weight gradients of the convolution."""
    xv, ldx = F.cl_view(x, "conv3d input")
    dav, ldda = F.cl_view(F._like(da, xv), "conv+norm grad")
    w = F._w32(weight, "conv3d weight")
    N, D, H, W, Cin, Cout, k, Do, Ho, Wo = _geom(xv, w, stride, padding)
    dev = xv.device
    L = lib()
    rows = N * Do * Ho * Wo
    ws = F.workspace(max(F._conv_ws(L, xv, N, D, H, W, Cin, Cout, k, stride, padding), L.query("mi355seg_norm_ws_bytes", rows, 1, Cout)), dev)
    dy = torch.empty_like(y)
    dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
    dbeta = torch.empty(Cout, dtype=torch.float32, device=dev)
    db = torch.empty(Cout if with_bias else 0, dtype=torch.float32, device=dev)
    L.call("mi355seg_norm_act_bwd_colsum_" + _sfx(xv), _p(dav), ldda, _p(y), Cout, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
           _p(dy), Cout, _p(dgamma), _p(dbeta), None, 0, _p(db) if with_bias else None, rows, 1, Cout, act, slope, _p(ws), ws.numel(), _stream())
    dx = torch.empty(xv.shape if need_dx else (0,), dtype=xv.dtype, device=dev)
    if need_dx:
        L.call("mi355seg_conv3d_dgrad_" + _sfx(xv), _p(dy), Cout, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, padding, _p(ws), ws.numel(), _stream())
    dw = torch.empty_like(w)
    L.call("mi355seg_conv3d_wgrad_" + _sfx(xv), _p(dy), Cout, _p(xv), ldx, _p(dw), None, N, D, H, W, Cin, Cout, k, stride, padding, 0, _p(ws), ws.numel(), _stream())
    return dx, dw, db, dgamma, dbeta


@conv_bn_act_backward.register_fake
def _(da, x, weight, y, mean, rstd, gamma, beta, stride, padding, act, slope, with_bias, need_dx):
    C = weight.shape[0]
    f = lambda n: torch.empty(n, dtype=torch.float32, device=x.device)
    return (torch.empty_like(x) if need_dx else torch.empty(0, dtype=x.dtype, device=x.device), torch.empty_like(weight), f(C if with_bias else 0), f(C), f(C))


def _cba_setup(ctx, inputs, output):
    x, w, b, gamma, beta, rm, rv, stride, pad, momentum, eps, act, slope = inputs
    _a, y, mean, rstd, _rm, _rv = output
    ctx.save_for_backward(x, w, y, mean, rstd, gamma, beta)
    ctx.cfg = (stride, pad, act, slope, b is not None)


def _cba_backward(ctx, da, _dy, _dm, _dr, _drm, _drv):
    x, w, y, mean, rstd, gamma, beta = ctx.saved_tensors
    stride, pad, act, slope, has_b = ctx.cfg
    dx, dw, db, dgamma, dbeta = torch.ops.mi355seg.conv_bn_act_backward(da, x, w, y, mean, rstd, gamma, beta, stride, pad, act, slope, has_b, ctx.needs_input_grad[0])
    return (dx if ctx.needs_input_grad[0] else None), dw, (db if has_b else None), dgamma, dbeta, None, None, None, None, None, None, None, None


torch.library.register_autograd("mi355seg::conv_bn_act", _cba_backward, setup_context=_cba_setup)


def conv_bn_act_train(x, conv, bn, act, slope=0.0):
    """One conv -> BatchNorm3d -> activation unit of the reference's blocks (unet3d.py:80-98) through the dispatcher, training mode."""
    stride = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
    pad = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
    a, _y, _m, _r, rm, rv = torch.ops.mi355seg.conv_bn_act(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                            int(stride), int(pad), float(bn.momentum), float(bn.eps), int(act), float(slope))
    with torch.no_grad():
        bn.running_mean.copy_(rm)
        bn.running_var.copy_(rv)
        bn.num_batches_tracked.add_(1)
    return a


# ------------------------------------------------------------------------------------------------ activation, channel concat
@torch.library.custom_op("mi355seg::activation", mutates_args=())
def activation(x: Tensor, act: int, slope: float) -> Tensor:
    return F._Act.apply(x.detach(), None, act, slope)


@activation.register_fake
def _(x, act, slope):
    return torch.empty_like(x)


@torch.library.custom_op("mi355seg::activation_backward", mutates_args=())
def activation_backward(dy: Tensor, x: Tensor, act: int, slope: float) -> Tensor:
    xv, ldx = F.cl_view(x, "activation input")
    dyv, lddy = F.cl_view(F._like(dy, xv), "activation grad")
    N, D, H, W, C = xv.shape
    dx = torch.empty(xv.shape, dtype=xv.dtype, device=xv.device)
    lib().call("mi355seg_act_bwd_" + _sfx(xv), _p(dyv), lddy, _p(xv), ldx, None, 0, _p(dx), C, N * D * H * W, C, act, slope, _stream())
    return dx


@activation_backward.register_fake
def _(dy, x, act, slope):
    return torch.empty_like(x)


def _act_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.cfg = (inputs[1], inputs[2])


torch.library.register_autograd("mi355seg::activation", lambda ctx, dy: (torch.ops.mi355seg.activation_backward(dy, ctx.saved_tensors[0], *ctx.cfg), None, None),
                                setup_context=_act_setup)


@torch.library.custom_op("mi355seg::cat_channels", mutates_args=())
def cat_channels(a: Tensor, b: Tensor) -> Tensor:
    """torch.cat((a, b), dim=channel) for channel-last tensors (unet3d.py:60-70): two strided row copies."""
    return F._CatChannels.apply(a.detach(), b.detach())


@cat_channels.register_fake
def _(a, b):
    return a.new_empty(tuple(a.shape[:-1]) + (a.shape[-1] + b.shape[-1],))


def _cat_setup(ctx, inputs, output):
    ctx.ca = inputs[0].shape[-1]


torch.library.register_autograd("mi355seg::cat_channels", lambda ctx, g: (g[..., :ctx.ca].contiguous(), g[..., ctx.ca:].contiguous()), setup_context=_cat_setup)


OPS = ["conv3d", "conv3d_dgrad", "conv3d_wgrad", "conv_transpose3d_k2s2", "conv_transpose3d_k2s2_backward", "max_pool3d_2x",
       "max_pool3d_2x_backward", "upsample_nearest_2x", "upsample_nearest_2x_backward", "bce_argmax_dice", "bce_with_logits_backward",
       "dice_counts", "to_channels_last", "to_channels_first", "norm_stats", "norm_apply_act", "norm_act_backward", "conv_bn_act",
       "conv_bn_act_backward", "activation", "activation_backward", "cat_channels"]
