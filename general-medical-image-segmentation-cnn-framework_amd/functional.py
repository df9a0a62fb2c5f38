"""torch.autograd wrappers over the C-ABI (libmi355seg.so).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and
the autograd graph.  Every numeric kernel is a hand-written gfx950 kernel behind
include/mi355seg.h; nothing in this file computes with torch ops on the hot path.

Activations inside the package are channel-last: a 5-D fp32 tensor of logical shape
[N, D, H, W, C] whose last stride is 1 and whose voxel pitch (stride of W) may exceed C
(a channel slice of a wider concat buffer).
"""
import os
import threading

import torch
from torch.autograd import Function

from ._lib import lib, Mi355SegError

ACT_NONE, ACT_RELU, ACT_ELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3, 4

_WS = {}

# ---- activation storage type.  fp32 by default; inside ``autocast(torch.bfloat16)`` the models' entry layout conversion
# (to_channels_last) produces bf16 activations and every op below follows its input's dtype: what
# ``Accelerator(mixed_precision="bf16")`` / torch.autocast does for the reference (parameters, their gradients, norm
# statistics and the loss stay fp32; conv products are bf16 MFMAs with fp32 accumulation).
class _AutocastState(threading.local):
    """Per-thread stack (a worker thread's forward must not inherit or leak another thread's bf16 mode)."""

    def __init__(self):
        self.stack = [torch.float32]


_AUTOCAST = _AutocastState()
_SFX = {torch.float32: "f32", torch.bfloat16: "bf16"}


class autocast:
    """``with mi355seg.functional.autocast(torch.bfloat16): pred = model(x)`` -- activations in bf16 (NDHWC in HBM)."""

    def __init__(self, dtype=torch.bfloat16, enabled=True):
        if dtype not in _SFX:
            raise Mi355SegError(f"autocast: dtype must be torch.float32 or torch.bfloat16, got {dtype}")
        self.dtype = dtype if enabled else torch.float32

    def __enter__(self):
        _AUTOCAST.stack.append(self.dtype)
        return self

    def __exit__(self, *exc):
        _AUTOCAST.stack.pop()
        return False


class _CounterBatch(threading.local):
    active = False
    seen = None


_COUNTERS = _CounterBatch()


class counters_batched:
    """While active, the fused BatchNorm entry points leave ``num_batches_tracked`` alone: engine.train_step has advanced the
    counters of ``modules`` (every training-mode BatchNorm of the model) with ONE multi-tensor launch (18-24 one-element torch
    kernels per step otherwise).  Calls are tallied, so a module used twice in one forward, or not at all, still ends with exactly
    the count nn.BatchNorm3d would hold."""

    def __init__(self, modules=()):
        self.modules = list(modules)

    def __enter__(self):
        self.prev = (_COUNTERS.active, _COUNTERS.seen)
        _COUNTERS.active, _COUNTERS.seen = True, {}

    def __exit__(self, *exc):
        seen = _COUNTERS.seen
        _COUNTERS.active, _COUNTERS.seen = self.prev
        for m in self.modules:
            k = seen.pop(id(m), (m, 0))[1]
            if k != 1:
                m.num_batches_tracked.add_(k - 1)
        for m, k in seen.values():                      # a BatchNorm outside the advanced set
            m.num_batches_tracked.add_(k)


_DEFERRED_WAITS = []


def defer_wait(work):
    """A collective launched ahead of the forward (distributed.broadcast_buffers(async_op=True)): waited for by the first norm layer."""
    _DEFERRED_WAITS.append(work)


def flush_deferred_waits():
    while _DEFERRED_WAITS:
        _DEFERRED_WAITS.pop().wait()


def bump_counter(bn):
    """nn.BatchNorm3d's per-forward ``num_batches_tracked += 1`` (training mode).  Also the point where a forward first touches a
    module buffer: a buffer broadcast launched ahead of the step is waited for here."""
    if _DEFERRED_WAITS:
        flush_deferred_waits()
    if bn.num_batches_tracked is None:
        return
    if _COUNTERS.active:
        _COUNTERS.seen[id(bn)] = (bn, _COUNTERS.seen.get(id(bn), (bn, 0))[1] + 1)
    else:
        bn.num_batches_tracked.add_(1)


def compute_dtype():
    return _AUTOCAST.stack[-1]


def _sfx(t):
    return _SFX[t.dtype]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes, device):
    """Per-device scratch buffer, grown on demand.  Stream-ordered reuse: all kernels of
    one process run on the current stream, so consecutive ops may share it."""
    buf = _WS.get(device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[device] = buf
    return buf


# ----------------------------------------------------------------------------- a step's weight packings in one launch (csrc/prepack.hip)
_PREPACK_OFF = bool(os.environ.get("MI355SEG_NO_PREPACK"))
_PREPACK_ACTIVE = [False]       # a recorded plan is being replayed for the running step (prepacked_weights)


def conv_math_signature():
    """The process-wide kernel policies that decide which packings a step's convolutions read (part of a prepack plan's key)."""
    L = lib()
    return (L.query("mi355seg_get_conv_math"), L.query("mi355seg_get_x3_shape"), L.query("mi355seg_get_b16_tiles"))


def _prepack_free(plans):
    try:
        L = lib()
        for rec in plans.values():
            L.call("mi355seg_prepack_free", rec["plan"])
    except Exception:            # interpreter shutdown
        pass


class prepacked_weights:
    """Context manager around ONE training iteration of ``model`` (forward AND backward inside it; engine.train_step).  The first
    iteration at a signature (``key`` = whatever fixes the kernels' plans: dtype, conv math, input shape; plus the parameters' storage
    addresses) runs as always while the library records every weight packing its convolutions launch; each later one starts with
    ALL of those packings formed by one launch into an arena this object owns (include/mi355seg.h, mi355seg_prepack_run) and its
    convolutions read them there -- 44 small launches off the cfg-2 step's stream, ~50 off UNETR's.  Bit-identical results.
    MI355SEG_NO_PREPACK=1 turns it off."""

    def __init__(self, model, key):
        self.model, self.key, self.mode = model, key, None

    @property
    def replaying(self):
        return self.mode == "run" and _PREPACK_ACTIVE[0]

    def __enter__(self):
        if _PREPACK_OFF:
            return self
        params = [p for p in self.model.parameters()]
        if not params or not params[0].is_cuda:
            return self
        sig = (self.key, tuple(p.data_ptr() for p in params))
        plans = self.model.__dict__.get("_seg_prepack")
        if plans is None:
            import weakref
            plans = self.model.__dict__["_seg_prepack"] = {}
            weakref.finalize(self.model, _prepack_free, plans)
        rec = plans.get(sig)
        L = lib()
        if rec is None:
            if torch.cuda.is_current_stream_capturing():     # (a capture without eager warm-up: nothing recorded, the packings stay in place)
                return self
            if len(plans) >= 4:                              # shapes keep changing: forget the oldest
                old = next(iter(plans))
                L.call("mi355seg_prepack_free", plans.pop(old)["plan"])
            L.call("mi355seg_prepack_record_begin")
            self.mode, self.sig, self.plans = "record", sig, plans
        elif rec["plan"]:
            L.call("mi355seg_prepack_run", rec["plan"], _p(rec["arena"]), rec["arena"].numel(), _stream())
            self.mode = "run"
            _PREPACK_ACTIVE[0] = L.query("mi355seg_prepack_active") != 0
        return self

    def __exit__(self, *exc):
        L = lib()
        if self.mode == "record":
            import ctypes
            plan, nbytes = ctypes.c_int(0), ctypes.c_size_t(0)
            L.call("mi355seg_prepack_record_end", ctypes.byref(plan), ctypes.byref(nbytes))
            if exc[0] is not None:
                if plan.value:
                    L.call("mi355seg_prepack_free", plan.value)
            else:
                dev = next(self.model.parameters()).device
                arena = torch.empty(max(int(nbytes.value), 256), dtype=torch.uint8, device=dev) if plan.value else None
                self.plans[self.sig] = {"plan": plan.value, "arena": arena}
        elif self.mode == "run":
            _PREPACK_ACTIVE[0] = False
            L.call("mi355seg_prepack_done", _stream())
        self.mode = None
        return False


def _require_cuda(t, what, allow_bf16=False):
    if not t.is_cuda:
        raise Mi355SegError(f"{what}: expected a tensor on an MI355X (cuda/HIP) device, got {t.device}; "
                            "there is no CPU fallback in this package")
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise Mi355SegError(f"{what}: expected float32{' or bfloat16' if allow_bf16 else ''}, got {t.dtype}")


def cl_view(t, what="tensor", allow_bf16=True):
    """Return (tensor, ld) with tensor laid out NDHWC (last stride 1, dense in N,D,H,W with
    voxel pitch ld >= C).  Copies only if the given tensor does not already satisfy that."""
    _require_cuda(t, what, allow_bf16=allow_bf16)
    if t.dim() != 5:
        raise Mi355SegError(f"{what}: expected 5-D [N,D,H,W,C], got shape {tuple(t.shape)}")
    N, D, H, W, C = t.shape
    s = t.stride()
    ld = s[3]
    ok = (C == 1 or s[4] == 1) and ld >= C and W > 1
    ok = ok and (H == 1 or s[2] == W * ld) and (D == 1 or s[1] == H * W * ld) and (N == 1 or s[0] == D * H * W * ld)
    ok = ok and t.data_ptr() % 16 == 0          # the float4 paths assume a 16-byte aligned base
    if not ok:
        t = t.contiguous()
        ld = C
    return t, ld


def _p(t):
    return None if t is None else t.data_ptr()


def _like(g, x):
    """An incoming gradient in the storage type of the tensor it belongs to (autograd may hand over fp32 zeros)."""
    return g if g.dtype == x.dtype else g.to(x.dtype)


def _conv_ws(L, x, N, D, H, W, Cin, Cout, k, stride, pad):
    name = "mi355seg_conv3d_ws_bytes_bf16" if x.dtype == torch.bfloat16 else "mi355seg_conv3d_ws_bytes"
    return L.query(name, N, D, H, W, Cin, Cout, k, stride, pad)


def _w32(w, what):
    if w.dtype != torch.float32:
        raise Mi355SegError(f"{what}: parameters are fp32 masters, got {w.dtype}")
    return w.contiguous()


def to_channels_last(x, dtype=None):
    """fp32 [N,C,D,H,W] -> [N,D,H,W,C] in ``dtype`` (default: the autocast compute dtype; a free view when C == 1 and
    the result stays fp32).  This is where a model's activations take their storage type."""
    return _ToNDHWC.apply(x, dtype or compute_dtype())


def to_channels_first(x):
    """[N,D,H,W,C] (fp32 or bf16) -> fp32 [N,C,D,H,W] contiguous (free view when C == 1 and fp32): the model boundary
    always hands fp32 NCDHW tensors to the caller (logits go to an fp32 loss, as under torch autocast)."""
    return _ToNCDHW.apply(x)


class _ToNDHWC(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        _require_cuda(x, "to_channels_last")
        if x.dtype != torch.float32:
            raise Mi355SegError(f"to_channels_last: the NCDHW side is fp32, got {x.dtype}")
        N, C, D, H, W = x.shape
        x = x.contiguous()
        if dtype == torch.float32:
            if C == 1:
                return x.view(N, D, H, W, 1)
            y = torch.empty((N, D, H, W, C), dtype=x.dtype, device=x.device)
            lib().call("mi355seg_ncdhw_to_ndhwc_f32", _p(x), _p(y), C, N, C, D * H * W, _stream())
            return y
        y = torch.empty((N, D, H, W, C), dtype=torch.bfloat16, device=x.device)
        lib().call("mi355seg_ncdhw_f32_to_ndhwc_bf16", _p(x), _p(y), C, N, C, D * H * W, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return _ToNCDHW.apply(g), None


class _ToNCDHW(Function):
    @staticmethod
    def forward(ctx, x):
        x, ld = cl_view(x, "to_channels_first")
        N, D, H, W, C = x.shape
        ctx.src_dtype = x.dtype
        if x.dtype == torch.float32:
            if C == 1 and ld == 1:
                return x.view(N, 1, D, H, W)
            y = torch.empty((N, C, D, H, W), dtype=x.dtype, device=x.device)
            lib().call("mi355seg_ndhwc_to_ncdhw_f32", _p(x), ld, _p(y), N, C, D * H * W, _stream())
            return y
        y = torch.empty((N, C, D, H, W), dtype=torch.float32, device=x.device)
        lib().call("mi355seg_ndhwc_bf16_to_ncdhw_f32", _p(x), ld, _p(y), N, C, D * H * W, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return _ToNDHWC.apply(g, ctx.src_dtype)


class _Cast(Function):
    """Storage-type change of a channel-last tensor (an autocast boundary inside a model, e.g. UNETR's fp32 token
    tensors entering the bf16 convolutional decoder); the gradient takes the opposite cast."""

    @staticmethod
    def forward(ctx, x, dtype):
        _require_cuda(x, "cast input", allow_bf16=True)
        ctx.src_dtype = x.dtype
        if x.dtype == dtype:
            return x.view_as(x)
        xc = x.contiguous()
        C = xc.shape[-1]
        rows = xc.numel() // C
        y = torch.empty(xc.shape, dtype=dtype, device=x.device)
        name = "mi355seg_cast_f32_to_bf16" if dtype == torch.bfloat16 else "mi355seg_cast_bf16_to_f32"
        lib().call(name, _p(xc), C, _p(y), C, rows, C, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return _Cast.apply(g, ctx.src_dtype), None


def cast(x, dtype=None):
    """x in the storage type ``dtype`` (default: the autocast compute dtype)."""
    return _Cast.apply(x, dtype or compute_dtype())


# ----------------------------------------------------------------------------- f16x3 operand maxima
# The default conv math on fp32 tensors ("f16x3", include/mi355seg.h) splits every operand into two fp16 parts under a per-tensor
# power-of-two scale and needs max |.| of both operands as device scalars.  The plain entry points measure them with a pass of
# their own; here they ride along instead: the norm + activation kernels emit the maximum of what they write, a max-pool keeps
# its input's maximum, and a tensor carries its scalar as the attribute ``_seg_amax`` = (1-element fp32 tensor, tensor version)
# to whatever convolution reads it (a missing / stale attribute only means that the library measures).  Weight maxima are taken
# in the forward (one small launch per layer) and reused by that layer's backward.
_AMAX_POOL = {}


def _amax_slot(device):
    """A zeroed 1-element fp32 device tensor (the kernels max-combine into it), cut from a chunk that is zero-filled once."""
    st = _AMAX_POOL.get(device)
    if st is None or st[1] >= st[0].numel():
        st = [torch.zeros(2048, dtype=torch.float32, device=device), 0]
        _AMAX_POOL[device] = st
    slot = st[0][st[1]:st[1] + 1]
    st[1] += 1
    return slot


def amax_pool_reset():
    """Forget the current slot chunks (engine.GraphedTrainStep: a captured step must zero-fill the chunk it draws from inside the capture)."""
    _AMAX_POOL.clear()


def _takes_amax(x):
    return x.dtype == torch.float32 and lib().query("mi355seg_conv_math_takes_amax") != 0


_AMAX_USE = {}


def _amax_use(x, N, D, H, W, Cin, Cout, k, stride, pad):
    """Which passes of this layer read operand maxima under the selected conv math (bit 0 forward, 1 input gradient, 2 weight gradient)."""
    if x.dtype != torch.float32:
        return 0
    L = lib()
    key = (L.query("mi355seg_get_conv_math"), L.query("mi355seg_get_x3_shape"), N, D, H, W, Cin, Cout, k, stride, pad)
    use = _AMAX_USE.get(key)
    if use is None:
        use = _AMAX_USE[key] = L.query("mi355seg_conv3d_amax_use_f32", N, D, H, W, Cin, Cout, k, stride, pad)
    return use


_CHECK_AMAX = bool(os.environ.get("MI355SEG_CHECK_AMAX"))


def check_amax(enable=True):
    """Debug mode (also MI355SEG_CHECK_AMAX=1): every operand maximum a convolution is handed is compared with the tensor's true
    maximum (one synchronising reduction per use).  INVARIANT the carried scalars rest on: ``_seg_amax`` is keyed by the tensor's
    version counter, and the library's kernels write through ``data_ptr()`` without bumping it -- so whatever writes an fp32
    activation in place through a raw pointer AFTER its maximum was recorded must delete or refresh the attribute.  A too-SMALL
    maximum makes the f16x3 high part overflow fp16 silently; a too-large one only costs headroom."""
    global _CHECK_AMAX
    _CHECK_AMAX = bool(enable)


def _assert_amax(value, slot, what):
    """value: the operand tensor (any layout; only its values matter) or its true maximum as a float."""
    if not _CHECK_AMAX or slot is None:
        return
    true = float(value.detach().abs().max()) if torch.is_tensor(value) else float(value)
    have = float(slot.reshape(-1)[0])
    if not (have >= true * (1.0 - 1e-6)):
        raise Mi355SegError(f"operand maximum handed to {what} is too small: carried {have!r}, tensor has {true!r}")


def _get_amax(t):
    rec = getattr(t, "_seg_amax", None)
    if rec is None or rec[1] != t._version or rec[0].device != t.device:
        return None
    _assert_amax(t, rec[0], "a convolution (the tensor's _seg_amax attribute)")
    return rec[0]


def _set_amax(t, slot):
    if slot is not None:
        t._seg_amax = (slot, t._version)
    return t


def _measure_amax(t, ld, rows, C, slot=None):
    """max |t| over rows x C at pitch ld into a fresh slot (or max-combined into ``slot``)."""
    if slot is None:
        slot = _amax_slot(t.device)
    lib().call("mi355seg_amax_f32", _p(t), ld, rows, C, _p(slot), _stream())
    return slot


_WEIGHT_AMAX = {}


def prefetch_weight_amax(model):
    """max |w| of every 3x3x3 convolution weight of ``model`` by ONE multi-tensor launch (torch._foreach_norm, inf-norm) instead of
    one small launch per layer and forward (19 per U-Net step).  engine.train_step calls it ahead of the forward and drops the
    table (clear_weight_amax) when the step ends: the entries never outlive the weights they were measured on.  Keyed by the
    storage address, so stand-in leaves that alias a parameter (engine.GraphedTrainStep) find their entry too."""
    _WEIGHT_AMAX.clear()
    if lib().query("mi355seg_conv_math_takes_amax") == 0:
        return
    ws = [m.weight for m in model.modules()
          if isinstance(m, torch.nn.Conv3d) and m.weight is not None and m.weight.is_cuda and m.weight.dtype == torch.float32
          and m.weight.is_contiguous() and tuple(m.weight.shape[2:]) == (3, 3, 3)]
    if len(ws) < 2:
        return
    with torch.no_grad():
        norms = torch._foreach_norm(ws, float("inf"))
    for w, a in zip(ws, norms):
        _WEIGHT_AMAX[w.data_ptr()] = (a.reshape(1), w.numel())


def clear_weight_amax():
    _WEIGHT_AMAX.clear()


def _weight_amax(w):
    if _PREPACK_ACTIVE[0]:       # the step's packings (and max |w| with them) were formed up front: the library hands the kernels its own scalars
        return None              # (a weight tensor the plan does not hold is measured by the entry point: NULL = measure)
    rec = _WEIGHT_AMAX.get(w.data_ptr())
    if rec is not None and rec[1] == w.numel() and rec[0].device == w.device:
        return rec[0]
    return _measure_amax(w, w.numel(), 1, w.numel())


# ----------------------------------------------------------------------------- conv
class _Conv3d(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, res=None):
        xa_in = _get_amax(x)
        x, ldx = cl_view(x, "conv3d input")
        N, D, H, W, Cin = x.shape
        Cout, Cin_w, k = w.shape[0], w.shape[1], w.shape[2]
        if Cin_w != Cin or w.shape[3] != k or w.shape[4] != k:
            raise Mi355SegError(f"conv3d: weight {tuple(w.shape)} does not match input channels {Cin} / cubic kernel")
        w = _w32(w, "conv3d weight")
        Do, Ho, Wo = [(e + 2 * pad - k) // stride + 1 for e in (D, H, W)]
        y = torch.empty((N, Do, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
        L = lib()
        ws = workspace(_conv_ws(L, x, N, D, H, W, Cin, Cout, k, stride, pad), x.device)
        ctx.amax = None
        use = _amax_use(x, N, D, H, W, Cin, Cout, k, stride, pad)
        if use:
            xa, wa = xa_in, (_weight_amax(w) if use & 3 else None)
            if xa is None and (use & 1) and (use & 4) and ctx.needs_input_grad[1]:    # forward and weight gradient both read x: measure once
                xa = _measure_amax(x, ldx, N * D * H * W, Cin)
            L.call("mi355seg_conv3d_fwd_ax_f32", _p(x), ldx, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                   None, None, _p(xa), _p(wa), _p(ws), ws.numel(), _stream())
            ctx.amax = (xa, wa)
        elif res is not None:                # bf16: conv(x) + res, the sum in the convolution's epilogue where the launch allows
            res, ldres = cl_view(res, "conv3d residual")
            if res.shape != y.shape or res.dtype != x.dtype:
                raise Mi355SegError(f"conv3d: residual {tuple(res.shape)} {res.dtype} does not match the output {tuple(y.shape)} {x.dtype}")
            L.call("mi355seg_conv3d_fwd_res_bf16", _p(x), ldx, _p(w), _p(b), _p(res), ldres, _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                   _p(ws), ws.numel(), _stream())
        else:
            L.call("mi355seg_conv3d_fwd_" + _sfx(x), _p(x), ldx, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                   None, None, _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(x, w)
        ctx.geom = (N, D, H, W, Cin, Cout, k, stride, pad, ldx, b is not None)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, D, H, W, Cin, Cout, k, stride, pad, ldx, has_b = ctx.geom
        dy, lddy = cl_view(_like(dy, x), "conv3d grad")
        L = lib()
        ws = workspace(_conv_ws(L, x, N, D, H, W, Cin, Cout, k, stride, pad), x.device)
        dx = dw = db = None
        want_dw = ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2])
        if ctx.amax is not None:
            xa, wa = ctx.amax
            da = _get_amax(dy)
            if da is None and ctx.needs_input_grad[0] and want_dw:      # both gradients read dy: measure it once
                da = _measure_amax(dy, lddy, N * D * H * W, Cout)
            if ctx.needs_input_grad[0]:
                dx = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=x.device)
                L.call("mi355seg_conv3d_dgrad_ax_f32", _p(dy), lddy, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, pad,
                       _p(da), _p(wa), _p(ws), ws.numel(), _stream())
            if want_dw:
                dw = torch.empty_like(w)
                db = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_b else None
                L.call("mi355seg_conv3d_wgrad_ax_f32", _p(dy), lddy, _p(x), ldx, _p(dw), _p(db), N, D, H, W, Cin, Cout, k, stride, pad,
                       0, _p(da), _p(xa), _p(ws), ws.numel(), _stream())
            return dx, dw, db, None, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=x.device)
            L.call("mi355seg_conv3d_dgrad_" + _sfx(x), _p(dy), lddy, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, pad,
                   _p(ws), ws.numel(), _stream())
        if want_dw:
            dw = torch.empty_like(w)
            db = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_b else None
            L.call("mi355seg_conv3d_wgrad_" + _sfx(x), _p(dy), lddy, _p(x), ldx, _p(dw), _p(db), N, D, H, W, Cin, Cout, k, stride, pad,
                   0, _p(ws), ws.numel(), _stream())
        return dx, dw, db, None, None, (dy if ctx.has_res else None)        # d(conv + res) / d res = dy itself


def conv3d(x, weight, bias=None, stride=1, padding=0, residual=None):
    """nn.Conv3d on a channel-last tensor (cubic kernel, isotropic stride/padding); residual: conv(x) + residual (bf16 tensors: one
    launch where the convolution's epilogue can take the sum, mi355seg_conv3d_fwd_res_bf16)."""
    if residual is None:
        return _Conv3d.apply(x, weight, bias, int(stride), int(padding))
    if x.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16 and not os.environ.get("MI355SEG_NO_RES_EPILOGUE"):
        return _Conv3d.apply(x, weight, bias, int(stride), int(padding), residual)
    return activation(_Conv3d.apply(x, weight, bias, int(stride), int(padding)), ACT_NONE, residual=residual)


class _ConvT3dK2S2(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, ldx = cl_view(x, "conv_transpose3d input")
        N, D, H, W, Cin = x.shape
        if tuple(w.shape[2:]) != (2, 2, 2) or w.shape[0] != Cin:
            raise Mi355SegError(f"conv_transpose3d_k2s2: weight {tuple(w.shape)} must be (Cin={Cin}, Cout, 2, 2, 2)")
        Cout = w.shape[1]
        w = w.contiguous()
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, Cout), dtype=x.dtype, device=x.device)
        L = lib()
        ws = workspace(L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout), x.device)
        L.call("mi355seg_convt3d_k2s2_fwd_" + _sfx(x), _p(x), ldx, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout,
               _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(x, w)
        ctx.geom = (N, D, H, W, Cin, Cout, ldx, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, D, H, W, Cin, Cout, ldx, has_b = ctx.geom
        dy, lddy = cl_view(_like(dy, x), "conv_transpose3d grad")
        L = lib()
        ws = workspace(L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout), x.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=x.device)
            L.call("mi355seg_convt3d_k2s2_dgrad_" + _sfx(x), _p(dy), lddy, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout,
                   _p(ws), ws.numel(), _stream())
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_b else None
            L.call("mi355seg_convt3d_k2s2_wgrad_" + _sfx(x), _p(dy), lddy, _p(x), ldx, _p(dw), _p(db), N, D, H, W, Cin, Cout,
                   _p(ws), ws.numel(), _stream())
        return dx, dw, db


class _ConvT3dK2S2Cat(Function):
    """cat(conv_transpose3d_k2s2(x), skip) along channels without the copy: ``skip`` must be the right channel slice
    of a buffer with exactly Cout free channels on its left (conv_bn_act(..., left_pad=Cout)); the up-convolution is
    written into those channels and the whole buffer is returned."""

    @staticmethod
    def forward(ctx, x, w, b, skip):
        x, ldx = cl_view(x, "conv_transpose3d input")
        N, D, H, W, Cin = x.shape
        Cout = w.shape[1]
        base = skip._base
        Cs = skip.shape[-1]
        ok = (base is not None and base.dim() == 5 and base.is_contiguous() and tuple(base.shape[:4]) == (N, 2 * D, 2 * H, 2 * W)
              and base.shape[-1] == Cout + Cs and skip.data_ptr() == base.data_ptr() + base.element_size() * Cout and skip.stride() == base.stride()
              and base.dtype == x.dtype)
        if not ok:
            raise Mi355SegError("conv_transpose3d_k2s2_cat: `skip` is not the right channel slice of a matching concat buffer")
        w = w.contiguous()
        L = lib()
        ws = workspace(L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout), x.device)
        sa = _get_amax(skip) if _takes_amax(x) else None
        if sa is not None:
            # max |cat| = max(max |up-convolution| (from that kernel's epilogue), max |skip| (its own scalar, max-combined))
            ca = _amax_slot(x.device)
            L.call("mi355seg_convt3d_k2s2_fwd_ax_f32", _p(x), ldx, _p(w), _p(b), _p(base), Cout + Cs, N, D, H, W, Cin, Cout, _p(ca),
                   _p(ws), ws.numel(), _stream())
            _measure_amax(sa, 1, 1, 1, ca)
            _set_amax(base, ca)
        else:
            L.call("mi355seg_convt3d_k2s2_fwd_" + _sfx(x), _p(x), ldx, _p(w), _p(b), _p(base), Cout + Cs, N, D, H, W, Cin, Cout,
                   _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(x, w)
        ctx.geom = (N, D, H, W, Cin, Cout, Cs, ldx, b is not None)
        return base

    @staticmethod
    def backward(ctx, dcat):
        x, w = ctx.saved_tensors
        N, D, H, W, Cin, Cout, Cs, ldx, has_b = ctx.geom
        dcat, ldd = cl_view(_like(dcat, x), "conv_transpose3d grad")
        L = lib()
        ws = workspace(L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout), x.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=x.device)
            L.call("mi355seg_convt3d_k2s2_dgrad_" + _sfx(x), _p(dcat), ldd, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout,
                   _p(ws), ws.numel(), _stream())
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_b else None
            L.call("mi355seg_convt3d_k2s2_wgrad_" + _sfx(x), _p(dcat), ldd, _p(x), ldx, _p(dw), _p(db), N, D, H, W, Cin, Cout,
                   _p(ws), ws.numel(), _stream())
        return dx, dw, db, dcat[..., Cout:]


def conv_transpose3d_k2s2_cat(x, weight, bias, skip):
    return _ConvT3dK2S2Cat.apply(x, weight, bias, skip)


class _ConvT3dAdjoint(Function):
    """nn.ConvTranspose3d with kernel_size = stride = k, no padding (csrnet.py:121-137 uses k = 4), computed as the
    adjoint of the Conv3d with the same weight tensor: forward = that conv's dgrad, input gradient = its forward,
    weight gradient = its wgrad with the operands swapped.  (Cin, Cout, k, k, k) of the transposed conv IS the
    (Cout_c, Cin_c, k, k, k) layout of the conv, so no repacking is involved."""

    @staticmethod
    def forward(ctx, x, w, b, k):
        x, ldx = cl_view(x, "conv_transpose3d input", allow_bf16=False)
        N, D, H, W, Cin = x.shape
        if w.shape[0] != Cin or tuple(w.shape[2:]) != (k, k, k):
            raise Mi355SegError(f"conv_transpose3d: weight {tuple(w.shape)} does not match input channels {Cin} / kernel {k}")
        w = w.contiguous()
        Cout = w.shape[1]
        geom = (N, k * D, k * H, k * W, Cout, Cin, k, k, 0)             # the adjoint conv: Cout -> Cin channels, stride k
        y = torch.empty((N, k * D, k * H, k * W, Cout), dtype=x.dtype, device=x.device)
        L = lib()
        ws = workspace(L.query("mi355seg_conv3d_ws_bytes", *geom), x.device)
        L.call("mi355seg_conv3d_dgrad_f32", _p(x), ldx, _p(w), _p(y), Cout, *geom, _p(ws), ws.numel(), _stream())
        if b is not None:
            L.call("mi355seg_add_bias_f32", _p(y), Cout, _p(b), y.numel() // Cout, Cout, _stream())
        ctx.save_for_backward(x, w)
        ctx.cfg = (geom, ldx, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        geom, ldx, has_b = ctx.cfg
        N, D2, H2, W2, Cout, Cin, k = geom[:7]
        dy, lddy = cl_view(dy, "conv_transpose3d grad")
        L = lib()
        ws = workspace(L.query("mi355seg_conv3d_ws_bytes", *geom), x.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            L.call("mi355seg_conv3d_fwd_f32", _p(dy), lddy, _p(w), None, _p(dx), Cin, *geom, None, None, _p(ws), ws.numel(), _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            L.call("mi355seg_conv3d_wgrad_f32", _p(x), ldx, _p(dy), lddy, _p(dw), None, *geom, 0, _p(ws), ws.numel(), _stream())
        if has_b and ctx.needs_input_grad[2]:
            rows = N * D2 * H2 * W2
            cws = workspace(L.query("mi355seg_norm_ws_bytes", rows, 1, Cout), x.device)
            db = torch.empty(Cout, dtype=x.dtype, device=x.device)
            L.call("mi355seg_colsum_f32", _p(dy), lddy, rows, Cout, _p(db), _p(cws), cws.numel(), _stream())
        return dx, dw, db, None


def conv_transpose3d_adjoint(x, weight, bias, k):
    return _ConvT3dAdjoint.apply(x, weight, bias, int(k))


def conv_transpose3d_k2s2(x, weight, bias=None):
    """nn.ConvTranspose3d(kernel_size=2, stride=2) on a channel-last tensor."""
    return _ConvT3dK2S2.apply(x, weight, bias)


# ----------------------------------------------------------------------------- norm + activation
class _NormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, training, momentum, eps, act, slope, instance, left_pad=0):
        # left_pad > 0: the result is the RIGHT channel slice of a buffer with left_pad free channels on its left, which a later node
        # fills (conv_in_act(..., cat_right=)): the skip concatenation of residual_unet3d.py:174-209 without the copies
        x, ldx = cl_view(x, "norm input")
        N, D, H, W, C = x.shape
        groups = N if instance else 1
        rows = D * H * W * (1 if instance else N)
        L = lib()
        dev = x.device
        ws = workspace(L.query("mi355seg_norm_ws_bytes", rows, groups, C), dev)
        if res is not None:
            res, ldres = cl_view(res, "norm residual")
        else:
            ldres = 0
        if training or instance:
            mean = torch.empty(groups * C, dtype=torch.float32, device=dev)
            rstd = torch.empty(groups * C, dtype=torch.float32, device=dev)
            upd = training and (running_mean is not None) and not instance
            L.call("mi355seg_norm_stats_" + _sfx(x), _p(x), ldx, rows, groups, C, eps, _p(mean), _p(rstd),
                   _p(running_mean) if upd else None, _p(running_var) if upd else None, momentum,
                   _p(ws), ws.numel(), _stream())
        else:
            mean = running_mean
            rstd = torch.empty(C, dtype=torch.float32, device=dev)
            L.call("mi355seg_rstd_from_var_f32", _p(running_var), eps, _p(rstd), C, _stream())
        if left_pad:
            full = torch.empty((N, D, H, W, left_pad + C), dtype=x.dtype, device=dev)
            y = full[..., left_pad:]
        else:
            y = torch.empty((N, D, H, W, C), dtype=x.dtype, device=dev)
        L.call("mi355seg_norm_act_fwd_" + _sfx(x), _p(x), ldx, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), ldres,
               y.data_ptr(), left_pad + C, rows, groups, C, act, slope, _stream())
        ctx.save_for_backward(x, mean, rstd, gamma, beta, res)
        ctx.cfg = (ldx, ldres, rows, groups, C, act, slope, bool(training or instance))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta, res = ctx.saved_tensors
        ldx, ldres, rows, groups, C, act, slope, batch_stats = ctx.cfg
        if not batch_stats:
            raise Mi355SegError("backward through eval-mode BatchNorm (running statistics) is not supported")
        dy, lddy = cl_view(_like(dy, x), "norm grad")
        L = lib()
        dev = x.device
        ws = workspace(L.query("mi355seg_norm_ws_bytes", rows, groups, C), dev)
        dx = torch.empty(x.shape, dtype=x.dtype, device=dev)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=dev) if gamma is not None else None
        dres = torch.empty(x.shape, dtype=x.dtype, device=dev) if res is not None else None
        L.call("mi355seg_norm_act_bwd_" + _sfx(x), _p(dy), lddy, _p(x), ldx, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), ldres,
               _p(dx), C, _p(dgamma), _p(dbeta), _p(dres), C, rows, groups, C, act, slope, _p(ws), ws.numel(), _stream())
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5,
                   act=ACT_NONE, slope=0.01, residual=None):
    """act(BatchNorm3d(x) [+ residual]); training mode updates running stats in place."""
    return _NormAct.apply(x, gamma, beta, residual, running_mean, running_var, bool(training), float(momentum),
                          float(eps), int(act), float(slope), False, 0)


def instance_norm_act(x, eps=1e-5, act=ACT_NONE, slope=0.01, left_pad=0):
    """act(InstanceNorm3d(x)) with affine=False, track_running_stats=False.  ``left_pad``: see _NormAct (the result is the right channel
    slice of a concat buffer)."""
    return _NormAct.apply(x, None, None, None, None, None, True, 0.0, float(eps), int(act), float(slope), True, int(left_pad))


class _ConvBnAct(Function):
    """act(BatchNorm3d(conv3d(x))) as one autograd node: the batch statistics come out of the convolution's
    epilogue (no separate pass over y) and the convolution's bias gradient comes out of the BatchNorm backward
    pass (no separate pass over dy).  Training mode only updates running stats exactly as nn.BatchNorm3d."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rmean, rvar, stride, pad, training, momentum, eps, act, slope, left_pad, inference=False, fork=False, cat_right=None):
        # cat_right (training): a tensor that IS the right channel slice of a concat buffer with exactly Cout free channels on its left
        # (instance_norm_act / activation_fork(..., left_pad=Cout)): the activation is written into those channels and the whole buffer
        # is returned -- cat((act(norm(conv(x))), cat_right), channels) without the two copies
        # fork (training): the result is (act(bn(conv(x))), x) -- x also continues unchanged (a residual block's input, vnet3d.py:61-104), and
        # the backward sums the pass-through's gradient into the input gradient it computes (bf16: in that kernel's epilogue)
        xa_in = _get_amax(x)
        x_arg = x
        x, ldx = cl_view(x, "conv3d input")
        N, D, H, W, Cin = x.shape
        Cout, k = w.shape[0], w.shape[2]
        if w.shape[1] != Cin:
            raise Mi355SegError(f"conv3d: weight {tuple(w.shape)} does not match input channels {Cin}")
        w = w.contiguous()
        Do, Ho, Wo = [(e + 2 * pad - k) // stride + 1 for e in (D, H, W)]
        dev = x.device
        L = lib()
        ws = workspace(max(_conv_ws(L, x, N, D, H, W, Cin, Cout, k, stride, pad),
                           L.query("mi355seg_norm_ws_bytes", N * Do * Ho * Wo, 1, Cout)), dev)
        rows = N * Do * Ho * Wo
        fused = (not training) and inference and x.data_ptr() % 16 == 0 and \
            L.query("mi355seg_conv3d_fused_supported_" + _sfx(x), N, D, H, W, Cin, Cout, k, stride, pad, ldx, left_pad + Cout)
        y = None if fused else torch.empty((N, Do, Ho, Wo, Cout), dtype=x.dtype, device=dev)
        ax = training and _takes_amax(x)          # f16x3: operand maxima ride along (this layer's input, weights, and its output for the next layer)
        ctx.amax = None
        if ax:
            sums = torch.empty(2 * Cout, dtype=torch.float64, device=dev)
            use = _amax_use(x, N, D, H, W, Cin, Cout, k, stride, pad)
            xa, wa = (xa_in if use else None), None
            if use & 3:
                wa = _weight_amax(w)
            if xa is None and (use & 1) and (use & 4):
                xa = _measure_amax(x, ldx, N * D * H * W, Cin)
            L.call("mi355seg_conv3d_fwd_ax_f32", _p(x), ldx, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                   sums.data_ptr(), sums.data_ptr() + 8 * Cout, _p(xa), _p(wa), _p(ws), ws.numel(), _stream())
            ctx.amax = (xa, wa)
            mean = torch.empty(Cout, dtype=torch.float32, device=dev)
            rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
            L.call("mi355seg_norm_stats_from_sums_f32", sums.data_ptr(), sums.data_ptr() + 8 * Cout, rows, Cout, eps,
                   _p(mean), _p(rstd), _p(rmean), _p(rvar), momentum, _stream())
        elif training:
            sums = torch.empty(2 * Cout, dtype=torch.float64, device=dev)
            L.call("mi355seg_conv3d_fwd_" + _sfx(x), _p(x), ldx, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                   sums.data_ptr(), sums.data_ptr() + 8 * Cout, _p(ws), ws.numel(), _stream())
            mean = torch.empty(Cout, dtype=torch.float32, device=dev)
            rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
            L.call("mi355seg_norm_stats_from_sums_f32", sums.data_ptr(), sums.data_ptr() + 8 * Cout, rows, Cout, eps,
                   _p(mean), _p(rstd), _p(rmean), _p(rvar), momentum, _stream())
        elif fused:
            # inference (model.eval() under no_grad, predict.py:79-81,133): eval-mode BatchNorm folded into the packed weights and
            # the bias slot, the activation in the convolution's epilogue -- one pass, nothing saved for a backward
            fold = torch.empty(2 * Cout, dtype=torch.float32, device=dev)
            L.call("mi355seg_bn_fold_f32", _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(b), eps, Cout, fold.data_ptr(), fold.data_ptr() + 4 * Cout, _stream())
            full = torch.empty((N, Do, Ho, Wo, left_pad + Cout), dtype=x.dtype, device=dev)
            a = full[..., left_pad:] if left_pad else full
            L.call("mi355seg_conv3d_fwd_fused_" + _sfx(x), _p(x), ldx, _p(w), fold.data_ptr(), fold.data_ptr() + 4 * Cout, act, slope,
                   a.data_ptr(), left_pad + Cout, N, D, H, W, Cin, Cout, k, stride, pad, _p(ws), ws.numel(), _stream())
            return a
        else:
            L.call("mi355seg_conv3d_fwd_" + _sfx(x), _p(x), ldx, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                   None, None, _p(ws), ws.numel(), _stream())
            mean = rmean
            rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
            L.call("mi355seg_rstd_from_var_f32", _p(rvar), eps, _p(rstd), Cout, _stream())
        lda = left_pad + Cout
        ctx.cat = 0
        if cat_right is not None:
            base = cat_right._base
            Cs = cat_right.shape[-1]
            if not (base is not None and base.dim() == 5 and base.is_contiguous() and tuple(base.shape) == (N, Do, Ho, Wo, Cout + Cs) and base.dtype == x.dtype
                    and cat_right.data_ptr() == base.data_ptr() + base.element_size() * Cout and cat_right.stride() == base.stride() and not left_pad):
                raise Mi355SegError("conv_bn_act: `cat_right` is not the right channel slice of a matching concat buffer")
            full, a, lda = base, base[..., :Cout], Cout + Cs
            ctx.cat = Cs
        elif left_pad:
            # the activation lands in the RIGHT channel slice of a wider buffer whose left `left_pad` channels a later
            # up-convolution fills (conv_transpose3d_k2s2_cat): the skip concatenation then costs no copy
            full = torch.empty((N, Do, Ho, Wo, left_pad + Cout), dtype=x.dtype, device=dev)
            a = full[..., left_pad:]
        else:
            a = torch.empty_like(y)
        if ax:
            aa = _amax_slot(dev)
            L.call("mi355seg_norm_act_fwd_ax_f32", _p(y), Cout, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
                   a.data_ptr(), lda, rows, 1, Cout, act, slope, _p(aa), _stream())
            if not ctx.cat:
                _set_amax(a, aa)
        else:
            L.call("mi355seg_norm_act_fwd_" + _sfx(x), _p(y), Cout, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
                   a.data_ptr(), lda, rows, 1, Cout, act, slope, _stream())
        ctx.save_for_backward(x, w, y, mean, rstd, gamma, beta)
        ctx.cfg = (N, D, H, W, Cin, Cout, k, stride, pad, ldx, b is not None, rows, act, slope, bool(training))
        if fork:
            ctx.set_materialize_grads(False)
            return a, x_arg.view_as(x_arg)
        if ctx.cat:
            return full
        return a

    @staticmethod
    def backward(ctx, da, dpass=None):
        x, w, y, mean, rstd, gamma, beta = ctx.saved_tensors
        N, D, H, W, Cin, Cout, k, stride, pad, ldx, has_b, rows, act, slope, training = ctx.cfg
        if not training:
            raise Mi355SegError("backward through eval-mode BatchNorm (running statistics) is not supported")
        if da is None:                       # (fork, only the pass-through was used)
            return (dpass,) + (None,) * 17
        if dpass is not None:
            dpass, lddp = cl_view(_like(dpass, x), "pass-through grad")
        dcat = None
        if ctx.cat:                          # da is the gradient of the whole concat buffer: ours is its left slice, the right one goes back to cat_right
            da = _like(da, x)
            dcat, da = da[..., Cout:], da[..., :Cout]
        da, ldda = cl_view(_like(da, x), "conv+norm grad")
        L = lib()
        dev = x.device
        ws = workspace(max(_conv_ws(L, x, N, D, H, W, Cin, Cout, k, stride, pad),
                           L.query("mi355seg_norm_ws_bytes", rows, 1, Cout)), dev)
        dy = torch.empty_like(y)
        dgamma = torch.empty(Cout, dtype=torch.float32, device=dev) if gamma is not None else None      # no affine: instance norm
        dbeta = torch.empty(Cout, dtype=torch.float32, device=dev) if gamma is not None else None
        db = torch.empty(Cout, dtype=torch.float32, device=dev) if has_b else None
        if ctx.amax is not None and (ctx.amax[0] is not None or ctx.amax[1] is not None):
            xa, wa = ctx.amax
            dya = _amax_slot(dev)
            L.call("mi355seg_norm_act_bwd_colsum_ax_f32", _p(da), ldda, _p(y), Cout, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
                   _p(dy), Cout, _p(dgamma), _p(dbeta), None, 0, _p(db), _p(dya), rows, 1, Cout, act, slope, _p(ws), ws.numel(), _stream())
            dx = dw = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=dev)
                L.call("mi355seg_conv3d_dgrad_ax_f32", _p(dy), Cout, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, pad,
                       _p(dya), _p(wa), _p(ws), ws.numel(), _stream())
                if dpass is not None:
                    L.call("mi355seg_act_fwd_f32", _p(dx), Cin, _p(dpass), lddp, _p(dx), Cin, N * D * H * W, Cin, ACT_NONE, 0.0, _stream())
            elif dpass is not None:
                dx = dpass
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w)
                L.call("mi355seg_conv3d_wgrad_ax_f32", _p(dy), Cout, _p(x), ldx, _p(dw), None, N, D, H, W, Cin, Cout, k, stride, pad,
                       0, _p(dya), _p(xa), _p(ws), ws.numel(), _stream())
            return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None, dcat
        L.call("mi355seg_norm_act_bwd_colsum_" + _sfx(x), _p(da), ldda, _p(y), Cout, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
               _p(dy), Cout, _p(dgamma), _p(dbeta), None, 0, _p(db), rows, 1, Cout, act, slope, _p(ws), ws.numel(), _stream())
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=dev)
            if dpass is not None and x.dtype == torch.bfloat16:       # input gradient + the pass-through's gradient: the sum in the kernel's epilogue
                L.call("mi355seg_conv3d_dgrad_res_bf16", _p(dy), Cout, _p(w), _p(dpass), lddp, _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, pad,
                       _p(ws), ws.numel(), _stream())
            else:
                L.call("mi355seg_conv3d_dgrad_" + _sfx(x), _p(dy), Cout, _p(w), _p(dx), Cin, N, D, H, W, Cin, Cout, k, stride, pad,
                       _p(ws), ws.numel(), _stream())
                if dpass is not None:
                    L.call("mi355seg_act_fwd_" + _sfx(x), _p(dx), Cin, _p(dpass), lddp, _p(dx), Cin, N * D * H * W, Cin, ACT_NONE, 0.0, _stream())
        elif dpass is not None:
            dx = dpass
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            L.call("mi355seg_conv3d_wgrad_" + _sfx(x), _p(dy), Cout, _p(x), ldx, _p(dw), None, N, D, H, W, Cin, Cout, k, stride, pad,
                   0, _p(ws), ws.numel(), _stream())
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None, dcat


def conv_in_act(x, conv, norm, act=ACT_NONE, slope=0.01, cat_right=None):
    """act(InstanceNorm3d(conv(x))) (residual_unet3d.py:82-107: conv_norm_lrelu and the tail of norm_lrelu_upscale_conv_norm_lrelu;
    affine=False, no running statistics).  With ONE sample per batch -- cfg 4 -- the per-(sample, channel) statistics are per-channel
    statistics over the whole tensor, i.e. exactly what the convolution's epilogue already reduces for BatchNorm: the node of
    conv_bn_act without affine parameters and buffers (no separate statistics pass over y).  N > 1: convolution, then the
    instance-norm node."""
    if x.shape[0] != 1 or norm.affine or norm.track_running_stats or not torch.is_grad_enabled():
        a = norm.forward_act(conv(x), act, slope)
        return a if cat_right is None else cat_channels(a, cat_right)
    stride = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
    pad = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
    # cat_right: cat((result, cat_right), channels) -- written in place when cat_right is the right slice of a matching concat buffer
    # (instance_norm_act / activation_fork(..., left_pad=)), by two slice copies otherwise
    base = getattr(cat_right, "_base", None) if cat_right is not None else None
    inplace = base is not None and base.dim() == 5 and base.is_contiguous() and base.shape[-1] == conv.out_channels + cat_right.shape[-1] and \
        cat_right.dtype == x.dtype and cat_right.data_ptr() == base.data_ptr() + base.element_size() * conv.out_channels and \
        cat_right.stride() == base.stride() and not os.environ.get("MI355SEG_NO_CAT_FUSION")
    out = _ConvBnAct.apply(x, conv.weight, conv.bias, None, None, None, None, int(stride), int(pad), True, 0.0, float(norm.eps),
                           int(act), float(slope), 0, False, False, cat_right if inplace else None)
    return out if (cat_right is None or inplace) else cat_channels(out, cat_right)


def conv_bn_act(x, conv, bn, act=ACT_NONE, slope=0.01, left_pad=0, fork=False):
    """act(bn(conv(x))) for a layers.Conv3d / layers.BatchNorm3d pair (module objects carry the parameters).  ``fork`` (training mode):
    returns (act(bn(conv(x))), x) -- x continues unchanged beside the convolution (the input of a residual block) and the node's backward
    adds the pass-through's gradient to the input gradient it computes instead of leaving the sum to autograd."""
    if bn.momentum is None or not bn.affine or not bn.track_running_stats:
        raise NotImplementedError("conv_bn_act: BatchNorm3d must be affine with running statistics and a momentum")
    stride = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
    pad = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
    if bn.training:
        bump_counter(bn)
    # eval mode under torch.no_grad() (predict.py:79-81,133) takes the folded one-pass form where the layer has one
    forked = bool(fork) and bn.training and torch.is_grad_enabled()
    out = _ConvBnAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, int(stride), int(pad),
                           bool(bn.training), float(bn.momentum), float(bn.eps), int(act), float(slope), int(left_pad),
                           not torch.is_grad_enabled(), forked)
    return (out, x) if (fork and not forked) else out


class _DoubleConvBnAct(Function):
    """act(BN2(conv2(act(BN1(conv1(x)))))) -- the double-conv block of unet3d.py:73-104 -- as ONE autograd node (training mode, fp32
    tensors).  Forward is the two fused layers of _ConvBnAct back to back.  Because the activation between the two convolutions
    has exactly one consumer here, the backward can let conv2's input-gradient kernel reduce BN1's two backward column sums in its
    epilogue (mi355seg_conv3d_dgrad_bnsums_f32) instead of a separate pass over d(act) and y1."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, rm1, rv1, w2, b2, g2, be2, rm2, rv2, geo1, geo2, mom1, eps1, mom2, eps2, act, slope, left_pad,
                wh=None, bh=None, pool=False):
        """wh / bh: the 1x1x1 output head behind the block (unet3d.py:46-48,71).  The node then returns the head's LOGITS: norm2 +
        activation + head run as one kernel and the block's activation is never written (csrc/bn_head.hip).
        pool: the MaxPool3d(2, 2) behind an encoder block (unet3d.py:51-58).  The node then returns (pooled, activation): norm2 +
        activation + pooling are one kernel, and the backward forms d(activation) = d(skip) + pool_backward(d(pooled)) inside the
        norm backward's two passes."""
        xa_in = _get_amax(x)
        x, ldx = cl_view(x, "conv3d input")
        L = lib()
        dev = x.device
        N = x.shape[0]
        ax = _takes_amax(x)           # f16x3: operand maxima ride along

        def layer(inp, ldin, w, b, g, be, rm, rv, geo, mom, eps, lp, xa, head=None, pool=False, fold=False, pro=None):
            """conv + batch statistics + (norm + activation).  fold: the norm + activation is NOT applied -- the layer hands back its
            folded form (al, be) and a bound on the activation's maximum for the next convolution's prologue; pro = (al, be): this
            convolution's input is the previous layer's RAW output, normalised + activated while its tiles are staged."""
            D, H, W, Cin = inp.shape[1:]
            Cout, k = w.shape[0], w.shape[2]
            stride, pad = geo
            if w.shape[1] != Cin:
                raise Mi355SegError(f"conv3d: weight {tuple(w.shape)} does not match input channels {Cin}")
            Do, Ho, Wo = [(e + 2 * pad - k) // stride + 1 for e in (D, H, W)]
            rows = N * Do * Ho * Wo
            ws = workspace(max(_conv_ws(L, inp, N, D, H, W, Cin, Cout, k, stride, pad), L.query("mi355seg_norm_ws_bytes", rows, 1, Cout)), dev)
            y = torch.empty((N, Do, Ho, Wo, Cout), dtype=inp.dtype, device=dev)
            sums = torch.empty(2 * Cout, dtype=torch.float64, device=dev)
            wa = aa = None
            use = _amax_use(inp, N, D, H, W, Cin, Cout, k, stride, pad) if ax else 0
            if use & 3:
                wa = _weight_amax(w)
            if not use:
                xa = None
            elif xa is None and (use & 1) and (use & 4):
                xa = _measure_amax(inp, ldin, N * D * H * W, Cin)
            ya = _amax_slot(dev) if fold else None
            if _CHECK_AMAX:
                _assert_amax(w, wa, "conv weights")
                if pro is not None:          # the bound on the prologue's output against the activation it stands for
                    z = inp * pro[0] + pro[1]
                    _assert_amax(torch.where(z > 0, z, z * (slope if act == ACT_LRELU else 0.0)), xa, "the norm prologue's bound")
                elif xa is not None:
                    _assert_amax(inp, xa, "conv input")
            if pro is not None:
                L.call("mi355seg_conv3d_fwd_pro_ax_f32", _p(inp), ldin, _p(pro[0]), _p(pro[1]), act, slope, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout,
                       k, stride, pad, sums.data_ptr(), sums.data_ptr() + 8 * Cout, _p(xa), _p(wa), _p(ws), ws.numel(), _stream())
            elif fold:
                L.call("mi355seg_conv3d_fwd_yamax_ax_f32", _p(inp), ldin, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                       sums.data_ptr(), sums.data_ptr() + 8 * Cout, _p(xa), _p(wa), _p(ya), _p(ws), ws.numel(), _stream())
            else:
                L.call("mi355seg_conv3d_fwd_ax_f32", _p(inp), ldin, _p(w), _p(b), _p(y), Cout, N, D, H, W, Cin, Cout, k, stride, pad,
                       sums.data_ptr(), sums.data_ptr() + 8 * Cout, _p(xa), _p(wa), _p(ws), ws.numel(), _stream())
            mean = torch.empty(Cout, dtype=torch.float32, device=dev)
            rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
            cfg = (N, D, H, W, Cin, Cout, k, stride, pad, ldin, b is not None, rows)
            if fold:                         # statistics + folded normalisation + the bound of the activation's maximum: one small launch
                alb = torch.empty(2 * Cout, dtype=torch.float32, device=dev)
                aa = _amax_slot(dev)
                L.call("mi355seg_norm_fold_f32", sums.data_ptr(), sums.data_ptr() + 8 * Cout, rows, Cout, eps, _p(g), _p(be), act,
                       _p(mean), _p(rstd), _p(rm), _p(rv), mom, _p(ya), alb.data_ptr(), alb.data_ptr() + 4 * Cout, _p(aa), _stream())
                return y, mean, rstd, (alb[:Cout], alb[Cout:]), cfg, (xa, wa, aa)
            L.call("mi355seg_norm_stats_from_sums_f32", sums.data_ptr(), sums.data_ptr() + 8 * Cout, rows, Cout, eps,
                   _p(mean), _p(rstd), _p(rm), _p(rv), mom, _stream())
            if head is not None:             # norm + activation + 1x1x1 head: logits, no activation tensor
                hw, hb = head
                K = hw.shape[0]
                lg = torch.empty((N, Do, Ho, Wo, K), dtype=inp.dtype, device=dev)
                L.call("mi355seg_bn_act_head_fwd_f32", _p(y), Cout, _p(mean), _p(rstd), _p(g), _p(be), act, slope, _p(hw), _p(hb),
                       _p(lg), K, rows, Cout, K, _stream())
                return y, mean, rstd, lg, cfg, (xa, wa, None)
            full = torch.empty((N, Do, Ho, Wo, lp + Cout), dtype=inp.dtype, device=dev)
            a = full[..., lp:] if lp else full
            if ax:
                aa = _amax_slot(dev)
            if pool:                         # norm + activation + MaxPool3d(2, 2): the skip tensor, the pooled tensor and the argmax codes
                pd = torch.empty((N, Do // 2, Ho // 2, Wo // 2, Cout), dtype=inp.dtype, device=dev)
                idx = torch.empty((N, Do // 2, Ho // 2, Wo // 2, Cout), dtype=torch.uint8, device=dev)
                L.call("mi355seg_bn_act_pool_fwd_f32", _p(y), Cout, _p(mean), _p(rstd), _p(g), _p(be), act, slope, a.data_ptr(), lp + Cout,
                       _p(pd), _p(idx), _p(aa), N, Do, Ho, Wo, Cout, _stream())
                return y, mean, rstd, (a, pd, idx), cfg, (xa, wa, aa)
            L.call("mi355seg_norm_act_fwd_ax_f32", _p(y), Cout, _p(mean), _p(rstd), _p(g), _p(be), None, 0,
                   a.data_ptr(), lp + Cout, rows, 1, Cout, act, slope, _p(aa), _stream())
            return y, mean, rstd, a, cfg, (xa, wa, aa)

        w1, w2 = w1.contiguous(), w2.contiguous()
        head = None
        if wh is not None:
            wh = wh.contiguous()
            head = (wh, bh)
        # norm1 + activation as a PROLOGUE of conv2 (its forward and its weight gradient read conv1's raw output): where both run the
        # f16x3 kernels the activation between the two convolutions is never written (mi355seg_conv3d_fwd_pro_ax_f32)
        C1, k1_, (s1_, p1_) = w1.shape[0], w1.shape[2], geo1
        e1 = [(e + 2 * p1_ - k1_) // s1_ + 1 for e in x.shape[1:4]]
        fuse1 = ax and not os.environ.get("MI355SEG_NO_PRO_FUSION") and \
            L.query("mi355seg_conv3d_pro_supported_f32", N, e1[0], e1[1], e1[2], C1, w2.shape[0], w2.shape[2], geo2[0], geo2[1], act) != 0
        y1, mean1, rstd1, a1, cfg1, am1 = layer(x, ldx, w1, b1, g1, be1, rm1, rv1, geo1, mom1, eps1, 0, xa_in, fold=fuse1)
        pro1 = None
        if fuse1:
            pro1, a1 = a1, None
            y2, mean2, rstd2, a2, cfg2, am2 = layer(y1, C1, w2, b2, g2, be2, rm2, rv2, geo2, mom2, eps2, left_pad, am1[2], head, pool, pro=pro1)
        else:
            y2, mean2, rstd2, a2, cfg2, am2 = layer(a1, a1.shape[-1], w2, b2, g2, be2, rm2, rv2, geo2, mom2, eps2, left_pad, am1[2], head, pool)
        idx = None
        if pool:
            a2, pd, idx = a2
        ctx.save_for_backward(x, w1, y1, mean1, rstd1, g1, be1, a1, w2, y2, mean2, rstd2, g2, be2, wh, idx,
                              pro1[0] if fuse1 else None, pro1[1] if fuse1 else None)
        ctx.cfg = (cfg1, cfg2, act, slope)
        ctx.amax = (am1, am2)
        # the prologue form exists under one conv math only: its backward must meet the policy the forward chose (a set_conv_math between
        # the two would otherwise make the weight gradient's entry point refuse -- the activation it would need was never written)
        ctx.math = L.query("mi355seg_get_conv_math") if fuse1 else None
        ctx.head_bias = head is not None and bh is not None
        if pool:                             # max |max_pool(a)| <= max |a| (equal for the non-negative outputs of a ReLU)
            return _set_amax(pd, am2[2]), _set_amax(a2, am2[2])
        return a2 if head is not None else _set_amax(a2, am2[2])

    @staticmethod
    def backward(ctx, da2, dskip=None):
        x, w1, y1, mean1, rstd1, g1, be1, a1, w2, y2, mean2, rstd2, g2, be2, wh, idx, al1, bl1 = ctx.saved_tensors
        cfg1, cfg2, act, slope = ctx.cfg
        N, D1, H1, W1, Cin1, C1, k1, st1, pd1, ldx, has_b1, rows1 = cfg1
        _, D2, H2, W2, _, C2, k2, st2, pd2, lda1, has_b2, rows2 = cfg2
        L = lib()
        dev = x.device
        if da2 is not None:
            da2, ldda2 = cl_view(_like(da2, x), "conv+norm grad")
        ws = workspace(max(_conv_ws(L, x, N, D1, H1, W1, Cin1, C1, k1, st1, pd1), _conv_ws(L, y1, N, D2, H2, W2, C1, C2, k2, st2, pd2),
                           L.query("mi355seg_norm_ws_bytes", rows1, 1, C1), L.query("mi355seg_norm_ws_bytes", rows2, 1, C2),
                           L.query("mi355seg_bn_act_head_ws_bytes", C2, wh.shape[0]) if wh is not None else 0,
                           L.query("mi355seg_bn_act_pool_ws_bytes", C2) if idx is not None else 0), dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # layer 2: BatchNorm + activation backward (its dx column sums are conv2's bias gradient)
        dy2 = torch.empty_like(y2)
        dg2, dbe2 = torch.empty(C2, **f32), torch.empty(C2, **f32)
        db2 = torch.empty(C2, **f32) if has_b2 else None
        (xa1, wa1, _), (xa2, wa2, _) = ctx.amax
        # f16x3: the two gradients d(conv output) take their maxima from the norm-backward kernels that write them
        dya2 = _amax_slot(dev) if (wa2 is not None or xa2 is not None) else None
        dya1 = _amax_slot(dev) if (wa1 is not None or xa1 is not None) else None
        dwh = dbh = None
        if idx is not None:
            # encoder block behind a max-pool: da2 is d(pooled), dskip the skip connection's gradient; d(activation) = dskip +
            # pool_backward(d(pooled)) exists only inside the two passes of the norm backward
            if dskip is None:                # (an unused skip output / pooled output: its gradient is zero)
                dskip = torch.zeros((N, D2, H2, W2, C2), **f32)
            if da2 is None:
                da2 = torch.zeros((N, D2 // 2, H2 // 2, W2 // 2, C2), **f32)
            ds, ldds = cl_view(_like(dskip, x), "skip grad")
            da2 = da2.contiguous()
            sp = torch.empty(2 * C2, **f32)
            L.call("mi355seg_bn_act_pool_bwd_f32", _p(ds), ldds, _p(da2), _p(idx), _p(y2), C2, _p(mean2), _p(rstd2), _p(g2), _p(be2), act, slope,
                   sp.data_ptr(), sp.data_ptr() + 4 * C2, _p(dg2), _p(dbe2), _p(dy2), C2, _p(db2), _p(dya2), N, D2, H2, W2, C2, _p(ws), ws.numel(), _stream())
        elif wh is not None:
            # da2 here is d(logits) [rows2, K]: the norm backward's column sums, the head's weight / bias gradients (one pass over
            # y2 and d(logits)), then dy2 with conv2's bias gradient and its f16x3 maximum (a second pass)
            K = wh.shape[0]
            sh = torch.empty(2 * C2, **f32)
            dwh = torch.empty_like(wh)
            dbh = torch.empty(K, **f32) if ctx.head_bias else None
            L.call("mi355seg_bn_act_head_bwd_sums_f32", _p(da2), ldda2, _p(y2), C2, _p(mean2), _p(rstd2), _p(g2), _p(be2), act, slope, _p(wh),
                   sh.data_ptr(), sh.data_ptr() + 4 * C2, _p(dg2), _p(dbe2), _p(dwh), _p(dbh), rows2, C2, K, _p(ws), ws.numel(), _stream())
            L.call("mi355seg_bn_act_head_bwd_apply_f32", _p(da2), ldda2, _p(y2), C2, _p(mean2), _p(rstd2), _p(g2), _p(be2), act, slope, _p(wh),
                   sh.data_ptr(), sh.data_ptr() + 4 * C2, _p(dy2), C2, _p(db2), _p(dya2), rows2, C2, K, _p(ws), ws.numel(), _stream())
        else:
            L.call("mi355seg_norm_act_bwd_colsum_ax_f32", _p(da2), ldda2, _p(y2), C2, _p(mean2), _p(rstd2), _p(g2), _p(be2), None, 0,
                   _p(dy2), C2, _p(dg2), _p(dbe2), None, 0, _p(db2), _p(dya2), rows2, 1, C2, act, slope, _p(ws), ws.numel(), _stream())
        # conv2 input gradient = d(act1); BN1's two column sums come out of the same kernel
        da1 = torch.empty((N, D2, H2, W2, C1), dtype=x.dtype, device=dev)
        s12 = torch.empty(2 * C1, **f32)
        dg1, dbe1 = torch.empty(C1, **f32), torch.empty(C1, **f32)
        L.call("mi355seg_conv3d_dgrad_bnsums_ax_f32", _p(dy2), C2, _p(w2), _p(da1), C1, N, D2, H2, W2, C1, C2, k2, st2, pd2,
               _p(y1), C1, _p(mean1), _p(rstd1), _p(g1), _p(be1), act, slope, s12.data_ptr(), s12.data_ptr() + 4 * C1, _p(dg1), _p(dbe1),
               _p(dya2), _p(wa2), _p(ws), ws.numel(), _stream())
        dw2 = torch.empty_like(w2)
        if al1 is not None:                  # conv2 read conv1's raw output through the norm + activation prologue: so does its weight gradient
            now = L.query("mi355seg_get_conv_math")
            if now != ctx.math:
                L.call("mi355seg_set_conv_math", ctx.math)
            try:
                L.call("mi355seg_conv3d_wgrad_pro_ax_f32", _p(dy2), C2, _p(y1), C1, _p(al1), _p(bl1), act, slope, _p(dw2), None,
                       N, D2, H2, W2, C1, C2, k2, st2, pd2, 0, _p(dya2), _p(xa2), _p(ws), ws.numel(), _stream())
            finally:
                if now != ctx.math:
                    L.call("mi355seg_set_conv_math", now)
        else:
            L.call("mi355seg_conv3d_wgrad_ax_f32", _p(dy2), C2, _p(a1), lda1, _p(dw2), None, N, D2, H2, W2, C1, C2, k2, st2, pd2,
                   0, _p(dya2), _p(xa2), _p(ws), ws.numel(), _stream())
        del dy2
        # the 1-channel stem whose input needs no gradient (enc1conv1, unet3d.py:80-89): d(conv1 output) has ONE consumer, the stem's weight
        # gradient, which forms it from d(act1) and y1 on the fly -- the apply pass and the 537-MB tensor it writes are gone
        if not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and Cin1 == 1 and not os.environ.get("MI355SEG_NO_STEM_FUSION") and \
                L.query("mi355seg_stem_wgrad_bnbwd_supported_f32", N, D1, H1, W1, Cin1, C1, k1, st1, pd1) != 0:
            dw1 = torch.empty_like(w1)
            db1 = torch.empty(C1, **f32) if has_b1 else None
            L.call("mi355seg_stem_wgrad_bnbwd_f32", _p(da1), C1, _p(y1), C1, _p(mean1), _p(rstd1), _p(g1), _p(be1), act, slope,
                   s12.data_ptr(), s12.data_ptr() + 4 * C1, _p(x), ldx, _p(dw1), _p(db1), N, D1, H1, W1, Cin1, C1, k1, st1, pd1,
                   _p(ws), ws.numel(), _stream())
            return (None, dw1, db1, dg1, dbe1, None, None, dw2, db2, dg2, dbe2, None, None) + (None,) * 9 + (dwh, dbh, None)
        # layer 1: the apply half of the norm backward (+ conv1's bias gradient), then conv1's gradients
        dy1 = torch.empty_like(y1)
        db1 = torch.empty(C1, **f32) if has_b1 else None
        L.call("mi355seg_norm_act_bwd_apply_ax_f32", _p(da1), C1, _p(y1), C1, _p(mean1), _p(rstd1), _p(g1), _p(be1), None, 0,
               s12.data_ptr(), s12.data_ptr() + 4 * C1, _p(dy1), C1, None, 0, _p(db1), _p(dya1), rows1, 1, C1, act, slope, _p(ws), ws.numel(), _stream())
        del da1
        dx = dw1 = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, D1, H1, W1, Cin1), dtype=x.dtype, device=dev)
            L.call("mi355seg_conv3d_dgrad_ax_f32", _p(dy1), C1, _p(w1), _p(dx), Cin1, N, D1, H1, W1, Cin1, C1, k1, st1, pd1,
                   _p(dya1), _p(wa1), _p(ws), ws.numel(), _stream())
        if ctx.needs_input_grad[1]:
            dw1 = torch.empty_like(w1)
            L.call("mi355seg_conv3d_wgrad_ax_f32", _p(dy1), C1, _p(x), ldx, _p(dw1), None, N, D1, H1, W1, Cin1, C1, k1, st1, pd1,
                   0, _p(dya1), _p(xa1), _p(ws), ws.numel(), _stream())
        return (dx, dw1, db1, dg1, dbe1, None, None, dw2, db2, dg2, dbe2, None, None) + (None,) * 9 + (dwh, dbh, None)


def double_conv_bn_act(x, conv1, bn1, conv2, bn2, act=ACT_NONE, slope=0.01, left_pad=0, head=None, pool=False):
    """act(bn2(conv2(act(bn1(conv1(x)))))): training mode on fp32 tensors runs as one autograd node (_DoubleConvBnAct), everything
    else as two conv_bn_act layers.  ``head`` (a layers.Conv3d with kernel_size 1: the output head behind the LAST block,
    unet3d.py:46-48,71): the result is ``head(block(x))``; in the fused node norm2 + activation + head are one kernel."""
    fused = bn1.training and bn2.training and torch.is_grad_enabled() and compute_dtype() == torch.float32 and x.dtype == torch.float32
    for bn in (bn1, bn2):
        fused = fused and bn.momentum is not None and bn.affine and bn.track_running_stats
    if pool and head is not None:
        raise Mi355SegError("double_conv_bn_act: a block feeds either the output head or a max-pool, not both")
    if not fused:
        a = conv_bn_act(conv_bn_act(x, conv1, bn1, act, slope), conv2, bn2, act, slope, left_pad=left_pad)
        if pool:
            return max_pool3d_2x_and_skip(a)
        return a if head is None else head(a)
    head_fused = False
    if head is not None:
        hk = head.kernel_size[0] if isinstance(head.kernel_size, (tuple, list)) else head.kernel_size
        hs = head.stride[0] if isinstance(head.stride, (tuple, list)) else head.stride
        hp = head.padding[0] if isinstance(head.padding, (tuple, list)) else head.padding
        head_fused = (hk, hs, hp) == (1, 1, 0) and head.groups == 1 and head.weight.dtype == torch.float32 and left_pad == 0 and \
            head.in_channels == conv2.out_channels and not os.environ.get("MI355SEG_NO_HEAD_FUSION") and \
            lib().query("mi355seg_bn_act_head_supported_f32", 1, conv2.out_channels, head.out_channels, conv2.out_channels) != 0

    def geo(conv):
        st = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
        pd = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
        return int(st), int(pd)
    bump_counter(bn1)
    bump_counter(bn2)
    args = (x, conv1.weight, conv1.bias, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
            conv2.weight, conv2.bias, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
            geo(conv1), geo(conv2), float(bn1.momentum), float(bn1.eps), float(bn2.momentum), float(bn2.eps),
            int(act), float(slope), int(left_pad))
    if head_fused:
        return _DoubleConvBnAct.apply(*args, head.weight, head.bias)
    if pool:
        # ``pool``: (max_pool3d_2x(block(x)), block(x)) -- in the fused node norm2 + activation + pooling are one kernel and the pool's
        # backward rides in the norm backward; geometry the fused kernels do not take (odd extents) pools separately
        _, D, H, W, _ = x.shape if x.dim() == 5 else (0, 0, 0, 0, 0)
        k2, (s2_, p2_) = conv2.weight.shape[2], geo(conv2)
        k1, (s1_, p1_) = conv1.weight.shape[2], geo(conv1)
        ext = [((e + 2 * p1_ - k1) // s1_ + 1 + 2 * p2_ - k2) // s2_ + 1 for e in (D, H, W)]
        # (the fused kernels address the skip slice -- full[..., left_pad:] -- and its gradient by 16-byte quads: left_pad and the buffer's
        # channel pitch must be multiples of four, else the separate pool node below takes the shape)
        if not os.environ.get("MI355SEG_NO_POOL_FUSION") and left_pad % 4 == 0 and (left_pad + conv2.out_channels) % 4 == 0 and \
                lib().query("mi355seg_bn_act_pool_supported_f32", int(x.shape[0]), ext[0], ext[1], ext[2], conv2.out_channels, conv2.out_channels) != 0:
            return _DoubleConvBnAct.apply(*args, None, None, True)
        return max_pool3d_2x_and_skip(_DoubleConvBnAct.apply(*args))
    a = _DoubleConvBnAct.apply(*args)
    return a if head is None else head(a)


class _Act(Function):
    @staticmethod
    def forward(ctx, x, res, act, slope):
        x, ldx = cl_view(x, "activation input")
        N, D, H, W, C = x.shape
        if res is not None:
            res, ldres = cl_view(res, "activation residual")
        else:
            ldres = 0
        y = torch.empty((N, D, H, W, C), dtype=x.dtype, device=x.device)
        rows = N * D * H * W
        lib().call("mi355seg_act_fwd_" + _sfx(x), _p(x), ldx, _p(res), ldres, _p(y), C, rows, C, act, slope, _stream())
        ctx.cfg = (ldx, ldres, rows, C, act, slope)
        ctx.has_res = res is not None
        ctx.x_dtype = x.dtype
        if act != ACT_NONE:                  # a plain sum needs nothing for its backward
            ctx.save_for_backward(x, res)
        return y

    @staticmethod
    def backward(ctx, dy):
        ldx, ldres, rows, C, act, slope = ctx.cfg
        if act == ACT_NONE:                  # x + residual: the gradient of both is dy itself -- no kernel, no copy
            g = dy if dy.dtype == ctx.x_dtype else dy.to(ctx.x_dtype)
            return g, (g if ctx.has_res else None), None, None
        x, res = ctx.saved_tensors
        dy, lddy = cl_view(_like(dy, x), "activation grad")
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("mi355seg_act_bwd_" + _sfx(x), _p(dy), lddy, _p(x), ldx, _p(res), ldres, _p(dx), C, rows, C, act, slope, _stream())
        return dx, (dx if res is not None else None), None, None


def activation(x, act, slope=0.01, residual=None):
    """act(x [+ residual]) for ReLU / ELU / LeakyReLU."""
    return _Act.apply(x, residual, int(act), float(slope))


class _ActFork(Function):
    """(act(x), x): the activation of a tensor that ALSO continues unchanged (a residual fork, residual_unet3d.py:110-121).  The backward forms
    d(x) = d(pass-through) + d(act) * act'(x) in one pass (mi355seg_act_bwd_add_*) -- autograd would run the activation's backward and then
    add the two gradients with a kernel of its own."""

    @staticmethod
    def forward(ctx, x, act, slope, left_pad=0):
        xv, ldx = cl_view(x, "activation input")
        N, D, H, W, C = xv.shape
        if left_pad:                         # the activation is the RIGHT channel slice of a concat buffer a later node fills on the left
            y = torch.empty((N, D, H, W, left_pad + C), dtype=xv.dtype, device=xv.device)[..., left_pad:]
        else:
            y = torch.empty((N, D, H, W, C), dtype=xv.dtype, device=xv.device)
        rows = N * D * H * W
        lib().call("mi355seg_act_fwd_" + _sfx(xv), _p(xv), ldx, None, 0, y.data_ptr(), left_pad + C, rows, C, act, slope, _stream())
        ctx.cfg = (ldx, rows, C, act, slope)
        ctx.save_for_backward(xv)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dpass):
        ldx, rows, C, act, slope = ctx.cfg
        (x,) = ctx.saved_tensors
        if dy is None:
            return dpass, None, None, None
        dy, lddy = cl_view(_like(dy, x), "activation grad")
        add = ldadd = None
        if dpass is not None:
            add, ldadd = cl_view(_like(dpass, x), "pass-through grad")
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("mi355seg_act_bwd_add_" + _sfx(x), _p(dy), lddy, _p(x), ldx, None, 0, _p(add), ldadd or 0, _p(dx), C, rows, C, act, slope, _stream())
        return dx, None, None, None


def activation_fork(x, act, slope=0.01, left_pad=0):
    """(act(x), x) for a tensor that feeds an activation and continues unchanged; see _ActFork.  ``left_pad``: act(x) is the right channel
    slice of a concat buffer with that many free channels on its left."""
    return _ActFork.apply(x, int(act), float(slope), int(left_pad))


class _PReLU(Function):
    @staticmethod
    def forward(ctx, x, res, slope):
        x, ldx = cl_view(x, "prelu input")
        N, D, H, W, C = x.shape
        if res is not None:
            res, ldres = cl_view(res, "prelu residual")
        else:
            ldres = 0
        slope = slope.contiguous().to(torch.float32)
        if slope.numel() != C:
            raise Mi355SegError(f"prelu: {slope.numel()} slopes for {C} channels")
        y = torch.empty((N, D, H, W, C), dtype=x.dtype, device=x.device)
        rows = N * D * H * W
        lib().call("mi355seg_prelu_fwd_" + _sfx(x), _p(x), ldx, _p(res), ldres, _p(slope), _p(y), C, rows, C, _stream())
        ctx.save_for_backward(x, res, slope)
        ctx.cfg = (ldx, ldres, rows, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, slope = ctx.saved_tensors
        ldx, ldres, rows, C = ctx.cfg
        dy, lddy = cl_view(_like(dy, x), "prelu grad")
        L = lib()
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        dslope = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(L.query("mi355seg_norm_ws_bytes", rows, 1, C), x.device)
        L.call("mi355seg_prelu_bwd_" + _sfx(x), _p(dy), lddy, _p(x), ldx, _p(res), ldres, _p(slope), _p(dx), C, _p(dslope), rows, C,
               _p(ws), ws.numel(), _stream())
        return dx, (dx if res is not None else None), dslope


def prelu(x, weight, residual=None):
    """nn.PReLU(C)(x [+ residual]) with one learnable slope per channel (vnet3d.py:14-18, elu=False)."""
    return _PReLU.apply(x, residual, weight)


class _ScaleChannels(Function):
    @staticmethod
    def forward(ctx, x, scale):
        x, ldx = cl_view(x, "dropout input")
        N, D, H, W, C = x.shape
        scale = scale.contiguous().to(torch.float32)
        if scale.numel() != N * C:
            raise Mi355SegError(f"scale_channels: scale must hold N*C = {N * C} values, got {scale.numel()}")
        y = torch.empty((N, D, H, W, C), dtype=x.dtype, device=x.device)
        lib().call("mi355seg_scale_channels_" + _sfx(x), _p(x), ldx, _p(scale), _p(y), C, D * H * W, N, C, _stream())
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy, lddy = cl_view(dy, "dropout grad")
        N, D, H, W, C = dy.shape
        dx = torch.empty((N, D, H, W, C), dtype=dy.dtype, device=dy.device)
        lib().call("mi355seg_scale_channels_" + _sfx(dy), _p(dy), lddy, _p(scale), _p(dx), C, D * H * W, N, C, _stream())
        return dx, None


def scale_channels(x, scale):
    """y[n,...,c] = x[n,...,c] * scale[n,c] -- the arithmetic of nn.Dropout3d given its mask."""
    return _ScaleChannels.apply(x, scale)


# ----------------------------------------------------------------------------- pool / upsample
class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        xa = _get_amax(x)
        x, ldx = cl_view(x, "max_pool3d input")
        N, D, H, W, C = x.shape
        y = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        idx = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)
        lib().call("mi355seg_maxpool2_fwd_" + _sfx(x), _p(x), ldx, _p(y), C, _p(idx), N, D, H, W, C, _stream())
        ctx.save_for_backward(idx)
        ctx.geom = (N, D, H, W, C)
        return _set_amax(y, xa)             # max |max_pool(x)| <= max |x|

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, D, H, W, C = ctx.geom
        dy, lddy = cl_view(dy, "max_pool3d grad")
        dx = torch.empty((N, D, H, W, C), dtype=dy.dtype, device=dy.device)
        lib().call("mi355seg_maxpool2_bwd_" + _sfx(dy), _p(dy), lddy, _p(idx), _p(dx), C, N, D, H, W, C, _stream())
        return dx


class _PoolAndSkip(Function):
    """(max_pool3d_2x(x), x): the pooled tensor for the next level and x itself for the skip connection, as ONE
    autograd node so that the two incoming gradients are summed inside the pool-backward kernel."""

    @staticmethod
    def forward(ctx, x):
        xv, ldx = cl_view(x, "max_pool3d input")
        N, D, H, W, C = xv.shape
        y = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=xv.dtype, device=xv.device)
        idx = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=torch.uint8, device=xv.device)
        lib().call("mi355seg_maxpool2_fwd_" + _sfx(xv), _p(xv), ldx, _p(y), C, _p(idx), N, D, H, W, C, _stream())
        ctx.save_for_backward(idx)
        ctx.geom = (N, D, H, W, C)
        skip = x.view_as(x)
        xa = _get_amax(x)
        if xa is not None:              # max |max_pool(x)| <= max |x| (equal for the non-negative outputs of a ReLU)
            _set_amax(y, xa)
            _set_amax(skip, xa)
        return y, skip

    @staticmethod
    def backward(ctx, dy, dskip):
        (idx,) = ctx.saved_tensors
        N, D, H, W, C = ctx.geom
        dy, lddy = cl_view(dy, "max_pool3d grad")
        ds, lds = cl_view(dskip, "skip grad")
        dx = torch.empty((N, D, H, W, C), dtype=dy.dtype, device=dy.device)
        lib().call("mi355seg_maxpool2_bwd_add_" + _sfx(dy), _p(dy), lddy, _p(idx), _p(ds), lds, _p(dx), C, N, D, H, W, C, _stream())
        return dx


def max_pool3d_2x_and_skip(x):
    return _PoolAndSkip.apply(x)


def max_pool3d_2x(x):
    """nn.MaxPool3d(kernel_size=2, stride=2)."""
    return _MaxPool2.apply(x)


class _Upsample2(Function):
    @staticmethod
    def forward(ctx, x):
        x, ldx = cl_view(x, "upsample input")
        N, D, H, W, C = x.shape
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
        lib().call("mi355seg_upsample2_fwd_" + _sfx(x), _p(x), ldx, _p(y), C, N, D, H, W, C, _stream())
        ctx.geom = (N, D, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, D, H, W, C = ctx.geom
        dy, lddy = cl_view(dy, "upsample grad")
        dx = torch.empty((N, D, H, W, C), dtype=dy.dtype, device=dy.device)
        lib().call("mi355seg_upsample2_bwd_" + _sfx(dy), _p(dy), lddy, _p(dx), C, N, D, H, W, C, _stream())
        return dx


def upsample_nearest_2x(x):
    """nn.Upsample(scale_factor=2, mode='nearest')."""
    return _Upsample2.apply(x)


# ----------------------------------------------------------------------------- losses / metric kernels
class _BCEWithLogits(Function):
    @staticmethod
    def forward(ctx, logits, target):
        _require_cuda(logits, "bce_with_logits input")
        logits = logits.contiguous()
        target = target.contiguous().to(torch.float32)
        if logits.shape != target.shape:
            raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({tuple(logits.shape)})")
        L = lib()
        ws = workspace(L.query("mi355seg_loss_ws_bytes", logits.numel()), logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        L.call("mi355seg_bce_logits_fwd_f32", _p(logits), _p(target), logits.numel(), _p(loss), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        d = torch.empty_like(logits)
        lib().call("mi355seg_bce_logits_bwd_f32", _p(logits), _p(target), _p(g), logits.numel(), _p(d), _stream())
        return d, None


def bce_with_logits(logits, target):
    """nn.BCEWithLogitsLoss() (mean reduction)."""
    return _BCEWithLogits.apply(logits, target)


def argmax_channels(logits):
    """pred.argmax(dim=1, keepdim=True) on an NCDHW tensor -> int64 [N,1,D,H,W]."""
    _require_cuda(logits, "argmax input")
    logits = logits.contiguous()
    N, K = logits.shape[0], logits.shape[1]
    S = logits[0, 0].numel()
    mask = torch.empty((N, 1) + tuple(logits.shape[2:]), dtype=torch.int64, device=logits.device)
    lib().call("mi355seg_argmax_ch_f32", _p(logits), N, K, S, _p(mask), _stream())
    return mask


def dice_counts(gt, pred):
    """Integer counters of utils/metric.py on two int64 device tensors -> int64[4]
    (sum gt, sum pred, nnz(gt & pred), nnz(gt | pred))."""
    if not gt.is_cuda or gt.dtype != torch.int64 or pred.dtype != torch.int64:
        raise Mi355SegError("dice_counts: expected int64 tensors on the GPU")
    gt, pred = gt.contiguous(), pred.contiguous()
    L = lib()
    ws = workspace(L.query("mi355seg_loss_ws_bytes", gt.numel()), gt.device)
    out = torch.empty(4, dtype=torch.int64, device=gt.device)
    L.call("mi355seg_dice_counts_i64", _p(gt), _p(pred), gt.numel(), _p(out), _p(ws), ws.numel(), _stream())
    return out


class _BCEArgmaxDice(Function):
    """train.py:204,209,221 in ONE pass over the logits: BCE-with-logits mean loss (differentiable w.r.t. the logits),
    pred.argmax(1, keepdim) and the four integer Dice counters of metric(gt.argmax, mask)."""

    @staticmethod
    def forward(ctx, logits, target):
        _require_cuda(logits, "bce_argmax_dice input")
        logits, target = logits.contiguous(), target.contiguous().to(torch.float32)
        if logits.shape != target.shape:
            raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({tuple(logits.shape)})")
        N, K = logits.shape[0], logits.shape[1]
        S = logits[0, 0].numel()
        L = lib()
        ws = workspace(L.query("mi355seg_loss_ws_bytes", logits.numel()), logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        mask = torch.empty((N, 1) + tuple(logits.shape[2:]), dtype=torch.int64, device=logits.device)
        counts = torch.empty(4, dtype=torch.int64, device=logits.device)
        L.call("mi355seg_bce_argmax_dice_f32", _p(logits), _p(target), N, K, S, _p(loss), _p(mask), _p(counts),
               _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(logits, target)
        ctx.mark_non_differentiable(mask, counts)
        return loss, mask, counts

    @staticmethod
    def backward(ctx, g, _gmask, _gcounts):
        logits, target = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        d = torch.empty_like(logits)
        lib().call("mi355seg_bce_logits_bwd_f32", _p(logits), _p(target), _p(g), logits.numel(), _p(d), _stream())
        return d, None


def bce_argmax_dice(logits, target):
    """Fused tail of the train step: (loss, mask int64 [N,1,...], counts int64[4]); the loss carries the autograd
    edge of nn.BCEWithLogitsLoss()."""
    return _BCEArgmaxDice.apply(logits, target)


def two_channel_gt(gt):
    """train.py:190-193 as one kernel: cat([(gt == 0), gt], dim=1) as float for a single-channel label volume."""
    if not gt.is_cuda:
        raise Mi355SegError(f"two_channel_gt input: expected a tensor on an MI355X (cuda/HIP) device, got {gt.device}")
    gt = gt.contiguous().to(torch.float32)
    if gt.dim() < 3 or gt.shape[1] != 1:
        raise Mi355SegError(f"two_channel_gt: expected [N,1,...], got {tuple(gt.shape)}")
    N = gt.shape[0]
    S = gt[0].numel()
    out = torch.empty((N, 2) + tuple(gt.shape[2:]), dtype=torch.float32, device=gt.device)
    lib().call("mi355seg_two_channel_gt_f32", _p(gt), _p(out), N, S, _stream())
    return out


def znormalize(x):
    """tio.ZNormalization() of one volume on the device: (x - mean) / std over all voxels (unbiased std)."""
    _require_cuda(x, "znormalize input")
    x = x.contiguous().to(torch.float32)
    L = lib()
    ws = workspace(L.query("mi355seg_znorm_ws_bytes", x.numel()), x.device)
    y = torch.empty_like(x)
    L.call("mi355seg_znorm_f32", _p(x), x.numel(), _p(y), _p(ws), ws.numel(), _stream())
    return y


def dice_sums(x, t, apply_sigmoid=False):
    """(sum a*b, sum a, sum b, sum a*a, sum b*b) as float64[5], a = sigmoid(x) if asked."""
    _require_cuda(x, "dice_sums input")
    x, t = x.contiguous(), t.contiguous().to(torch.float32)
    L = lib()
    ws = workspace(L.query("mi355seg_loss_ws_bytes", x.numel()), x.device)
    out = torch.empty(5, dtype=torch.float64, device=x.device)
    L.call("mi355seg_dice_sums_f32", _p(x), _p(t), x.numel(), int(bool(apply_sigmoid)), _p(out), _p(ws), ws.numel(), _stream())
    return out


class _DiceSums(Function):
    """S = (sum a*t, sum a, sum t, sum a*a, sum t*t) with autograd w.r.t. x (a = sigmoid(x) or x)."""

    @staticmethod
    def forward(ctx, x, t, apply_sigmoid):
        x, t = x.contiguous(), t.contiguous().to(torch.float32)
        out = dice_sums(x, t, apply_sigmoid)
        ctx.save_for_backward(x, t)
        ctx.apply_sigmoid = bool(apply_sigmoid)
        return out

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = g.contiguous().to(torch.float64)
        dx = torch.empty_like(x)
        lib().call("mi355seg_dice_sums_bwd_f32", _p(x), _p(t), _p(g), x.numel(), int(ctx.apply_sigmoid), _p(dx), _stream())
        return dx, None, None


def dice_sums_autograd(x, t, apply_sigmoid=False):
    return _DiceSums.apply(x, t, apply_sigmoid)


class _DiceRows(Function):
    """S[r] = (sum a*t, sum a, sum t, sum a^p, sum t^p) over row r of a [R, L] view, every row in one launch, with autograd
    w.r.t. x (a = sigmoid(x) or x)."""

    @staticmethod
    def forward(ctx, x, t, apply_sigmoid, p):
        _require_cuda(x, "dice_rows input")
        x, t = x.contiguous().to(torch.float32), t.contiguous().to(torch.float32)
        R, Ln = x.shape
        if tuple(t.shape) != (R, Ln):
            raise Mi355SegError(f"dice_rows: target {tuple(t.shape)} must match input {tuple(x.shape)}")
        L = lib()
        ws = workspace(L.query("mi355seg_dice_rows_ws_bytes", R, Ln), x.device)
        out = torch.empty((R, 5), dtype=torch.float64, device=x.device)
        L.call("mi355seg_dice_rows_f32", _p(x), _p(t), R, Ln, int(bool(apply_sigmoid)), float(p), _p(out), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(x, t)
        ctx.cfg = (R, Ln, int(bool(apply_sigmoid)), float(p))
        return out

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        R, Ln, sg, p = ctx.cfg
        g = g.contiguous().to(torch.float64)
        dx = torch.empty_like(x)
        lib().call("mi355seg_dice_rows_bwd_f32", _p(x), _p(t), _p(g), R, Ln, sg, float(p), _p(dx), _stream())
        return dx, None, None, None


def dice_rows_autograd(x, t, apply_sigmoid=False, p=2.0):
    """Per-row Dice sums of a [R, L] pair as float64 [R, 5] (one reduction launch + one finalise for all rows; more rows than
    a launch grid holds -- 65535 -- go in slices)."""
    R = x.shape[0]
    if R <= 65535:
        return _DiceRows.apply(x, t, apply_sigmoid, p)
    return torch.cat([_DiceRows.apply(x[i:i + 65535], t[i:i + 65535], apply_sigmoid, p) for i in range(0, R, 65535)], dim=0)


class _SoftmaxCh(Function):
    @staticmethod
    def forward(ctx, x):
        _require_cuda(x, "softmax input")
        x = x.contiguous()
        N, K = x.shape[0], x.shape[1]
        S = x[0, 0].numel()
        y = torch.empty_like(x)
        lib().call("mi355seg_softmax_ch_f32", _p(x), _p(y), N, K, S, _stream())
        ctx.save_for_backward(y)
        ctx.geom = (N, K, S)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        N, K, S = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        lib().call("mi355seg_softmax_ch_bwd_f32", _p(y), _p(dy), _p(dx), N, K, S, _stream())
        return dx


def softmax_channels(x):
    """torch.softmax(x, dim=1) for an NCDHW tensor."""
    return _SoftmaxCh.apply(x)


class _CE3D(Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, size_average):
        _require_cuda(logits, "cross_entropy_3D input")
        logits = logits.contiguous()
        labels = labels.contiguous().to(torch.int64)
        N, K = logits.shape[0], logits.shape[1]
        S = logits[0, 0].numel()
        if labels.numel() != N * S:
            raise ValueError(f"Expected target size {N * S}, got {labels.numel()}")
        if weight is not None:
            weight = weight.to(device=logits.device, dtype=torch.float32).contiguous()
        L = lib()
        ws = workspace(L.query("mi355seg_loss_ws_bytes", logits.numel()), logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        L.call("mi355seg_ce3d_fwd_f32", _p(logits), _p(labels), _p(weight), N, K, S, int(bool(size_average)), _p(loss),
               _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(logits, labels, weight)
        ctx.cfg = (N, K, S, int(bool(size_average)))
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, weight = ctx.saved_tensors
        N, K, S, sa = ctx.cfg
        g = g.contiguous().to(torch.float32)
        d = torch.empty_like(logits)
        lib().call("mi355seg_ce3d_bwd_f32", _p(logits), _p(labels), _p(weight), _p(g), N, K, S, sa, _p(d), _stream())
        return d, None, None, None


def cross_entropy_3d(logits, labels, weight=None, size_average=True):
    return _CE3D.apply(logits, labels, weight, size_average)


# ----------------------------------------------------------------------------- reverse-attention gate
class _Gate(Function):
    """enc * (2 - sigmoid(t)) with a one-channel ``t``: ``(1 - sigmoid(t)).expand(C).mul(enc) + enc`` of
    RE_net.py:104-107 / ER_net.py in one pass (and one backward pass with the per-voxel channel reduction for dt)."""

    @staticmethod
    def forward(ctx, enc, t):
        enc, lde = cl_view(enc, "gate features", allow_bf16=False)
        t, ldt = cl_view(t, "gate map", allow_bf16=False)
        N, D, H, W, C = enc.shape
        if tuple(t.shape) != (N, D, H, W, 1):
            raise Mi355SegError(f"gate: map {tuple(t.shape)} must be one channel over the voxels of {tuple(enc.shape)}")
        y = torch.empty((N, D, H, W, C), dtype=enc.dtype, device=enc.device)
        lib().call("mi355seg_gate_fwd_f32", _p(enc), lde, _p(t), ldt, _p(y), C, N * D * H * W, C, _stream())
        ctx.save_for_backward(enc, t)
        ctx.cfg = (lde, ldt, N * D * H * W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        enc, t = ctx.saved_tensors
        lde, ldt, rows, C = ctx.cfg
        dy, lddy = cl_view(dy, "gate grad")
        denc = torch.empty(enc.shape, dtype=enc.dtype, device=enc.device)
        dt = torch.empty(t.shape, dtype=t.dtype, device=t.device)
        lib().call("mi355seg_gate_bwd_f32", _p(dy), lddy, _p(enc), lde, _p(t), ldt, _p(denc), C, _p(dt), rows, C, _stream())
        return denc, dt


def reverse_attention_gate(enc, t):
    return _Gate.apply(enc, t)


# ----------------------------------------------------------------------------- selective fusion (ER_Net's SFConv)
class _SFPool(Function):
    """(x1 + x2).mean over the voxels -> [N, C]  (``fea_U.mean(-1).mean(-1).mean(-1)`` of ER_net.py:57-58)."""

    @staticmethod
    def forward(ctx, x1, x2):
        x1, ld1 = cl_view(x1, "sf_pool input", allow_bf16=False)
        x2, ld2 = cl_view(x2, "sf_pool input", allow_bf16=False)
        N, D, H, W, C = x1.shape
        if x2.shape != x1.shape:
            raise Mi355SegError(f"sf_pool: shapes differ: {tuple(x1.shape)} vs {tuple(x2.shape)}")
        V = D * H * W
        s = torch.empty((N, C), dtype=x1.dtype, device=x1.device)
        L = lib()
        ws = workspace(L.query("mi355seg_norm_ws_bytes", V, 1, C) + 8 * C + 512, x1.device)
        L.call("mi355seg_group_sums_f32", _p(x1), ld1, V, N, C, 1.0 / V, _p(s), 0, _p(ws), ws.numel(), _stream())
        L.call("mi355seg_group_sums_f32", _p(x2), ld2, V, N, C, 1.0 / V, _p(s), 1, _p(ws), ws.numel(), _stream())
        ctx.shape = (N, D, H, W, C)
        return s

    @staticmethod
    def backward(ctx, gs):
        N, D, H, W, C = ctx.shape
        V = D * H * W
        g = torch.empty((N, D, H, W, C), dtype=gs.dtype, device=gs.device)
        lib().call("mi355seg_broadcast_channels_f32", _p(gs.contiguous()), 1.0 / V, _p(g), C, V, N, C, _stream())
        return g, g


class _SFMix(Function):
    """x1 * a[n, c] + x2 * b[n, c]  (``(feas * attention_vectors).sum(dim=1)`` of ER_net.py:67-69)."""

    @staticmethod
    def forward(ctx, x1, x2, a, b):
        x1, ld1 = cl_view(x1, "sf_mix input", allow_bf16=False)
        x2, ld2 = cl_view(x2, "sf_mix input", allow_bf16=False)
        N, D, H, W, C = x1.shape
        a, b = a.contiguous(), b.contiguous()
        if x2.shape != x1.shape or a.numel() != N * C or b.numel() != N * C:
            raise Mi355SegError("sf_mix: x1 / x2 must match and the two attention vectors hold N*C values each")
        y = torch.empty((N, D, H, W, C), dtype=x1.dtype, device=x1.device)
        lib().call("mi355seg_mix_channels_f32", _p(x1), ld1, _p(a), _p(x2), ld2, _p(b), _p(y), C, D * H * W, N, C, _stream())
        ctx.save_for_backward(x1, x2, a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, a, b = ctx.saved_tensors
        N, D, H, W, C = x1.shape
        V = D * H * W
        dy, lddy = cl_view(dy, "sf_mix grad")
        L = lib()
        dx1, dx2 = torch.empty_like(dy.contiguous()), None
        dx2 = torch.empty_like(dx1)
        L.call("mi355seg_scale_channels_f32", _p(dy), lddy, _p(a), _p(dx1), C, V, N, C, _stream())
        L.call("mi355seg_scale_channels_f32", _p(dy), lddy, _p(b), _p(dx2), C, V, N, C, _stream())
        ws = workspace(L.query("mi355seg_norm_ws_bytes", V, 1, C) + 8 * C + 512, dy.device)
        dyc = dy.contiguous()
        grads = []
        for x in (x1, x2):                                   # da[n, c] = sum_v dy * x
            prod = _mul(dyc, x.contiguous())
            d = torch.empty((N, C), dtype=dy.dtype, device=dy.device)
            L.call("mi355seg_group_sums_f32", _p(prod), C, V, N, C, 1.0, _p(d), 0, _p(ws), ws.numel(), _stream())
            grads.append(d.view_as(a))
        return dx1, dx2, grads[0], grads[1]


class _SoftmaxLast(Function):
    """softmax over the last dimension of a small 2-D tensor (the two-branch attention softmax, ER_net.py:66)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x, "softmax input")
        x = x.contiguous()
        y = torch.empty_like(x)
        lib().call("mi355seg_softmax_rows_f32", _p(x), _p(y), x.shape[0], x.shape[1], _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        lib().call("mi355seg_softmax_rows_bwd_f32", _p(y), _p(dy.contiguous()), _p(dx), y.shape[0], y.shape[1], _stream())
        return dx


def sf_pool(x1, x2):
    return _SFPool.apply(x1, x2)


def sf_mix(x1, x2, a, b):
    return _SFMix.apply(x1, x2, a, b)


def softmax_last(x):
    return _SoftmaxLast.apply(x)


# ----------------------------------------------------------------------------- channel concat / repeat
class _CatChannels(Function):
    """torch.cat((a, b), dim=1) of the reference (vnet3d.py:101, residual_unet3d.py:183-209, unetr.py:286-293) in
    channel-last form: two strided slice copies into one buffer; the backward hands out the two slices as views."""

    @staticmethod
    def forward(ctx, a, b):
        a, lda = cl_view(a, "cat_channels first input")
        b, ldb = cl_view(b, "cat_channels second input")
        if a.shape[:4] != b.shape[:4]:
            raise Mi355SegError(f"cat_channels: spatial shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        N, D, H, W, Ca = a.shape
        Cb = b.shape[4]
        out = torch.empty((N, D, H, W, Ca + Cb), dtype=a.dtype, device=a.device)
        rows = N * D * H * W
        L = lib()
        if a.dtype != b.dtype:
            raise Mi355SegError(f"cat_channels: storage types differ: {a.dtype} vs {b.dtype}")
        L.call("mi355seg_copy_rows_" + _sfx(a), _p(a), lda, _p(out), Ca + Cb, rows, Ca, _stream())
        L.call("mi355seg_copy_rows_" + _sfx(a), _p(b), ldb, out.data_ptr() + out.element_size() * Ca, Ca + Cb, rows, Cb, _stream())
        ctx.ca = Ca
        return out

    @staticmethod
    def backward(ctx, dcat):
        return dcat[..., :ctx.ca], dcat[..., ctx.ca:]


def cat_channels(a, b):
    return _CatChannels.apply(a, b)


class _BnActCatScaled(Function):
    """cat((act(BatchNorm3d(y)), skip * scale), channels) -- V-Net's up-transition head (vnet3d.py:97-101: relu1(bn1(up_conv(x))),
    do2(skipx), torch.cat) -- as one node: the normalise pass writes the left channel slice of the concat buffer and the dropout
    scaling the right one (no separate Dropout3d tensor, no concat copies); the backward reads the two slices of the incoming
    gradient in place.  ``scale`` None: the skip is copied (eval mode / p = 0)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, rmean, rvar, skip, scale, training, momentum, eps, act, slope):
        y, ldy = cl_view(y, "norm input")
        skip, lds = cl_view(skip, "concat skip input")
        if y.shape[:4] != skip.shape[:4] or y.dtype != skip.dtype:
            raise Mi355SegError(f"up-transition concat: {tuple(y.shape)} {y.dtype} vs {tuple(skip.shape)} {skip.dtype}")
        N, D, H, W, Cu = y.shape
        Cs = skip.shape[4]
        rows = N * D * H * W
        L = lib()
        dev = y.device
        ws = workspace(L.query("mi355seg_norm_ws_bytes", rows, 1, Cu), dev)
        if training:
            mean = torch.empty(Cu, dtype=torch.float32, device=dev)
            rstd = torch.empty(Cu, dtype=torch.float32, device=dev)
            L.call("mi355seg_norm_stats_" + _sfx(y), _p(y), ldy, rows, 1, Cu, eps, _p(mean), _p(rstd), _p(rmean), _p(rvar), momentum,
                   _p(ws), ws.numel(), _stream())
        else:
            mean = rmean
            rstd = torch.empty(Cu, dtype=torch.float32, device=dev)
            L.call("mi355seg_rstd_from_var_f32", _p(rvar), eps, _p(rstd), Cu, _stream())
        both = torch.empty((N, D, H, W, Cu + Cs), dtype=y.dtype, device=dev)
        right = both.data_ptr() + both.element_size() * Cu
        L.call("mi355seg_norm_act_fwd_" + _sfx(y), _p(y), ldy, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
               both.data_ptr(), Cu + Cs, rows, 1, Cu, act, slope, _stream())
        if scale is not None:
            scale = scale.contiguous().to(torch.float32)
            L.call("mi355seg_scale_channels_" + _sfx(y), _p(skip), lds, _p(scale), right, Cu + Cs, D * H * W, N, Cs, _stream())
        else:
            L.call("mi355seg_copy_rows_" + _sfx(y), _p(skip), lds, right, Cu + Cs, rows, Cs, _stream())
        ctx.save_for_backward(y, mean, rstd, gamma, beta, scale)
        ctx.cfg = (ldy, rows, Cu, Cs, act, slope, bool(training), (N, D, H, W))
        return both

    @staticmethod
    def backward(ctx, dboth):
        y, mean, rstd, gamma, beta, scale = ctx.saved_tensors
        ldy, rows, Cu, Cs, act, slope, training, (N, D, H, W) = ctx.cfg
        if not training:
            raise Mi355SegError("backward through eval-mode BatchNorm (running statistics) is not supported")
        dboth, ldd = cl_view(_like(dboth, y), "up-transition concat grad")
        L = lib()
        dev = y.device
        ws = workspace(L.query("mi355seg_norm_ws_bytes", rows, 1, Cu), dev)
        dy = torch.empty(y.shape, dtype=y.dtype, device=dev)
        dgamma = torch.empty(Cu, dtype=torch.float32, device=dev)
        dbeta = torch.empty(Cu, dtype=torch.float32, device=dev)
        L.call("mi355seg_norm_act_bwd_" + _sfx(y), _p(dboth), ldd, _p(y), ldy, _p(mean), _p(rstd), _p(gamma), _p(beta), None, 0,
               _p(dy), Cu, _p(dgamma), _p(dbeta), None, 0, rows, 1, Cu, act, slope, _p(ws), ws.numel(), _stream())
        right = dboth.data_ptr() + dboth.element_size() * Cu
        if scale is not None:
            dskip = torch.empty((N, D, H, W, Cs), dtype=y.dtype, device=dev)
            L.call("mi355seg_scale_channels_" + _sfx(y), right, ldd, _p(scale), _p(dskip), Cs, D * H * W, N, Cs, _stream())
        else:
            dskip = dboth[..., Cu:]
        return dy, dgamma, dbeta, None, None, dskip, None, None, None, None, None, None


def bn_act_cat_scaled(y, bn, skip, scale, act=ACT_NONE, slope=0.01):
    """cat((act(bn(y)), skip * scale), channel axis) for a layers.BatchNorm3d; ``scale`` = Dropout3d.draw_scale(...) or None."""
    if bn.momentum is None or not bn.affine or not bn.track_running_stats:
        raise NotImplementedError("bn_act_cat_scaled: BatchNorm3d must be affine with running statistics and a momentum")
    if bn.training:
        bump_counter(bn)
    return _BnActCatScaled.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, skip, scale, bool(bn.training),
                                 float(bn.momentum), float(bn.eps), int(act), float(slope))


class _RepeatChannels(Function):
    """x.repeat(1, rep, 1, 1, 1) (vnet3d.py:55-56) in channel-last form."""

    @staticmethod
    def forward(ctx, x, rep):
        x, ldx = cl_view(x, "repeat_channels input")
        N, D, H, W, C = x.shape
        y = torch.empty((N, D, H, W, C * rep), dtype=x.dtype, device=x.device)
        lib().call("mi355seg_repeat_channels_" + _sfx(x), _p(x), ldx, _p(y), C * rep, N * D * H * W, C, rep, _stream())
        ctx.cfg = (C, rep)
        return y

    @staticmethod
    def backward(ctx, dy):
        C, rep = ctx.cfg
        dy, lddy = cl_view(dy, "repeat_channels grad")
        N, D, H, W, _ = dy.shape
        dx = torch.empty((N, D, H, W, C), dtype=dy.dtype, device=dy.device)
        lib().call("mi355seg_repeat_channels_bwd_" + _sfx(dy), _p(dy), lddy, _p(dx), C, N * D * H * W, C, rep, _stream())
        return dx, None


def repeat_channels(x, rep):
    return _RepeatChannels.apply(x, int(rep))


def _mul(a, b, out=None):
    """Elementwise product of two equally shaped contiguous fp32 tensors (no autograd; used inside Functions)."""
    a, b = a.contiguous(), b.contiguous()
    if a.shape != b.shape:
        raise Mi355SegError(f"mul: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    if out is None:
        out = torch.empty_like(a)
    lib().call("mi355seg_mul_f32", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def _softmax_keep(scores, keep, rows, Lr):
    """(softmax(scores), softmax(scores) * keep) over the last dim -- one launch; keep None: the second is the first."""
    probs = torch.empty_like(scores)
    if keep is None:
        lib().call("mi355seg_softmax_rows_f32", _p(scores), _p(probs), rows, Lr, _stream())
        return probs, probs
    keep = keep.contiguous()
    if keep.shape != scores.shape:
        raise Mi355SegError(f"attention: dropout mask {tuple(keep.shape)} does not match the scores {tuple(scores.shape)}")
    pd = torch.empty_like(scores)
    lib().call("mi355seg_softmax_rows_keep_f32", _p(scores), _p(keep), _p(probs), _p(pd), rows, Lr, _stream())
    return probs, pd


def _softmax_keep_bwd(probs, dpd, keep, rows, Lr):
    ds = torch.empty_like(probs)
    if keep is None:
        lib().call("mi355seg_softmax_rows_bwd_f32", _p(probs), _p(dpd), _p(ds), rows, Lr, _stream())
    else:
        lib().call("mi355seg_softmax_rows_keep_bwd_f32", _p(probs), _p(dpd), _p(keep.contiguous()), _p(ds), rows, Lr, _stream())
    return ds


# ---- element-wise dropout masks of one training step from ONE draw (r5).  nn.Dropout layers on token tensors (unetr.py:70-71,93,100,
# 124,133,149) each drew their keep / (1 - p) mask with a launch of their own: 37 per UNETR step.  Between two `dropout_pool_begin_step()`
# calls (engine.train_step makes them) the layers' requests are tallied; from the next step on one launch draws that many values and the
# layers take consecutive slices.  A request the pool cannot serve (no begin_step caller, a first step, a changed shape) draws by itself.
class _MaskPool(threading.local):
    def __init__(self):
        self.pools = {}                      # (p, device) -> [buffer or None, offset, values taken since the last begin_step]


_MASKS = _MaskPool()
# (values, device) -> the tensor of ones the pooled draw reads.  Entries are never dropped: a captured GraphedTrainStep has the buffer's
# address baked into its dropout node, so freeing it on a later request of another size (another model or patch shape in the same process)
# would let replays draw their masks from recycled memory.  One float per mask value and distinct size: a few MB per model.
_POOL_ONES = {}


def dropout_pool_begin_step():
    for (p, dev), st in _MASKS.pools.items():
        need = st[2]
        st[0], st[1], st[2] = None, 0, 0
        if need > 0 and not os.environ.get("MI355SEG_NO_MASK_POOL"):
            ones = _POOL_ONES.get((need, dev))
            if ones is None:
                ones = _POOL_ONES.setdefault((need, dev), torch.ones(need, device=dev))
            st[0] = torch.nn.functional.dropout(ones, p, True)       # keep / (1 - p), one launch for the whole step


def dropout_pool_take(shape, p, device, fallback):
    """A keep / (1 - p) mask of ``shape``: a slice of the step's pooled draw, else ``fallback()`` (an individual draw)."""
    n = 1
    for e in shape:
        n *= int(e)
    st = _MASKS.pools.setdefault((float(p), str(device)), [None, 0, 0])
    st[2] += n
    if st[0] is not None and st[1] + n <= st[0].numel() and n % 4 == 0:
        m = st[0][st[1]:st[1] + n].view(*shape)
        st[1] += n
        return m
    return fallback()


# ----------------------------------------------------------------------------- UNETR encoder ops
def _lowp_now():
    """Inside autocast(torch.bfloat16) the token path's GEMMs take bf16 products (fp32 accumulate), as the reference's nn.Linear / matmul do
    under torch.autocast; a Function records the mode at its forward and keeps it for its backward."""
    return _AUTOCAST.stack[-1] == torch.bfloat16 and not os.environ.get("MI355SEG_TOKEN_GEMM_FP32")       # (the variable: fp32 products, for A/B timing)


def _gemm(A, a_rs, a_cs, a_b0, a_b1, B, b_rs, b_cs, b_b0, b_b1, C, c_rs, c_b0, c_b1, bias, M, N, K, nb0=1, nb1=1,
          alpha=1.0, relu=0, accumulate=0, lowp=False):
    L = lib()
    need = L.query("mi355seg_gemm_ws_bytes", M, N, K, nb0, nb1)
    ws = workspace(need, torch.device("cuda", torch.cuda.current_device())) if need else None
    L.call("mi355seg_gemm_lowp_f32" if lowp else "mi355seg_gemm_f32", A, a_rs, a_cs, a_b0, a_b1, B, b_rs, b_cs, b_b0, b_b1, C, c_rs, c_b0, c_b1, bias,
           M, N, K, nb0, nb1, alpha, relu, accumulate, _p(ws), ws.numel() if ws is not None else 0, _stream())


class _Linear(Function):
    """y[M,N] = x[M,K] @ W[N,K]^T + b, optionally followed -- inside the GEMM's epilogue -- by ReLU, an element-wise factor ``mask``
    (a dropout layer's keep / (1 - p)) and the sum with ``residual``: y = relu?(x W^T + b) * mask + residual (r5,
    mi355seg_linear_fwd_f32: the token encoder's out-projection + dropout + residual and its two feed-forward layers, unetr.py:98-100,
    120-138,159-166, as one launch each)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, mask=None, residual=None):
        lowp = ctx.lowp = _lowp_now()
        _require_cuda(x, "linear input")
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        w = w.contiguous()
        M, K, N = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty((M, N), dtype=x.dtype, device=x.device)
        if relu and residual is not None:
            raise Mi355SegError("linear: ReLU and a residual sum in one call are not supported (the backward keys the ReLU on the output)")
        if mask is None and residual is None:
            _gemm(_p(x2), K, 1, 0, 0, _p(w), 1, K, 0, 0, _p(y), N, 0, 0, _p(b), M, N, K, relu=int(relu), lowp=lowp)
        else:
            if mask is not None:
                mask = mask.contiguous()
                if mask.numel() != M * N or mask.dtype != x.dtype:
                    raise Mi355SegError(f"linear: mask of {mask.numel()} {mask.dtype} values for an output of {M} x {N} {x.dtype}")
            if residual is not None:
                residual = residual.contiguous()
                if residual.numel() != M * N or residual.dtype != x.dtype:
                    raise Mi355SegError(f"linear: residual of {residual.numel()} {residual.dtype} values for an output of {M} x {N} {x.dtype}")
            L = lib()
            need = L.query("mi355seg_gemm_ws_bytes", M, N, K, 1, 1)
            ws = workspace(need, x.device) if need else None
            L.call("mi355seg_linear_fwd_f32", int(lowp), _p(x2), K, _p(w), _p(b), int(relu), _p(mask), _p(residual), _p(y), M, N, K,
                   _p(ws), ws.numel() if ws is not None else 0, _stream())
        ctx.save_for_backward(x2, w, y if relu else None, mask)
        ctx.cfg = (shp, M, N, K, bool(relu), b is not None, residual is not None)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        lowp = ctx.lowp
        x2, w, yrelu, mask = ctx.saved_tensors
        shp, M, N, K, relu, has_b, has_res = ctx.cfg
        dy2 = dy.contiguous().view(M, N)
        dres = dy if has_res else None                       # d(... + residual) / d residual = dy itself: no kernel
        L = lib()
        # r6: the whole backward as ONE launch -- dx = dyf W and dw = dyf^T x (+ db) are independent, latency-bound GEMMs at the token
        # encoder's sizes and run as two problems of one grid; the dropout factors and the ReLU gate (dyf = dy * mask * [y > 0]) are folded
        # into their loads of dy (mi355seg_linear_bwd_f32; bit-identical to the separate launches)
        if lowp and not os.environ.get("MI355SEG_NO_GEMM_PAIRS") and L.query("mi355seg_linear_bwd_supported_f32", 1, M, N, K) != 0 and \
                all(t is None or (t.is_contiguous() and t.data_ptr() % 16 == 0) for t in (dy2, mask, yrelu if relu else None, x2, w)):
            dx = torch.empty((M, K), dtype=dy2.dtype, device=dy2.device)
            dw = torch.empty_like(w)
            db = torch.empty(N, dtype=dy2.dtype, device=dy2.device) if has_b else None
            L.call("mi355seg_linear_bwd_f32", 1, _p(dy2), N, _p(mask), _p(yrelu) if relu else None, _p(x2), K, _p(w), _p(dx), _p(dw), _p(db), M, N, K, _stream())
            return dx.view(*shp), dw, db, None, None, dres
        if mask is not None:
            dy2 = _mul(dy2, mask.view(M, N))
        if relu:                                            # dy * 1[y > 0]: the ReLU backward kernel, keyed on the saved output (mask 0 => y 0 and dy 0)
            g = torch.empty_like(dy2)
            lib().call("mi355seg_act_bwd_f32", _p(dy2), N, _p(yrelu), N, None, 0, _p(g), N, M, N, ACT_RELU, 0.0, _stream())
            dy2 = g
        dx = torch.empty((M, K), dtype=dy2.dtype, device=dy2.device)
        _gemm(_p(dy2), N, 1, 0, 0, _p(w), K, 1, 0, 0, _p(dx), K, 0, 0, None, M, K, N, lowp=lowp)
        # weight gradient; the bias gradient (column sums of dy) rides in the same launch on the token encoder's shapes
        dw = torch.empty_like(w)
        db = torch.empty(N, dtype=dy2.dtype, device=dy2.device) if has_b else None
        ws = workspace(max(L.query("mi355seg_gemm_ws_bytes", N, K, M, 1, 1), L.query("mi355seg_norm_ws_bytes", M, 1, N)), dy2.device)
        L.call("mi355seg_linear_wgrad_f32", int(lowp), _p(dy2), N, _p(x2), K, _p(dw), _p(db), M, N, K, _p(ws), ws.numel(), _stream())
        return dx.view(*shp), dw, db, None, None, dres


def linear(x, weight, bias=None, relu=False, mask=None, residual=None):
    """nn.Linear; ``mask`` / ``residual``: see _Linear (y = relu?(x W^T + b) * mask + residual in one launch)."""
    return _Linear.apply(x, weight, bias, relu, mask, residual)


def qkv_params_are_fused(wq, wk, wv, bq, bk, bv):
    """The three projections' parameters are consecutive slices of ONE buffer (models.three_d.unetr.SelfAttention seats them so): the
    [3E, K] weight and the [3E] bias of the fused projection exist in place."""
    if bq is None or bk is None or bv is None or wq.dim() != 2 or wq.shape != wk.shape or wq.shape != wv.shape:
        return False
    E, K = wq.shape
    ts = (wq, wk, wv, bq, bk, bv)
    if not all(t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda for t in ts) or bq.numel() != E or bk.numel() != E or bv.numel() != E:
        return False
    if wk.data_ptr() != wq.data_ptr() + 4 * E * K or wv.data_ptr() != wq.data_ptr() + 8 * E * K:
        return False
    if bk.data_ptr() != bq.data_ptr() + 4 * E or bv.data_ptr() != bq.data_ptr() + 8 * E:
        return False
    room = lambda t, n: t.untyped_storage().nbytes() >= 4 * (t.storage_offset() + n)
    return room(wq, 3 * E * K) and room(bq, 3 * E)


class _LinearQKV(Function):
    """The query / key / value projections of unetr.py:60-75 as ONE GEMM on parameters that are slices of one buffer
    (qkv_params_are_fused): no concatenation of the weights in the forward, and the backward hands each parameter its rows of the fused
    weight / bias gradient as views (r6; the concatenated form cost two copy launches per layer and step)."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, bq, bk, bv):
        E, K = wq.shape
        wf = torch.as_strided(wq.detach(), (3 * E, K), (K, 1))
        bf = torch.as_strided(bq.detach(), (3 * E,), (1,))
        ctx.E = E
        return _Linear.forward(ctx, x, wf, bf, False)

    @staticmethod
    def backward(ctx, dy):
        dx, dw, db = _Linear.backward(ctx, dy)[:3]
        E = ctx.E
        return dx, dw[:E], dw[E:2 * E], dw[2 * E:], db[:E], db[E:2 * E], db[2 * E:]


def linear_qkv(x, wq, wk, wv, bq, bk, bv):
    """[x Wq^T + bq | x Wk^T + bk | x Wv^T + bv] for parameters seated as slices of one buffer; see _LinearQKV."""
    return _LinearQKV.apply(x, wq, wk, wv, bq, bk, bv)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _require_cuda(x, "layer_norm input")
        shp = x.shape
        E = shp[-1]
        x2 = x.contiguous().view(-1, E)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        lib().call("mi355seg_layernorm_fwd_f32", _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, E, eps, _stream())
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shp = shp
        return y.view(*shp)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        rows, E = x2.shape
        dy2 = dy.contiguous().view(rows, E)
        dx = torch.empty_like(x2)
        dg = torch.empty(E, dtype=torch.float32, device=x2.device)
        db = torch.empty(E, dtype=torch.float32, device=x2.device)
        lib().call("mi355seg_layernorm_bwd_f32", _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows, E, _stream())
        return dx.view(*ctx.shp), dg, db, None


class _LayerNormFork(Function):
    """(LayerNorm(x), x): the norm of a tensor that also continues unchanged -- a pre-norm transformer block's x + f(LN(x)) (unetr.py:159-166).
    The backward writes d(x) = d(pass-through) + LN-backward(d(norm)) from the kernel that forms the LayerNorm gradient
    (mi355seg_layernorm_bwd_add_f32); autograd would add the two with a launch of its own."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _require_cuda(x, "layer_norm input")
        shp = x.shape
        E = shp[-1]
        x2 = x.contiguous().view(-1, E)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        lib().call("mi355seg_layernorm_fwd_f32", _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, E, eps, _stream())
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shp = shp
        ctx.set_materialize_grads(False)
        return y.view(*shp), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dpass):
        x2, gamma, mean, rstd = ctx.saved_tensors
        rows, E = x2.shape
        if dy is None:
            return dpass, None, None, None
        dy2 = dy.contiguous().view(rows, E)
        add = dpass.contiguous().view(rows, E) if dpass is not None else None
        dx = torch.empty_like(x2)
        dg = torch.empty(E, dtype=torch.float32, device=x2.device)
        db = torch.empty(E, dtype=torch.float32, device=x2.device)
        lib().call("mi355seg_layernorm_bwd_add_f32", _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(dg), _p(db), rows, E, _stream())
        return dx.view(*ctx.shp), dg, db, None


def layer_norm_fork(x, gamma, beta, eps=1e-5):
    """(LayerNorm(x), x) for a tensor that is normalised and also continues unchanged; see _LayerNormFork."""
    return _LayerNormFork.apply(x, gamma, beta, float(eps))


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, float(eps))


class _Attention(Function):
    """softmax(Q K^T / sqrt(d)) V per (batch, head) with Q,K,V stored [B, P, H*d] (unetr.py:74-98).
    ``keep`` (optional, [B,H,P,P], already divided by 1-p) is the attention-dropout mask."""

    @staticmethod
    def forward(ctx, q, k, v, heads, keep):
        lowp = ctx.lowp = _lowp_now()
        _require_cuda(q, "attention input")
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, P, E = q.shape
        d = E // heads
        alpha = 1.0 / (d ** 0.5)
        scores = torch.empty((B, heads, P, P), dtype=q.dtype, device=q.device)
        _gemm(_p(q), E, 1, P * E, d, _p(k), 1, E, P * E, d, _p(scores), P, heads * P * P, P * P, None, P, P, d, B, heads, alpha, lowp=lowp)
        probs, pd = _softmax_keep(scores, keep, B * heads * P, P)
        ctxl = torch.empty((B, P, E), dtype=q.dtype, device=q.device)
        _gemm(_p(pd), P, 1, heads * P * P, P * P, _p(v), E, 1, P * E, d, _p(ctxl), E, P * E, d, None, P, d, P, B, heads, lowp=lowp)
        ctx.save_for_backward(q, k, v, probs, keep, pd if keep is not None else None)
        ctx.cfg = (B, P, E, heads, d, alpha)
        return ctxl

    @staticmethod
    def backward(ctx, do):
        lowp = ctx.lowp
        q, k, v, probs, keep, pd = ctx.saved_tensors
        B, P, E, heads, d, alpha = ctx.cfg
        do = do.contiguous()
        if pd is None:
            pd = probs
        HPP, PP = heads * P * P, P * P
        dpd = torch.empty_like(probs)                                   # dP = dO V^T
        _gemm(_p(do), E, 1, P * E, d, _p(v), 1, E, P * E, d, _p(dpd), P, HPP, PP, None, P, P, d, B, heads, lowp=lowp)
        dv = torch.empty_like(v)                                        # dV = Pd^T dO
        _gemm(_p(pd), 1, P, HPP, PP, _p(do), E, 1, P * E, d, _p(dv), E, P * E, d, None, P, d, P, B, heads, lowp=lowp)
        ds = _softmax_keep_bwd(probs, dpd, keep, B * heads * P, P)
        dq = torch.empty_like(q)                                        # dQ = alpha dS K
        _gemm(_p(ds), P, 1, HPP, PP, _p(k), E, 1, P * E, d, _p(dq), E, P * E, d, None, P, d, P, B, heads, alpha, lowp=lowp)
        dk = torch.empty_like(k)                                        # dK = alpha dS^T Q
        _gemm(_p(ds), 1, P, HPP, PP, _p(q), E, 1, P * E, d, _p(dk), E, P * E, d, None, P, d, P, B, heads, alpha, lowp=lowp)
        return dq, dk, dv, None, None


class _AttentionQKV(Function):
    """_Attention on a FUSED projection output qkv[B, P, 3E] = (Q | K | V) (one GEMM for the three projections of unetr.py:60-75):
    the batched GEMMs read the three channel slices in place (row pitch 3E) and the backward writes dQ | dK | dV into the slices of
    one [B, P, 3E] gradient, so the projection's backward is one input-gradient GEMM, one weight-gradient GEMM and one bias sum."""

    @staticmethod
    def forward(ctx, qkv, heads, keep):
        lowp = ctx.lowp = _lowp_now()
        _require_cuda(qkv, "attention input")
        qkv = qkv.contiguous()
        B, P, E3 = qkv.shape
        E = E3 // 3
        d = E // heads
        alpha = 1.0 / (d ** 0.5)
        q, k, v = _p(qkv), _p(qkv) + 4 * E, _p(qkv) + 8 * E
        scores = torch.empty((B, heads, P, P), dtype=qkv.dtype, device=qkv.device)
        _gemm(q, E3, 1, P * E3, d, k, 1, E3, P * E3, d, _p(scores), P, heads * P * P, P * P, None, P, P, d, B, heads, alpha, lowp=lowp)
        probs, pd = _softmax_keep(scores, keep, B * heads * P, P)
        ctxl = torch.empty((B, P, E), dtype=qkv.dtype, device=qkv.device)
        _gemm(_p(pd), P, 1, heads * P * P, P * P, v, E3, 1, P * E3, d, _p(ctxl), E, P * E, d, None, P, d, P, B, heads, lowp=lowp)
        ctx.save_for_backward(qkv, probs, keep, pd if keep is not None else None)
        ctx.cfg = (B, P, E, heads, d, alpha)
        return ctxl

    @staticmethod
    def backward(ctx, do):
        lowp = ctx.lowp
        qkv, probs, keep, pd = ctx.saved_tensors
        B, P, E, heads, d, alpha = ctx.cfg
        E3 = 3 * E
        q, k, v = _p(qkv), _p(qkv) + 4 * E, _p(qkv) + 8 * E
        do = do.contiguous()
        if pd is None:
            pd = probs
        HPP, PP = heads * P * P, P * P
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = _p(dqkv), _p(dqkv) + 4 * E, _p(dqkv) + 8 * E
        dpd = torch.empty_like(probs)                                   # dP = dO V^T
        L = lib()
        # r6: the two pairs of independent GEMMs (dP with dV: both read dO; dQ with dK: both read dS) as one launch each
        pairs = lowp and not os.environ.get("MI355SEG_NO_GEMM_PAIRS") and \
            L.query("mi355seg_gemm_pair_supported_f32", P, P, d, P, d, P, B, heads) != 0 and L.query("mi355seg_gemm_pair_supported_f32", P, d, P, P, d, P, B, heads) != 0
        if pairs:
            L.call("mi355seg_gemm_pair_lowp_f32", _p(do), E, 1, P * E, d, v, 1, E3, P * E3, d, _p(dpd), P, HPP, PP, P, P, d, 1.0,
                   _p(pd), 1, P, HPP, PP, _p(do), E, 1, P * E, d, dv, E3, P * E3, d, P, d, P, 1.0, B, heads, _stream())
        else:
            _gemm(_p(do), E, 1, P * E, d, v, 1, E3, P * E3, d, _p(dpd), P, HPP, PP, None, P, P, d, B, heads, lowp=lowp)
            _gemm(_p(pd), 1, P, HPP, PP, _p(do), E, 1, P * E, d, dv, E3, P * E3, d, None, P, d, P, B, heads, lowp=lowp)         # dV = Pd^T dO
        ds = _softmax_keep_bwd(probs, dpd, keep, B * heads * P, P)
        if pairs:
            L.call("mi355seg_gemm_pair_lowp_f32", _p(ds), P, 1, HPP, PP, k, E3, 1, P * E3, d, dq, E3, P * E3, d, P, d, P, alpha,
                   _p(ds), 1, P, HPP, PP, q, E3, 1, P * E3, d, dk, E3, P * E3, d, P, d, P, alpha, B, heads, _stream())
        else:
            _gemm(_p(ds), P, 1, HPP, PP, k, E3, 1, P * E3, d, dq, E3, P * E3, d, None, P, d, P, B, heads, alpha, lowp=lowp)      # dQ = alpha dS K
            _gemm(_p(ds), 1, P, HPP, PP, q, E3, 1, P * E3, d, dk, E3, P * E3, d, None, P, d, P, B, heads, alpha, lowp=lowp)      # dK = alpha dS^T Q
        return dqkv, None, None


def attention_qkv(qkv, heads, keep=None):
    """Multi-head attention on the fused projection output [B, P, 3E] (fp32)."""
    return _AttentionQKV.apply(qkv, int(heads), keep)


def attention(q, k, v, heads, keep=None):
    return _Attention.apply(q, k, v, int(heads), keep)
