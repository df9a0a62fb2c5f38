"""Train-step engine: the reference's per-iteration glue (train.py:187-221) on the GPU.

zero_grad -> 2-channel gt -> forward -> argmax -> BCE -> backward -> optimizer.step ->
Dice, with the Dice counters reduced on the device (four int64 cross PCIe instead of the
reference's two full int64 volumes, train.py:221).  Optimizer stays torch.optim (north_star:
"autograd/optimizer plumbing stays Python")."""
import torch

from . import functional as F
from .utils.metric import metric_from_counts


def two_channel_gt(gt):
    """train.py:190-193: gt_back = (gt == 0); gt = cat([gt_back, gt], dim=1) as float (one kernel, no torch.cat)."""
    return F.two_channel_gt(gt)


def mixed_precision_dtype(config=None):
    """The activation storage type of a run: ``Accelerator()`` (train.py:167) takes its mixed-precision mode from
    ``accelerate config`` / the ACCELERATE_MIXED_PRECISION environment variable ("no" | "bf16"); the same variable, or
    ``config.mixed_precision`` on the command line, selects it here.  "bf16" = bf16 activations in HBM + bf16 MFMA
    convolutions (fp32 parameters, gradients, statistics and loss); fp16 is not offered (no loss scaler on this path)."""
    import os
    mode = None
    if config is not None:
        mode = config.get("mixed_precision") if hasattr(config, "get") else getattr(config, "mixed_precision", None)
    mode = str(mode if mode is not None else os.environ.get("ACCELERATE_MIXED_PRECISION", "no")).lower()
    if mode in ("no", "none", "false", "fp32", "f32", ""):
        return torch.float32
    if mode in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError(f"mixed_precision must be 'no' or 'bf16', got {mode!r}")


def make_adam(params, lr, **kw):
    """The optimizer of train.py:109 (``torch.optim.Adam(model.parameters(), lr=config.init_lr)``) as PyTorch's single-kernel
    implementation when every parameter lives on the GPU (``fused=True``: the same update rule, one multi-tensor launch
    instead of the five foreach passes whose host-side dispatch alone costs 1.2 ms per step on the U-Net and 5-7 ms on UNETR's
    few hundred parameter tensors); plain ``torch.optim.Adam`` otherwise."""
    params = list(params)
    if "fused" not in kw and "foreach" not in kw and params and all(p.is_cuda and p.is_floating_point() for p in params):
        try:
            return torch.optim.Adam(params, lr=lr, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):        # a build without the fused kernel: the default implementation
            pass
    return torch.optim.Adam(params, lr=lr, **kw)


def _forward(model, leaves, *args):
    """``model(*args)``, or the same forward on ``leaves`` (name -> tensor aliasing the parameter's storage) in place of the
    module's parameters (torch.func.functional_call): GraphedTrainStep's capture, see there."""
    if leaves is None:
        return model(*args)
    return torch.func.functional_call(model, leaves, args)


def train_step(model, optimizer, x, gt, criterion=None, sync_metric=True, grad_hook=None, dtype=None, step_optimizer=True, _leaves=None):
    """One iteration.  ``gt`` is the single-channel label volume [N,1,D,H,W].  ``dtype`` = torch.bfloat16 runs the
    forward under mi355seg.autocast (bf16 activations; the loss and everything after it stay fp32).
    ``grad_hook`` (if given) runs between backward and optimizer.step -- the data-parallel
    gradient all-reduce plugs in there.  Returns a dict with pred, mask, loss and, when
    ``sync_metric``, python floats jaccard/dice (one tiny D2H copy); otherwise the raw
    int64[4] counters stay on the device under 'counts'."""
    optimizer.zero_grad(set_to_none=True)
    gt2 = two_channel_gt(gt)
    x = x.to(torch.float32)
    # every training-mode BatchNorm counter advanced by one multi-tensor launch AFTER the forward (the modules tally their calls
    # meanwhile): the forward's first access to a module buffer is then its first norm layer, which is where a buffer broadcast
    # launched ahead of the step (distributed.broadcast_buffers(async_op=True)) is waited for
    bns = [m for m in model.modules()
           if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training and m.num_batches_tracked is not None]
    F.dropout_pool_begin_step()              # the element-wise dropout masks of the step from one draw (once their total is known)
    # the step's weight packings by one launch at its top (recorded during the model's first step at this signature)
    with F.prepacked_weights(model, (str(dtype or F.compute_dtype()), F.conv_math_signature(), tuple(x.shape))) as pp:
        if (dtype or F.compute_dtype()) == torch.float32 and not pp.replaying:
            F.prefetch_weight_amax(model)        # f16x3: the k3 weights' maxima by one multi-tensor launch (valid for this step only; a replayed plan measured them)
        try:
            with F.autocast(dtype or F.compute_dtype()), F.counters_batched(bns):
                if getattr(model, "takes_frequency_bands", False):      # the IS network, train.py:198-201: second output discarded
                    from .models.three_d.IS import frequency_bands
                    low_x, high_x = frequency_bands(x)
                    pred, _ = _forward(model, _leaves, x, low_x, high_x)
                else:
                    pred = _forward(model, _leaves, x)
        finally:
            F.clear_weight_amax()                # (the backward reuses the scalars the forward handed to each layer, not the table)
        F.flush_deferred_waits()                                     # (a model without norm layers)
        if bns:
            torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        if criterion is None:
            # nn.BCEWithLogitsLoss + pred.argmax + gt.argmax + the Dice counters in one pass over the logits
            loss, mask, counts = F.bce_argmax_dice(pred, gt2)
        else:
            loss = criterion(pred, gt2)
            with torch.no_grad():
                mask = F.argmax_channels(pred)
                gt_lab = F.argmax_channels(gt2)
                counts = F.dice_counts(gt_lab, mask)
        loss.backward()
    if _leaves is not None:                  # the capture's stand-in leaves: their gradients ARE the parameters' (static tensors of the graph pool)
        for name, p in model.named_parameters():
            p.grad = _leaves[name].grad
    if grad_hook is not None:
        grad_hook(model)
    if step_optimizer:                       # (GraphedTrainStep with a gradient hook captures the optimizer step as a graph of its own)
        optimizer.step()
    out = {"pred": pred, "mask": mask, "loss": loss.detach(), "counts": counts}
    if sync_metric:
        out["jaccard"], out["dice"] = metric_from_counts(counts.cpu().tolist())
    return out


class GraphedTrainStep:
    """``train_step`` captured once into a HIP graph and replayed: one ``hipGraphLaunch`` per iteration instead of
    ~400 kernel launches from Python.  On 64^3 patches (the reference's default ``patch_size``) the eager step is
    launch-bound -- the GPU finishes its kernels faster than one host thread can enqueue them -- and the graph
    removes that; at 128^3 the step is GPU-bound either way.

    Constraints of stream capture: fixed input shapes (the patch pipeline already guarantees them), an optimizer
    constructed with ``capturable=True`` (its step counter lives on the device) and the in-library kernel profiler off.
    Data parallel (``grad_hook`` = the gradient reducer): the collectives are host calls, so the iteration is captured as TWO
    graphs -- zero_grad ... backward, and the optimizer step -- with the reducer run eagerly between their replays (its
    post-accumulate-grad hooks, which only fire while Python runs a backward, are suspended: every bucket goes out at the call;
    the launch-bound part of the step is still one ``hipGraphLaunch``).  Three eager warm-up steps run
    on a side stream first -- they size the workspace and set the kernels' LDS attributes -- and, like the captured
    step, they DO update the model, so a freshly built instance has already taken ``warmup`` optimiser steps (``first`` holds
    the loss and the Dice counters of the last of them).
    Returned tensors are static buffers overwritten by the next call.

    **Capturing after eager iterations of the same model** (r5; it used to take the process down in ``hipStreamEndCapture``, where no
    try / except reaches).  Cause: a parameter's gradient accumulator (autograd's AccumulateGrad node) is bound to the stream that
    was current when the node was CREATED, and it lives as long as any autograd graph that points at it -- an eager iteration on the
    default stream whose ``pred`` / ``loss`` somebody still holds keeps default-stream accumulators alive.  A backward inside the
    capture then hands them their gradients across streams: the engine makes the (legacy) default stream wait for an event of the
    capturing stream, which a global-mode capture forbids (hipErrorStreamCaptureImplicit); the capture is invalidated and ending it
    fails inside a destructor.  The accumulators cannot be re-bound or dropped from Python, so the captured iteration does not use
    them: its forward runs on **stand-in leaves** -- fresh ``detach()`` aliases of the parameters' storage
    (``torch.func.functional_call``), whose accumulators are created inside the capture, on the capturing stream -- and the stand-ins'
    gradient tensors are installed as the parameters' ``.grad`` before the captured optimizer step.  Same kernels, same memory, no
    copies; whatever ran before the capture, on whatever stream, no longer matters (tests/test_gpu_unet.py::
    test_graphed_step_captures_after_an_eager_step_with_a_live_output)."""

    def __init__(self, model, optimizer, x, gt, criterion=None, warmup=3, dtype=None, grad_hook=None):
        from ._lib import lib
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainStep: build the optimizer with capturable=True (e.g. torch.optim.Adam(..., capturable=True))")
        lib().call("mi355seg_prof_enable", 0)
        self.model, self.optimizer, self.criterion, self.grad_hook = model, optimizer, criterion, grad_hook
        self.x = x.detach().to(torch.float32).clone()
        self.gt = gt.detach().clone()
        if grad_hook is not None and hasattr(grad_hook, "suspend_hooks"):
            grad_hook.suspend_hooks(True)            # from here on every bucket is launched at the reducer's call
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.first = None
            for _ in range(warmup):
                o = train_step(model, optimizer, self.x, self.gt, criterion, sync_metric=False, dtype=dtype, grad_hook=grad_hook)
                self.first = {"loss": o["loss"].detach().clone(), "counts": o["counts"].clone()}
                del o
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.opt_graph = None
        F.amax_pool_reset()              # the captured step zero-fills the chunk of operand-maximum slots it draws from INSIDE the capture
        leaves = {name: p.detach().requires_grad_(p.requires_grad) for name, p in model.named_parameters()}
        if grad_hook is None:
            with torch.cuda.graph(self.graph):
                self.out = train_step(model, optimizer, self.x, self.gt, criterion, sync_metric=False, dtype=dtype, _leaves=leaves)
        else:
            with torch.cuda.graph(self.graph):
                self.out = train_step(model, optimizer, self.x, self.gt, criterion, sync_metric=False, dtype=dtype, step_optimizer=False, _leaves=leaves)
            # a parameter the backward does not reach has no static gradient tensor: the reducer would allocate one at its first call
            # (and the optimizer graph must already see it), so give it zeros now -- no collective here: the captured backward has
            # not executed, the static gradients hold nothing yet
            for p in model.parameters():
                if p.requires_grad and p.grad is None:
                    p.grad = torch.zeros_like(p)
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph, pool=self.graph.pool()):
                optimizer.step()
        F.amax_pool_reset()
        # the captured step reads its weight packings (and the job table that forms them) in the arenas of the model's prepack plans: keep those
        # tensors alive for as long as the graph is, whatever later eager steps at other shapes evict from the model's own table
        self._prepack_arenas = [rec.get("arena") for rec in (model.__dict__.get("_seg_prepack") or {}).values()]

    def __call__(self, x, gt, sync_metric=True):
        self.x.copy_(x, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)
        # a buffer broadcast launched ahead of the step (distributed.broadcast_buffers(async_op=True)): a replay runs no Python forward,
        # so nothing else would wait for it before the captured BatchNorm kernels read and update the running statistics
        F.flush_deferred_waits()
        self.graph.replay()
        if self.opt_graph is not None:
            self.grad_hook(self.model)               # mean all-reduce of the static gradient tensors, in place
            self.opt_graph.replay()
        out = dict(self.out)
        if sync_metric:
            out["jaccard"], out["dice"] = metric_from_counts(out["counts"].cpu().tolist())
        return out


def weights_init_normal(init_type):
    """The reference's init policy (train.py:33-61) for ``model.apply(...)``: Conv*/Linear
    weights by ``init_type``, their biases zeroed; only classes named *BatchNorm2d* get the
    norm branch, so BatchNorm3d keeps its default (1, 0)."""
    from torch.nn import init

    def init_func(m):
        name = type(m).__name__
        if "BatchNorm2d" in name:
            if getattr(m, "weight", None) is not None:
                init.normal_(m.weight.data, 1.0, 0.02)
            if getattr(m, "bias", None) is not None:
                init.constant_(m.bias.data, 0.0)
            return
        if not (hasattr(m, "weight") and ("Conv" in name or "Linear" in name)):
            return
        table = {
            "normal": lambda w: init.normal_(w, 0.0, 0.02),
            "xavier": lambda w: init.xavier_normal_(w, gain=0.02),
            "xavier_uniform": lambda w: init.xavier_uniform_(w, gain=1.0),
            "kaiming": lambda w: init.kaiming_normal_(w, a=0, mode="fan_in"),
            "orthogonal": lambda w: init.orthogonal_(w, gain=0.02),
        }
        if init_type == "none":
            m.reset_parameters()
        elif init_type in table:
            table[init_type](m.weight.data)
        else:
            raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
        if getattr(m, "bias", None) is not None:
            init.constant_(m.bias.data, 0.0)

    return init_func
