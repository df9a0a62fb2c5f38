"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" ==
RCCL over xGMI on ROCm; "gloo" on CPU for the multi-process tests).

The reference delegates multi-GPU to accelerate -> DistributedDataParallel
(train.py:167-169,211): rank-local BatchNorm statistics, a gradient all-reduce (mean) per
step, and a broadcast of module buffers from rank 0 at every forward.  The same three
semantics are kept here.  Gradient payload is 90-585 MB per step for the models in scope:
on point-to-point xGMI (7 links x ~153 GB/s per GPU) that is ~1 ms against a >=40 ms
fp32 step, so the gradients travel as a few large flat buckets (32 MB: large messages are
what the per-link ring bandwidth wants), each launched from a gradient hook as soon as
backward has filled it, so the transfer hides under the remaining backward kernels."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun
    contract).  Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def self_launch(nproc, argv, env=None):
    """Start ``nproc`` ranks of the running script (``argv`` = [script, args...]) as child processes, one per GPU, with
    the torchrun environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), wait for them and
    return the worst exit code.  This is what ``accelerate launch`` does for the reference (train.py:167-169 runs under
    it).  The parent never touches the GPU (a process that has initialised HIP must not exec or fork workers), it only
    waits; the children inherit stdout / stderr, so rank 0's output is the job's output."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(nproc):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
        e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // nproc)))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    import time
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                if p.poll() not in (None, 0) and not rc:         # one rank died: the others would hang in a collective
                    rc = p.returncode
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()
            time.sleep(0.05)
        for p in procs:
            rc = rc or p.returncode
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise
    return rc


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class CommTimer:
    """Accumulates how long the step spent WAITING on communication, measured where the step waits: HIP events on the compute
    stream (the collectives run on the communicator's stream; ``work.wait()`` only makes the compute stream wait for them, so
    the time between the two events is the exposed part -- what backward did not hide -- plus the scatter-back copy), or the
    host clock for the CPU (gloo) rehearsal.  ``total_ms()`` synchronises once, at the end of the timed region.

    OFF by default: a training run pays nothing for it (no events recorded on the compute stream, nothing kept per step); only
    the call tally runs.  ``enable()`` -- bench.py, the tests -- turns the measurement on; finished event pairs are folded into a
    scalar every ``FOLD`` calls, so even an enabled timer holds a bounded number of live events."""

    FOLD = 64

    def __init__(self, enabled=False):
        self.enabled = enabled
        self.pairs, self.host_ms, self.calls, self.folded_ms = [], 0.0, 0, 0.0

    def enable(self, on=True):
        self.enabled = bool(on)
        return self

    def start(self, device):
        import time
        self.calls += 1
        if not self.enabled:
            return None
        if device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            return (e0, e1)
        return time.perf_counter()

    def stop(self, token):
        import time
        if token is None:
            return
        if isinstance(token, tuple):
            token[1].record()
            self.pairs.append(token)
            if len(self.pairs) >= self.FOLD:
                self._fold()
        else:
            self.host_ms += (time.perf_counter() - token) * 1e3

    def _fold(self, wait=False):
        """Move the pairs whose end event has completed (all of them when ``wait``) into the scalar and drop their events."""
        keep = []
        for a, b in self.pairs:
            if wait:
                b.synchronize()
            if wait or b.query():
                self.folded_ms += a.elapsed_time(b)
            else:
                keep.append((a, b))
        if len(keep) >= self.FOLD:               # nothing has completed for FOLD calls: wait for the oldest half rather than grow
            for a, b in keep[:self.FOLD // 2]:
                b.synchronize()
                self.folded_ms += a.elapsed_time(b)
            keep = keep[self.FOLD // 2:]
        self.pairs = keep

    def reset(self):
        self.pairs, self.host_ms, self.calls, self.folded_ms = [], 0.0, 0, 0.0

    def total_ms(self):
        self._fold(wait=True)
        return self.host_ms + self.folded_ms


BUFFER_BROADCAST_TIMER = CommTimer()


def comm_report(reducer, steps):
    """The `comm` block of bench.py's line (rank 0): what one step sends and how long it waited for it."""
    params = reducer.params if reducer is not None else []
    dev = params[0].device.type if params else "cpu"
    return {
        "grad_bytes_per_step": int(sum(p.numel() * p.element_size() for p in params)),
        "buckets": len(reducer.buckets) if reducer is not None else 0,
        "bucket_cap_mb": getattr(reducer, "bucket_mb", None),
        "allreduce_wait_ms_per_step": (reducer.timer.total_ms() / max(1, steps)) if reducer is not None else 0.0,
        # the reducer's own copies: gradients gathered into the flat buckets (torch.cat, on the compute stream inside backward) and
        # the means scattered back into p.grad (_foreach_copy_, part of the wait above): 2 x grad_bytes of extra traffic each
        "bucket_pack_ms_per_step": (reducer.pack_timer.total_ms() / max(1, steps)) if reducer is not None else 0.0,
        "bucket_unpack_ms_per_step": (reducer.unpack_timer.total_ms() / max(1, steps)) if reducer is not None else 0.0,
        "broadcast_buffers_ms_per_step": BUFFER_BROADCAST_TIMER.total_ms() / max(1, steps),
        "buffer_broadcast_collectives_per_step": BUFFER_BROADCAST_TIMER.calls / max(1, steps),
        "timer": "HIP events on the compute stream of rank 0: time the step waited in GradAllReducer.__call__ (exposed all-reduce + scatter-back) "
                 "and in broadcast_buffers" if dev == "cuda" else "host clock (CPU / gloo rehearsal)",
    }


class GradAllReducer:
    """Mean all-reduce of every parameter gradient through flat fp32 buckets, overlapped with backward.

    Buckets are filled in reverse parameter order (the order backward produces gradients) up to ``bucket_mb``.
    A post-accumulate-grad hook on every parameter counts down its bucket; the moment a bucket's last gradient
    lands, the bucket is flattened on the compute stream and its asynchronous all-reduce starts on the
    communicator's stream while backward keeps producing the earlier layers' gradients.  Calling the reducer
    after ``loss.backward()`` (``train_step(..., grad_hook=reducer)``) launches whatever was not triggered by a
    hook (parameters that received no gradient), waits for every collective and scatters the means back into
    ``p.grad``.  ``overlap=False`` skips the hooks (everything is launched at the call).  ``always=True`` runs
    the collectives even in a single-rank group (used to exercise the RCCL path on a one-GPU box)."""

    def __init__(self, model, bucket_mb=32.0, overlap=True, always=False):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.always = always
        self.bucket_mb = bucket_mb
        self.timer, self.pack_timer, self.unpack_timer = CommTimer(), CommTimer(), CommTimer()
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > cap:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._handles = []
        if overlap:
            for bi, bucket in enumerate(self.buckets):
                for p in bucket:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _active(self):
        return self.always or world_size() > 1

    def suspend_hooks(self, on=True):
        """Hooks off: nothing is launched from backward, every bucket goes out at the reducer's call (engine.GraphedTrainStep: a
        captured backward runs no Python, and a collective must not be issued inside a capture)."""
        self._suspended = bool(on)
        if on:
            self._pending = [len(b) for b in self.buckets]

    def _make_hook(self, bi):
        def hook(_param):
            if not self._active() or getattr(self, "_suspended", False):
                return
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        bucket = self.buckets[bi]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
        total = sum(g.numel() for g in grads)
        flat = self._flat[bi]
        if flat is None or flat.numel() != total or flat.device != grads[0].device:
            flat = torch.empty(total, dtype=grads[0].dtype, device=grads[0].device)
            self._flat[bi] = flat
        token = self.pack_timer.start(flat.device)
        torch.cat([g.reshape(-1) for g in grads], out=flat)
        self.pack_timer.stop(token)
        # RCCL averages inside the collective (ReduceOp.AVG); gloo only sums, the division follows the wait
        self._avg_in_collective = dist.get_backend() == "nccl"
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        self._works[bi] = dist.all_reduce(flat, op=op, async_op=True)

    def detach(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def __call__(self, model=None):
        if not self._active():
            return
        for bi in range(len(self.buckets)):
            if self._works[bi] is None:
                self._launch(bi)
        token = self.timer.start(self.params[0].device)
        inv = 1.0 / world_size()
        for bi, bucket in enumerate(self.buckets):
            self._works[bi].wait()
            flat = self._flat[bi]
            if inv != 1.0 and not getattr(self, "_avg_in_collective", False):
                flat.mul_(inv)
            off, dsts, srcs = 0, [], []
            for p in bucket:
                n = p.numel()
                view = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = view.clone()
                else:
                    dsts.append(p.grad)
                    srcs.append(view)
                off += n
            if dsts:
                tk = self.unpack_timer.start(flat.device)
                torch._foreach_copy_(dsts, srcs)              # one multi-tensor launch per bucket instead of one copy per parameter
                self.unpack_timer.stop(tk)
            self._works[bi] = None
            self._pending[bi] = len(bucket)
        self.timer.stop(token)


def broadcast_parameters(model, src=0):
    """DDP's wrap-time parameter broadcast (accelerator.prepare -> DistributedDataParallel.__init__, reference
    train.py:167-169): every rank starts from rank ``src``'s parameters, whatever its own random init drew.  One flat
    collective per dtype; a no-op for a single process.  Returns the number of parameters sent."""
    if world_size() == 1:
        return 0
    params = [p for p in model.parameters()]
    sent = 0
    for dtype in sorted({p.dtype for p in params}, key=str):
        group = [p for p in params if p.dtype == dtype]
        flat = torch.cat([p.detach().reshape(-1) for p in group])
        dist.broadcast(flat, src=src)
        off = 0
        with torch.no_grad():
            for p in group:
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n
        sent += off
    return sent


def setup_replica(model, bucket_mb=32.0, overlap=True):
    """Everything ``accelerator.prepare(model)`` does for the data-parallel replicas (train.py:167-169): rank 0's
    parameters and buffers overwrite every rank's, the buffers are flattened for the per-step broadcast, and the
    bucketed gradient reducer is armed.  Returns the reducer (None for a single process)."""
    if world_size() == 1:
        return None
    broadcast_parameters(model)
    flatten_buffers(model)
    broadcast_buffers(model)
    return GradAllReducer(model, bucket_mb=bucket_mb, overlap=overlap)


def flatten_buffers(model):
    """Re-seat every module buffer as a typed view into ONE flat byte tensor (each buffer at a 16-byte aligned offset), so that
    ``broadcast_buffers`` is a single collective with no gather / scatter copies around it -- float running statistics and the
    int64 ``num_batches_tracked`` counters travel together.  The kernels update running statistics through ``data_ptr()``, so
    views are transparent to them.  Idempotent; returns the flat tensor (None for a model without buffers)."""
    if getattr(model, "_mi355seg_flat_buffers", None) is not None:
        return model._mi355seg_flat_buffers
    owners = [(mod, name, b) for mod in model.modules() for name, b in mod._buffers.items() if b is not None]
    if not owners:
        model._mi355seg_flat_buffers = None
        return None
    offs, total = [], 0
    for _, _, b in owners:
        offs.append(total)
        total += (b.numel() * b.element_size() + 15) // 16 * 16
    flat = torch.zeros(total, dtype=torch.uint8, device=owners[0][2].device)
    for (m, n, b), off in zip(owners, offs):
        nbytes = b.numel() * b.element_size()
        view = flat[off:off + nbytes].view(b.dtype).view(b.shape)
        view.copy_(b.detach())
        m._buffers[n] = view
    model._mi355seg_flat_buffers = flat
    return flat


def broadcast_buffers(model, src=0, async_op=False):
    """DDP(broadcast_buffers=True): rank ``src``'s BatchNorm running statistics (and
    num_batches_tracked) overwrite every rank's before the forward -- one collective on the flattened buffers.
    ``async_op``: the collective is only LAUNCHED here (on the communicator's stream); the compute stream waits for it where the
    first norm layer of the forward touches a buffer (functional.bump_counter), so zero_grad, the label kernel, the layout change
    and the first convolution run beside it instead of behind it."""
    if world_size() == 1:
        return
    bufs = [b for b in model.buffers()]
    if not bufs:
        return
    token = BUFFER_BROADCAST_TIMER.start(bufs[0].device)
    flat = getattr(model, "_mi355seg_flat_buffers", None)
    if flat is not None and {b.untyped_storage().data_ptr() for b in bufs} == {flat.untyped_storage().data_ptr()}:
        if async_op:
            from . import functional as F
            F.defer_wait(dist.broadcast(flat, src=src, async_op=True))
        else:
            dist.broadcast(flat, src=src)
        BUFFER_BROADCAST_TIMER.stop(token)
        return
    model._mi355seg_flat_buffers = None           # the module was moved / re-created since: gather + scatter, one collective per dtype class
    fl = [b for b in bufs if b.is_floating_point()]
    it = [b for b in bufs if not b.is_floating_point()]
    for group in (fl, it):
        if not group:
            continue
        flat = torch.cat([b.reshape(-1) for b in group])
        dist.broadcast(flat, src=src)
        off = 0
        for b in group:
            n = b.numel()
            b.copy_(flat[off:off + n].view_as(b))
            off += n
    BUFFER_BROADCAST_TIMER.stop(token)


def all_reduce_metric(counts, loss):
    """Global Dice: sum the integer counters and average the loss over ranks (the
    reference leaves this as a TODO, train.py:220-224, and logs per-rank values)."""
    ws = world_size()
    if ws == 1:
        return counts, loss
    c = counts.clone()
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    l = loss.detach().clone().reshape(1)
    dist.all_reduce(l, op=dist.ReduceOp.SUM)
    return c, (l / ws).reshape(())
