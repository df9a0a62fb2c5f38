"""Data-parallel plumbing on CPU: two gloo ranks (the N > 1 path of bench.py / engine).
Checks the three DDP semantics the reference gets from accelerate (train.py:167-169,211):
mean gradient all-reduce, rank-0 buffer broadcast, global metric reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.BatchNorm3d(4), torch.nn.ReLU(),
                               torch.nn.Conv3d(4, 2, 1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import mi355seg
    from mi355seg import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world
    model = _make_model().train()
    # make rank-1's buffers differ, then broadcast rank 0's
    if rank == 1:
        for b in model.buffers():
            b.add_(1)
    D.broadcast_buffers(model)                                 # gather / broadcast / scatter path
    ref = _make_model()
    for a, b in zip(model.buffers(), ref.buffers()):
        assert torch.equal(a, b)
    if rank == 1:
        for b in model.buffers():
            b.add_(2)
    flat = D.flatten_buffers(model)                            # every buffer (float statistics AND the int64 counter) a view of ONE byte tensor
    assert flat is not None and flat.dtype == torch.uint8 and D.flatten_buffers(model) is flat
    assert all(b.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() and b.data_ptr() % 16 == 0 for b in model.buffers())
    D.BUFFER_BROADCAST_TIMER.reset()
    D.BUFFER_BROADCAST_TIMER.enable()                          # (the wait timers are opt-in: a training run records nothing)
    D.broadcast_buffers(model)                                 # one collective, no copies
    assert D.BUFFER_BROADCAST_TIMER.calls == 1 and D.BUFFER_BROADCAST_TIMER.total_ms() > 0.0
    for (k, a), b in zip(model.named_buffers(), ref.buffers()):
        assert torch.equal(a, b) and a.dtype == b.dtype, k
    model.train()(torch.zeros(2, 1, 4, 4, 4))                  # BatchNorm keeps updating the (view) buffers in place
    assert int(model[1].num_batches_tracked) == 1 and model[1].num_batches_tracked.dtype == torch.int64
    model[1].reset_running_stats()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(2, 1, 4, 4, 4, generator=g)
    hooked = D.GradAllReducer(model, bucket_mb=0.0001)         # tiny buckets -> several collectives, launched from hooks
    assert not hooked.timer.enabled
    hooked.timer.enable()
    results = []
    for step in range(2):                                      # twice: the per-bucket countdown must re-arm
        for p in model.parameters():
            p.grad = None
        model(x + step).square().mean().backward()
        local = [p.grad.clone() for p in model.parameters()]   # p.grad stays local until the reducer is called
        assert all(w is not None for w in hooked._works)       # every bucket went out during backward
        hooked(model)
        results.append((local, [p.grad.clone() for p in model.parameters()]))
    rep = D.comm_report(hooked, steps=2)                       # the `comm` block of bench.py's line
    assert rep["grad_bytes_per_step"] == sum(p.numel() * 4 for p in model.parameters()) and rep["buckets"] == len(hooked.buckets) > 1
    assert rep["allreduce_wait_ms_per_step"] > 0.0 and "host clock" in rep["timer"] and rep["buffer_broadcast_collectives_per_step"] >= 0
    hooked.detach()
    for p in model.parameters():
        p.grad = None
    model(x).square().mean().backward()
    late = D.GradAllReducer(model, bucket_mb=0.0001, overlap=False)   # no hooks: everything launched at the call
    late(model)
    for a, b in zip(results[0][1], model.parameters()):
        assert torch.equal(a, b.grad)
    local, red = results[1]
    counts = torch.tensor([1 + rank, 2, 3, 4 * (rank + 1)], dtype=torch.int64)
    c, l = D.all_reduce_metric(counts, torch.tensor(float(rank + 1)))
    out[rank] = (local, red, c, l)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_buffer_broadcast():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    l0, r0, c0, m0 = out[0]
    l1, r1, c1, m1 = out[1]
    for a, b, ra, rb in zip(l0, l1, r0, r1):
        assert torch.allclose(ra, (a + b) / 2, atol=1e-7)
        assert torch.equal(ra, rb)                              # every rank holds the same reduced gradient
    assert c0.tolist() == [3, 4, 6, 12] and c1.tolist() == c0.tolist()
    assert abs(float(m0) - 1.5) < 1e-7


def test_single_process_is_a_noop():
    import mi355seg
    from mi355seg import distributed as D
    assert D.world_size() == 1
    m = _make_model()
    m(torch.randn(1, 1, 4, 4, 4)).sum().backward()
    g = [p.grad.clone() for p in m.parameters()]
    D.GradAllReducer(m)(m)
    D.broadcast_buffers(m)
    assert all(torch.equal(a, p.grad) for a, p in zip(g, m.parameters()))


def _unet_worker(rank, world, port, out):
    """One data-parallel train step of the oracle U-Net on this rank's shard (rank-local BatchNorm statistics, as plain
    nn.BatchNorm3d under DDP), gradients mean-all-reduced by the product's reducer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import mi355seg
    from mi355seg import distributed as D
    from oracle.fill import fill_module_, make_input, make_labels
    from oracle.losses import bce_with_logits
    from oracle.nets import UNet3D
    from oracle.step import two_channel_gt
    D.init_from_env(backend="gloo")
    torch.set_num_threads(2)
    model = fill_module_(UNet3D(1, 2, 4)).train()
    reducer = D.GradAllReducer(model, bucket_mb=0.05)
    x = make_input((2, 1, 16, 16, 16), freq=0.05, phase=float(rank))
    gt = make_labels((2, 1, 16, 16, 16), thresh=0.8 - 0.3 * rank)
    D.broadcast_buffers(model)
    bce_with_logits(model(x), two_channel_gt(gt).float()).backward()
    reducer(model)
    out[rank] = {k: p.grad.clone() for k, p in model.named_parameters()}
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradients_equal_the_mean_of_per_shard_oracle_gradients():
    """the CPU oracle on each rank's shard from identical weights, per-shard gradients averaged,
    against the all-reduced gradients -- not a single batch-4 run (BatchNorm statistics would differ)."""
    from oracle.fill import fill_module_, make_input, make_labels
    from oracle.losses import bce_with_logits
    from oracle.nets import UNet3D
    from oracle.step import two_channel_gt
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_unet_worker, args=(world, port, out), nprocs=world, join=True)
    shard_grads = []
    for rank in range(world):
        m = fill_module_(UNet3D(1, 2, 4)).train()
        x = make_input((2, 1, 16, 16, 16), freq=0.05, phase=float(rank))
        gt = make_labels((2, 1, 16, 16, 16), thresh=0.8 - 0.3 * rank)
        bce_with_logits(m(x), two_channel_gt(gt).float()).backward()
        shard_grads.append({k: p.grad for k, p in m.named_parameters()})
    scale = max(float(g.abs().max()) for g in shard_grads[0].values())     # conv biases in front of BatchNorm have ~0 gradient
    for k, g0 in shard_grads[0].items():
        want = (g0 + shard_grads[1][k]) / 2
        for rank in range(world):
            got = out[rank][k]
            assert (got - want).abs().max() <= 1e-4 * max(float(want.abs().max()), 1e-3 * scale), (k, rank)
        assert torch.equal(out[0][k], out[1][k]), k


def _init_worker(rank, world, port, out):
    """Ranks that drew DIFFERENT random initial weights (train.py applies weights_init_normal with no seed) must hold
    rank 0's parameters and buffers after setup_replica -- DDP's wrap-time broadcast, reference train.py:167-169."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import mi355seg
    from mi355seg import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(1000 + rank)
    model = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.BatchNorm3d(4), torch.nn.ReLU(), torch.nn.Conv3d(4, 2, 1)).train()
    with torch.no_grad():
        model[1].running_mean.add_(rank)
    before = [p.detach().clone() for p in model.parameters()]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    reducer = D.setup_replica(model, bucket_mb=0.0001)
    after_setup = [p.detach().clone() for p in model.parameters()]
    g = torch.Generator().manual_seed(7 + rank)               # each rank its own shard
    x = torch.randn(2, 1, 4, 4, 4, generator=g)
    D.broadcast_buffers(model)
    opt.zero_grad(set_to_none=True)
    model(x).square().mean().backward()
    reducer(model)
    opt.step()
    out[rank] = (before, after_setup, [p.detach().clone() for p in model.parameters()], [b.clone() for b in model.buffers()])
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_start_from_rank0_parameters_and_stay_identical():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_init_worker, args=(world, port, out), nprocs=world, join=True)
    b0, s0, a0, buf0 = out[0]
    b1, s1, a1, buf1 = out[1]
    assert any(not torch.equal(x, y) for x, y in zip(b0, b1))          # the ranks really started apart
    for x, y, z in zip(b0, s0, s1):
        assert torch.equal(x, y) and torch.equal(x, z)                  # rank 0's values everywhere, bit for bit
    for x, y in zip(a0, a1):
        assert torch.equal(x, y)                                        # and still one model after a step on different shards
    assert any(not torch.equal(x, y) for x, y in zip(s0, a0))          # the step did move the parameters
    # (running statistics are rank-local between two forwards, as under DDP: rank 0's are re-broadcast at the next one)


def test_comm_timer_is_opt_in_and_bounded():
    """A long world > 1 training run must not accumulate timing events: the wait timers record nothing unless enabled (bench.py
    enables them for its timed steps), and an enabled timer folds finished measurements into a scalar."""
    from mi355seg.distributed import CommTimer
    t = CommTimer()
    dev = torch.device("cpu")
    for _ in range(1000):
        t.stop(t.start(dev))
    assert t.calls == 1000 and not t.pairs and t.host_ms == 0.0 and t.total_ms() == 0.0

    class FakeEvent:                       # stands in for a HIP event (no GPU in this suite): completes after `lag` queries
        def __init__(self, lag):
            self.lag = lag
        def record(self):
            pass
        def query(self):
            self.lag -= 1
            return self.lag < 0
        def synchronize(self):
            self.lag = -1
        def elapsed_time(self, other):
            return 0.5
    t = CommTimer(enabled=True)
    for i in range(10 * CommTimer.FOLD):
        t.pairs.append((FakeEvent(0), FakeEvent(3 if i % 7 == 0 else 0)))      # what stop() does for a device token
        if len(t.pairs) >= t.FOLD:
            t._fold()
        assert len(t.pairs) < 2 * CommTimer.FOLD
    assert abs(t.total_ms() - 0.5 * 10 * CommTimer.FOLD) < 1e-9 and not t.pairs
