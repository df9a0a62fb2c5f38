"""The RCCL code path on a one-GPU box: a single-rank "nccl" group, with the gradient reducer forced on
(``always=True``) so the flat-bucket all-reduces, the hook-driven overlap with backward, the buffer broadcast
and the metric reduction all run through RCCL on device tensors.  With one rank the mean is the identity, so
the train step must be bit-identical to the undistributed one.  The two-rank arithmetic of the HIP path runs here as two
replicas sharing the box's one GPU over gloo (device gradients, staged by the backend); the same arithmetic of the oracle network
is covered on CPU by tests/test_distributed.py; N > 1 GPUs over RCCL are the driver's to launch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_train_step_is_identity():
    import mi355seg
    from mi355seg import distributed as D
    from mi355seg.engine import train_step
    from mi355seg.models.three_d.unet3d import UNet3D
    from oracle.fill import fill_module_, make_input, make_labels

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = make_input((2, 1, 32, 32, 32)).cuda()
        gt = make_labels((2, 1, 32, 32, 32)).cuda()

        def run(distributed):
            model = fill_module_(UNet3D(1, 2, 8)).cuda().train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            reducer = D.GradAllReducer(model, bucket_mb=0.5, always=True) if distributed else None
            outs = []
            for _ in range(2):
                outs.append(train_step(model, opt, x, gt, grad_hook=reducer))
            if distributed:
                assert len(reducer.buckets) > 1
                c, l = outs[-1]["counts"].clone(), outs[-1]["loss"].detach().clone().reshape(1)
                dist.all_reduce(c)                       # int64 counters and the loss through RCCL
                dist.all_reduce(l)
                assert torch.equal(c, outs[-1]["counts"]) and torch.equal(l.reshape(()), outs[-1]["loss"].detach())
                flat = torch.cat([b.reshape(-1).float() for b in model.buffers()])
                dist.broadcast(flat, src=0)
            return model, outs

        m0, o0 = run(False)
        m1, o1 = run(True)
        for a, b in zip(o0, o1):
            assert torch.equal(a["loss"], b["loss"]) and torch.equal(a["counts"], b["counts"])
        for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        dist.destroy_process_group()


def _hip_ddp_worker(rank, world, port, out):
    """One data-parallel train step of the HIP U-Net on this rank's shard, both ranks on cuda:0 (the box has one GPU), gradients
    mean-all-reduced by the product's reducer over gloo (device tensors staged through the host by the backend)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import mi355seg
    from mi355seg import distributed as D
    from mi355seg.engine import train_step
    from mi355seg.models.three_d.unet3d import UNet3D
    from oracle.fill import fill_module_, make_input, make_labels
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = fill_module_(UNet3D(1, 2, 4)).cuda().train()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)           # the gradients are what is compared: leave the weights alone
    reducer = D.GradAllReducer(model, bucket_mb=0.05)
    x = make_input((2, 1, 16, 16, 16), freq=0.05, phase=float(rank)).cuda()
    gt = make_labels((2, 1, 16, 16, 16), thresh=0.8 - 0.3 * rank).cuda()
    D.broadcast_buffers(model)
    res = train_step(model, opt, x, gt, grad_hook=reducer)
    counts, loss = D.all_reduce_metric(res["counts"], res["loss"].detach())
    out[rank] = ({k: p.grad.detach().cpu() for k, p in model.named_parameters()}, len(reducer.buckets), counts.cpu(), float(loss),
                 res["counts"].cpu(), float(res["loss"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_path_gradients_equal_the_mean_of_per_shard_oracle_gradients():
    """SURVEY section 8(e) on the HIP path: two replicas (sharing the box's one GPU) each run the library's forward / backward on
    their own shard with rank-local BatchNorm statistics; the reducer's bucketed mean all-reduce must leave, on both ranks, the mean of
    the CPU oracle's per-shard gradients; the Dice counters and the loss reduce to the global values."""
    import torch.multiprocessing as mp
    from oracle.fill import fill_module_, make_input, make_labels
    from oracle.losses import bce_with_logits
    from oracle.nets import UNet3D
    from oracle.step import two_channel_gt
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_hip_ddp_worker, args=(world, port, out), nprocs=world, join=True)
    shard = []
    for rank in range(world):
        m = fill_module_(UNet3D(1, 2, 4)).train()
        x = make_input((2, 1, 16, 16, 16), freq=0.05, phase=float(rank))
        gt = make_labels((2, 1, 16, 16, 16), thresh=0.8 - 0.3 * rank)
        loss = bce_with_logits(m(x), two_channel_gt(gt).float())
        loss.backward()
        shard.append(({k: p.grad for k, p in m.named_parameters()}, float(loss.detach())))
    scale = max(float(g.abs().max()) for g in shard[0][0].values())
    assert out[0][1] > 1                                                   # several buckets went out
    for k, g0 in shard[0][0].items():
        want = (g0 + shard[1][0][k]) / 2
        for rank in range(world):
            got = out[rank][0][k]
            assert (got - want).abs().max() <= 2e-4 * max(float(want.abs().max()), 1e-3 * scale), (k, rank)
        assert torch.equal(out[0][0][k], out[1][0][k]), k                  # both replicas hold the same reduced gradient
    assert torch.equal(out[0][2], out[0][4] + out[1][4]) and torch.equal(out[0][2], out[1][2])      # global integer Dice counters
    assert abs(out[0][3] - (shard[0][1] + shard[1][1]) / 2) < 1e-5


def _graph_ddp_worker(rank, world, port, out):
    """Data-parallel steps of the HIP U-Net on two ranks sharing cuda:0, once eager (reducer from hooks) and once as captured graphs
    with the reducer between the two replays (engine.GraphedTrainStep(grad_hook=...))."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import mi355seg
    from mi355seg import distributed as D
    from mi355seg.engine import GraphedTrainStep, train_step
    from mi355seg.models.three_d.unet3d import UNet3D
    from oracle.fill import fill_module_, make_input, make_labels
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = make_input((1, 1, 16, 16, 16), freq=0.05, phase=float(rank)).cuda()
    gt = make_labels((1, 1, 16, 16, 16), thresh=0.8 - 0.3 * rank).cuda()
    res = {}
    for mode in ("eager", "graph"):
        model = fill_module_(UNet3D(1, 2, 4)).cuda().train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        reducer = D.setup_replica(model, bucket_mb=0.05)
        losses = []
        if mode == "eager":
            for i in (0, 0, 2, 3):                                                         # the graphed object's two warm-up steps run on x
                D.broadcast_buffers(model, async_op=True)
                losses.append(float(train_step(model, opt, x + 0.1 * i, gt, sync_metric=False, grad_hook=reducer)["loss"]))
        else:
            g = GraphedTrainStep(model, opt, x, gt, warmup=2, grad_hook=reducer)          # two eager steps (x), then replays
            losses = [None, float(g.first["loss"])]
            from mi355seg import functional as F
            for i in (2, 3):
                D.broadcast_buffers(model, async_op=True)                                  # as train.py launches it: the replay must wait for it
                assert len(F._DEFERRED_WAITS) == 1
                losses.append(float(g(x + 0.1 * i, gt, sync_metric=False)["loss"]))
                assert not F._DEFERRED_WAITS
        res[mode] = (losses, {k: v.detach().cpu() for k, v in model.state_dict().items()})
        reducer.detach()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_graphed_step_matches_the_eager_data_parallel_step():
    """the HIP-graph step under data parallelism -- backward and optimizer step as two captured graphs, the bucketed
    gradient all-reduce between their replays, buffers broadcast ahead of the step -- lands on the eager data-parallel run's
    parameters (both ranks, which must also agree with each other)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_graph_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in (0, 1):
        le, lg = out[rank]["eager"][0], out[rank]["graph"][0]
        assert le[1:] == lg[1:], (rank, le, lg)                                         # same losses step by step (the first warm-up loss is not kept)
        for k, v in out[rank]["eager"][1].items():
            assert torch.equal(v, out[rank]["graph"][1][k]), (rank, k)                  # and the same parameters / buffers after four steps
    for mode in ("graph", "eager"):                                                     # the replicas' PARAMETERS stay identical (their BatchNorm
        for k, v in out[0][mode][1].items():                                            # running statistics are rank-local until the next broadcast)
            if "running_" not in k:
                assert torch.equal(v, out[1][mode][1][k]), (mode, k)


def _bench(extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MI355SEG_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "unet3d_f32_1x64", "--no-workloads", "--no-cpu-baseline",
                        "--steps", "8", "--warmup", "2"] + extra, capture_output=True, text=True, env=env, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_whole_hip_path_on_one_gpu():
    """First contact for the N > 1 bench nobody could run on hardware yet: the REAL ``bench.py --gpus 2`` -- self-launch,
    rendezvous, setup_replica, the HIP train step with the bucketed reducer and the ahead-of-step buffer broadcast, the in-library
    HIP-event bracketing, max-over-ranks timing, one JSON line -- with both ranks on this box's one GPU and gloo in RCCL's place
    (MI355SEG_DIST_BACKEND).  Checked: two ranks formed, the `comm` block (with the reducer's pack / unpack copies timed), `roofline`
    present, and an aggregate rate that is neither absurdly below nor above two one-rank runs sharing the card."""
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2"])
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 1 and "comm" not in one and one["roofline"]["frac"] > 0
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["dist_backend"] == "gloo" and two["config"]["parallelism"] == "dp2"
    assert two["config"]["global_batch"] == 2 and two["scaling"] == "weak" and len(two["rank_devices"]) == 2
    c = two["comm"]
    assert c["grad_bytes_per_step"] == 4 * 22581250 and c["buckets"] >= 2                     # UNet3D(1, 2, 32): 90.3 MB of gradients
    assert c["allreduce_wait_ms_per_step"] > 0 and c["bucket_pack_ms_per_step"] > 0 and c["bucket_unpack_ms_per_step"] > 0
    assert c["buffer_broadcast_collectives_per_step"] == 1.0
    assert two["roofline"]["bound"] == "mfma" and two["roofline"]["frac"] > 0 and two["roofline"]["launches"] > 0
    assert 0.0 <= two["dice"] <= 1.0 and two["loss"] > 0
    # two ranks time-share one GPU and stage 90 MB of gradients through the host (gloo): the aggregate is at best the one-rank rate and
    # must stay within 2x of "twice as many voxels in twice the time plus the gloo round trip" -- an order-of-magnitude plumbing check
    assert two["value"] > 0.1 * one["value"] and two["value"] < 2.0 * 2.0 * one["value"], (one["value"], two["value"])
