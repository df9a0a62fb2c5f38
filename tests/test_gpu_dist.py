"""The RCCL code path on a one-GPU box: a single-rank "nccl" group, with the gradient reducer forced on
(``always=True``) so the flat-bucket all-reduces, the hook-driven overlap with backward, the buffer broadcast
and the metric reduction all run through RCCL on device tensors.  With one rank the mean is the identity, so
the train step must be bit-identical to the undistributed one.  (The two-rank arithmetic is covered on CPU by
tests/test_distributed.py; N > 1 GPUs are the driver's to launch.)"""
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
