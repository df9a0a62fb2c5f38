"""BASELINE-size checks (cfg 2: UNet3D(1,2,32), x = [2,1,128,128,128]): (0) ONE full-size train step against the CPU oracle's
(about 10-25 s of host time: logits, loss, Dice, masks, stem / head / bottleneck gradients -- the comparison bench.py also puts
on its line as `parity_vs_cpu`), and size-independent properties: (a) bitwise determinism of a
whole train step (no atomics anywhere), (b) exact linearity of the MFMA convolution under power-of-two scaling,
(c) crops of the full-size result against ATen-CPU on the crop's receptive field (catches 32-bit index overflow
and tile-edge errors at full extent), (d) invariants of BatchNorm / metric."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mi355seg
    mi355seg.lib()
    return mi355seg


def _rnd(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def test_full_size_first_step_vs_cpu_oracle(seg):
    """North_star's bar at the benchmark configuration itself: UNet3D(1, 2, 32) on [2, 1, 128^3], first train step from kaiming weights,
    under every conv math -- logits within 1e-4 of the CPU oracle's (== the reference's PyTorch-CPU arithmetic, /root/reference/
    models/three_d/unet3d.py:50-71, train.py:187-221), loss and Dice within 1e-4, masks identical wherever the oracle's logit margin is
    decisive, gradients of the stem, the head, the bottleneck, an up-convolution and a BatchNorm scale within 1e-3 of the tensor's
    maximum (sums over 4.2 M voxels; the three maths' measured figures are printed).  The batch is bench.py's (labels = a
    thresholded low-frequency field of the input), so Dice is a number that could disagree."""
    import bench
    from mi355seg.engine import make_adam, train_step, weights_init_normal
    from mi355seg.models.three_d.unet3d import UNet3D
    from mi355seg.utils.metric import metric_from_counts
    from oracle.nets import UNet3D as OracleUNet
    from oracle.step import train_step as oracle_step, weights_init_normal as oracle_init
    x, gt = bench.synthetic_batch((2, 1, 128, 128, 128), 1234)
    torch.manual_seed(0)
    ref = OracleUNet(1, 2, 32)
    ref.apply(oracle_init("kaiming"))
    ref.train()
    pr, mr, lr, (jr, dr) = oracle_step(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), x, gt)
    pr = pr.detach()
    assert 0.05 < dr < 0.95, dr
    rgrads = {k: p.grad.clone() for k, p in ref.named_parameters() if k in bench.PARITY_GRADS}
    del ref
    margin = (pr[:, 1] - pr[:, 0]).abs()
    decisive = margin > 2e-4
    xg, gg = x.cuda(), gt.cuda()
    try:
        for math in ("f16x3", "bf16x6", "fp32"):
            seg.set_conv_math(math)
            torch.manual_seed(0)
            m = UNet3D(1, 2, 32)
            m.apply(weights_init_normal("kaiming"))
            m = m.cuda().train()
            out = train_step(m, make_adam(m.parameters(), lr=1e-3), xg, gg)
            pg = out["pred"].detach().cpu()
            dl = float((pg - pr).abs().max())
            differ = (pg.argmax(1) != pr.argmax(1))
            named = dict(m.named_parameters())
            gerr = {k: float((named[k].grad.cpu() - g).abs().max() / g.abs().max()) for k, g in rgrads.items()}
            print(f"[{math}] dlogit_max {dl:.3e} dloss {abs(out['loss'].item() - lr.item()):.3e} ddice {abs(out['dice'] - dr):.3e} (dice {dr:.4f}) "
                  f"masks differ {int(differ.sum())} (decisive {int((differ & decisive).sum())}, excluded {1 - float(decisive.float().mean()):.2e}) "
                  f"grad rel err {max(gerr.values()):.3e}")
            assert dl < 1e-4, (math, dl)
            assert abs(out["loss"].item() - lr.item()) < 1e-5, math
            assert abs(out["dice"] - dr) < 1e-4 and abs(out["jaccard"] - jr) < 1e-4, (math, out["dice"], dr)
            assert int((differ & decisive).sum()) == 0, math
            assert float(decisive.float().mean()) > 0.99
            for k, e in gerr.items():
                assert e < 1e-3, (math, k, e)
            del m, out
    finally:
        seg.set_conv_math("f16x3")


def test_full_size_train_step_is_bitwise_deterministic_and_consistent(seg):
    from mi355seg.engine import train_step, weights_init_normal
    from mi355seg.models.three_d.unet3d import UNet3D

    def run():
        torch.manual_seed(0)
        m = UNet3D(1, 2, 32)
        m.apply(weights_init_normal("kaiming"))
        m = m.cuda().train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(1234)
        x = torch.randn((2, 1, 128, 128, 128), generator=g).cuda()
        gt = (torch.rand((2, 1, 128, 128, 128), generator=g) > 0.9).float().cuda()
        out = train_step(m, opt, x, gt)
        grads = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        return out, grads, m, gt

    o1, g1, m1, gt = run()
    o2, g2, _, _ = run()
    assert torch.equal(o1["pred"], o2["pred"]) and torch.equal(g1, g2) and o1["loss"].item() == o2["loss"].item()
    assert torch.isfinite(g1).all() and np.isfinite(o1["loss"].item())
    # metric invariants: counters are exact integers of the two masks
    c = o1["counts"].cpu().tolist()
    assert c[0] == int(gt.sum().item()) and c[1] == int(o1["mask"].sum().item())
    assert c[2] <= min(c[0], c[1]) and c[3] >= max(c[0], c[1]) and c[2] + c[3] == c[0] + c[1]
    assert torch.equal(o1["mask"], o1["pred"].argmax(1, keepdim=True))
    # every BatchNorm saw exactly one batch; running_var moved off its initial 1.0
    for k, b in m1.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == 1
    assert float((m1.encoder1.enc1norm1.running_var - 1).abs().max()) > 1e-3


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32)])
def test_full_size_conv_linearity_and_crops_vs_cpu(seg, cin, cout):
    F = seg.functional
    N, D = 2, 128
    x = _rnd((N, D, D, D, cin), 1).cuda()
    w = (_rnd((cout, cin, 3, 3, 3), 2) * 0.05).cuda()
    b = _rnd((cout,), 3).cuda()
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    y = F.conv3d(xg, wg, b, 1, 1)
    # (b) scaling by 2 is exact in fp32: conv(2x) - bias == 2 (conv(x) - bias), bit for bit
    y2 = F.conv3d(2 * x, w, None, 1, 1)
    y0 = F.conv3d(x, w, None, 1, 1)
    assert torch.equal(y2, 2 * y0)
    # (c) crops against ATen-CPU, including the last voxels of the last sample (largest linear indices)
    for (n, z, yy, xx) in [(0, 0, 0, 0), (1, 120, 120, 120), (0, 60, 3, 125), (1, 127 - 8, 0, 64)]:
        zs, ys, xs = [slice(max(0, s - 1), min(D, s + 9)) for s in (z, yy, xx)]
        crop = x[n, zs, ys, xs].permute(3, 0, 1, 2)[None].cpu()
        ref = TF.conv3d(crop, w.cpu(), b.cpu(), padding=1)
        oz, oy, ox = [s - max(0, s - 1) for s in (z, yy, xx)]
        ref = ref[0, :, oz:oz + 8, oy:oy + 8, ox:ox + 8].permute(1, 2, 3, 0)
        got = y[n, z:z + 8, yy:yy + 8, xx:xx + 8].detach().cpu()
        assert (got - ref).abs().max() < 1e-4
    # backward at full size: wgrad / dgrad of a sparse upstream gradient equal the CPU result on its support
    gy = torch.zeros_like(y)
    gy[1, 100:104, 64:68, 120:124] = _rnd((4, 4, 4, cout), 4).cuda()
    y.backward(gy)
    zs, ys, xs = slice(99, 105), slice(63, 69), slice(119, 125)
    crop = x[1, zs, ys, xs].permute(3, 0, 1, 2)[None].cpu().requires_grad_(True)
    wc = w.cpu().requires_grad_(True)
    yc = TF.conv3d(crop, wc, None, padding=1)
    gcrop = torch.zeros_like(yc)
    gcrop[0, :, 1:5, 1:5, 1:5] = gy[1, 100:104, 64:68, 120:124].permute(3, 0, 1, 2).cpu()
    yc.backward(gcrop)
    assert (wg.grad.cpu() - wc.grad).abs().max() < 1e-4 * max(1.0, float(wc.grad.abs().max()))
    assert (xg.grad[1, zs, ys, xs].permute(3, 0, 1, 2).cpu() - crop.grad[0]).abs().max() < 1e-4
    assert float(xg.grad[0].abs().max()) == 0.0          # no gradient leaks into the other sample


def test_full_size_batchnorm_output_is_standardised(seg):
    F = seg.functional
    x = (_rnd((2, 128, 128, 128, 32), 5) * 3 + 7).cuda()
    gamma, beta = torch.ones(32, device="cuda"), torch.zeros(32, device="cuda")
    rm, rv = torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")
    y = F.batch_norm_act(x, gamma, beta, rm, rv, True, 0.1, 1e-5, F.ACT_NONE)
    yd = y.double().reshape(-1, 32)
    assert yd.mean(0).abs().max() < 1e-5 and (yd.var(0, unbiased=False) - 1).abs().max() < 1e-4
    assert (rm - 0.7).abs().max() < 1e-2 and (rv - (0.9 + 0.9)).abs().max() < 2e-2
