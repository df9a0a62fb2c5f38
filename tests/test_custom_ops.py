"""``torch.ops.mi355seg.*``: the C-ABI entry points registered as PyTorch custom operators (north_star: "exposed as torch
custom ops").  CPU part: the operators exist with the expected schemas and propagate shapes through their fake
implementations (meta tensors: no kernel runs).  GPU part: the registered autograd wiring equals the autograd.Function
wrappers the models use, and ``torch.library.opcheck`` accepts the registrations."""
import pytest
import torch


def test_custom_ops_are_registered_with_schemas_and_fake_impls():
    import mi355seg
    from mi355seg import custom_ops
    for name in custom_ops.OPS:
        op = getattr(torch.ops.mi355seg, name)
        assert op.default._schema.name == "mi355seg::" + name
    x = torch.empty((2, 8, 8, 16, 32), device="meta")
    w = torch.empty((64, 32, 3, 3, 3), device="meta")
    assert torch.ops.mi355seg.conv3d(x, w, None, 1, 1).shape == (2, 8, 8, 16, 64)
    assert torch.ops.mi355seg.conv3d(x, w, None, 2, 1).shape == (2, 4, 4, 8, 64)
    assert torch.ops.mi355seg.conv3d_dgrad(torch.empty((2, 8, 8, 16, 64), device="meta"), w, 8, 8, 16, 1, 1).shape == x.shape
    dw, db = torch.ops.mi355seg.conv3d_wgrad(torch.empty((2, 8, 8, 16, 64), device="meta"), x, 3, 1, 1, True)
    assert dw.shape == w.shape and db.shape == (64,) and dw.dtype == torch.float32
    wt = torch.empty((32, 16, 2, 2, 2), device="meta")
    assert torch.ops.mi355seg.conv_transpose3d_k2s2(x, wt, None).shape == (2, 16, 16, 32, 16)
    y, idx = torch.ops.mi355seg.max_pool3d_2x(x)
    assert y.shape == (2, 4, 4, 8, 32) and idx.dtype == torch.uint8
    assert torch.ops.mi355seg.upsample_nearest_2x(x).shape == (2, 16, 16, 32, 32)
    lg = torch.empty((2, 2, 8, 8, 8), device="meta")
    loss, mask, counts = torch.ops.mi355seg.bce_argmax_dice(lg, lg)
    assert loss.shape == () and mask.shape == (2, 1, 8, 8, 8) and mask.dtype == torch.int64 and counts.shape == (4,)
    assert "Tensor? bias" in str(torch.ops.mi355seg.conv3d.default._schema)
    v = torch.empty(64, device="meta")
    outs = torch.ops.mi355seg.conv_bn_act(x, w, None, v, v, v, v, 1, 1, 0.1, 1e-5, 1, 0.0)
    assert [tuple(o.shape) for o in outs] == [(2, 8, 8, 16, 64)] * 2 + [(64,)] * 4
    assert torch.ops.mi355seg.cat_channels(x, x).shape == (2, 8, 8, 16, 64) and torch.ops.mi355seg.to_channels_first(x).shape == (2, 32, 8, 8, 16)
    mean, rstd = torch.ops.mi355seg.norm_stats(x, None, None, 0.1, 1e-5, True)
    assert mean.shape == (64,) and torch.ops.mi355seg.norm_apply_act(x, mean, rstd, None, None, None, 1, 0.0, True).shape == x.shape
    assert torch.ops.mi355seg.activation(x, 2, 1.0).shape == x.shape


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_custom_ops_match_the_autograd_function_wrappers(dtype):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mi355seg
    F = mi355seg.functional
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1, 8, 8, 16, 32), generator=g).cuda().to(dtype)
    w = (torch.randn((64, 32, 3, 3, 3), generator=g) * 0.05).cuda()
    b = torch.randn((64,), generator=g).cuda()
    outs = []
    for fn in (lambda a, ww, bb: torch.ops.mi355seg.conv3d(a, ww, bb, 1, 1), lambda a, ww, bb: F.conv3d(a, ww, bb, 1, 1)):
        xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = fn(xa, wa, ba)
        y.float().square().sum().backward()
        outs.append((y.detach(), xa.grad, wa.grad, ba.grad))
    for a, c in zip(*outs):
        assert a.dtype == c.dtype and torch.equal(a, c)
    # ConvT k2 s2 + pool + upsample through the dispatcher
    wt = (torch.randn((32, 16, 2, 2, 2), generator=g) * 0.1).cuda()
    xa, wa = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    up = torch.ops.mi355seg.conv_transpose3d_k2s2(xa, wa, None)
    pooled, _ = torch.ops.mi355seg.max_pool3d_2x(up)
    back = torch.ops.mi355seg.upsample_nearest_2x(pooled)
    back.float().sum().backward()
    xb, wb = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = F.upsample_nearest_2x(F.max_pool3d_2x(F.conv_transpose3d_k2s2(xb, wb, None)))
    ref.float().sum().backward()
    assert torch.equal(back, ref) and torch.equal(xa.grad, xb.grad) and torch.equal(wa.grad, wb.grad)
    if dtype == torch.float32:
        # fused tail + opcheck of the registrations (schema, fake tensor propagation, autograd registration)
        lg = torch.randn((2, 2, 8, 8, 8), generator=g).cuda().requires_grad_(True)
        tgt = (torch.rand((2, 2, 8, 8, 8), generator=g) > 0.5).float().cuda()
        loss, mask, counts = torch.ops.mi355seg.bce_argmax_dice(lg, tgt)
        loss.backward()
        lg2 = lg.detach().clone().requires_grad_(True)
        l2 = F.bce_with_logits(lg2, tgt)
        l2.backward()
        assert abs(loss.item() - l2.item()) < 1e-6 and torch.allclose(lg.grad, lg2.grad, atol=1e-9)
        assert torch.equal(mask, lg.argmax(1, keepdim=True)) and torch.equal(counts, torch.ops.mi355seg.dice_counts(tgt.argmax(1, keepdim=True), mask))
        torch.library.opcheck(torch.ops.mi355seg.conv3d.default, (x.float(), w, b, 1, 1), test_utils=("test_schema", "test_faketensor"))
        torch.library.opcheck(torch.ops.mi355seg.max_pool3d_2x.default, (x.float(),), test_utils=("test_schema", "test_faketensor"))


@pytest.mark.gpu
def test_whole_unet3d_train_forward_backward_through_torch_ops_matches_the_reference_fixture(golden_dir):
    """A whole UNet3D(1, 2, 8) train-mode forward + BCE + backward written ONLY in ``torch.ops.mi355seg.*`` calls (the dispatcher
    view of the C-ABI): layout in, (conv + BatchNorm + ReLU) units, max-pool, ConvTranspose k2 s2, channel concat, the pointwise
    head, layout out, the fused BCE / argmax / Dice tail -- against the reference fixture unet3d_f8_32.npz (unet3d.py:50-71,
    train.py:204-221): logits 1e-4, loss 1e-5, sampled weight gradients 2e-4 relative, BatchNorm running statistics 1e-5."""
    import os
    import numpy as np
    import mi355seg
    from mi355seg import custom_ops as C
    from mi355seg.models.three_d.unet3d import UNet3D
    from oracle.fill import fill_module_, make_input, make_labels
    from oracle.step import two_channel_gt
    ops, RELU = torch.ops.mi355seg, mi355seg.functional.ACT_RELU
    g = np.load(os.path.join(golden_dir, "unet3d_f8_32.npz"))
    m = fill_module_(UNet3D(1, 2, 8)).cuda().train()        # parameter container only: its forward() is not called

    def block(h, blk):
        conv1, norm1, _r1, conv2, norm2, _r2 = blk.children()
        return C.conv_bn_act_train(C.conv_bn_act_train(h, conv1, norm1, RELU), conv2, norm2, RELU)

    x = make_input((2, 1, 32, 32, 32)).cuda()
    gt2 = two_channel_gt(make_labels((2, 1, 32, 32, 32))).float().cuda()
    h = ops.to_channels_last(x)
    skips = []
    for enc in (m.encoder1, m.encoder2, m.encoder3, m.encoder4):
        e = block(h, enc)
        skips.append(e)
        h, _idx = ops.max_pool3d_2x(e)
    h = block(h, m.bottleneck)
    for up, dec, skip in ((m.upconv4, m.decoder4, skips[3]), (m.upconv3, m.decoder3, skips[2]), (m.upconv2, m.decoder2, skips[1]),
                          (m.upconv1, m.decoder1, skips[0])):
        h = block(ops.cat_channels(ops.conv_transpose3d_k2s2(h, up.weight, up.bias), skip), dec)
    logits = ops.to_channels_first(ops.conv3d(h, m.conv.weight, m.conv.bias, 1, 0))
    loss, mask, counts = ops.bce_argmax_dice(logits, gt2)
    loss.backward()
    assert np.abs(logits.detach().cpu().numpy() - g["pred"]).max() < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    margin = np.abs(g["pred"][:, 0] - g["pred"][:, 1])[:, None]
    assert (mask.cpu().numpy().astype(np.uint8) == g["mask"])[margin > 2e-4].all()
    params, bufs = dict(m.named_parameters()), dict(m.named_buffers())

    def sample(t, k=4096):
        f = t.detach().reshape(-1)
        return f[::max(1, f.numel() // k)][:k].cpu().numpy()
    ngrad = 0
    for k in g.files:
        if k.startswith("grad/"):
            ref = g[k]
            assert np.abs(sample(params[k[5:]].grad) - ref).max() <= 2e-4 * max(1e-3, np.abs(ref).max()), k
            ngrad += 1
        elif k.startswith("buf/"):
            assert np.abs(bufs[k[4:]].cpu().numpy() - g[k]).max() < 1e-5, k
    assert ngrad >= 10 and all(p.grad is not None for p in m.parameters())
    assert all(int(b) == 1 for k, b in bufs.items() if k.endswith("num_batches_tracked"))
    # the norm ops on their own (BatchNorm and InstanceNorm forms) against the autograd.Function wrappers
    F = mi355seg.functional
    t = torch.randn(2, 6, 6, 8, 16, device="cuda")
    gam, bet = torch.rand(16, device="cuda") + 0.5, torch.randn(16, device="cuda")
    rm1, rv1, rm2, rv2 = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda"), torch.zeros(16, device="cuda"), torch.ones(16, device="cuda")
    a, ga, ba = t.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ya = C.batch_norm_act(a, ga, ba, rm1, rv1, 0.1, 1e-5, RELU, 0.0)
    ya.square().sum().backward()
    b, gb, bb = t.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yb = F.batch_norm_act(b, gb, bb, rm2, rv2, True, 0.1, 1e-5, RELU, 0.0)
    yb.square().sum().backward()
    assert torch.equal(ya, yb) and torch.equal(a.grad, b.grad) and torch.equal(ga.grad, gb.grad) and torch.equal(ba.grad, bb.grad)
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    mean, rstd = ops.norm_stats(t, None, None, 0.0, 1e-5, True)
    assert torch.equal(ops.norm_apply_act(t, mean, rstd, None, None, None, F.ACT_LRELU, 0.01, True), F.instance_norm_act(t, 1e-5, F.ACT_LRELU, 0.01))
    assert torch.equal(ops.activation(t, F.ACT_ELU, 1.0), F.activation(t, F.ACT_ELU, 1.0))
    torch.library.opcheck(ops.norm_apply_act.default, (t, mean[:16].contiguous(), rstd[:16].contiguous(), gam, bet, None, RELU, 0.0, False),
                          test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(ops.cat_channels.default, (t, t), test_utils=("test_schema", "test_faketensor"))
