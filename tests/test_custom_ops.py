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
