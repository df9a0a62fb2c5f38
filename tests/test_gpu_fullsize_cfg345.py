"""BASELINE-size checks for cfg 3 / 4 / 5 (V-Net [2,1,128^3], Residual U-Net [1,4,160,192,160], UNETR [1,1,96^3]) through
size-independent properties, in the arithmetic the configs name (bf16) and in fp32:
  (a) a whole train step is bitwise deterministic (no atomics anywhere) and finite;
  (b) crops of full-size k5 / k3-stride-2 / InstanceNorm / nearest-upsample results, taken at the LAST voxels of the
      largest tensors (1.26 G elements at 160x192x160x64: where 32-bit offsets would wrap), equal ATen-CPU on the crop's
      receptive field;
  (c) InstanceNorm output is standardised per (sample, channel) at full extent."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mi355seg
    mi355seg.lib()
    return mi355seg


def _rnd(shape, seed, device="cpu"):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(device)


def _step_twice(seg, build, xshape, classes, dtype):
    from mi355seg.engine import weights_init_normal
    F = seg.functional

    def run():
        torch.manual_seed(0)
        m = build()
        m.apply(weights_init_normal("kaiming"))
        m = m.cuda().train()
        # (Dropout3d / Dropout stay on: both runs seed the device RNG identically, so the keep-masks are the same)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(99)
        x = torch.randn(xshape, generator=g).cuda()
        lab = torch.randint(0, classes, (xshape[0], 1) + tuple(xshape[2:]), generator=g).cuda()
        tgt = torch.cat([(lab == i).float() for i in range(classes)], dim=1)
        opt.zero_grad(set_to_none=True)
        with seg.autocast(dtype):
            pred = m(x)
        loss = F.bce_with_logits(pred, tgt)
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])
        opt.step()
        return pred.detach(), loss.item(), grads
    p1, l1, g1 = run()
    p2, l2, g2 = run()
    assert p1.dtype == torch.float32 and tuple(p1.shape) == (xshape[0], classes) + tuple(xshape[2:])
    assert torch.equal(p1, p2) and l1 == l2 and torch.equal(g1, g2)
    assert np.isfinite(l1) and bool(torch.isfinite(g1).all()) and bool(torch.isfinite(p1).all())
    return l1


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_cfg3_vnet_full_size_step_is_deterministic(seg, dtype):
    from mi355seg.models.three_d.vnet3d import VNet
    _step_twice(seg, lambda: VNet(in_channels=1, classes=2), (2, 1, 128, 128, 128), 2, dtype)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_cfg4_resunet_full_size_step_is_deterministic(seg, dtype):
    from mi355seg.models.three_d.residual_unet3d import UNet
    _step_twice(seg, lambda: UNet(4, 4, 32), (1, 4, 160, 192, 160), 4, dtype)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_cfg5_unetr_full_size_step_is_deterministic(seg, dtype):
    from mi355seg.models.three_d.unetr import UNETR
    _step_twice(seg, lambda: UNETR(img_shape=(96, 96, 96), input_dim=1, output_dim=2), (1, 1, 96, 96, 96), 2, dtype)


def _crop_check(x, w, b, y, k, stride, pad, spots, bf16):
    """y[n, z:z+4, ...] against ATen-CPU fp64 on the receptive field of the crop (x: channel-last GPU tensor)."""
    N, D, H, W, _ = x.shape
    for (n, z, yy, xx) in spots:
        lo = [s * stride - pad for s in (z, yy, xx)]
        hi = [(s + 3) * stride - pad + k for s in (z, yy, xx)]
        sl = [slice(max(0, a), min(e, b_)) for a, b_, e in zip(lo, hi, (D, H, W))]
        crop = x[n, sl[0], sl[1], sl[2]].float().permute(3, 0, 1, 2)[None].cpu().double()
        padl = [max(0, -a) for a in lo]
        padr = [max(0, b_ - e) for b_, e in zip(hi, (D, H, W))]
        crop = TF.pad(crop, (padl[2], padr[2], padl[1], padr[1], padl[0], padr[0]))
        wd = (w.to(BF).double() if bf16 else w.double()).cpu()
        ref = TF.conv3d(crop, wd, None if b is None else b.double().cpu(), stride=stride)[0].permute(1, 2, 3, 0)
        got = y[n, z:z + 4, yy:yy + 4, xx:xx + 4].float().cpu().double()
        tol = (2.0 ** -8 * ref.abs() + 1e-4) if bf16 else torch.full_like(ref, 1e-4)
        assert bool(((got - ref).abs() <= tol).all()), (n, z, yy, xx, float((got - ref).abs().max()))


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_cfg3_k5_conv_crops_at_the_last_voxels(seg, dtype):
    """V-Net's LUConv (vnet3d.py:21-31): k5 p2 32 -> 32 at [2, 128^3]; crops incl. the last voxels of the last sample."""
    F = seg.functional
    x = _rnd((2, 128, 128, 128, 32), 1, "cuda").to(dtype)
    w = (_rnd((32, 32, 5, 5, 5), 2) * (2.0 / 4000) ** 0.5).cuda()
    b = _rnd((32,), 3).cuda()
    y = F.conv3d(x, w, b, 1, 2)
    _crop_check(x, w, b, y, 5, 1, 2, [(0, 0, 0, 0), (1, 124, 124, 124), (0, 61, 2, 124), (1, 124, 0, 64)], dtype == BF)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_cfg4_strided_conv_norm_upsample_at_full_extent(seg, dtype):
    """Residual U-Net at [1, 160, 192, 160]: the k3 s2 p1 32 -> 64 down-convolution (residual_unet3d.py:30-33), the k3 s1
    64 -> 64 convolution on its 1.26 G-element neighbour tensor, InstanceNorm3d + LeakyReLU and nearest upsampling."""
    F = seg.functional
    bf = dtype == BF
    x = _rnd((1, 160, 192, 160, 32), 1, "cuda").to(dtype)
    w = (_rnd((64, 32, 3, 3, 3), 2) * (2.0 / 864) ** 0.5).cuda()
    y = F.conv3d(x, w, None, 2, 1)
    assert tuple(y.shape) == (1, 80, 96, 80, 64)
    _crop_check(x, w, None, y, 3, 2, 1, [(0, 0, 0, 0), (0, 76, 92, 76), (0, 40, 1, 76)], bf)
    del y
    # the largest activation of cfg 4: [1, 160, 192, 160, 64] = 314 M elements (629 MB in bf16, 1.26 GB in fp32)
    big = _rnd((1, 160, 192, 160, 64), 4, "cuda").to(dtype)
    w2 = (_rnd((64, 64, 3, 3, 3), 5) * (2.0 / 1728) ** 0.5).cuda()
    y2 = F.conv3d(big, w2, None, 1, 1)
    _crop_check(big, w2, None, y2, 3, 1, 1, [(0, 156, 188, 156), (0, 0, 0, 0), (0, 80, 188, 2)], bf)
    del y2
    # InstanceNorm3d + LeakyReLU(0.01) over 4.9 M voxels per channel: standardised before the activation
    n = F.instance_norm_act(big, 1e-5, F.ACT_NONE)
    nd = n.float().reshape(-1, 64).double()
    assert nd.mean(0).abs().max() < (2e-3 if bf else 1e-5) and (nd.var(0, unbiased=False) - 1).abs().max() < (5e-3 if bf else 1e-4)
    a = F.instance_norm_act(big, 1e-5, F.ACT_LRELU, 0.01)
    tail = a[0, -2:, -2:, -2:].float().cpu()
    want = TF.leaky_relu(n[0, -2:, -2:, -2:].float().cpu(), 0.01)
    assert (tail - want).abs().max() <= (2.0 ** -7 if bf else 1e-6)
    del n, a
    # nearest x2 upsampling of [1, 80, 96, 80, 64]: every child equals its parent, at the far corner too
    small = big[:, :80, :96, :80].contiguous()
    up = F.upsample_nearest_2x(small)
    assert tuple(up.shape) == (1, 160, 192, 160, 64)
    assert torch.equal(up[0, -1, -1, -1], small[0, -1, -1, -1]) and torch.equal(up[0, 158, 191, 1], small[0, 79, 95, 0])
    assert torch.equal(up[0, ::2, ::2, ::2], small[0]) and torch.equal(up[0, 1::2, 1::2, 1::2], small[0])


def test_cfg3_cfg4_narrow_layers_at_the_last_voxels(seg):
    """The layers that run on the narrow-axis MFMA kernels, at BASELINE extents, crops incl. the far corner (bf16):
    V-Net's k5 two-channel head at [2, 128^3] (conv_head2_lowp), the Residual U-Net's four-channel stem and its pointwise
    four-channel head at [1, 160, 192, 160] (conv_stem4_lowp, conv_headpw_lowp) -- forward values against ATen-CPU on the
    crop's receptive field, and the weight gradients against a chunked fp64 reduction over the whole volume."""
    F = seg.functional
    # V-Net head: Conv3d(32, 2, k5, p2)
    x = _rnd((2, 128, 128, 128, 32), 11, "cuda").to(BF)
    w = (_rnd((2, 32, 5, 5, 5), 12) * (2.0 / 4000) ** 0.5).cuda()
    b = _rnd((2,), 13).cuda()
    y = F.conv3d(x, w, b, 1, 2)
    assert tuple(y.shape) == (2, 128, 128, 128, 2)
    _crop_check(x, w, b, y, 5, 1, 2, [(0, 0, 0, 0), (1, 124, 124, 124), (0, 63, 1, 110), (1, 124, 0, 26)], True)
    del x, y
    # Residual U-Net stem: Conv3d(4, 32, k3, p1, bias=False)
    x = _rnd((1, 160, 192, 160, 4), 14, "cuda").to(BF)
    w = (_rnd((32, 4, 3, 3, 3), 15) * (2.0 / 108) ** 0.5).cuda()
    y = F.conv3d(x, w, None, 1, 1)
    _crop_check(x, w, None, y, 3, 1, 1, [(0, 0, 0, 0), (0, 156, 188, 156), (0, 77, 3, 129)], True)
    # ... its weight gradient over all 4.9 M voxels: dW[co][ci][tap] = sum_v dy[v][co] x[v + tap][ci], by z-slabs in fp64
    xr = x.detach().requires_grad_(True)
    wr = w.detach().requires_grad_(True)
    g = _rnd((1, 160, 192, 160, 32), 16, "cuda").to(BF)
    F.conv3d(xr, wr, None, 1, 1).backward(g)
    want = torch.zeros(32, 4, 3, 3, 3, dtype=torch.float64)
    xp = TF.pad(x[0].float().permute(3, 0, 1, 2), (1, 1, 1, 1, 1, 1)).double().cpu()        # [4, 162, 194, 162]
    gd = g[0].float().permute(3, 0, 1, 2).double().cpu()                                    # [32, 160, 192, 160]
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                want[:, :, dz, dy, dx] = torch.einsum("ozyx,izyx->oi", gd, xp[:, dz:dz + 160, dy:dy + 192, dx:dx + 160])
    err = (wr.grad.double().cpu() - want).abs().max()
    assert err <= 2e-5 * max(float(want.abs().max()), (160 * 192 * 160) ** 0.5), float(err)
    del x, y, xr, g, xp, gd
    # Residual U-Net head: Conv3d(32, 4, k1)
    x = _rnd((1, 160, 192, 160, 32), 17, "cuda").to(BF)
    w = (_rnd((4, 32, 1, 1, 1), 18) * (2.0 / 32) ** 0.5).cuda()
    b = _rnd((4,), 19).cuda()
    y = F.conv3d(x, w, b, 1, 0)
    _crop_check(x, w, b, y, 1, 1, 0, [(0, 0, 0, 0), (0, 156, 188, 156), (0, 5, 188, 77)], True)
