"""GPU parity tests: every HIP entry point (called through the C-ABI via the package's
autograd wrappers) against the CPU oracle arithmetic (ATen CPU ops == the reference's
arithmetic) on the same seeded inputs.  Tolerance: 1e-4 absolute on O(1) outputs
(north_star: "logits/Dice within 1e-4 fp32"), bit-exact for integer work."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

TOL = 1e-4
DEFAULT_MATH = "f16x3"         # the library default (include/mi355seg.h, MI355SEG_MATH_DEFAULT)


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mi355seg
    mi355seg.lib()          # raises if the HIP library is missing -- no fallback
    return mi355seg


def cl(x):   # NCDHW cpu -> channel-last gpu
    return x.permute(0, 2, 3, 4, 1).contiguous().cuda()


def cf(y):   # channel-last gpu -> NCDHW cpu
    return y.detach().cpu().permute(0, 4, 1, 2, 3).contiguous()


def rel_err(a, b):
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # N, D, H, W, Cin, Cout, k, s, p
    (2, 8, 8, 8, 1, 8, 3, 1, 1),
    (1, 6, 10, 12, 8, 16, 3, 1, 1),
    (2, 16, 16, 16, 32, 32, 3, 1, 1),
    (1, 8, 8, 32, 32, 64, 3, 1, 1),
    (1, 16, 16, 16, 64, 32, 3, 1, 1),
    (2, 8, 8, 8, 128, 128, 3, 1, 1),
    (1, 4, 4, 4, 256, 512, 3, 1, 1),
    (1, 32, 32, 32, 32, 32, 3, 1, 1),
    (1, 32, 64, 64, 32, 64, 3, 1, 1),      # MFMA tile BX=32, MB=2, NBW=2
    (1, 64, 64, 16, 16, 128, 3, 1, 1),     # BX=16, MB=2
    (1, 64, 64, 8, 16, 256, 3, 1, 1),      # BX=8, MB=2
    (2, 16, 16, 16, 48, 96, 3, 1, 1),      # 3 chunks, NBW=1
    (1, 8, 8, 16, 4, 32, 3, 1, 1),         # direct stem kernel, Cin=4
    (1, 4, 8, 32, 1, 32, 3, 1, 1),         # LDS-tiled Cin=1 stem (TX=32)
    (2, 6, 8, 64, 1, 16, 3, 1, 1),         # LDS-tiled Cin=1 stem (TX=64)
    (2, 8, 8, 8, 32, 2, 1, 1, 0),          # direct head kernel, Cout=2
    (1, 4, 4, 8, 256, 4, 1, 1, 0),         # direct head kernel, 64 lanes per voxel
    (1, 16, 16, 32, 64, 32, 1, 1, 0),      # k1 on the MFMA igemm + MFMA pointwise wgrad
    (2, 5, 7, 9, 96, 48, 1, 1, 0),         # k1, ragged tile and half-empty channel blocks (bf16x6: pw_wgrad_lowp planes)
    (1, 8, 8, 64, 1, 16, 5, 1, 2),         # V-Net k5 stem: z-marching fwd + LDS-tiled k5 wgrad (TX = 64)
    (2, 6, 8, 10, 2, 2, 1, 1, 0),          # tiny pointwise 2 -> 2 (V-Net out_tr.conv2)
    (1, 4, 6, 8, 3, 4, 1, 1, 0),           # tiny pointwise 3 -> 4
    (2, 8, 8, 8, 32, 64, 1, 1, 0),
    (1, 6, 6, 6, 64, 64, 3, 1, 1),         # partial MFMA tiles (UNETR 6^3 level)
    (2, 12, 12, 12, 32, 64, 3, 1, 1),      # partial tiles, W = 12
    (1, 10, 12, 20, 32, 32, 3, 1, 1),      # cfg-4 style 20 x 24 x 20 family
    (1, 5, 7, 9, 32, 32, 3, 1, 1),         # odd extents
    (1, 6, 6, 6, 64, 32, 1, 1, 0),         # k1, partial tiles
    (1, 8, 8, 8, 16, 16, 5, 1, 2),
    (2, 8, 8, 16, 1, 16, 5, 1, 2),         # V-Net in_tr: small-Cin direct wgrad, k5
    (1, 8, 8, 16, 32, 2, 5, 1, 2),         # V-Net out_tr.conv1: small-Cout direct wgrad, k5
    (2, 8, 8, 8, 2, 2, 1, 1, 0),           # V-Net out_tr.conv2: 2 -> 2, k1
    (1, 37, 11, 45, 32, 2, 5, 1, 2),       # z-marching head kernels: two z segments, ragged x / y tiles, 2 channel passes
    (2, 9, 10, 33, 16, 2, 3, 1, 1),        # z-marching head kernels, k3, one pass
    (1, 6, 9, 7, 48, 2, 5, 1, 2),          # three channel passes (12 input quads -> two dgrad wave groups)
    (1, 35, 9, 37, 2, 8, 5, 1, 2),         # k5 stem with two input channels on the narrow -> wide z-march kernel
    (1, 34, 8, 32, 1, 48, 5, 1, 2),        # k5 stem, 12 output quads (two wave groups), two z segments
    (1, 8, 8, 8, 4, 8, 2, 2, 0),           # small-Cin strided
    (1, 8, 12, 32, 32, 64, 5, 1, 2),       # k5 on the MFMA igemm (V-Net LUConv), CK = 8
    (2, 6, 6, 6, 8, 32, 5, 1, 2),          # k5, partial tiles, single chunk
    (1, 16, 16, 16, 128, 128, 5, 1, 2),    # k5 plane-wise MFMA wgrad, BX = 16
    (2, 5, 9, 8, 64, 32, 5, 1, 2),         # k5 MFMA wgrad, BX = 8, odd extents
    (1, 3, 6, 40, 32, 32, 5, 1, 2),        # k5 MFMA wgrad, BX = 8 (W = 40), D smaller than the halo
    (2, 8, 8, 8, 8, 16, 2, 2, 0),
    (1, 16, 16, 32, 32, 64, 3, 2, 1),      # gather igemm: k3 s2 p1 (Res-U-Net down conv), dgrad in 8 phases
    (1, 9, 11, 34, 16, 32, 3, 2, 1),       # gather igemm, odd extents
    (1, 8, 16, 32, 32, 64, 2, 2, 0),       # gather igemm: k2 s2 (V-Net down conv)
    (2, 8, 8, 16, 16, 32, 2, 2, 0),        # V-Net down_tr32: MFMA fwd, generic dgrad (Cin = 16)
    (1, 8, 8, 32, 16, 32, 4, 4, 0),        # k4 s4, 64 taps
    (1, 6, 6, 12, 16, 32, 2, 1, 0),        # even kernel, stride 1
    (1, 9, 9, 17, 64, 32, 3, 2, 0),        # k3 s2 without padding, CK = 64
    (1, 8, 12, 8, 16, 32, 3, 2, 1),
    (2, 8, 8, 8, 16, 4, 1, 1, 0),
    (1, 16, 16, 16, 1, 8, 16, 16, 0),
    (1, 7, 9, 11, 3, 5, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd_bwd(seg, case):
    N, D, H, W, Cin, Cout, k, s, p = case
    F = seg.functional
    x = rnd(N, Cin, D, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, k, seed=2, scale=(2.0 / (Cin * k ** 3)) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = TF.conv3d(xr, wr, br, stride=s, padding=p)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)

    xg, wg, bg = cl(x).requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = F.conv3d(xg, wg, bg, s, p)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < TOL
    assert rel_err(cf(xg.grad), xr.grad) < TOL
    assert rel_err(wg.grad.cpu(), wr.grad) < TOL
    assert rel_err(bg.grad.cpu(), br.grad) < TOL


def test_conv3d_fused_batch_statistics(seg):
    """conv epilogue statistics (sum, sum of squares per output channel) == those of y."""
    import ctypes
    F = seg.functional
    L = seg.lib()
    for (N, D, H, W, Cin, Cout) in [(2, 16, 16, 32, 32, 64), (1, 8, 8, 8, 4, 8), (2, 4, 8, 64, 1, 32)]:
        x = cl(rnd(N, Cin, D, H, W, seed=1))
        w = rnd(Cout, Cin, 3, 3, 3, seed=2, scale=0.1).cuda()
        b = rnd(Cout, seed=3).cuda()
        y = torch.empty(N, D, H, W, Cout, device="cuda")
        ssum = torch.zeros(Cout, dtype=torch.float64, device="cuda")
        ssq = torch.zeros(Cout, dtype=torch.float64, device="cuda")
        ws = F.workspace(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, 3, 1, 1), x.device)
        L.call("mi355seg_conv3d_fwd_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout,
               3, 1, 1, ssum.data_ptr(), ssq.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        yd = y.double().reshape(-1, Cout)
        assert ((ssum - yd.sum(0)).abs() / yd.abs().sum(0)).max() < 1e-6
        assert ((ssq - (yd * yd).sum(0)).abs() / (yd * yd).sum(0)).max() < 1e-6


def test_conv3d_channel_slices_and_no_bias(seg):
    """Inputs / grads that are channel slices of wider buffers (pitch > C), bias=None."""
    F = seg.functional
    N, D, H, W, Cin, Cout = 1, 8, 8, 8, 32, 32
    big = rnd(N, 2 * Cin, D, H, W, seed=5)
    w = rnd(Cout, Cin, 3, 3, 3, seed=6, scale=0.05)
    xr = big[:, Cin:].clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = TF.conv3d(xr, wr, None, padding=1)
    gbig = rnd(N, 2 * Cout, D, H, W, seed=7)
    yr.backward(gbig[:, :Cout])
    bigg = cl(big)
    xg = bigg[..., Cin:].detach().requires_grad_(True)        # strided view, pitch 2*Cin
    wg = w.cuda().requires_grad_(True)
    yg = F.conv3d(xg, wg, None, 1, 1)
    yg.backward(cl(gbig)[..., :Cout])
    assert (cf(yg) - yr.detach()).abs().max() < TOL
    assert rel_err(cf(xg.grad), xr.grad) < TOL
    assert rel_err(wg.grad.cpu(), wr.grad) < TOL


@pytest.mark.parametrize("case", [(2, 4, 4, 4, 16, 8), (1, 8, 8, 8, 64, 32), (1, 3, 5, 4, 6, 10), (2, 2, 2, 2, 512, 256),
                                  (2, 8, 8, 8, 512, 256), (1, 16, 16, 16, 128, 64), (1, 16, 32, 32, 64, 32),
                                  (1, 6, 6, 6, 64, 32), (1, 3, 5, 6, 32, 32), (2, 12, 12, 12, 128, 64),
                                  (1, 8, 8, 16, 64, 16), (2, 5, 6, 7, 32, 16), (1, 4, 4, 8, 32, 4),     # narrow Cout: flat (child, cout) tiles
                                  (1, 8, 8, 8, 256, 64), (1, 8, 8, 8, 128, 16)])         # direct GEMM, streaming form at K = 256 / 128 (convt_direct.hip)
def test_conv_transpose3d_k2s2(seg, case):
    N, D, H, W, Cin, Cout = case
    F = seg.functional
    x = rnd(N, Cin, D, H, W, seed=1)
    w = rnd(Cin, Cout, 2, 2, 2, seed=2, scale=(2.0 / (Cin * 8)) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = TF.conv_transpose3d(xr, wr, br, stride=2)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)
    xg, wg, bg = cl(x).requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = F.conv_transpose3d_k2s2(xg, wg, bg)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < TOL
    assert rel_err(cf(xg.grad), xr.grad) < TOL
    assert rel_err(wg.grad.cpu(), wr.grad) < TOL
    assert rel_err(bg.grad.cpu(), br.grad) < TOL


@pytest.mark.parametrize("case", [
    # (N, D, H, W, Cin, Cout, lddx, expected K-splits): K = 8 Cout in 64-wide chunks, split while a half still holds two chunks
    (1, 5, 5, 4, 64, 32, 64, 2), (1, 5, 5, 4, 64, 64, 64, 4), (1, 4, 4, 4, 128, 128, 128, 8), (2, 8, 8, 8, 512, 256, 512, 8),
    (1, 3, 7, 5, 64, 64, 96, 4),        # ragged voxel count, dx written into a wider buffer
])
def test_conv_transpose3d_input_gradient_split_k(seg, case):
    """convt_direct.hip, GEMM-form input gradient of ConvTranspose3d k2 s2 on fp32 tensors (unet3d.py:47-53 backward): the deep levels
    split K over adjacent workgroups into slabs + a reduce.  Each split launch against the SAME entry point handed a workspace too small
    for the slabs (the library then runs the unsplit launch: identical products, another summation order) and against ATen-CPU."""
    N, D, H, W, Cin, Cout, lddx, ks = case
    F = seg.functional
    L = seg.lib()
    dev = "cuda"
    nvox = N * D * H * W
    dy = rnd(N, 2 * D, 2 * H, 2 * W, Cout, seed=1).to(dev)
    w = rnd(Cin, Cout, 2, 2, 2, seed=2, scale=(2.0 / (Cout * 8)) ** 0.5).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    full = L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout)
    slab = ks * nvox * Cin * 4
    assert full >= slab + 8 * Cin * Cout * 6, "the workspace query must cover the slabs"
    out = []
    packed = (8 * Cin * Cout * 6 + 255) // 256 * 256          # the bf16x6 planes of the packed weights: all the unsplit launch needs
    for nbytes in (full, packed + slab // 2):     # with room for the slabs / without: split / unsplit
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dx = torch.full((nvox, lddx), 7.0, device=dev)
        L.call("mi355seg_convt3d_k2s2_dgrad_f32", dy.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), lddx, N, D, H, W, Cin, Cout, ws.data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        assert bool((dx[:, Cin:] == 7.0).all()), "columns beyond Cin must stay untouched"
        out.append(dx[:, :Cin].clone())
    want = TF.conv3d(dy.permute(0, 4, 1, 2, 3).cpu(), w.cpu(), stride=2).permute(0, 2, 3, 4, 1).reshape(nvox, Cin)
    sc = float(want.abs().max())
    assert (out[0].cpu() - want).abs().max() < 2e-5 * sc
    assert (out[1].cpu() - want).abs().max() < 2e-5 * sc
    assert (out[0] - out[1]).abs().max() < 4e-6 * sc and not torch.equal(out[0], out[1]), "the two launches must differ in summation order only"


@pytest.mark.parametrize("case", [(1, 4, 4, 8, 64, 32, 4), (2, 3, 2, 5, 8, 4, 4), (1, 3, 4, 5, 16, 32, 3), (1, 2, 2, 4, 512, 128, 4)])
def test_conv_transpose3d_kernel_equals_stride(seg, case):
    """nn.ConvTranspose3d(k, stride=k) (csrnet.py:121-137, k = 4) as the adjoint of the matching Conv3d: MFMA gather
    paths when the channels allow (64 phase launches for k = 4), generic kernels otherwise."""
    N, D, H, W, Cin, Cout, k = case
    F = seg.functional
    x = rnd(N, Cin, D, H, W, seed=1)
    w = rnd(Cin, Cout, k, k, k, seed=2, scale=(2.0 / Cin) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = TF.conv_transpose3d(xr, wr, br, stride=k)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)
    xg, wg, bg = cl(x).requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = F.conv_transpose3d_adjoint(xg, wg, bg, k)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < TOL
    assert rel_err(cf(xg.grad), xr.grad) < TOL
    assert rel_err(wg.grad.cpu(), wr.grad) < TOL
    assert rel_err(bg.grad.cpu(), br.grad) < TOL


@pytest.mark.parametrize("C,act", [(32, "relu"), (16, "elu"), (2, "elu"), (64, "none"), (6, "relu")])
def test_batchnorm_act_train(seg, C, act):
    F = seg.functional
    N, D, H, W = 2, 8, 6, 10
    x = rnd(N, C, D, H, W, seed=1) * 2.0 + 0.5
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    rm, rv = 0.05 * rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=5).abs()
    res = rnd(N, C, D, H, W, seed=6) if act == "elu" else None
    actf = {"relu": torch.relu, "elu": TF.elu, "none": lambda z: z}[act]
    code = {"relu": F.ACT_RELU, "elu": F.ACT_ELU, "none": F.ACT_NONE}[act]

    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if res is not None else None
    rm_r, rv_r = rm.clone(), rv.clone()
    z = TF.batch_norm(xr, rm_r, rv_r, gr, br, training=True, momentum=0.1, eps=1e-5)
    yr = actf(z + rr if rr is not None else z)
    g = rnd(*yr.shape, seed=7)
    yr.backward(g)

    xg, gg, bg = cl(x).requires_grad_(True), gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
    rg = cl(res).requires_grad_(True) if res is not None else None
    rm_g, rv_g = rm.cuda(), rv.cuda()
    yg = F.batch_norm_act(xg, gg, bg, rm_g, rv_g, True, 0.1, 1e-5, code, 0.01, rg)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < TOL
    assert (rm_g.cpu() - rm_r).abs().max() < 1e-6 and (rv_g.cpu() - rv_r).abs().max() < 1e-5
    assert rel_err(cf(xg.grad), xr.grad) < 2e-4
    assert rel_err(gg.grad.cpu(), gr.grad) < TOL and rel_err(bg.grad.cpu(), br.grad) < TOL
    if res is not None:
        assert rel_err(cf(rg.grad), rr.grad) < TOL


def test_batchnorm_eval_and_instancenorm(seg):
    F = seg.functional
    N, C, D, H, W = 2, 16, 4, 6, 8
    x = rnd(N, C, D, H, W, seed=1) * 1.5 - 0.3
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    rm, rv = 0.05 * rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=5).abs()
    yr = torch.relu(TF.batch_norm(x, rm, rv, gamma, beta, training=False, eps=1e-5))
    yg = F.batch_norm_act(cl(x), gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), False, 0.1, 1e-5, F.ACT_RELU)
    assert (cf(yg) - yr).abs().max() < TOL
    # InstanceNorm3d (no affine) + LeakyReLU, forward/backward
    xr = x.clone().requires_grad_(True)
    yr = TF.leaky_relu(TF.instance_norm(xr, eps=1e-5), 0.01)
    g = rnd(*yr.shape, seed=6)
    yr.backward(g)
    xg = cl(x).requires_grad_(True)
    yg = F.instance_norm_act(xg, 1e-5, F.ACT_LRELU, 0.01)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < TOL
    assert rel_err(cf(xg.grad), xr.grad) < 2e-4


@pytest.mark.parametrize("act", ["relu", "elu", "lrelu"])
def test_activation(seg, act):
    F = seg.functional
    x = rnd(2, 8, 4, 4, 6, seed=1)
    res = rnd(2, 8, 4, 4, 6, seed=2)
    fn = {"relu": torch.relu, "elu": TF.elu, "lrelu": lambda z: TF.leaky_relu(z, 0.01)}[act]
    code = {"relu": F.ACT_RELU, "elu": F.ACT_ELU, "lrelu": F.ACT_LRELU}[act]
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = fn(xr + rr)
    g = rnd(*yr.shape, seed=3)
    yr.backward(g)
    xg, rg = cl(x).requires_grad_(True), cl(res).requires_grad_(True)
    yg = F.activation(xg, code, 0.01, rg)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < 1e-6
    assert (cf(xg.grad) - xr.grad).abs().max() < 1e-6 and (cf(rg.grad) - rr.grad).abs().max() < 1e-6


@pytest.mark.parametrize("C", [32, 3])
def test_maxpool_and_upsample(seg, C):
    F = seg.functional
    x = rnd(2, C, 8, 6, 10, seed=1)
    x[0, 0, :2, :2, :2] = 1.0          # exact ties: the first maximum in scan order must win
    xr = x.clone().requires_grad_(True)
    yr = TF.max_pool3d(xr, 2, 2)
    g = rnd(*yr.shape, seed=2)
    yr.backward(g)
    xg = cl(x).requires_grad_(True)
    yg = F.max_pool3d_2x(xg)
    yg.backward(cl(g))
    assert torch.equal(cf(yg), yr.detach())
    assert torch.equal(cf(xg.grad), xr.grad)
    xr = x.clone().requires_grad_(True)
    yr = TF.interpolate(xr, scale_factor=2, mode="nearest")
    g = rnd(*yr.shape, seed=3)
    yr.backward(g)
    xg = cl(x).requires_grad_(True)
    yg = F.upsample_nearest_2x(xg)
    yg.backward(cl(g))
    assert torch.equal(cf(yg), yr.detach())
    assert (cf(xg.grad) - xr.grad).abs().max() < 1e-5


def test_cat_and_repeat_channels(seg):
    """torch.cat(dim=1) / x.repeat(1, rep, 1, 1, 1) of the reference in channel-last form: bit-exact copies, exact adjoints."""
    F = seg.functional
    for ca, cb in ((8, 8), (3, 5), (16, 2)):
        a, b = rnd(2, 3, 4, 5, ca, seed=2).cuda().requires_grad_(True), rnd(2, 3, 4, 5, cb, seed=3).cuda().requires_grad_(True)
        y = F.cat_channels(a, b)
        assert torch.equal(y, torch.cat((a, b), dim=-1))
        g = rnd(2, 3, 4, 5, ca + cb, seed=4).cuda()
        y.backward(g)
        assert torch.equal(a.grad, g[..., :ca]) and torch.equal(b.grad, g[..., ca:])
    for c, rep in ((1, 16), (2, 8), (4, 4)):
        x = rnd(2, 3, 4, 5, c, seed=5).cuda().requires_grad_(True)
        y = F.repeat_channels(x, rep)
        assert torch.equal(y, x.repeat(1, 1, 1, 1, rep))
        g = rnd(2, 3, 4, 5, c * rep, seed=6).cuda()
        y.backward(g)
        want = g.double().view(2, 3, 4, 5, rep, c).sum(4)
        assert (x.grad.double() - want).abs().max() < 1e-5
    u, v = rnd(7, 13, 11, seed=7).cuda(), rnd(7, 13, 11, seed=8).cuda()
    assert torch.equal(F._mul(u, v), u * v)


GEMM_CASES = [
    # M, N, K, a_transposed, b_transposed, bias, relu, accumulate
    (216, 768, 768, False, True, True, False, False),     # Linear fwd (x @ W^T): 48 tiles -> split-K
    (216, 2048, 768, False, True, True, True, False),     # FFN w_1 + ReLU epilogue
    (216, 768, 2048, False, False, False, False, False),  # Linear dgrad (dy @ W)
    (768, 768, 216, True, False, False, False, False),    # Linear wgrad (dy^T @ x), K = tokens
    (37, 53, 29, False, False, True, False, True),        # ragged, unaligned strides -> scalar reads, accumulate
    (130, 70, 260, True, True, False, False, True),       # ragged split-K with both operands transposed
    (100, 72, 264, True, True, True, False, True),        # LDS-free 32x32 kernel, A along its outer index / B along k, edge tiles
    (45, 40, 64, False, True, False, True, False),        # ... both along k, one k-block per wave, edge tiles in M and N
    (33, 31, 8, False, False, True, False, False),        # ... a single k-block: three of the four waves have nothing to add
]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_strided(seg, case):
    """mi355seg_gemm_f32 (nn.Linear / torch.matmul of unetr.py:61-121 and their adjoints) against an fp64 matmul."""
    M, N, K, ta, tb, has_bias, relu, acc = case
    F = seg.functional
    A = rnd(*((K, M) if ta else (M, K)), seed=1).cuda()
    B = rnd(*((N, K) if tb else (K, N)), seed=2).cuda()
    bias = rnd(N, seed=3).cuda() if has_bias else None
    C0 = rnd(M, N, seed=4).cuda()
    C = C0.clone()
    a_rs, a_cs = (1, M) if ta else (K, 1)
    b_rs, b_cs = (1, K) if tb else (N, 1)
    F._gemm(A.data_ptr(), a_rs, a_cs, 0, 0, B.data_ptr(), b_rs, b_cs, 0, 0, C.data_ptr(), N, 0, 0,
            None if bias is None else bias.data_ptr(), M, N, K, alpha=0.5, relu=int(relu), accumulate=int(acc))
    want = 0.5 * ((A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()))
    if has_bias:
        want = want + bias.double()
    if acc:
        want = want + C0.double()
    if relu:
        want = want.clamp_min(0)
    assert (C.double() - want).abs().max() < 2e-5 * max(1.0, K ** 0.5)
    C2 = C0.clone()                                  # split-K sums in a fixed order: bitwise repeatable
    F._gemm(A.data_ptr(), a_rs, a_cs, 0, 0, B.data_ptr(), b_rs, b_cs, 0, 0, C2.data_ptr(), N, 0, 0,
            None if bias is None else bias.data_ptr(), M, N, K, alpha=0.5, relu=int(relu), accumulate=int(acc))
    assert torch.equal(C, C2)


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_lowp_matches_bf16_rounded_operands(seg, case):
    """mi355seg_gemm_lowp_f32 (the token GEMMs inside functional.autocast(torch.bfloat16): fp32 tensors, operands rounded to bf16 in
    registers, v_mfma_f32_32x32x16_bf16, fp32 accumulation -- the arithmetic of the reference's nn.Linear / matmul under
    torch.autocast(bfloat16), unetr.py:59-138) against an fp64 matmul of the bf16-ROUNDED operands: the only difference left is the fp32
    accumulation order.  The cases include K = 216 and 264 (not multiples of the MFMA's 16: the tail lanes read zeros), K = 8 and the
    shapes outside the small-GEMM kernel's range (those run the fp32 kernels: also within the bar against rounded operands? no --
    they are compared with the un-rounded product at the bf16 bound)."""
    M, N, K, ta, tb, has_bias, relu, acc = case
    F = seg.functional
    A = rnd(*((K, M) if ta else (M, K)), seed=1).cuda()
    B = rnd(*((N, K) if tb else (K, N)), seed=2).cuda()
    bias = rnd(N, seed=3).cuda() if has_bias else None
    C0 = rnd(M, N, seed=4).cuda()
    C = C0.clone()
    a_rs, a_cs = (1, M) if ta else (K, 1)
    b_rs, b_cs = (1, K) if tb else (N, 1)
    F._gemm(A.data_ptr(), a_rs, a_cs, 0, 0, B.data_ptr(), b_rs, b_cs, 0, 0, C.data_ptr(), N, 0, 0,
            None if bias is None else bias.data_ptr(), M, N, K, alpha=0.5, relu=int(relu), accumulate=int(acc), lowp=True)

    def product(a, b):
        w = 0.5 * ((a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double()))
        if has_bias:
            w = w + bias.double()
        if acc:
            w = w + C0.double()
        return w.clamp_min(0) if relu else w
    rounded = product(A.bfloat16().float(), B.bfloat16().float())
    exact = product(A, B)
    err_r, err_e = float((C.double() - rounded).abs().max()), float((C.double() - exact).abs().max())
    # either the bf16 kernel ran (fp32-accumulation distance from the rounded product) or the shape fell to the fp32 kernels (exact product)
    assert min(err_r, err_e) < 2e-5 * max(1.0, K ** 0.5), (err_r, err_e)
    assert err_e < 2.0 ** -7 * K ** 0.5 * 3.0               # and never further from the true product than bf16 rounding of the operands allows
    C2 = C0.clone()
    F._gemm(A.data_ptr(), a_rs, a_cs, 0, 0, B.data_ptr(), b_rs, b_cs, 0, 0, C2.data_ptr(), N, 0, 0,
            None if bias is None else bias.data_ptr(), M, N, K, alpha=0.5, relu=int(relu), accumulate=int(acc), lowp=True)
    assert torch.equal(C, C2)


@pytest.mark.parametrize("P", [216, 27])
def test_gemm_batched_attention_shapes(seg, P):
    """Per-(batch, head) Q K^T and P V with the head split expressed in strides (unetr.py:74-98)."""
    F = seg.functional
    Bn, H, d = 2, 3, 64
    E = H * d
    q, k, v = (rnd(Bn, P, E, seed=s).cuda() for s in (1, 2, 3))
    scores = torch.empty(Bn, H, P, P, device="cuda")
    F._gemm(q.data_ptr(), E, 1, P * E, d, k.data_ptr(), 1, E, P * E, d, scores.data_ptr(), P, H * P * P, P * P, None, P, P, d, Bn, H, 0.125)
    qh, kh, vh = (t.double().view(Bn, P, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
    assert (scores.double() - 0.125 * qh @ kh.transpose(-1, -2)).abs().max() < 1e-4
    out = torch.empty(Bn, P, E, device="cuda")
    F._gemm(scores.data_ptr(), P, 1, H * P * P, P * P, v.data_ptr(), E, 1, P * E, d, out.data_ptr(), E, P * E, d, None, P, d, P, Bn, H)
    want = (scores.double() @ vh).permute(0, 2, 1, 3).reshape(Bn, P, E)
    assert (out.double() - want).abs().max() < 1e-3


@pytest.mark.parametrize("C", [32, 64, 128, 12])
def test_reverse_attention_gate_and_sigmoid(seg, C):
    """enc * (1 - sigmoid(t)) + enc with a one-channel map (RE_net.py:104-107) and the sigmoid output activation."""
    F = seg.functional
    enc, t = rnd(2, C, 4, 5, 6, seed=1), rnd(2, 1, 4, 5, 6, seed=2) * 3
    er, tr = enc.clone().double().requires_grad_(True), t.clone().double().requires_grad_(True)
    yr = (-1 * torch.sigmoid(tr) + 1).expand(-1, C, -1, -1, -1).mul(er) + er
    g = rnd(*yr.shape, seed=3)
    yr.backward(g.double())
    eg, tg = cl(enc).requires_grad_(True), cl(t).requires_grad_(True)
    yg = F.reverse_attention_gate(eg, tg)
    yg.backward(cl(g))
    assert (cf(yg).double() - yr.detach()).abs().max() < 1e-5
    assert (cf(eg.grad).double() - er.grad).abs().max() < 1e-5
    assert (cf(tg.grad).double() - tr.grad).abs().max() < 1e-5 * max(1.0, float(tr.grad.abs().max()))
    xs = cl(enc).requires_grad_(True)
    ys = F.activation(xs, F.ACT_SIGMOID)
    ys.backward(cl(g))
    xr = enc.clone().double().requires_grad_(True)
    torch.sigmoid(xr).backward(g.double())
    assert (cf(ys).double() - torch.sigmoid(enc.double())).abs().max() < 1e-6
    assert (cf(xs.grad).double() - xr.grad).abs().max() < 1e-6


def test_selective_fusion_pieces(seg):
    """SFConv's voxel mean, two-branch softmax and weighted mix (ER_net.py:52-69) with their adjoints."""
    F = seg.functional
    N, C, D, H, W = 2, 32, 4, 6, 8
    x1, x2 = rnd(N, D, H, W, C, seed=1), rnd(N, D, H, W, C, seed=2)
    a, b = rnd(N, C, seed=3).abs(), rnd(N, C, seed=4).abs()
    g, gs = rnd(N, D, H, W, C, seed=5), rnd(N, C, seed=6)
    leaves = [t.clone().double().requires_grad_(True) for t in (x1, x2, a, b)]
    want_mix = leaves[0] * leaves[2][:, None, None, None, :] + leaves[1] * leaves[3][:, None, None, None, :]
    want_pool = (leaves[0] + leaves[1]).mean(dim=(1, 2, 3))
    ((want_mix * g.double()).sum() + (want_pool * gs.double()).sum()).backward()
    dev = [t.cuda().requires_grad_(True) for t in (x1, x2, a, b)]
    mix, pool = F.sf_mix(*dev), F.sf_pool(dev[0], dev[1])
    ((mix * g.cuda()).sum() + (pool * gs.cuda()).sum()).backward()
    assert (mix.detach().cpu().double() - want_mix.detach()).abs().max() < 1e-6
    assert (pool.detach().cpu().double() - want_pool.detach()).abs().max() < 1e-6
    for got, ref in zip(dev, leaves):
        assert (got.grad.cpu().double() - ref.grad).abs().max() < 2e-5
    logits = rnd(64, 2, seed=7)
    lg = logits.cuda().requires_grad_(True)
    lr = logits.clone().double().requires_grad_(True)
    w = rnd(64, 2, seed=8)
    (F.softmax_last(lg) * w.cuda()).sum().backward()
    (torch.softmax(lr, dim=1) * w.double()).sum().backward()
    assert (lg.grad.cpu().double() - lr.grad).abs().max() < 1e-6


def test_c_abi_rejects_bad_arguments(seg):
    """Error behaviour of the boundary: invalid geometry, pitches below the channel count, null pointers and short
    workspaces come back as error codes with a message (raised as Mi355SegError by the binding), never as a launch."""
    from mi355seg._lib import Mi355SegError
    L = seg.lib()
    x = torch.zeros(1, 4, 4, 4, 16, device="cuda")
    w = torch.zeros(32, 16, 3, 3, 3, device="cuda")
    y = torch.zeros(1, 4, 4, 4, 32, device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ok = (x.data_ptr(), 16, w.data_ptr(), None, y.data_ptr(), 32, 1, 4, 4, 4, 16, 32, 3, 1, 1, None, None, ws.data_ptr(), ws.numel(), st)
    L.call("mi355seg_conv3d_fwd_f32", *ok)
    bad = {
        "pitch below channels": ok[:1] + (8,) + ok[2:],
        "null input": (None,) + ok[1:],
        "zero depth": ok[:7] + (0,) + ok[8:],
        "kernel larger than the padded input": ok[:12] + (11,) + ok[13:],
        "short workspace": ok[:18] + (64,) + ok[19:],
    }
    for what, args in bad.items():
        with pytest.raises(Mi355SegError) as err:
            L.call("mi355seg_conv3d_fwd_f32", *args)
        assert "conv" in str(err.value) or "workspace" in str(err.value), what
    with pytest.raises(Mi355SegError):
        seg.functional.conv3d(x.cpu(), w, None, 1, 1)                      # CPU tensor: no fallback
    with pytest.raises(Mi355SegError):
        seg.functional.conv3d(x.double(), w, None, 1, 1)                   # fp32 only
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(1, 8, 8, 32, 32, 64, 3), (2, 6, 9, 40, 32, 32, 3), (1, 12, 12, 16, 64, 32, 3), (2, 8, 8, 8, 128, 128, 3),
                                  (1, 4, 4, 4, 256, 512, 3), (1, 5, 7, 13, 32, 64, 3), (1, 10, 12, 16, 32, 32, 5)])
def test_split_precision_conv_is_fp32_accurate(seg, case):
    """Conv math "bf16x6" (three bf16 parts per fp32 operand, six bf16 MFMAs per product, fp32 accumulate) against an fp64
    convolution of the UNROUNDED fp32 tensors, for the forward, the input gradient and the weight gradient: it must be as
    close to the truth as the exact-fp32 MFMA path is (the same 1e-6-level error), through the ordinary entry points."""
    N, D, H, W, Cin, Cout, k = case
    F = seg.functional
    pad = k // 2
    x, w, b = rnd(N, Cin, D, H, W, seed=1), rnd(Cout, Cin, k, k, k, seed=2, scale=(2.0 / (k ** 3 * Cin)) ** 0.5), rnd(Cout, seed=3, scale=0.1)
    g = rnd(N, Cout, D, H, W, seed=4)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    want = TF.conv3d(xd, wd, b.double(), padding=pad)
    want.backward(g.double())
    errs = {}
    try:
        # bf16x6 on both MFMA shapes: 16 = conv_x3s.hip (v_mfma_f32_16x16x32_bf16, where its 16-wide tiles apply), 32 = the generic kernel
        for math, shape in (("fp32", 16), ("bf16x6", 32), ("bf16x6", 16), ("f16x3", 16)):
            seg.set_conv_math(math)
            seg.set_x3_shape(shape)
            assert seg.get_conv_math() == math and seg.get_x3_shape() == shape
            xg = cl(x).requires_grad_(True)
            wg = w.cuda().requires_grad_(True)
            y = F.conv3d(xg, wg, b.cuda(), 1, pad)
            y.backward(cl(g))
            errs[(math, shape)] = ((cf(y).double() - want.detach()).abs().max().item(), (cf(xg.grad).double() - xd.grad).abs().max().item(),
                                   (wg.grad.cpu().double() - wd.grad).abs().max().item())
    finally:
        seg.set_conv_math(DEFAULT_MATH)
        seg.set_x3_shape(16)
    scales = (float(want.abs().max()), float(xd.grad.abs().max()), float(wd.grad.abs().max()))
    # f16x3 (two fp16 parts per operand under a per-tensor power-of-two scale, three fp16 MFMAs per product): the same criteria, unchanged
    for key in (("bf16x6", 32), ("bf16x6", 16), ("f16x3", 16)):
        for what, e6, e32, sc in zip(("fwd", "dgrad", "wgrad"), errs[key], errs[("fp32", 16)], scales):
            assert e6 < 3e-6 * max(1.0, sc), (key, what, e6, e32, sc)
            assert e6 < 4.0 * e32 + 2e-7 * max(1.0, sc), (key, what, e6, e32, sc)


@pytest.mark.parametrize("case", [(1, 32, 32, 32, 32, 32, 0), (1, 16, 16, 16, 32, 64, 0), (1, 16, 16, 16, 64, 64, 0), (2, 16, 16, 16, 128, 128, 0),
                                  (1, 9, 7, 20, 32, 64, 0), (2, 5, 6, 17, 64, 32, 0), (1, 16, 16, 48, 64, 32, 32), (1, 12, 8, 16, 32, 32, 16)])
def test_bf16x6_16x16x32_kernel_against_fp64_and_the_32x32x16_kernel(seg, case):
    """conv_x3s.hip (bf16x6 on v_mfma_f32_16x16x32_bf16: tap-paired K = 32 steps, piece-major LDS tile) through the C-ABI on every
    variant it has -- four / two lines per wave, 32- / 64-channel tiles, split-K slabs (few tiles), ragged extents, channel-slice
    pitches (ld > C) -- forward with bias + BatchNorm statistics and input gradient: error statistics against an fp64 convolution
    must match the generic 32x32x16 kernel's (same six products: the two are the same arithmetic in a different summation
    order) and the exact-fp32 MFMA path's."""
    N, D, H, W, Cin, Cout, extra = case
    F, L = seg.functional, seg.lib()
    ldx, ldy = Cin + extra, Cout + extra
    xw = rnd(N, D, H, W, ldx, seed=1)
    w = rnd(Cout, Cin, 3, 3, 3, seed=2, scale=(2.0 / (27 * Cin)) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    gw = rnd(N, D, H, W, ldy, seed=4)
    x, g = xw[..., :Cin], gw[..., :Cout]
    want = TF.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 4, 1)
    want_dx = torch.nn.grad.conv3d_input((N, Cin, D, H, W), w.double(), g.permute(0, 4, 1, 2, 3).double(), padding=1).permute(0, 2, 3, 4, 1)
    xg, gg, wg, bg = xw.cuda(), gw.cuda(), w.cuda(), b.cuda()
    ws = F.workspace(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, 3, 1, 1), xg.device)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for math, shape in (("fp32", 16), ("bf16x6", 32), ("bf16x6", 16), ("f16x3", 16)):
            seg.set_conv_math(math)
            seg.set_x3_shape(shape)
            y = torch.full((N, D, H, W, ldy), 7.0, device="cuda")
            dx = torch.full((N, D, H, W, ldx), 7.0, device="cuda")
            ssum = torch.zeros(Cout, dtype=torch.float64, device="cuda")
            ssq = torch.zeros_like(ssum)
            L.call("mi355seg_conv3d_fwd_f32", xg.data_ptr(), ldx, wg.data_ptr(), bg.data_ptr(), y.data_ptr(), ldy, N, D, H, W, Cin, Cout, 3, 1, 1,
                   ssum.data_ptr(), ssq.data_ptr(), ws.data_ptr(), ws.numel(), st)
            L.call("mi355seg_conv3d_dgrad_f32", gg.data_ptr(), ldy, wg.data_ptr(), dx.data_ptr(), ldx, N, D, H, W, Cin, Cout, 3, 1, 1,
                   ws.data_ptr(), ws.numel(), st)
            torch.cuda.synchronize()
            if extra:                                            # the channels beyond C of a slice pitch are not the kernel's to touch
                assert bool((y[..., Cout:] == 7.0).all()) and bool((dx[..., Cin:] == 7.0).all())
            ey, ed = y[..., :Cout].cpu().double() - want, dx[..., :Cin].cpu().double() - want_dx
            res[(math, shape)] = (ey, ed, ssum.cpu(), ssq.cpu())
    finally:
        seg.set_conv_math(DEFAULT_MATH)
        seg.set_x3_shape(16)
    sy, sd = float(want.abs().max()), float(want_dx.abs().max())
    rms = lambda e: float(e.pow(2).mean().sqrt())
    e32, ef = res[("bf16x6", 32)], res[("fp32", 16)]
    nvox = N * D * H * W
    for key in (("bf16x6", 16), ("f16x3", 16)):                                        # the two forms of conv_x3s.hip
        e16 = res[key]
        for i, sc in ((0, sy), (1, sd)):
            assert float(e16[i].abs().max()) < 3e-6 * sc, key
            assert rms(e16[i]) <= 1.25 * max(rms(e32[i]), rms(ef[i])) + 1e-9 * sc, (key, rms(e16[i]), rms(e32[i]), rms(ef[i]))   # same error level as the other two maths
            assert abs(float(e16[i].mean())) <= 2e-7 * sc, key                          # no systematic offset
        assert torch.allclose(e16[2], want.sum(dim=(0, 1, 2, 3)), rtol=0, atol=2e-6 * sy * nvox)
        assert torch.allclose(e16[3], want.pow(2).sum(dim=(0, 1, 2, 3)), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("case", [(1, 8, 16, 32, 32, 64), (2, 5, 9, 40, 64, 128), (1, 3, 7, 21, 32, 64), (1, 16, 48, 64, 128, 128),
                                  (2, 16, 16, 32, 64, 32), (1, 9, 7, 40, 128, 32), (1, 32, 32, 64, 64, 32)],
                         ids=lambda c: "x".join(map(str, c)))
def test_f16x3_wide_weight_gradient(seg, case):
    """(r5: the last three cases have Cin % 64 == 0 and Cout = 32 -- the wide kernel then runs with the operands' roles SWAPPED, x the
    centred 64-channel operand, dy the haloed one, mirrored taps, slabs transposed back by wgrad_reduce_swapped: dec1conv1 of cfg 2.)
    conv_wgrad_f16w_kernel (f16x3, k3 s1, Cout % 64 == 0: four waves, a 32 x 64 channel block per workgroup; include/mi355seg.h,
    mi355seg_set_wgrad_wide): forced on wherever the geometry allows (mode 2: full, ragged and BX = 8 tiles, D smaller than a tile,
    and -- last case -- a shape the default mode 1 sends there), against the fp64 weight gradient by the criteria of
    test_split_precision_conv_is_fp32_accurate, and against the 32 x 32 kernel (mode 0): both sum the same products."""
    N, D, H, W, Cin, Cout = case
    F = seg.functional
    x, w = rnd(N, Cin, D, H, W, seed=11), rnd(Cout, Cin, 3, 3, 3, seed=12, scale=(2.0 / (27 * Cin)) ** 0.5)
    g = rnd(N, Cout, D, H, W, seed=13) * 1e-6                   # gradients of a mean loss are small: the scale must not matter
    wd = w.double().requires_grad_(True)
    TF.conv3d(x.double(), wd, None, padding=1).backward(g.double())
    got = {}
    try:
        for mode in (0, 2, 1):
            seg.set_wgrad_wide(mode)
            assert seg.get_wgrad_wide() == mode
            xg, wg = cl(x).requires_grad_(True), w.cuda().requires_grad_(True)
            F.conv3d(xg, wg, None, 1, 1).backward(cl(g))
            got[mode] = wg.grad.cpu().double()
    finally:
        seg.set_wgrad_wide(1)
    sc = float(wd.grad.abs().max())
    for mode in (0, 2, 1):
        assert float((got[mode] - wd.grad).abs().max()) < 3e-6 * sc, mode
    assert float((got[2] - got[0]).abs().max()) < 1e-6 * sc


def test_f16x3_wide_weight_gradient_at_cfg2_size(seg):
    """BASELINE cfg 2's largest layer the wide kernel takes (2 x 128^3, 32 -> 64) at full size, through two size-independent properties:
    the wide and the 8-wave kernel sum the same 8.4 M products per element in different fp32 orders (each is within 3e-6 of the fp64 value
    by the small-shape tests: 6e-6 of the scale between them), and the weight gradient is linear in the output gradient
    (dW(g1 + 2 g2) = dW(g1) + 2 dW(g2) to the same bar)."""
    F = seg.functional
    N, D, Cin, Cout = 2, 128, 32, 64
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, D, D, D, Cin, device="cuda", generator=gen)
    g1 = torch.randn(N, D, D, D, Cout, device="cuda", generator=gen) * 1e-6
    g2 = torch.randn(N, D, D, D, Cout, device="cuda", generator=gen) * 1e-6
    w = (torch.randn(Cout, Cin, 3, 3, 3, device="cuda", generator=gen) * 0.05).requires_grad_(True)

    def dw(g, mode):
        seg.set_wgrad_wide(mode)
        try:
            w.grad = None
            F.conv3d(x, w, None, 1, 1).backward(g)
            return w.grad.double()
        finally:
            seg.set_wgrad_wide(1)

    a1, a0 = dw(g1, 1), dw(g1, 0)
    sc = float(a0.abs().max())
    assert float((a1 - a0).abs().max()) < 6e-6 * sc
    b1 = dw(g2, 1)
    c1 = dw(g1 + 2 * g2, 1)
    assert float((c1 - (a1 + 2 * b1)).abs().max()) < 6e-6 * float(c1.abs().max())


@pytest.mark.parametrize("xs,ws,tail", [(1.0, 1.0, 0.0), (1e-8, 1.0, 0.0), (3e-30, 0.02, 0.0), (1e6, 1e-3, 0.0), (1e18, 1e12, 0.0), (1.0, 1.0, 1e3), (1e-9, 1.0, 3e4),
                                        (0.0, 1.0, 0.0)])
def test_f16x3_is_scale_free(seg, xs, ws, tail):
    """The two-piece fp16 split runs under per-tensor power-of-two scales taken from the tensors' own maxima, so its accuracy must
    not depend on where the values sit in the fp32 range: gradients of ~1e-8 (mean-BCE), tensors near either end of the fp32
    exponent range, a tensor whose largest value is 1e3 - 3e4 times its typical one (everything else then sits that far down in
    the fp16 range: the low parts go subnormal and must still count), an all-zero tensor.  Forward and input gradient of
    conv_x3s.hip against fp64, graded against the exact-fp32 path's own error on the same inputs."""
    N, D, H, W, Cin, Cout = 1, 8, 8, 32, 64, 64
    F, L = seg.functional, seg.lib()
    x = rnd(N, D, H, W, Cin, seed=1) * xs
    if tail:
        x.view(-1)[12345] = tail * xs
        x.view(-1)[777] = -0.7 * tail * xs
    w = rnd(Cout, Cin, 3, 3, 3, seed=2, scale=(2.0 / (27 * Cin)) ** 0.5) * ws
    want = TF.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), None, padding=1).permute(0, 2, 3, 4, 1)
    xg, wg = x.cuda(), w.cuda()
    ws_ = F.workspace(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, 3, 1, 1), xg.device)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for math in ("fp32", "f16x3"):
            seg.set_conv_math(math)
            y = torch.empty((N, D, H, W, Cout), device="cuda")
            dx = torch.empty((N, D, H, W, Cin), device="cuda")
            L.call("mi355seg_conv3d_fwd_f32", xg.data_ptr(), Cin, wg.data_ptr(), None, y.data_ptr(), Cout, N, D, H, W, Cin, Cout, 3, 1, 1,
                   None, None, ws_.data_ptr(), ws_.numel(), st)
            # the same tensor as an incoming gradient of the transposed-role convolution (Cin == Cout here)
            L.call("mi355seg_conv3d_dgrad_f32", xg.data_ptr(), Cout, wg.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, 3, 1, 1,
                   ws_.data_ptr(), ws_.numel(), st)
            torch.cuda.synchronize()
            res[math] = (y.cpu().double() - want, dx.cpu().double() - torch.nn.grad.conv3d_input((N, Cin, D, H, W), w.double(), x.permute(0, 4, 1, 2, 3).double(), padding=1).permute(0, 2, 3, 4, 1))
    finally:
        seg.set_conv_math(DEFAULT_MATH)
    rms = lambda e: float(e.pow(2).mean().sqrt())
    for i in (0, 1):
        eh, ef = res["f16x3"][i], res["fp32"][i]
        assert bool(torch.isfinite(eh).all())
        if xs == 0.0:
            assert float(eh.abs().max()) == 0.0
            continue
        assert rms(eh) <= 1.5 * rms(ef), (i, rms(eh), rms(ef))
        assert float(eh.abs().max()) <= 2.0 * float(ef.abs().max()), (i, float(eh.abs().max()), float(ef.abs().max()))


@pytest.mark.parametrize("shape", [(2, 5, 4, 6, 7), (2, 2, 4, 6, 8), (1, 3, 4, 6, 8), (3, 4, 2, 6, 10), (2, 2, 3, 5, 7), (1, 1, 4, 4, 4)])
def test_layout_roundtrip(seg, shape):
    """32 x 32 transpose tiles, the narrow-channel kernel (C <= 4, S % 4 == 0) and its fall-back (S % 4 != 0)."""
    F = seg.functional
    x = rnd(*shape, seed=1)
    y = F.to_channels_last(x.cuda())
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 4, 1).contiguous())
    assert torch.equal(F.to_channels_first(y).cpu(), x)


def test_bce_argmax_dice(seg):
    from oracle.metric import confusion_counts, metric as ometric
    F = seg.functional
    N, K, D, H, W = 2, 2, 8, 10, 12
    logits = rnd(N, K, D, H, W, seed=1) * 3
    logits[0, :, 0, 0, :4] = 0.25                         # exact ties -> class 0 must win
    lab = (rnd(N, 1, D, H, W, seed=2) > 0.8).float()
    tgt = torch.cat([1 - lab, lab], 1)
    lr = logits.clone().requires_grad_(True)
    loss_r = TF.binary_cross_entropy_with_logits(lr, tgt)
    (loss_r * 1.7).backward()
    lg = logits.cuda().requires_grad_(True)
    loss_g = F.bce_with_logits(lg, tgt.cuda())
    (loss_g * 1.7).backward()
    assert abs(loss_g.item() - loss_r.item()) < 1e-6
    assert (lg.grad.cpu() - lr.grad).abs().max() < 1e-9 + 1e-5 * lr.grad.abs().max()
    mask_r = logits.argmax(1, keepdim=True)
    mask_g = F.argmax_channels(logits.cuda())
    assert torch.equal(mask_g.cpu(), mask_r)
    gt_r = tgt.argmax(1, keepdim=True)
    cnt = F.dice_counts(gt_r.cuda(), mask_g).cpu().tolist()
    c = confusion_counts(gt_r.numpy(), mask_r.numpy())
    assert cnt == [c["gdth_sum"], c["pred_sum"], c["intersection_sum"], c["union_sum"]]
    loss_f, mask_f, cnt_f = F.bce_argmax_dice(logits.cuda(), tgt.cuda())
    assert abs(loss_f.item() - loss_r.item()) < 1e-6
    assert torch.equal(mask_f.cpu(), mask_r) and cnt_f.cpu().tolist() == cnt
    from mi355seg.utils.metric import metric
    j, d = metric(gt_r.cuda(), mask_g)
    jo, do = ometric(gt_r, mask_r)
    assert abs(j - jo) < 1e-12 and abs(d - do) < 1e-12
    # multi-class labels: bitwise & / | quirk of utils/metric.py:40-41
    a = torch.randint(0, 4, (2, 1, 4, 4, 4), generator=torch.Generator().manual_seed(3))
    b = torch.randint(0, 4, (2, 1, 4, 4, 4), generator=torch.Generator().manual_seed(4))
    assert metric(a.cuda(), b.cuda()) == ometric(a, b)
    s = F.dice_sums(logits.cuda(), tgt.cuda(), apply_sigmoid=True).cpu()
    pr = torch.sigmoid(logits.double())
    ref = torch.stack([(pr * tgt).sum(), pr.sum(), tgt.double().sum(), (pr * pr).sum(), (tgt * tgt).double().sum()])
    assert ((s - ref).abs() / ref.abs().clamp_min(1)).max() < 1e-6


def test_dice_counters_bit_exact_against_reference_metric_fixture(seg, golden_dir):
    """mi355seg_dice_counts_i64 (through utils.metric.metric and F.dice_counts) against what the reference's own
    utils/metric.py:20-75 counted and returned for eleven mask pairs (tests/golden/metric.npz, generated by executing the
    reference function lifted from its file): the four integer counters and the (jaccard, dice) doubles, bit for bit."""
    import os
    from mi355seg.utils.metric import metric
    F = seg.functional
    g = np.load(os.path.join(golden_dir, "metric.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) >= 8
    for n in names:
        gt, pred = torch.from_numpy(g[n + "/gt"]), torch.from_numpy(g[n + "/pred"])
        cnt = F.dice_counts(gt.to(torch.int64).cuda(), pred.to(torch.int64).cuda()).cpu().tolist()
        assert cnt == g[n + "/counts"].tolist(), n
        assert list(metric(gt.cuda(), pred.cuda())) == g[n + "/jaccard_dice"].tolist(), n


def test_missing_library_fails_loudly(seg, monkeypatch):
    import importlib
    L = importlib.import_module(seg.__name__ + "._lib")
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libmi355seg.so")
    with pytest.raises(L.Mi355SegError):
        L._Lib()
    with pytest.raises(L.Mi355SegError):
        seg.functional.conv3d(torch.zeros(1, 2, 2, 2, 1), torch.zeros(1, 1, 3, 3, 3), None, 1, 1)   # CPU tensor


@pytest.mark.parametrize("offset,scale", [(0.3, 4.5), (50.0, 0.5), (-3.0, 1e-2)])
@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 128), (1, 16, 12, 20, 32), (2, 4, 6, 10, 6)])
def test_norm_statistics_are_accurate_and_shift_invariant(seg, shape, offset, scale):
    """mean / rstd / running_var against float64, including channels whose |mean| >> std."""
    F = seg.functional
    L = seg.lib()
    N, D, H, W, C = shape
    x = (rnd(N, D, H, W, C, seed=9) * scale + offset).cuda()
    rows = N * D * H * W
    mean = torch.empty(C, device="cuda"); rstd = torch.empty(C, device="cuda")
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    ws = F.workspace(L.query("mi355seg_norm_ws_bytes", rows, 1, C), x.device)
    L.call("mi355seg_norm_stats_f32", x.data_ptr(), C, rows, 1, C, 1e-5, mean.data_ptr(), rstd.data_ptr(), rm.data_ptr(), rv.data_ptr(),
           0.1, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    xd = x.double().reshape(rows, C).cpu()
    m64, v64 = xd.mean(0), xd.var(0, unbiased=False)
    assert ((mean.cpu().double() - m64).abs() / (m64.abs() + v64.sqrt())).max() < 1e-6
    assert ((rstd.cpu().double() * (v64 + 1e-5).sqrt()) - 1).abs().max() < 2e-6
    ref_rv = 0.9 + 0.1 * xd.var(0, unbiased=True)
    assert ((rv.cpu().double() - ref_rv).abs() / ref_rv).max() < 1e-6


@pytest.mark.parametrize("C,res", [(32, True), (16, False), (6, True)])
def test_prelu_against_aten(seg, C, res):
    """nn.PReLU(C)(x [+ res]) (vnet3d.py:14-18, elu=False): values, dx, d(res) and the slope gradient."""
    F = seg.functional
    N, D, H, W = 2, 5, 6, 7
    x, r = rnd(N, C, D, H, W, seed=1), (rnd(N, C, D, H, W, seed=2) if res else None)
    a = 0.25 + 0.2 * rnd(C, seed=3)
    g = rnd(N, C, D, H, W, seed=4)
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    rr = None if r is None else r.clone().requires_grad_(True)
    yr = TF.prelu(xr + rr if res else xr, ar)
    yr.backward(g)
    xg, ag = cl(x).requires_grad_(True), a.cuda().requires_grad_(True)
    rg = None if r is None else cl(r).requires_grad_(True)
    yg = F.prelu(xg, ag, residual=rg)
    yg.backward(cl(g))
    assert (cf(yg) - yr.detach()).abs().max() < 1e-6
    assert (cf(xg.grad) - xr.grad).abs().max() < 1e-6
    if res:
        assert (cf(rg.grad) - rr.grad).abs().max() < 1e-6
    assert rel_err(ag.grad.cpu(), ar.grad) < 1e-5


@pytest.mark.parametrize("shape,C,K,act", [((2, 7, 9, 11), 32, 2, "relu"), ((1, 4, 4, 6), 8, 3, "relu"), ((1, 5, 6, 7), 64, 4, "lrelu"),
                                            ((1, 3, 3, 5), 256, 1, "relu"), ((1, 1, 1, 5), 32, 2, "relu"), ((2, 16, 16, 16), 32, 2, "elu")])
def test_bn_act_head_fused_kernels(seg, shape, C, K, act):
    """csrc/bn_head.hip through the C-ABI: BatchNorm + activation + 1x1x1 head as one forward kernel and the backward's two passes
    (column sums + head gradients; dy + its column sums + its maximum), against the chain of torch-CPU ops in fp64
    (/root/reference/models/three_d/unet3d.py:46-48,68-71,100-101: relu2(norm2(.)) then ``self.conv``) and, for the forward, BIT-FOR-BIT
    against the library's own unfused chain (norm_act_fwd + the k1 convolution) where that runs the same head kernel (K = 2, 4)."""
    F = seg.functional
    L = seg.lib()
    N, D, H, W = shape
    rows = N * D * H * W
    y = (rnd(rows, C, seed=1) * 1.7 + 0.3)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma, beta = 1 + 0.2 * rnd(C, seed=2), 0.3 * rnd(C, seed=3)
    wh, bh = rnd(K, C, seed=4) * 0.3, rnd(K, seed=5) * 0.1
    dl = rnd(rows, K, seed=6)
    code, slope = {"relu": (F.ACT_RELU, 0.0), "lrelu": (F.ACT_LRELU, 0.01), "elu": (F.ACT_ELU, 1.0)}[act]
    actf = {"relu": torch.relu, "lrelu": lambda z: TF.leaky_relu(z, 0.01), "elu": TF.elu}[act]
    # fp64 reference of the whole chain, training-mode batch statistics
    y64 = y.double().requires_grad_(True)
    g64, b64, w64, hb64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), wh.double().requires_grad_(True), bh.double().requires_grad_(True)
    m64 = y64.mean(0)
    xh = (y64 - m64) / torch.sqrt(y64.var(0, unbiased=False) + 1e-5)
    a64 = actf(xh * g64 + b64)
    lg64 = a64 @ w64.t() + hb64
    lg64.backward(dl.double())
    dev = "cuda"
    yg, mg, rg, gg, bg, wg, hbg, dlg = [t.to(dev).contiguous() for t in (y, mean, rstd, gamma, beta, wh, bh, dl)]
    st = torch.cuda.current_stream().cuda_stream
    lg = torch.empty(rows, K, device=dev)
    L.call("mi355seg_bn_act_head_fwd_f32", yg.data_ptr(), C, mg.data_ptr(), rg.data_ptr(), gg.data_ptr(), bg.data_ptr(), code, slope,
           wg.data_ptr(), hbg.data_ptr(), lg.data_ptr(), K, rows, C, K, st)
    assert (lg.cpu().double() - lg64.detach()).abs().max() < 2e-5 * max(1.0, float(lg64.abs().max()))
    if K in (2, 4) and C >= 16:
        a = torch.empty(rows, C, device=dev)
        L.call("mi355seg_norm_act_fwd_f32", yg.data_ptr(), C, mg.data_ptr(), rg.data_ptr(), gg.data_ptr(), bg.data_ptr(), None, 0,
               a.data_ptr(), C, rows, 1, C, code, slope, st)
        lg2 = F.conv3d(a.view(N, D, H, W, C), wg.view(K, C, 1, 1, 1), hbg, 1, 0).reshape(rows, K)
        assert torch.equal(lg, lg2)                             # same arithmetic as the unfused chain: identical logits
    ws = F.workspace(L.query("mi355seg_bn_act_head_ws_bytes", C, K), torch.device(dev))
    out = torch.zeros(4 * C + K * C + K, device=dev)
    s1, s2, dg, db = [out[i * C:(i + 1) * C] for i in range(4)]
    dwh, dbh = out[4 * C:4 * C + K * C], out[4 * C + K * C:]
    L.call("mi355seg_bn_act_head_bwd_sums_f32", dlg.data_ptr(), K, yg.data_ptr(), C, mg.data_ptr(), rg.data_ptr(), gg.data_ptr(), bg.data_ptr(), code, slope,
           wg.data_ptr(), s1.data_ptr(), s2.data_ptr(), dg.data_ptr(), db.data_ptr(), dwh.data_ptr(), dbh.data_ptr(), rows, C, K, ws.data_ptr(), ws.numel(), st)
    dy = torch.empty(rows, C, device=dev)
    col, amax = torch.zeros(C, device=dev), torch.zeros(1, device=dev)
    L.call("mi355seg_bn_act_head_bwd_apply_f32", dlg.data_ptr(), K, yg.data_ptr(), C, mg.data_ptr(), rg.data_ptr(), gg.data_ptr(), bg.data_ptr(), code, slope,
           wg.data_ptr(), s1.data_ptr(), s2.data_ptr(), dy.data_ptr(), C, col.data_ptr(), amax.data_ptr(), rows, C, K, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    tol = lambda ref: 3e-5 * max(1e-3, float(ref.abs().max()))
    assert (dg.cpu().double() - g64.grad).abs().max() < tol(g64.grad) and (db.cpu().double() - b64.grad).abs().max() < tol(b64.grad)
    assert torch.equal(s1, db) and torch.equal(s2, dg)
    assert (dwh.cpu().double().view(K, C) - w64.grad).abs().max() < tol(w64.grad)
    assert (dbh.cpu().double() - hb64.grad).abs().max() < tol(hb64.grad)
    assert (dy.cpu().double() - y64.grad).abs().max() < tol(y64.grad) * 3
    assert (col.cpu().double() - dy.cpu().double().sum(0)).abs().max() < 1e-4 * max(1e-3, float(dy.abs().max()))
    assert float(amax) == float(dy.abs().max())


@pytest.mark.parametrize("shape,C,lp,act", [((2, 8, 6, 10), 32, 32, "relu"), ((1, 4, 4, 4), 8, 0, "relu"), ((1, 2, 6, 18), 64, 64, "lrelu"),
                                             ((1, 4, 2, 2), 256, 0, "relu"), ((2, 16, 16, 16), 32, 32, "relu")])
def test_bn_act_pool_fused_kernels(seg, shape, C, lp, act):
    """csrc/bn_head.hip, the encoder form: BatchNorm + activation + MaxPool3d(2, 2) as one forward kernel -- BIT-FOR-BIT the library's
    own chain norm_act_fwd + maxpool2_fwd (activation into a channel slice of a wider buffer, pooled tensor, argmax codes, maximum) -- and the
    norm backward with d(act) = dskip + pool_backward(dpooled) formed on the fly, against the unfused chain maxpool2_bwd_add +
    norm_act_bwd_colsum (same per-element arithmetic; the column sums in another order).  /root/reference/models/three_d/unet3d.py:19-25,51-58,100-101."""
    F = seg.functional
    L = seg.lib()
    N, D, H, W = shape
    rows, prow = N * D * H * W, N * (D // 2) * (H // 2) * (W // 2)
    dev = "cuda"
    y = (rnd(rows, C, seed=1) * 1.7 + 0.3).to(dev)
    y[3, :4] = y[2, :4]                                       # an exact tie inside a window: the first maximum must win
    mean, var = y.mean(0), y.var(0, unbiased=False)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).contiguous()
    gamma, beta = (1 + 0.2 * rnd(C, seed=2)).to(dev), (0.3 * rnd(C, seed=3)).to(dev)
    code, slope = {"relu": (F.ACT_RELU, 0.0), "lrelu": (F.ACT_LRELU, 0.01)}[act]
    st = torch.cuda.current_stream().cuda_stream
    ld = lp + C
    full1, full2 = torch.zeros(rows, ld, device=dev), torch.zeros(rows, ld, device=dev)
    a1, a2 = full1[:, lp:], full2[:, lp:]
    p1, p2 = torch.empty(prow, C, device=dev), torch.empty(prow, C, device=dev)
    i1, i2 = torch.empty(prow, C, dtype=torch.uint8, device=dev), torch.empty(prow, C, dtype=torch.uint8, device=dev)
    am1, am2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    L.call("mi355seg_bn_act_pool_fwd_f32", y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), code, slope,
           a1.data_ptr(), ld, p1.data_ptr(), i1.data_ptr(), am1.data_ptr(), N, D, H, W, C, st)
    L.call("mi355seg_norm_act_fwd_ax_f32", y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0,
           a2.data_ptr(), ld, rows, 1, C, code, slope, am2.data_ptr(), st)
    L.call("mi355seg_maxpool2_fwd_f32", a2.data_ptr(), ld, p2.data_ptr(), C, i2.data_ptr(), N, D, H, W, C, st)
    torch.cuda.synchronize()
    assert torch.equal(full1, full2) and torch.equal(p1, p2) and torch.equal(i1, i2) and torch.equal(am1, am2)
    # backward
    dskip_full = rnd(rows, ld, seed=5).to(dev)
    dskip = dskip_full[:, lp:]
    dp = rnd(prow, C, seed=6).to(dev)
    ws = F.workspace(max(L.query("mi355seg_bn_act_pool_ws_bytes", C), L.query("mi355seg_norm_ws_bytes", rows, 1, C)), torch.device(dev))
    s12, dgb = torch.empty(2 * C, device=dev), torch.empty(2 * C, device=dev)
    dy1, col1, dam1 = torch.empty(rows, C, device=dev), torch.empty(C, device=dev), torch.zeros(1, device=dev)
    L.call("mi355seg_bn_act_pool_bwd_f32", dskip.data_ptr(), ld, dp.data_ptr(), i1.data_ptr(), y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(),
           gamma.data_ptr(), beta.data_ptr(), code, slope, s12.data_ptr(), s12.data_ptr() + 4 * C, dgb.data_ptr(), dgb.data_ptr() + 4 * C,
           dy1.data_ptr(), C, col1.data_ptr(), dam1.data_ptr(), N, D, H, W, C, ws.data_ptr(), ws.numel(), st)
    da = torch.empty(rows, C, device=dev)
    L.call("mi355seg_maxpool2_bwd_add_f32", dp.data_ptr(), C, i2.data_ptr(), dskip.data_ptr(), ld, da.data_ptr(), C, N, D, H, W, C, st)
    dy2, dg2, db2, col2, dam2 = torch.empty(rows, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev), torch.zeros(1, device=dev)
    L.call("mi355seg_norm_act_bwd_colsum_ax_f32", da.data_ptr(), C, y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0,
           dy2.data_ptr(), C, dg2.data_ptr(), db2.data_ptr(), None, 0, col2.data_ptr(), dam2.data_ptr(), rows, 1, C, code, slope, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    scale = lambda t: max(1e-3, float(t.abs().max()))
    assert (dgb[:C] - dg2).abs().max() < 2e-5 * scale(dg2) and (dgb[C:] - db2).abs().max() < 2e-5 * scale(db2)
    assert torch.equal(s12[:C], dgb[C:]) and torch.equal(s12[C:], dgb[:C])           # s1 = dbeta, s2 = dgamma
    assert (dy1 - dy2).abs().max() < 2e-5 * scale(dy2)
    assert (col1 - col2).abs().max() < 1e-4 * scale(dy2) and abs(float(dam1) - float(dam2)) < 2e-5 * scale(dy2)


@pytest.mark.parametrize("shape,chans", [((2, 16, 16, 32), (32, 64, 64)), ((1, 9, 7, 33), (64, 32, 32)), ((2, 8, 8, 16), (32, 32, 32)), ((1, 16, 16, 16), (128, 128, 128))])
def test_double_conv_with_norm_prologue_matches_the_unfused_block(seg, shape, chans, monkeypatch):
    """r5: conv2 of a double-conv block reads conv1's RAW output and applies norm1 + ReLU while it stages its tiles (forward:
    mi355seg_conv3d_fwd_pro_ax_f32, weight gradient: mi355seg_conv3d_wgrad_pro_ax_f32; /root/reference/models/three_d/unet3d.py:80-101).
    Same per-element arithmetic as norm_act_fwd + the plain kernels -- only the f16x3 scale of conv2's input differs (a bound from
    max |y1| instead of the measured maximum) -- so the block's output, every gradient and the running statistics must agree with the
    unfused block to fp32 rounding; interior tiles, border tiles (zero padding must stay zero behind the prologue) and ragged extents."""
    import os
    F = seg.functional
    from mi355seg.layers import BatchNorm3d, Conv3d
    N, D, H, W = shape
    c0, c1, c2 = chans
    torch.manual_seed(3)
    conv1, bn1, conv2, bn2 = Conv3d(c0, c1, 3, padding=1).cuda(), BatchNorm3d(c1).cuda(), Conv3d(c1, c2, 3, padding=1).cuda(), BatchNorm3d(c2).cuda()
    with torch.no_grad():
        bn1.weight.copy_(1 + 0.3 * torch.randn(c1)); bn1.bias.copy_(0.5 * torch.randn(c1))
        bn2.weight.copy_(1 + 0.3 * torch.randn(c2)); bn2.bias.copy_(0.2 * torch.randn(c2))
    x = (rnd(N, D, H, W, c0, seed=5) * 1.3).cuda()
    g = rnd(N, D, H, W, c2, seed=6).cuda()
    params = [conv1.weight, conv1.bias, bn1.weight, bn1.bias, conv2.weight, conv2.bias, bn2.weight, bn2.bias]

    def run(no_fusion):
        if no_fusion:
            monkeypatch.setenv("MI355SEG_NO_PRO_FUSION", "1")
        else:
            monkeypatch.delenv("MI355SEG_NO_PRO_FUSION", raising=False)
        for m in (bn1, bn2):
            m.running_mean.zero_(); m.running_var.fill_(1.0); m.num_batches_tracked.zero_()
        for p in params:
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.cuda.reset_peak_memory_stats()
        y = F.double_conv_bn_act(xi, conv1, bn1, conv2, bn2, F.ACT_RELU)
        y.backward(g)
        torch.cuda.synchronize()
        return (y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in params],
                [b.clone() for m in (bn1, bn2) for b in (m.running_mean, m.running_var)])

    fused = seg.lib().query("mi355seg_conv3d_pro_supported_f32", N, D, H, W, c1, c2, 3, 1, 1, F.ACT_RELU) == 1
    assert fused or shape == (1, 9, 7, 33)                    # (a geometry outside the f16x3 kernels' plans simply stays unfused)
    yf, dxf, gf, rf = run(False)
    yu, dxu, gu, ru = run(True)
    sc = lambda t: max(1e-6, float(t.abs().max()))
    assert (yf - yu).abs().max() < 2e-5 * sc(yu)
    assert (dxf - dxu).abs().max() < 1e-4 * sc(dxu)
    for name, a, b in zip(["w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2"], gf, gu):
        if name in ("b1", "b2"):             # bias gradients in front of a BatchNorm are rounding noise around an exact zero
            assert (a - b).abs().max() < 1e-3 * sc(gu[0])
        else:
            assert (a - b).abs().max() < 1e-4 * sc(b), name
    for a, b in zip(rf, ru):
        assert (a - b).abs().max() < 1e-6 * max(1.0, sc(b))


@pytest.mark.parametrize("shape,C,act", [((2, 8, 8, 32), 32, "relu"), ((1, 4, 8, 64), 16, "relu"), ((1, 6, 12, 32), 32, "lrelu"), ((2, 16, 16, 64), 32, "relu"),
                                         ((1, 4, 8, 128), 8, "relu"), ((1, 2, 8, 256), 4, "relu"), ((1, 4, 4, 16), 64, "lrelu")])
def test_stem_weight_gradient_with_norm_backward_prologue(seg, shape, C, act):
    """r5, mi355seg_stem_wgrad_bnbwd_f32: the 1-channel stem's weight + bias gradient formed straight from d(activation) and the pre-norm tensor
    (the norm backward's apply half inside the kernel; /root/reference/models/three_d/unet3d.py:80-89) against the library's own two-step chain
    norm_act_bwd_apply + conv3d_wgrad (same per-element expression: the weight gradients agree to fp32 summation order) and against fp64."""
    F = seg.functional
    L = seg.lib()
    N, D, H, W = shape
    rows = N * D * H * W
    dev = "cuda"
    x = rnd(rows, 1, seed=1).to(dev)
    y = (rnd(rows, C, seed=2) * 1.5 + 0.2).to(dev)
    da = rnd(rows, C, seed=3).to(dev)
    mean, rstd = y.mean(0), (1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)).contiguous()
    gamma, beta = (1 + 0.2 * rnd(C, seed=4)).to(dev), (0.3 * rnd(C, seed=5)).to(dev)
    code, slope = {"relu": (F.ACT_RELU, 0.0), "lrelu": (F.ACT_LRELU, 0.01)}[act]
    assert L.query("mi355seg_stem_wgrad_bnbwd_supported_f32", N, D, H, W, 1, C, 3, 1, 1) == 1
    st = torch.cuda.current_stream().cuda_stream
    ws = F.workspace(max(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, 1, C, 3, 1, 1), L.query("mi355seg_norm_ws_bytes", rows, 1, C)), torch.device(dev))
    s12, dgb = torch.empty(2 * C, device=dev), torch.empty(2 * C, device=dev)
    L.call("mi355seg_norm_act_bwd_sums_f32", da.data_ptr(), C, y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0,
           s12.data_ptr(), s12.data_ptr() + 4 * C, dgb.data_ptr(), dgb.data_ptr() + 4 * C, rows, 1, C, code, slope, ws.data_ptr(), ws.numel(), st)
    dw1, db1 = torch.empty(C, 1, 3, 3, 3, device=dev), torch.empty(C, device=dev)
    L.call("mi355seg_stem_wgrad_bnbwd_f32", da.data_ptr(), C, y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), code, slope,
           s12.data_ptr(), s12.data_ptr() + 4 * C, x.data_ptr(), 1, dw1.data_ptr(), db1.data_ptr(), N, D, H, W, 1, C, 3, 1, 1, ws.data_ptr(), ws.numel(), st)
    dy, db2 = torch.empty(rows, C, device=dev), torch.empty(C, device=dev)
    L.call("mi355seg_norm_act_bwd_apply_f32", da.data_ptr(), C, y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0,
           s12.data_ptr(), s12.data_ptr() + 4 * C, dy.data_ptr(), C, None, 0, db2.data_ptr(), rows, 1, C, code, slope, ws.data_ptr(), ws.numel(), st)
    dw2 = torch.empty(C, 1, 3, 3, 3, device=dev)
    L.call("mi355seg_conv3d_wgrad_f32", dy.data_ptr(), C, x.data_ptr(), 1, dw2.data_ptr(), None, N, D, H, W, 1, C, 3, 1, 1, 0, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    sc = max(1e-6, float(dw2.abs().max()))
    assert (dw1 - dw2).abs().max() < 2e-5 * sc
    assert (db1 - db2).abs().max() < 2e-4 * max(1e-6, float(dy.abs().max())) + 1e-5
    # fp64: dw[co][0][tap] = sum_v x[v + tap] dy[v][co] with dy from the unfused library kernel (itself covered by test_batchnorm_act_train)
    xv = x.view(N, 1, D, H, W).double().cpu()
    dyv = dy.view(N, D, H, W, C).permute(0, 4, 1, 2, 3).double().cpu()
    want = torch.nn.grad.conv3d_weight(xv, (C, 1, 3, 3, 3), dyv, padding=1)
    assert (dw1.cpu().double() - want).abs().max() < 3e-5 * float(want.abs().max())


@pytest.mark.parametrize("lowp", [0, 1])
@pytest.mark.parametrize("shape", [(216, 768, 768), (216, 3072, 768), (216, 768, 3072), (150, 200, 96), (64, 4096, 4096), (1100, 40, 64)])
def test_linear_weight_and_bias_gradient_in_one_call(seg, shape, lowp):
    """r5, mi355seg_linear_wgrad_f32: dw = dy^T x and db = column sums of dy (the backward of nn.Linear, unetr.py:61-66,120-121) -- on the
    token encoder's shapes db is summed inside the weight-gradient GEMM from the dy values it loads anyway.  dw must be BIT-identical to
    the plain GEMM entry point's (same kernel, same order), db equal to the separate column-sum kernel's to fp32 summation order and to
    fp64.  The last two shapes take the fallback inside the entry point (too many 32 x 32 tiles; rows beyond the small-GEMM kernel's)."""
    F, L = seg.functional, seg.lib()
    M, N, K = shape
    dev = "cuda"
    dy = rnd(M, N, seed=41).to(dev)
    x = rnd(M, K, seed=42).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = F.workspace(max(L.query("mi355seg_gemm_ws_bytes", N, K, M, 1, 1), L.query("mi355seg_norm_ws_bytes", M, 1, N)), torch.device(dev))
    dw1, db1 = torch.empty(N, K, device=dev), torch.full((N,), float("nan"), device=dev)
    L.call("mi355seg_linear_wgrad_f32", lowp, dy.data_ptr(), N, x.data_ptr(), K, dw1.data_ptr(), db1.data_ptr(), M, N, K, ws.data_ptr(), ws.numel(), st)
    dw2, db2 = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    L.call("mi355seg_gemm_lowp_f32" if lowp else "mi355seg_gemm_f32", dy.data_ptr(), 1, N, 0, 0, x.data_ptr(), K, 1, 0, 0, dw2.data_ptr(), K, 0, 0, None,
           N, K, M, 1, 1, 1.0, 0, 0, ws.data_ptr(), ws.numel(), st)
    L.call("mi355seg_colsum_f32", dy.data_ptr(), N, M, N, db2.data_ptr(), ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2)
    want = dy.double().sum(0)
    bound = 4e-6 * dy.double().abs().sum(0) + 1e-30
    assert ((db1.double() - want).abs() <= bound).all() and ((db2.double() - want).abs() <= bound).all()
    # without a bias: db = NULL is allowed
    dw3 = torch.empty(N, K, device=dev)
    L.call("mi355seg_linear_wgrad_f32", lowp, dy.data_ptr(), N, x.data_ptr(), K, dw3.data_ptr(), None, M, N, K, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    assert torch.equal(dw3, dw2)


@pytest.mark.parametrize("mask,gate", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("shape", [(216, 768, 768), (216, 2048, 768), (216, 768, 2048), (216, 2304, 768), (64, 40, 96)])
def test_linear_backward_as_one_launch(seg, shape, mask, gate):
    """r6, mi355seg_linear_bwd_f32: the backward of nn.Linear (unetr.py:61-66,98-100,120-138) as one launch -- dx = dyf W and dw = dyf^T x
    (+ db) as two problems of one grid, dyf = dy * mask * [y > 0] formed in the loads.  Every output must be BIT-identical to the chain it
    replaces (element-wise kernels, mi355seg_gemm_lowp_f32, mi355seg_linear_wgrad_f32)."""
    F, L = seg.functional, seg.lib()
    M, N, K = shape
    dev = "cuda"
    assert L.query("mi355seg_linear_bwd_supported_f32", 1, M, N, K) == 1
    dy = rnd(M, N, seed=61).to(dev)
    x = rnd(M, K, seed=62).to(dev)
    w = (rnd(N, K, seed=63) * 0.05).to(dev)
    mk = ((rnd(M, N, seed=64) > -0.8).float() / 0.9).to(dev) if mask else None
    y = (rnd(M, N, seed=65).clamp_min(0.0).to(dev) * (mk if mk is not None else 1.0)) if gate else None
    st = torch.cuda.current_stream().cuda_stream
    dx1, dw1, db1 = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    L.call("mi355seg_linear_bwd_f32", 1, dy.data_ptr(), N, mk.data_ptr() if mask else None, y.data_ptr() if gate else None, x.data_ptr(), K, w.data_ptr(),
           dx1.data_ptr(), dw1.data_ptr(), db1.data_ptr(), M, N, K, st)
    dyf = dy.clone()
    if mask:
        L.call("mi355seg_mul_f32", dyf.data_ptr(), mk.data_ptr(), dyf.data_ptr(), M * N, st)
    if gate:
        g = torch.empty_like(dyf)
        L.call("mi355seg_act_bwd_f32", dyf.data_ptr(), N, y.data_ptr(), N, None, 0, g.data_ptr(), N, M, N, F.ACT_RELU, 0.0, st)
        dyf = g
    ws = F.workspace(max(L.query("mi355seg_gemm_ws_bytes", N, K, M, 1, 1), L.query("mi355seg_norm_ws_bytes", M, 1, N)), torch.device(dev))
    dx2, dw2, db2 = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    L.call("mi355seg_gemm_lowp_f32", dyf.data_ptr(), N, 1, 0, 0, w.data_ptr(), K, 1, 0, 0, dx2.data_ptr(), K, 0, 0, None, M, K, N, 1, 1, 1.0, 0, 0, ws.data_ptr(), ws.numel(), st)
    L.call("mi355seg_linear_wgrad_f32", 1, dyf.data_ptr(), N, x.data_ptr(), K, dw2.data_ptr(), db2.data_ptr(), M, N, K, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2) and torch.equal(db1, db2)
    # and against fp64 on the bf16-rounded operands (fp32 accumulation)
    want = dyf.bfloat16().double() @ w.bfloat16().double()
    assert (dx1.double() - want).abs().max() < 2e-5 * float(want.abs().max()) + 1e-6


def test_two_batched_gemms_in_one_launch(seg):
    """r6, mi355seg_gemm_pair_lowp_f32: attention's backward pairs (unetr.py:74-98) -- dP = dO V^T with dV = Pd^T dO, dQ = a dS K with
    dK = a dS^T Q -- on the fused [B, P, 3E] layout, each bit-identical to its own mi355seg_gemm_lowp_f32 launch."""
    F, L = seg.functional, seg.lib()
    dev = "cuda"
    B, P, heads, d = 2, 216, 12, 64
    E, E3 = heads * d, 3 * heads * d
    qkv = rnd(B, P, E3, seed=71).to(dev)
    do = rnd(B, P, E, seed=72).to(dev)
    pd = rnd(B, heads, P, P, seed=73).to(dev)
    HPP, PP = heads * P * P, P * P
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * E, qkv.data_ptr() + 8 * E
    st = torch.cuda.current_stream().cuda_stream
    alpha = 0.125
    assert L.query("mi355seg_gemm_pair_supported_f32", P, P, d, P, d, P, B, heads) == 1
    dpd1, dqkv1 = torch.empty_like(pd), torch.zeros_like(qkv)
    dq1, dk1, dv1 = dqkv1.data_ptr(), dqkv1.data_ptr() + 4 * E, dqkv1.data_ptr() + 8 * E
    L.call("mi355seg_gemm_pair_lowp_f32", do.data_ptr(), E, 1, P * E, d, v, 1, E3, P * E3, d, dpd1.data_ptr(), P, HPP, PP, P, P, d, 1.0,
           pd.data_ptr(), 1, P, HPP, PP, do.data_ptr(), E, 1, P * E, d, dv1, E3, P * E3, d, P, d, P, 1.0, B, heads, st)
    L.call("mi355seg_gemm_pair_lowp_f32", pd.data_ptr(), P, 1, HPP, PP, k, E3, 1, P * E3, d, dq1, E3, P * E3, d, P, d, P, alpha,
           pd.data_ptr(), 1, P, HPP, PP, q, E3, 1, P * E3, d, dk1, E3, P * E3, d, P, d, P, alpha, B, heads, st)
    dpd2, dqkv2 = torch.empty_like(pd), torch.zeros_like(qkv)
    dq2, dk2, dv2 = dqkv2.data_ptr(), dqkv2.data_ptr() + 4 * E, dqkv2.data_ptr() + 8 * E

    def gemm(*a):
        L.call("mi355seg_gemm_lowp_f32", *a, None, 0, st)
    gemm(do.data_ptr(), E, 1, P * E, d, v, 1, E3, P * E3, d, dpd2.data_ptr(), P, HPP, PP, None, P, P, d, B, heads, 1.0, 0, 0)
    gemm(pd.data_ptr(), 1, P, HPP, PP, do.data_ptr(), E, 1, P * E, d, dv2, E3, P * E3, d, None, P, d, P, B, heads, 1.0, 0, 0)
    gemm(pd.data_ptr(), P, 1, HPP, PP, k, E3, 1, P * E3, d, dq2, E3, P * E3, d, None, P, d, P, B, heads, alpha, 0, 0)
    gemm(pd.data_ptr(), 1, P, HPP, PP, q, E3, 1, P * E3, d, dk2, E3, P * E3, d, None, P, d, P, B, heads, alpha, 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(dpd1, dpd2) and torch.equal(dqkv1, dqkv2)
    assert float(dqkv1.abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 8, 12, 16, 16), (2, 5, 6, 7, 12), (1, 4, 4, 8, 64)])
def test_activation_fork_sums_both_gradients_in_one_pass(seg, shape, dtype):
    """r5, functional.activation_fork / mi355seg_act_bwd_add_*: (lrelu(x), x) whose backward is d(pass-through) + d(act) * lrelu'(x)
    (the residual forks of residual_unet3d.py:110-121) against autograd on the two separate uses; 12 channels: the fallback inside the
    entry point (activation backward, then the in-place sum)."""
    F = seg.functional
    x = rnd(*shape, seed=51).cuda().to(dtype)
    g1, g2 = rnd(*shape, seed=52).cuda().to(dtype), rnd(*shape, seed=53).cuda().to(dtype)
    xa = x.clone().requires_grad_(True)
    a, p = F.activation_fork(xa, F.ACT_LRELU, 0.01)
    assert torch.equal(p.detach(), x)
    (a.float() * g1.float()).sum().backward(retain_graph=True)
    only_act = xa.grad.clone()
    xa.grad = None
    ((a.float() * g1.float()).sum() + (p.float() * g2.float()).sum()).backward()
    xr = x.clone().float().requires_grad_(True)
    ar = torch.nn.functional.leaky_relu(xr, 0.01)
    (ar * g1.float()).sum().backward(retain_graph=True)
    ref_act = xr.grad.clone()
    xr.grad = None
    ((ar * g1.float()).sum() + (xr * g2.float()).sum()).backward()
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -7
    assert (a.detach().float() - ar.detach()).abs().max() <= tol * max(1.0, float(ar.abs().max()))
    assert (only_act.float() - ref_act).abs().max() <= tol * max(1.0, float(ref_act.abs().max()))
    assert (xa.grad.float() - xr.grad).abs().max() <= tol * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("lowp", [False, True])
@pytest.mark.parametrize("case", [(216, 768, 768, False), (216, 3072, 768, True), (150, 200, 96, False), (64, 4096, 4096, False)])
def test_linear_with_mask_and_residual_in_the_gemm_epilogue(seg, case, lowp):
    """r5, functional.linear(..., mask=, residual=) / mi355seg_linear_fwd_f32: y = relu?(x W^T + b) * mask + residual in the GEMM's epilogue
    (unetr.py:98-100,120-138,159-166) against the same chain in ATen, forward and all gradients (x, W, b, residual); the last shape takes
    the fallback inside the entry point (the GEMM, then the element-wise kernels in place).  lowp: bf16 products (compared with ATen on
    bf16-rounded operands at a correspondingly looser tolerance)."""
    import torch.nn.functional as TF
    F = seg.functional
    M, N, K, relu = case
    g = torch.Generator().manual_seed(7)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g) * 0.1
    keep = (torch.rand(M, N, generator=g) > 0.1).float() / 0.9
    res = None if relu else torch.randn(M, N, generator=g)
    go = torch.randn(M, N, generator=g)
    rd = lambda t: t.to(torch.bfloat16).float() if lowp else t
    xr, wr, br = rd(x).clone().requires_grad_(True), rd(w).clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr = None if res is None else res.clone().requires_grad_(True)
    yr = TF.linear(xr, wr, br)
    yr = (torch.relu(yr) if relu else yr) * keep
    if rr is not None:
        yr = yr + rr
    yr.backward(go)
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rg = None if res is None else res.cuda().requires_grad_(True)
    with F.autocast(torch.bfloat16 if lowp else torch.float32):
        yg = F.linear(xg, wg, bg, relu=relu, mask=keep.cuda(), residual=rg)
    yg.backward(go.cuda())
    tol = 2e-2 if lowp else 2e-5
    sc = lambda t: max(1.0, float(t.abs().max()))
    assert (yg.detach().cpu() - yr.detach()).abs().max() < tol * sc(yr)
    for a, r in ((xg, xr), (wg, wr), (bg, br)) + (((rg, rr),) if rr is not None else ()):
        assert (a.grad.cpu() - r.grad).abs().max() < tol * sc(r.grad)
    if rr is not None:
        assert torch.equal(rg.grad.cpu(), go)          # the residual's gradient is dy itself


def test_layer_norm_fork_sums_both_gradients_in_the_norm_backward(seg):
    """r5, functional.layer_norm_fork / mi355seg_layernorm_bwd_add_f32: (LayerNorm(x), x) whose backward is d(pass-through) + LN-backward
    (x + f(LN(x)), unetr.py:159-166) against ATen's layer_norm with the two uses kept apart; one output unused: the other's gradient alone."""
    import torch.nn.functional as TF
    F = seg.functional
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 108, 768, generator=g) * 2 + 0.5
    ga, be = torch.rand(768, generator=g) + 0.5, torch.randn(768, generator=g)
    g1, g2 = torch.randn(2, 108, 768, generator=g), torch.randn(2, 108, 768, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    yr = TF.layer_norm(xr, (768,), gr, br, 1e-6)
    ((yr * g1).sum() + (xr * g2).sum()).backward()
    xg, gg, bg = x.cuda().requires_grad_(True), ga.cuda().requires_grad_(True), be.cuda().requires_grad_(True)
    n, p = F.layer_norm_fork(xg, gg, bg, 1e-6)
    assert torch.equal(p.detach().cpu(), x) and (n.detach().cpu() - yr.detach()).abs().max() < 1e-5
    ((n * g1.cuda()).sum() + (p * g2.cuda()).sum()).backward()
    assert (xg.grad.cpu() - xr.grad).abs().max() < 2e-5 * max(1.0, float(xr.grad.abs().max()))
    assert (gg.grad.cpu() - gr.grad).abs().max() < 1e-4 * max(1.0, float(gr.grad.abs().max()))
    assert (bg.grad.cpu() - br.grad).abs().max() < 1e-4 * max(1.0, float(br.grad.abs().max()))
    # only the pass-through is used: its gradient comes back as it is
    x2 = x.cuda().requires_grad_(True)
    _, p2 = F.layer_norm_fork(x2, gg.detach(), bg.detach(), 1e-6)
    (p2 * g2.cuda()).sum().backward()
    assert torch.equal(x2.grad.cpu(), g2)
