"""Every weight packing of a step in one launch (csrc/prepack.hip, functional.prepacked_weights): a model trained with its packings formed
all at once must end up with bit-identical parameters to the same model packing in place, under every conv math, for the fp32
U-Net (conv_x3s / generic / ConvT packings), the bf16 Residual U-Net (the one packing of a stride-2 input gradient's phases, modules
applied twice per step) and inside a captured HIP graph."""
import pytest
import torch

from oracle.fill import fill_module_, make_input, make_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mi355seg
    mi355seg.lib()
    return mi355seg


def _train(model_fn, x, gt, steps, prepack, monkeypatch, dtype=None, graphed=False):
    from mi355seg import functional as F
    from mi355seg.engine import train_step, GraphedTrainStep
    monkeypatch.setattr(F, "_PREPACK_OFF", not prepack)
    m = fill_module_(model_fn()).cuda().train()
    if graphed:
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
        g = GraphedTrainStep(m, opt, x, gt, warmup=2, dtype=dtype)
        losses = [float(g.first["loss"])]
        for _ in range(steps - 2):
            losses.append(float(g(x, gt, sync_metric=False)["loss"]))
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        losses = [float(train_step(m, opt, x, gt, sync_metric=False, dtype=dtype)["loss"]) for _ in range(steps)]
    torch.cuda.synchronize()
    plans = m.__dict__.get("_seg_prepack") or {}
    jobs = sum(F.lib().query("mi355seg_prepack_jobs", r["plan"]) for r in plans.values())
    return [p.detach().clone() for p in m.parameters()], losses, jobs


@pytest.mark.parametrize("math", ["f16x3", "bf16x6", "fp32"])
def test_unet_prepacked_steps_are_bit_identical(seg, monkeypatch, math):
    from mi355seg.models.three_d.unet3d import UNet3D
    seg.set_conv_math(math)
    try:
        x = make_input((2, 1, 32, 32, 32)).cuda()
        gt = make_labels((2, 1, 32, 32, 32)).cuda()
        fn = lambda: UNet3D(in_channels=1, out_channels=2, init_features=16)
        pa, la, ja = _train(fn, x, gt, 4, True, monkeypatch)
        pb, lb, jb = _train(fn, x, gt, 4, False, monkeypatch)
    finally:
        seg.set_conv_math(seg.DEFAULT_CONV_MATH)
    # forward + input-gradient packings of the k3 convolutions and the four ConvT layers (the exact-fp32 MFMA layout is not a replayed one)
    assert ja >= (20 if math != "fp32" else 0) and jb == 0, (ja, jb)
    assert la == lb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_res_unet_bf16_prepacked_steps_are_bit_identical(seg, monkeypatch):
    from mi355seg.models.three_d.residual_unet3d import UNet
    x = make_input((1, 4, 32, 48, 32)).cuda()
    gt = make_labels((1, 1, 32, 48, 32)).cuda()
    from mi355seg import functional as F
    # Dropout3d draws from the device generator: the same seed in front of both runs
    fn = lambda: UNet(in_channels=4, n_classes=2, base_n_filter=16)
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    pa, la, ja = _train(fn, x, gt, 3, True, monkeypatch, dtype=torch.bfloat16)
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    pb, lb, jb = _train(fn, x, gt, 3, False, monkeypatch, dtype=torch.bfloat16)
    assert ja >= 20 and jb == 0, (ja, jb)
    assert la == lb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_prepacked_steps_inside_a_captured_graph(seg, monkeypatch):
    from mi355seg.models.three_d.unet3d import UNet3D
    x = make_input((2, 1, 32, 32, 32)).cuda()
    gt = make_labels((2, 1, 32, 32, 32)).cuda()
    fn = lambda: UNet3D(in_channels=1, out_channels=2, init_features=16)
    pa, la, ja = _train(fn, x, gt, 5, True, monkeypatch, graphed=True)
    pb, lb, jb = _train(fn, x, gt, 5, False, monkeypatch, graphed=True)
    assert ja >= 20 and jb == 0
    assert la == lb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_prepack_plan_follows_the_input_shape(seg, monkeypatch):
    """A second input shape records a second plan; going back to the first replays the first (no stale geometry)."""
    from mi355seg import functional as F
    from mi355seg.engine import train_step
    from mi355seg.models.three_d.unet3d import UNet3D
    monkeypatch.setattr(F, "_PREPACK_OFF", False)
    m = fill_module_(UNet3D(in_channels=1, out_channels=2, init_features=16)).cuda().train()
    ref = fill_module_(UNet3D(in_channels=1, out_channels=2, init_features=16)).cuda().train()
    opt, opt_ref = torch.optim.Adam(m.parameters(), lr=1e-3), torch.optim.Adam(ref.parameters(), lr=1e-3)
    shapes = [(2, 1, 32, 32, 32), (1, 1, 48, 32, 32), (2, 1, 32, 32, 32), (1, 1, 48, 32, 32)]
    for shp in shapes:
        x, gt = make_input(shp).cuda(), make_labels(shp).cuda()
        monkeypatch.setattr(F, "_PREPACK_OFF", False)
        a = train_step(m, opt, x, gt, sync_metric=False)["loss"]
        monkeypatch.setattr(F, "_PREPACK_OFF", True)
        b = train_step(ref, opt_ref, x, gt, sync_metric=False)["loss"]
        assert float(a) == float(b)
    assert len(m.__dict__["_seg_prepack"]) == 2
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.equal(p, q)
