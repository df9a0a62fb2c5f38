"""Build-container-only check: the oracle restatement equals the live reference
modules (imported from /root/reference) on identical closed-form weights/inputs.
Skipped wherever the reference tree is absent (e.g. the GPU box)."""
import sys

import numpy as np
import pytest
import torch

from conftest import REFERENCE, has_reference
from oracle.fill import fill_module_, make_input
from oracle import nets

pytestmark = pytest.mark.skipif(not has_reference(), reason="/root/reference not present")


def _ref(modname, cls):
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import importlib
    return getattr(importlib.import_module(modname), cls)


def _same_keys(a, b):
    ka, kb = a.state_dict(), b.state_dict()
    assert set(ka) == set(kb)
    for k in ka:
        assert ka[k].shape == kb[k].shape, k


def _fwd_bwd(m, x):
    y = m(x)
    y.square().mean().backward()
    return y.detach(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}


def test_unet3d_bit_identical():
    R = _ref("models.three_d.unet3d", "UNet3D")
    a = fill_module_(R(1, 2, 8)).train()
    b = fill_module_(nets.UNet3D(1, 2, 8)).train()
    _same_keys(a, b)
    x = make_input((2, 1, 16, 16, 16))
    ya, ga = _fwd_bwd(a, x)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_vnet_bit_identical():
    R = _ref("models.three_d.vnet3d", "VNet")
    a = fill_module_(R(in_channels=1, classes=2)).train()
    b = fill_module_(nets.VNet(in_channels=1, classes=2)).train()
    _same_keys(a, b)
    x = make_input((2, 1, 16, 16, 16))
    torch.manual_seed(3)
    ya, ga = _fwd_bwd(a, x)
    torch.manual_seed(3)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_vnet_prelu_branch_bit_identical():
    """elu=False: nn.PReLU(nchan) in place of every ELU (vnet3d.py:14-18), incl. the state_dict keys of the slopes."""
    R = _ref("models.three_d.vnet3d", "VNet")
    a = fill_module_(R(elu=False, in_channels=1, classes=2)).train()
    b = fill_module_(nets.VNet(elu=False, in_channels=1, classes=2)).train()
    _same_keys(a, b)
    assert "down_tr64.relu2.weight" in b.state_dict() and b.state_dict()["up_tr128.relu1.weight"].shape == (64,)
    x = make_input((2, 1, 16, 16, 16))
    torch.manual_seed(3)
    ya, ga = _fwd_bwd(a, x)
    torch.manual_seed(3)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_resunet_bit_identical():
    R = _ref("models.three_d.residual_unet3d", "UNet")
    a = fill_module_(R(in_channels=4, n_classes=4, base_n_filter=4)).train()
    b = fill_module_(nets.ResUNet(in_channels=4, n_classes=4, base_n_filter=4)).train()
    _same_keys(a, b)
    x = make_input((1, 4, 16, 32, 16))
    torch.manual_seed(5)
    ya, ga = _fwd_bwd(a, x)
    torch.manual_seed(5)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_unetr_bit_identical():
    R = _ref("models.three_d.unetr", "UNETR")
    kw = dict(img_shape=(32, 32, 32), input_dim=1, output_dim=2, embed_dim=96, patch_size=16, num_heads=4, dropout=0.1)
    a = fill_module_(R(**kw)).train()
    b = fill_module_(nets.UNETR(**kw)).train()
    _same_keys(a, b)
    x = make_input((2, 1, 32, 32, 32))
    torch.manual_seed(9)
    ya, ga = _fwd_bwd(a, x)
    torch.manual_seed(9)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        if ga[k] is None:
            assert gb[k] is None
            continue
        assert torch.equal(ga[k], gb[k]), k


def test_isnet_bit_identical_and_frequency_bands():
    """IS.py (three parameter sets, shared encoder) and the FFT band split of train.py:76-88.  train.py cannot be
    imported (hydra / torchio absent), so its two filter functions are lifted out of its syntax tree and executed as
    they stand -- the reference itself, not a copy -- to pin oracle.step.frequency_bands."""
    import ast
    import torch.fft as fft
    from oracle.step import frequency_bands
    tree = ast.parse(open(f"{REFERENCE}/train.py").read())
    ns = {"torch": torch, "fft": fft}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("low_pass_torch", "high_pass_torch"):
            exec(compile(ast.Module([node], []), "train.py", "exec"), ns)
    x = make_input((1, 1, 16, 16, 16), freq=0.37)
    low, high = frequency_bands(x)
    assert torch.equal(low, ns["low_pass_torch"](x, 0.04)) and torch.equal(high, ns["high_pass_torch"](x, 0.04))
    xb = make_input((2, 1, 8, 12, 16), freq=0.21)                     # batch 2: the all-axes forward transform quirk
    lb, hb = frequency_bands(xb)
    assert torch.equal(lb, ns["low_pass_torch"](xb, 0.04)) and torch.equal(hb, ns["high_pass_torch"](xb, 0.04))

    R = _ref("models.three_d.IS", "UNet3D")
    a = fill_module_(R(1, 2, 4)).train()
    b = fill_module_(nets.ISUNet3D(1, 2, 4)).train()
    _same_keys(a, b)
    assert list(a.state_dict()) == list(b.state_dict())              # registration order (optimizer state index)
    outs = []
    x = make_input((1, 1, 32, 32, 32), freq=0.37)                    # 32^3: the bottleneck still sees 2^3 voxels for BatchNorm
    low, high = frequency_bands(x)
    for m in (a, b):
        o1, o2 = m(x, low, high)
        (o1.square().mean() + 0.5 * o2.square().mean()).backward()
        outs.append((o1.detach(), o2.detach(), {k: p.grad for k, p in m.named_parameters()}, dict(m.named_buffers())))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k, g in outs[0][2].items():
        assert (g is None) == (outs[1][2][k] is None), k               # the never-called suffixed encoders get no gradient
        if g is not None:
            assert torch.equal(g, outs[1][2][k]), k
    for k, v in outs[0][3].items():
        assert torch.equal(v, outs[1][3][k]), k                        # shared-encoder BN stats advanced three times


def test_csrnet_bit_identical():
    R = _ref("models.three_d.csrnet", "CSRNet")
    a = fill_module_(R(1, 2, 4)).train()
    b = fill_module_(nets.CSRNet(1, 2, 4)).train()
    _same_keys(a, b)
    assert list(a.state_dict()) == list(b.state_dict())
    x = make_input((2, 1, 32, 32, 32), freq=0.23)
    ya, ga = _fwd_bwd(a, x)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_renet_bit_identical():
    from oracle.fill import make_input_rough
    R = _ref("models.three_d.RE_net", "RE_Net")
    a = fill_module_(R()).train()
    b = fill_module_(nets.RE_Net()).train()
    _same_keys(a, b)
    x = make_input_rough((1, 1, 32, 32, 32), seed=5.0)
    ya, ga = _fwd_bwd(a, x)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_ernet_bit_identical():
    from oracle.fill import make_input_rough
    R = _ref("models.three_d.ER_net", "ER_Net")
    a = fill_module_(R(classes=2, channels=1)).train()
    b = fill_module_(nets.ER_Net(classes=2, channels=1)).train()
    _same_keys(a, b)
    x = make_input_rough((1, 1, 32, 32, 32), seed=7.0)
    ya, ga = _fwd_bwd(a, x)
    yb, gb = _fwd_bwd(b, x)
    assert torch.equal(ya, yb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
