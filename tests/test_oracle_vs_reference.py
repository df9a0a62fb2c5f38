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
