"""V-Net on the MI355X kernels -- drop-in for the reference's models/three_d/vnet3d.py.

Interface contract kept from the reference: ``VNet(elu=True, in_channels=1, classes=2)`` (vnet3d.py:129), the
``state_dict`` keys ``in_tr.conv1.weight`` ... ``out_tr.conv2.bias`` and the forward
semantics of vnet3d.py:146-157.  Everything else is organised for the kernels: activations stay channel-last,
every BatchNorm is one fused launch with the residual add and the ELU that follow it (vnet3d.py:57-58,79,103),
and the channel concat / repeat are strided slice copies.  ``elu=False`` (nn.PReLU per unit, vnet3d.py:14-18) keeps the same
kernels with the activation un-fused: BatchNorm without activation, then the per-channel-slope PReLU launch.
"""
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, Dropout3d, ELU, PReLU

_K5 = dict(kernel_size=5, padding=2)
_STEM_WIDTH = 16
# (attribute, input channels, number of k5 conv units) for the encoder; (attribute, in, out, units) for the decoder
_ENCODER = (("down_tr32", 16, 1), ("down_tr64", 32, 2), ("down_tr128", 64, 3), ("down_tr256", 128, 2))
_DECODER = (("up_tr256", 256, 256, 2), ("up_tr128", 256, 128, 2), ("up_tr64", 128, 64, 1), ("up_tr32", 64, 32, 1))


def _act_module(elu, nchan):
    """ELUCons of vnet3d.py:14-18."""
    return ELU(inplace=True) if elu else PReLU(nchan)


def _bn_act(bn, conv_out, relu, residual=None):
    """relu(bn(conv_out) [+ residual]): one fused launch for ELU; BatchNorm, then the PReLU launch otherwise."""
    if isinstance(relu, ELU):
        return bn.forward_act(conv_out, F.ACT_ELU, residual=residual)
    return relu(bn.forward_act(conv_out, F.ACT_NONE), residual=residual)


def _conv_bn_act(x, conv, bn, relu):
    """relu(bn(conv(x))) with the batch statistics out of the convolution's epilogue."""
    if isinstance(relu, ELU):
        return F.conv_bn_act(x, conv, bn, F.ACT_ELU)
    return relu(F.conv_bn_act(x, conv, bn, F.ACT_NONE))


def _units_and_input(ops, x, elu_mode):
    """(units(x), x) for a residual block: with the fused ELU units the FIRST unit's node hands x on unchanged, so that its backward adds
    the block sum's gradient of x inside the input-gradient kernel (functional.conv_bn_act(..., fork=True)) instead of autograd's own sum."""
    units = list(ops.children()) if isinstance(ops, nn.Sequential) else []
    if not (elu_mode and units and isinstance(units[0], LUConv)):
        return ops(x), x
    first = units[0]
    h, x = F.conv_bn_act(x, first.conv1, first.bn1, F.ACT_ELU, fork=True)
    for unit in units[1:]:
        h = unit(h)
    return h, x


def _add_act(x, residual, relu):
    if isinstance(relu, ELU):
        return F.activation(x, F.ACT_ELU, residual=residual)
    return relu(x, residual=residual)


class LUConv(nn.Module):
    """One k5 unit: ELU(BN(conv(x)))  (vnet3d.py:21-31)."""

    def __init__(self, width, elu):
        super().__init__()
        self.relu1 = _act_module(elu, width)
        self.conv1 = Conv3d(width, width, **_K5)
        self.bn1 = BatchNorm3d(width)

    def forward(self, x):
        # conv + batch statistics (conv epilogue) + BN + ELU as one autograd node; the bias gradient comes out of the BN backward
        return _conv_bn_act(x, self.conv1, self.bn1, self.relu1)


def _make_nConv(width, units, elu):
    return nn.Sequential(*(LUConv(width, elu) for _ in range(units)))


class InputTransition(nn.Module):
    """ELU(BN(conv k5(x)) + x tiled to 16 channels)  (vnet3d.py:41-58)."""

    def __init__(self, in_channels, elu):
        super().__init__()
        self.in_channels, self.num_features = in_channels, _STEM_WIDTH
        self.conv1 = Conv3d(in_channels, _STEM_WIDTH, **_K5)
        self.bn1 = BatchNorm3d(_STEM_WIDTH)
        self.relu1 = _act_module(elu, _STEM_WIDTH)

    def forward(self, x):
        tiled = F.repeat_channels(x, self.num_features // self.in_channels)
        return _bn_act(self.bn1, self.conv1(x), self.relu1, residual=tiled)


class DownTransition(nn.Module):
    """down = ELU(BN(conv k2 s2)); ELU(units(down) + down)  (vnet3d.py:61-80)."""

    def __init__(self, cin, units, elu, dropout=False):
        super().__init__()
        self.down_conv = Conv3d(cin, 2 * cin, kernel_size=2, stride=2)
        self.bn1 = BatchNorm3d(2 * cin)
        self.relu1, self.relu2 = _act_module(elu, 2 * cin), _act_module(elu, 2 * cin)
        self.do1 = Dropout3d() if dropout else nn.Identity()
        self.ops = _make_nConv(2 * cin, units, elu)

    def forward(self, x):
        down = _conv_bn_act(x, self.down_conv, self.bn1, self.relu1)
        if isinstance(self.do1, nn.Identity):
            out, down = _units_and_input(self.ops, down, isinstance(self.relu2, ELU))
            return _add_act(out, down, self.relu2)
        return _add_act(self.ops(self.do1(down)), down, self.relu2)


class UpTransition(nn.Module):
    """cat = [ELU(BN(ConvT k2 s2(x))), Dropout3d(skip)]; ELU(units(cat) + cat)  (vnet3d.py:83-104)."""

    def __init__(self, cin, cout, units, elu, dropout=False):
        super().__init__()
        self.up_conv = ConvTranspose3d(cin, cout // 2, kernel_size=2, stride=2)
        self.bn1 = BatchNorm3d(cout // 2)
        self.do1 = Dropout3d() if dropout else nn.Identity()
        self.do2 = Dropout3d()
        self.relu1, self.relu2 = _act_module(elu, cout // 2), _act_module(elu, cout)
        self.ops = _make_nConv(cout, units, elu)

    def forward(self, x, skipx):
        if isinstance(self.relu1, ELU):
            # one node: the skip's dropout mask first (as the reference draws it, vnet3d.py:99), then BN + ELU of the up-convolution
            # into the left channel slice of the concat buffer and the scaled skip into the right one
            scale = self.do2.draw_scale(skipx.shape[0], skipx.shape[-1], skipx.device)
            both = F.bn_act_cat_scaled(self.up_conv(self.do1(x)), self.bn1, skipx, scale, F.ACT_ELU)
        else:
            kept_skip = self.do2(skipx)                  # the reference draws the skip mask first (vnet3d.py:99)
            up = _bn_act(self.bn1, self.up_conv(self.do1(x)), self.relu1)
            both = F.cat_channels(up, kept_skip)
        out, both = _units_and_input(self.ops, both, isinstance(self.relu2, ELU))
        return _add_act(out, both, self.relu2)


class OutputTransition(nn.Module):
    """conv k1(ELU(BN(conv k5(x))))  (vnet3d.py:107-121)."""

    def __init__(self, in_channels, classes, elu):
        super().__init__()
        self.classes = classes
        self.conv1 = Conv3d(in_channels, classes, **_K5)
        self.bn1 = BatchNorm3d(classes)
        self.conv2 = Conv3d(classes, classes, kernel_size=1)
        self.relu1 = _act_module(elu, classes)

    def forward(self, x):
        return self.conv2(_conv_bn_act(x, self.conv1, self.bn1, self.relu1))


class VNet(nn.Module):
    def __init__(self, elu=True, in_channels=1, classes=2):
        super().__init__()
        if _STEM_WIDTH % in_channels:
            raise ValueError("VNet: in_channels must divide 16 (vnet3d.py:55 repeat_rate)")
        self.classes, self.in_channels = classes, in_channels
        self.in_tr = InputTransition(in_channels, elu=elu)
        for attr, cin, units in _ENCODER:
            setattr(self, attr, DownTransition(cin, units, elu))
        for attr, cin, cout, units in _DECODER:
            setattr(self, attr, UpTransition(cin, cout, units, elu))
        self.out_tr = OutputTransition(32, classes, elu)

    def dropout_layers(self):
        """The four always-on skip dropouts in call order (for mask injection in parity tests)."""
        return [getattr(self, attr).do2 for attr, *_ in _DECODER]

    def forward(self, x):
        feats = [self.in_tr(F.to_channels_last(x))]
        for attr, *_ in _ENCODER:
            feats.append(getattr(self, attr)(feats[-1]))
        h = feats.pop()
        for attr, *_ in _DECODER:
            h = getattr(self, attr)(h, feats.pop())
        return F.to_channels_first(self.out_tr(h))
