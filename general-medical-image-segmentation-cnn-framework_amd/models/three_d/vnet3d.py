"""V-Net on the MI355X kernels -- drop-in for the reference's models/three_d/vnet3d.py.

Same constructor (``VNet(elu=True, in_channels=1, classes=2)``, vnet3d.py:129), same
``state_dict`` keys (``in_tr.conv1.weight`` ... ``out_tr.conv2.bias``) and
forward semantics (vnet3d.py:146-157).  Activations are channel-last; every BatchNorm runs fused
with the residual add and ELU that follow it (vnet3d.py:57-58,79,103).  Only the default
``elu=True`` branch is implemented (the PReLU branch is dead at the reference's defaults).
"""
import torch
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, Dropout3d, ELU


def _elu_only(elu):
    if not elu:
        raise NotImplementedError("VNet(elu=False) (PReLU) is not implemented; the reference default is elu=True")
    return ELU(inplace=True)


class LUConv(nn.Module):
    """conv k5 p2 -> BN -> ELU (vnet3d.py:21-31)."""

    def __init__(self, nchan, elu):
        super().__init__()
        self.relu1 = _elu_only(elu)
        self.conv1 = Conv3d(nchan, nchan, kernel_size=5, padding=2)
        self.bn1 = BatchNorm3d(nchan)

    def forward(self, x):
        return self.bn1.forward_act(self.conv1(x), F.ACT_ELU)


def _make_nConv(nchan, depth, elu):
    return nn.Sequential(*[LUConv(nchan, elu) for _ in range(depth)])


class InputTransition(nn.Module):
    """ELU(BN(conv k5(x)) + x repeated to 16 channels)  (vnet3d.py:41-58)."""

    def __init__(self, in_channels, elu):
        super().__init__()
        self.num_features = 16
        self.in_channels = in_channels
        self.conv1 = Conv3d(in_channels, self.num_features, kernel_size=5, padding=2)
        self.bn1 = BatchNorm3d(self.num_features)
        self.relu1 = _elu_only(elu)

    def forward(self, x):
        rep = int(self.num_features / self.in_channels)
        x16 = x.repeat(1, 1, 1, 1, rep)                       # channel-last: last dim is the channel
        return self.bn1.forward_act(self.conv1(x), F.ACT_ELU, residual=x16)


class DownTransition(nn.Module):
    """conv k2 s2 -> BN -> ELU -> n x LUConv -> (+ down) -> ELU  (vnet3d.py:61-80)."""

    def __init__(self, inChans, nConvs, elu, dropout=False):
        super().__init__()
        outChans = 2 * inChans
        self.down_conv = Conv3d(inChans, outChans, kernel_size=2, stride=2)
        self.bn1 = BatchNorm3d(outChans)
        self.relu1 = _elu_only(elu)
        self.relu2 = _elu_only(elu)
        self.do1 = Dropout3d() if dropout else nn.Identity()
        self.ops = _make_nConv(outChans, nConvs, elu)

    def forward(self, x):
        down = self.bn1.forward_act(self.down_conv(x), F.ACT_ELU)
        out = self.ops(self.do1(down))
        return F.activation(out, F.ACT_ELU, residual=down)


class UpTransition(nn.Module):
    """Dropout3d(skip); ConvT k2 s2 -> BN -> ELU; cat; n x LUConv; (+ cat) -> ELU  (vnet3d.py:83-104)."""

    def __init__(self, inChans, outChans, nConvs, elu, dropout=False):
        super().__init__()
        self.up_conv = ConvTranspose3d(inChans, outChans // 2, kernel_size=2, stride=2)
        self.bn1 = BatchNorm3d(outChans // 2)
        self.do1 = Dropout3d() if dropout else nn.Identity()
        self.do2 = Dropout3d()
        self.relu1 = _elu_only(elu)
        self.relu2 = _elu_only(elu)
        self.ops = _make_nConv(outChans, nConvs, elu)

    def forward(self, x, skipx):
        skip = self.do2(skipx)
        up = self.bn1.forward_act(self.up_conv(self.do1(x)), F.ACT_ELU)
        xcat = torch.cat((up, skip), dim=-1)
        return F.activation(self.ops(xcat), F.ACT_ELU, residual=xcat)


class OutputTransition(nn.Module):
    """conv k5 -> BN -> ELU -> conv k1  (vnet3d.py:107-121)."""

    def __init__(self, in_channels, classes, elu):
        super().__init__()
        self.classes = classes
        self.conv1 = Conv3d(in_channels, classes, kernel_size=5, padding=2)
        self.bn1 = BatchNorm3d(classes)
        self.conv2 = Conv3d(classes, classes, kernel_size=1)
        self.relu1 = _elu_only(elu)

    def forward(self, x):
        return self.conv2(self.bn1.forward_act(self.conv1(x), F.ACT_ELU))


class VNet(nn.Module):
    def __init__(self, elu=True, in_channels=1, classes=2):
        super().__init__()
        if 16 % in_channels:
            raise ValueError("VNet: in_channels must divide 16 (vnet3d.py:55 repeat_rate)")
        self.classes = classes
        self.in_channels = in_channels
        self.in_tr = InputTransition(in_channels, elu=elu)
        self.down_tr32 = DownTransition(16, 1, elu)
        self.down_tr64 = DownTransition(32, 2, elu)
        self.down_tr128 = DownTransition(64, 3, elu, dropout=False)
        self.down_tr256 = DownTransition(128, 2, elu, dropout=False)
        self.up_tr256 = UpTransition(256, 256, 2, elu, dropout=False)
        self.up_tr128 = UpTransition(256, 128, 2, elu, dropout=False)
        self.up_tr64 = UpTransition(128, 64, 1, elu)
        self.up_tr32 = UpTransition(64, 32, 1, elu)
        self.out_tr = OutputTransition(32, classes, elu)

    def dropout_layers(self):
        """The four always-on skip dropouts in call order (for mask injection in parity tests)."""
        return [self.up_tr256.do2, self.up_tr128.do2, self.up_tr64.do2, self.up_tr32.do2]

    def forward(self, x):
        h = F.to_channels_last(x)
        out16 = self.in_tr(h)
        out32 = self.down_tr32(out16)
        out64 = self.down_tr64(out32)
        out128 = self.down_tr128(out64)
        out256 = self.down_tr256(out128)
        h = self.up_tr256(out256, out128)
        h = self.up_tr128(h, out64)
        h = self.up_tr64(h, out32)
        h = self.up_tr32(h, out16)
        return F.to_channels_first(self.out_tr(h))
