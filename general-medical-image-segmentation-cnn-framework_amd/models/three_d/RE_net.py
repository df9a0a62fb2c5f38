"""RE_Net on the MI355X kernels -- drop-in for the reference's models/three_d/RE_net.py (registry key ``re_net``,
train.py:336-339: ``RE_Net()``, fixed 1 -> 2 channels, widths 32 / 64 / 128 / 256).

Three residual encoders + a bridge, each ``relu(relu(bn2(conv2(relu(bn1(conv1 x))))) + conv1x1(x))`` (RE_net.py:20-35);
three plain double-conv decoders; and three *reverse-attention* skips (RE_net.py:101-127): a 1x1x1 conv squeezes the
next-deeper feature map to one channel, a 1 -> 1 k2 s2 transposed conv lifts it to the skip's resolution, and the skip
becomes ``enc * (1 - sigmoid(map)) + enc`` -- one gate kernel here.  The output is ``sigmoid(final(dec1))``
(RE_net.py:159-160; train.py then feeds that into BCEWithLogitsLoss, which is kept as is).
``initialize_weights`` (RE_net.py:10-19) is reproduced: kaiming-normal Conv3d / Linear weights with zero bias,
BatchNorm (1, 0); transposed convolutions keep PyTorch's default init, as upstream.
"""
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, MaxPool3d, ReLU


_WIDTHS = (32, 64, 128, 256)            # encoder1..3 + bridge


def downsample():
    return MaxPool3d(kernel_size=2, stride=2)


def deconv(in_channels, out_channels):
    return ConvTranspose3d(in_channels, out_channels, kernel_size=2, stride=2)


def initialize_weights(*models):
    """RE_net.py:10-19: kaiming-normal Conv3d / Linear weights, zero biases, BatchNorm (1, 0); ConvTranspose3d untouched."""
    for net in models:
        for mod in net.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.data.fill_(1)
                mod.bias.data.zero_()
            elif isinstance(mod, (nn.Conv3d, nn.Linear)):
                nn.init.kaiming_normal_(mod.weight)
                if mod.bias is not None:
                    mod.bias.data.zero_()


def _k3(cin, cout):
    return Conv3d(cin, cout, kernel_size=3, padding=1)


class ResEncoder(nn.Module):
    """relu(relu(bn2(conv2(relu(bn1(conv1 x))))) + conv1x1(x))  (RE_net.py:20-35); both conv+BN+ReLU pairs are fused nodes."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1, self.bn1 = _k3(in_channels, out_channels), BatchNorm3d(out_channels)
        self.conv2, self.bn2 = _k3(out_channels, out_channels), BatchNorm3d(out_channels)
        self.relu = ReLU(inplace=False)
        self.conv1x1 = Conv3d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        shortcut = self.conv1x1(x)
        h = F.conv_bn_act(x, self.conv1, self.bn1, F.ACT_RELU)
        h = F.conv_bn_act(h, self.conv2, self.bn2, F.ACT_RELU)
        return F.activation(h, F.ACT_RELU, residual=shortcut)


class Decoder(nn.Module):
    """(conv k3 -> BN -> ReLU) x 2 held in ``self.conv`` with indices 0..5  (RE_net.py:36-50)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        stages = []
        for cin in (in_channels, out_channels):
            stages += [_k3(cin, out_channels), BatchNorm3d(out_channels), ReLU(inplace=True)]
        self.conv = nn.Sequential(*stages)

    def forward(self, x):
        c1, b1, _, c2, b2, _ = self.conv.children()
        return F.conv_bn_act(F.conv_bn_act(x, c1, b1, F.ACT_RELU), c2, b2, F.ACT_RELU)


class RE_Net(nn.Module):
    def __init__(self):
        super().__init__()
        cin = 1
        for name, width in zip(("encoder1", "encoder2", "encoder3", "bridge"), _WIDTHS):
            setattr(self, name, ResEncoder(cin, width))
            cin = width
        for k, width in zip((1, 2, 3), reversed(_WIDTHS[1:])):          # conv1_1: 256 -> 1, conv2_2: 128 -> 1, conv3_3: 64 -> 1
            setattr(self, f"conv{k}_{k}", Conv3d(width, 1, kernel_size=1))
        for k in (1, 2, 3):
            setattr(self, f"convTrans{k}", ConvTranspose3d(1, 1, kernel_size=2, stride=2))
        for k, width in zip((3, 2, 1), reversed(_WIDTHS[1:])):          # decoder3: 256 -> 128, decoder2: 128 -> 64, decoder1: 64 -> 32
            setattr(self, f"decoder{k}", Decoder(width, width // 2))
        self.down = downsample()
        for k, width in zip((3, 2, 1), reversed(_WIDTHS[1:])):
            setattr(self, f"up{k}", deconv(width, width // 2))
        self.final = Conv3d(_WIDTHS[0], 2, kernel_size=1, padding=0)
        initialize_weights(self)

    def _skip(self, enc, deeper, k):
        """enc * (1 - sigmoid(up(squeeze(deeper)))) + enc  (RE_net.py:101-127)."""
        squeezed = getattr(self, f"conv{k}_{k}")(deeper)
        return F.reverse_attention_gate(enc, getattr(self, f"convTrans{k}")(squeezed))

    def forward(self, x):
        enc1 = self.encoder1(F.to_channels_last(x))
        enc2 = self.encoder2(self.down(enc1))
        skip1 = self._skip(enc1, enc2, 3)
        enc3 = self.encoder3(self.down(enc2))
        skip2 = self._skip(enc2, enc3, 2)
        h = self.bridge(self.down(enc3))
        skip3 = self._skip(enc3, h, 1)
        for k, skip in ((3, skip3), (2, skip2), (1, skip1)):
            h = getattr(self, f"decoder{k}")(F.cat_channels(getattr(self, f"up{k}")(h), skip))
        return F.to_channels_first(F.activation(self.final(h), F.ACT_SIGMOID))
