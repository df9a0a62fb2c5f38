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


def downsample():
    return MaxPool3d(kernel_size=2, stride=2)


def deconv(in_channels, out_channels):
    return ConvTranspose3d(in_channels, out_channels, kernel_size=2, stride=2)


def initialize_weights(*models):
    for model in models:
        for m in model.modules():
            if isinstance(m, (nn.Conv3d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class ResEncoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = Conv3d(in_channels, out_channels, kernel_size=3, padding=1)
        self.bn1 = BatchNorm3d(out_channels)
        self.conv2 = Conv3d(out_channels, out_channels, kernel_size=3, padding=1)
        self.bn2 = BatchNorm3d(out_channels)
        self.relu = ReLU(inplace=False)
        self.conv1x1 = Conv3d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        shortcut = self.conv1x1(x)
        h = F.conv_bn_act(x, self.conv1, self.bn1, F.ACT_RELU)
        h = F.conv_bn_act(h, self.conv2, self.bn2, F.ACT_RELU)
        return F.activation(h, F.ACT_RELU, residual=shortcut)


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(
            Conv3d(in_channels, out_channels, kernel_size=3, padding=1), BatchNorm3d(out_channels), ReLU(inplace=True),
            Conv3d(out_channels, out_channels, kernel_size=3, padding=1), BatchNorm3d(out_channels), ReLU(inplace=True))

    def forward(self, x):
        c1, b1, _, c2, b2, _ = self.conv.children()
        return F.conv_bn_act(F.conv_bn_act(x, c1, b1, F.ACT_RELU), c2, b2, F.ACT_RELU)


class RE_Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder1 = ResEncoder(1, 32)
        self.encoder2 = ResEncoder(32, 64)
        self.encoder3 = ResEncoder(64, 128)
        self.bridge = ResEncoder(128, 256)
        self.conv1_1 = Conv3d(256, 1, kernel_size=1)
        self.conv2_2 = Conv3d(128, 1, kernel_size=1)
        self.conv3_3 = Conv3d(64, 1, kernel_size=1)
        self.convTrans1 = ConvTranspose3d(1, 1, kernel_size=2, stride=2)
        self.convTrans2 = ConvTranspose3d(1, 1, kernel_size=2, stride=2)
        self.convTrans3 = ConvTranspose3d(1, 1, kernel_size=2, stride=2)
        self.decoder3 = Decoder(256, 128)
        self.decoder2 = Decoder(128, 64)
        self.decoder1 = Decoder(64, 32)
        self.down = downsample()
        self.up3 = deconv(256, 128)
        self.up2 = deconv(128, 64)
        self.up1 = deconv(64, 32)
        self.final = Conv3d(32, 2, kernel_size=1, padding=0)
        initialize_weights(self)

    def forward(self, x):
        enc1 = self.encoder1(F.to_channels_last(x))
        enc2 = self.encoder2(self.down(enc1))
        skip1 = F.reverse_attention_gate(enc1, self.convTrans3(self.conv3_3(enc2)))
        enc3 = self.encoder3(self.down(enc2))
        skip2 = F.reverse_attention_gate(enc2, self.convTrans2(self.conv2_2(enc3)))
        bridge = self.bridge(self.down(enc3))
        skip3 = F.reverse_attention_gate(enc3, self.convTrans1(self.conv1_1(bridge)))
        h = self.decoder3(F.cat_channels(self.up3(bridge), skip3))
        h = self.decoder2(F.cat_channels(self.up2(h), skip2))
        h = self.decoder1(F.cat_channels(self.up1(h), skip1))
        return F.to_channels_first(F.activation(self.final(h), F.ACT_SIGMOID))
