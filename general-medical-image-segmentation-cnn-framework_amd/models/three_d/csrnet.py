"""CSRNet on the MI355X kernels -- drop-in for the reference's models/three_d/csrnet.py (registry key ``csrnet``,
train.py:362-365).

A U-Net (same ``_block`` double convolutions, pools, k2 s2 up-convolutions and 1x1x1 head as unet3d.py) with
cross-scale residual links (csrnet.py:46-69): three encoder links ``encoder_r_k`` = Conv3d k3 **stride 4, no padding**
-> BN -> ReLU that jump two levels down, and three decoder links ``dncoder_r_k`` = ConvTranspose3d **k4 stride 4** ->
BN -> ReLU that jump two levels up; each is added to the feature map it lands on.  Constructor, state_dict keys and
forward semantics are the reference's; the links run on the gather implicit-GEMM (stride-4 conv) and on the
conv-adjoint form of the transposed conv, everything else on the U-Net kernels.
"""
from collections import OrderedDict

import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, MaxPool3d, ReLU
from .unet3d import UNet3D as _PlainUNet


class _DownLink(nn.Sequential):
    """conv k3 s4 -> BN -> ReLU (csrnet.py:101-119), conv + statistics + BN + ReLU as one autograd node."""

    def forward(self, x):
        conv, bn, _relu = self.children()
        return F.conv_bn_act(x, conv, bn, F.ACT_RELU)


class _UpLink(nn.Sequential):
    """ConvT k4 s4 -> BN -> ReLU (csrnet.py:121-137)."""

    def forward(self, x):
        up, bn, _relu = self.children()
        return bn.forward_act(up(x), F.ACT_RELU)


def _add(a, b):
    return F.activation(a, F.ACT_NONE, residual=b)


class CSRNet(nn.Module):
    def __init__(self, in_channels=1, out_channels=3, init_features=64):
        super().__init__()
        f = init_features
        w = {1: f, 2: 2 * f, 3: 4 * f, 4: 8 * f}
        prev = in_channels
        for lvl in (1, 2, 3, 4):
            setattr(self, f"encoder{lvl}", _PlainUNet._block(prev, w[lvl], name=f"enc{lvl}"))
            setattr(self, f"pool{lvl}", MaxPool3d(kernel_size=2, stride=2))
            prev = w[lvl]
        for j in (1, 2, 3):                                  # enc_j -> level j + 2
            setattr(self, f"encoder_r_{j}", CSRNet._block_r(w[j], 4 * w[j], name=f"enc{j}_r"))
        self.bottleneck = _PlainUNet._block(8 * f, 16 * f, name="bottleneck")
        prev = 16 * f
        for lvl in (4, 3, 2, 1):
            setattr(self, f"upconv{lvl}", ConvTranspose3d(prev, w[lvl], kernel_size=2, stride=2))
            setattr(self, f"decoder{lvl}", _PlainUNet._block(2 * w[lvl], w[lvl], name=f"dec{lvl}"))
            prev = w[lvl]
        self.conv = Conv3d(in_channels=f, out_channels=out_channels, kernel_size=1)
        for j, cin in ((1, 16 * f), (2, 8 * f), (3, 4 * f)):
            setattr(self, f"dncoder_r_{j}", CSRNet._block_rr(cin, cin // 4, name=f"dnc{j}_r"))

    @staticmethod
    def _block_r(in_channels, features, name):
        return _DownLink(OrderedDict([
            (name + "conv1", Conv3d(in_channels, features, kernel_size=3, stride=4, bias=True)),
            (name + "norm1", BatchNorm3d(num_features=features)),
            (name + "relu1", ReLU(inplace=True)),
        ]))

    @staticmethod
    def _block_rr(in_channels, features, name):
        return _UpLink(OrderedDict([
            (name + "conv1", ConvTranspose3d(in_channels, features, kernel_size=4, stride=4, bias=True)),
            (name + "norm1", BatchNorm3d(num_features=features)),
            (name + "relu1", ReLU(inplace=True)),
        ]))

    def forward(self, x):
        h = F.to_channels_last(x)
        enc1 = self.encoder1(h)
        enc2 = self.encoder2(self.pool1(enc1))
        enc3 = _add(self.encoder3(self.pool2(enc2)), self.encoder_r_1(enc1))
        enc4 = _add(self.encoder4(self.pool3(enc3)), self.encoder_r_2(enc2))
        bott = _add(self.bottleneck(self.pool4(enc4)), self.encoder_r_3(enc3))
        dec4 = self.decoder4(F.cat_channels(self.upconv4(bott), enc4))
        dec3 = self.decoder3(F.cat_channels(_add(self.upconv3(dec4), self.dncoder_r_1(bott)), enc3))
        dec2 = self.decoder2(F.cat_channels(_add(self.upconv2(dec3), self.dncoder_r_2(dec4)), enc2))
        dec1 = self.decoder1(F.cat_channels(_add(self.upconv1(dec2), self.dncoder_r_3(dec3)), enc1))
        return F.to_channels_first(self.conv(dec1))
