"""3D U-Net on the MI355X kernels -- drop-in for the reference's models/three_d/unet3d.py.

Same constructor (``UNet3D(in_channels=1, out_channels=3, init_features=64)``,
unet3d.py:10), same ``state_dict`` keys/shapes (``encoder1.enc1conv1.weight`` ...
``upconv4.weight``, ``conv.bias``), same ``forward(x[N,C,D,H,W])
-> logits[N,K,D,H,W]`` (unet3d.py:50-71).  Internally activations are channel-last and
every op is a HIP kernel from libmi355seg.so; BatchNorm+ReLU run as one fused kernel.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, MaxPool3d, ReLU


class _DoubleConv(nn.Sequential):
    """(conv k3 p1 -> BN -> ReLU) x 2 with the reference's child names (unet3d.py:73-104).
    ``forward`` fuses each BN with its ReLU."""

    def forward(self, x, left_pad=0, head=None, pool=False):
        """``left_pad`` > 0: the block's output is the right channel slice of a buffer with ``left_pad`` free channels
        on its left, ready for the decoder's concat-free up-convolution (encoder blocks only).  ``head``: the 1x1x1 output
        convolution behind the last block -- the result is then ``head(block(x))`` (in training: norm2 + ReLU + head as one kernel,
        the block's activation never written).  ``pool``: the result is (max_pool3d_2x(block(x)), block(x)) -- the pooled tensor for
        the next level and the block's output for the skip connection (in training: norm2 + ReLU + pooling as one kernel)."""
        conv1, norm1, _r1, conv2, norm2, _r2 = self.children()
        # (conv + batch statistics + BN + ReLU) x 2; in training the whole block is one autograd node whose backward takes norm1's
        # column sums out of conv2's input-gradient kernel
        return F.double_conv_bn_act(x, conv1, norm1, conv2, norm2, F.ACT_RELU, left_pad=left_pad, head=head, pool=pool)


class UNet3D(nn.Module):
    def __init__(self, in_channels=1, out_channels=3, init_features=64):
        super().__init__()
        f = init_features
        self.encoder1 = UNet3D._block(in_channels, f, name="enc1")
        self.pool1 = MaxPool3d(kernel_size=2, stride=2)
        self.encoder2 = UNet3D._block(f, f * 2, name="enc2")
        self.pool2 = MaxPool3d(kernel_size=2, stride=2)
        self.encoder3 = UNet3D._block(f * 2, f * 4, name="enc3")
        self.pool3 = MaxPool3d(kernel_size=2, stride=2)
        self.encoder4 = UNet3D._block(f * 4, f * 8, name="enc4")
        self.pool4 = MaxPool3d(kernel_size=2, stride=2)
        self.bottleneck = UNet3D._block(f * 8, f * 16, name="bottleneck")
        self.upconv4 = ConvTranspose3d(f * 16, f * 8, kernel_size=2, stride=2)
        self.decoder4 = UNet3D._block(f * 16, f * 8, name="dec4")
        self.upconv3 = ConvTranspose3d(f * 8, f * 4, kernel_size=2, stride=2)
        self.decoder3 = UNet3D._block(f * 8, f * 4, name="dec3")
        self.upconv2 = ConvTranspose3d(f * 4, f * 2, kernel_size=2, stride=2)
        self.decoder2 = UNet3D._block(f * 4, f * 2, name="dec2")
        self.upconv1 = ConvTranspose3d(f * 2, f, kernel_size=2, stride=2)
        self.decoder1 = UNet3D._block(f * 2, f, name="dec1")
        self.conv = Conv3d(in_channels=f, out_channels=out_channels, kernel_size=1)

    @staticmethod
    def _block(in_channels, features, name):
        return _DoubleConv(OrderedDict([
            (name + "conv1", Conv3d(in_channels, features, kernel_size=3, padding=1, bias=True)),
            (name + "norm1", BatchNorm3d(num_features=features)),
            (name + "relu1", ReLU(inplace=True)),
            (name + "conv2", Conv3d(features, features, kernel_size=3, padding=1, bias=True)),
            (name + "norm2", BatchNorm3d(num_features=features)),
            (name + "relu2", ReLU(inplace=True)),
        ]))

    def forward(self, x):
        h = F.to_channels_last(x)
        # each encoder output is written as the RIGHT half of its level's concat buffer; the matching up-convolution
        # later fills the LEFT half (torch.cat((up, skip), dim=1) of unet3d.py:59-68 without the copy)
        # pool + skip leave each encoder block as one autograd node (their two gradients are summed in the pool backward)
        p1, enc1 = self.encoder1(h, left_pad=self.upconv1.out_channels, pool=True)
        p2, enc2 = self.encoder2(p1, left_pad=self.upconv2.out_channels, pool=True)
        p3, enc3 = self.encoder3(p2, left_pad=self.upconv3.out_channels, pool=True)
        p4, enc4 = self.encoder4(p3, left_pad=self.upconv4.out_channels, pool=True)
        h = self.bottleneck(p4)
        for up, dec, skip in ((self.upconv4, self.decoder4, enc4), (self.upconv3, self.decoder3, enc3), (self.upconv2, self.decoder2, enc2)):
            h = dec(F.conv_transpose3d_k2s2_cat(h, up.weight, up.bias, skip))
        # decoder1 and the output head (unet3d.py:68-71) together: head(decoder1(cat))
        h = self.decoder1(F.conv_transpose3d_k2s2_cat(h, self.upconv1.weight, self.upconv1.bias, enc1), head=self.conv)
        return F.to_channels_first(h)
