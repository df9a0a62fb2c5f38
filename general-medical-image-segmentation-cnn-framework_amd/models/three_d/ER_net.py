"""ER_Net on the MI355X kernels -- drop-in for the reference's models/three_d/ER_net.py (registry key ``er_net``,
train.py:332-335: ``ER_Net(classes=config.out_classes, channels=config.in_classes)``).

Encoder side = RE_Net's (residual encoders, reverse-attention skips; ER_net.py:112-165).  Decoder side: each level
fuses the up-convolved features and the gated skip with a *selective-fusion* unit ``SFConv`` (ER_net.py:36-70): the
voxel mean of their sum goes through ``fc`` and one ``fcs[i]`` per branch, a softmax across the two branches gives
per-(sample, channel) weights, and the output is the weighted sum of the two inputs; then BN + ReLU and a residual
decoder block (ER_net.py:71-81, 20-35).  The voxel statistics and the mixing are library kernels
(``mi355seg_group_sums / mix_channels / broadcast_channels``), the tiny [N, C] algebra runs on the Linear / softmax
kernels.  Logits are returned (no output sigmoid, unlike RE_Net).
"""
import torch
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, Linear, ReLU
from .RE_net import ResEncoder, deconv, downsample


class ResDecoder(nn.Module):
    """relu(relu(bn2(conv2(relu(bn1(conv1 x))))) + conv1x1(x)), all at ``in_channels`` (ER_net.py:20-35)."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = Conv3d(in_channels, in_channels, kernel_size=3, padding=1)
        self.bn1 = BatchNorm3d(in_channels)
        self.conv2 = Conv3d(in_channels, in_channels, kernel_size=3, padding=1)
        self.bn2 = BatchNorm3d(in_channels)
        self.relu = ReLU(inplace=False)
        self.conv1x1 = Conv3d(in_channels, in_channels, kernel_size=1)

    def forward(self, x):
        shortcut = self.conv1x1(x)
        h = F.conv_bn_act(x, self.conv1, self.bn1, F.ACT_RELU)
        h = F.conv_bn_act(h, self.conv2, self.bn2, F.ACT_RELU)
        return F.activation(h, F.ACT_RELU, residual=shortcut)


class SFConv(nn.Module):
    def __init__(self, features, M=2, r=4, L=32):
        super().__init__()
        if M != 2:
            raise NotImplementedError("SFConv: two branches, as ER_Net uses it")
        d = max(int(features / r), L)
        self.M, self.features = M, features
        self.fc = Linear(features, d)
        self.fcs = nn.ModuleList([Linear(d, features) for _ in range(M)])
        self.softmax = nn.Softmax(dim=1)               # interface parity; the branch softmax runs on softmax_rows

    def forward(self, x1, x2):
        n, c = x1.shape[0], self.features
        z = self.fc(F.sf_pool(x1, x2))                                            # [N, d]
        logits = torch.stack([fc(z) for fc in self.fcs], dim=-1)                  # [N, C, 2] (copy of 2*N*C values)
        att = F.softmax_last(logits.reshape(n * c, 2)).reshape(n, c, 2)
        return F.sf_mix(x1, x2, att[..., 0], att[..., 1])


class SF_Decoder(nn.Module):
    def __init__(self, out_channels):
        super().__init__()
        self.conv1 = SFConv(out_channels)
        self.bn1 = BatchNorm3d(out_channels)
        self.relu = ReLU(inplace=True)
        self.ResDecoder = ResDecoder(out_channels)

    def forward(self, x1, x2):
        return self.ResDecoder(self.bn1.forward_act(self.conv1(x1, x2), F.ACT_RELU))


class ER_Net(nn.Module):
    def __init__(self, classes, channels):
        super().__init__()
        self.encoder1 = ResEncoder(channels, 32)
        self.encoder2 = ResEncoder(32, 64)
        self.encoder3 = ResEncoder(64, 128)
        self.bridge = ResEncoder(128, 256)
        self.conv1_1 = Conv3d(256, 1, kernel_size=1)
        self.conv2_2 = Conv3d(128, 1, kernel_size=1)
        self.conv3_3 = Conv3d(64, 1, kernel_size=1)
        self.convTrans1 = ConvTranspose3d(1, 1, kernel_size=2, stride=2)
        self.convTrans2 = ConvTranspose3d(1, 1, kernel_size=2, stride=2)
        self.convTrans3 = ConvTranspose3d(1, 1, kernel_size=2, stride=2)
        self.decoder3 = SF_Decoder(128)
        self.decoder2 = SF_Decoder(64)
        self.decoder1 = SF_Decoder(32)
        self.down = downsample()
        self.up3 = deconv(256, 128)
        self.up2 = deconv(128, 64)
        self.up1 = deconv(64, 32)
        self.final = Conv3d(32, classes, kernel_size=1, padding=0)

    def forward(self, x):
        enc1 = self.encoder1(F.to_channels_last(x))
        enc2 = self.encoder2(self.down(enc1))
        skip1 = F.reverse_attention_gate(enc1, self.convTrans3(self.conv3_3(enc2)))
        enc3 = self.encoder3(self.down(enc2))
        skip2 = F.reverse_attention_gate(enc2, self.convTrans2(self.conv2_2(enc3)))
        bridge = self.bridge(self.down(enc3))
        skip3 = F.reverse_attention_gate(enc3, self.convTrans1(self.conv1_1(bridge)))
        h = self.decoder3(self.up3(bridge), skip3)
        h = self.decoder2(self.up2(h), skip2)
        h = self.decoder1(self.up1(h), skip1)
        return F.to_channels_first(self.final(h))
