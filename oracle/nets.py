"""Oracle networks: plain PyTorch-CPU restatements of the reference's U-Net family.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Each class keeps the reference's
constructor signature, forward semantics and ``state_dict`` key names so that
closed-form weights (oracle/fill.py) land on the same tensors on both sides.

Follows (reference file:line):
  * UNet3D      -- models/three_d/unet3d.py:10-48 (ctor), :50-71 (forward), :73-104 (block)
  * VNet        -- models/three_d/vnet3d.py:21-31, :41-58, :61-80, :83-104, :107-121, :124-157
  * ResUNet     -- models/three_d/residual_unet3d.py:11-80 (ctor), :82-107 (factories), :109-204 (forward)
  * CSRNet      -- models/three_d/csrnet.py:5-45 (ctor), :46-69 (forward), :101-137 (cross-scale link blocks)
  * RE_Net      -- models/three_d/RE_net.py:20-35 (ResEncoder), :36-50 (Decoder), :51-100 (ctor), :101-161 (forward)
  * ER_Net      -- models/three_d/ER_net.py:20-35 (ResDecoder), :36-70 (SFConv), :71-81 (SF_Decoder), :98-165 (ER_Net)
  * ISUNet3D    -- models/three_d/IS.py:10-130 (ctor: three parameter sets), :132-190 (forward: shared encoder, three decoders)
"""
from collections import OrderedDict

import torch
import torch.nn as nn


# --------------------------------------------------------------------------- U-Net
def _double_conv(tag, cin, cout):
    """(Conv3d k3 p1 bias -> BatchNorm3d -> ReLU) x 2, keys '<tag>conv1' ... '<tag>relu2'
    (unet3d.py:73-104)."""
    mods = OrderedDict()
    for i, c_in in ((1, cin), (2, cout)):
        mods[f"{tag}conv{i}"] = nn.Conv3d(c_in, cout, kernel_size=3, padding=1, bias=True)
        mods[f"{tag}norm{i}"] = nn.BatchNorm3d(cout)
        mods[f"{tag}relu{i}"] = nn.ReLU(inplace=True)
    return nn.Sequential(mods)


class UNet3D(nn.Module):
    """3D U-Net (unet3d.py:10-71): 4 encoder levels + bottleneck + 4 decoder levels,
    MaxPool3d(2,2) down, ConvTranspose3d k2s2 up, channel concat (up first, skip
    second), 1x1x1 head."""

    def __init__(self, in_channels=1, out_channels=3, init_features=64):
        super().__init__()
        f = init_features
        widths = [f, 2 * f, 4 * f, 8 * f]
        prev = in_channels
        for lvl, w in enumerate(widths, start=1):
            setattr(self, f"encoder{lvl}", _double_conv(f"enc{lvl}", prev, w))
            setattr(self, f"pool{lvl}", nn.MaxPool3d(kernel_size=2, stride=2))
            prev = w
        self.bottleneck = _double_conv("bottleneck", prev, 16 * f)
        prev = 16 * f
        for lvl in (4, 3, 2, 1):
            w = widths[lvl - 1]
            setattr(self, f"upconv{lvl}", nn.ConvTranspose3d(prev, w, kernel_size=2, stride=2))
            setattr(self, f"decoder{lvl}", _double_conv(f"dec{lvl}", 2 * w, w))
            prev = w
        self.conv = nn.Conv3d(f, out_channels, kernel_size=1)

    def forward(self, x):
        skips = []
        h = x
        for lvl in (1, 2, 3, 4):
            h = getattr(self, f"encoder{lvl}")(h)
            skips.append(h)
            h = getattr(self, f"pool{lvl}")(h)
        h = self.bottleneck(h)
        for lvl in (4, 3, 2, 1):
            h = getattr(self, f"upconv{lvl}")(h)
            h = torch.cat((h, skips[lvl - 1]), dim=1)
            h = getattr(self, f"decoder{lvl}")(h)
        return self.conv(h)


# --------------------------------------------------------------------------- CSRNet (U-Net + cross-scale residual links)
def _link(tag, op):
    return nn.Sequential(OrderedDict([(f"{tag}conv1", op), (f"{tag}norm1", nn.BatchNorm3d(op.out_channels)),
                                      (f"{tag}relu1", nn.ReLU(inplace=True))]))


class CSRNet(nn.Module):
    """csrnet.py:5-69.  Encoder links: Conv3d k3 stride 4 (no padding) from level j to level j+2; decoder links:
    ConvTranspose3d k4 stride 4 from the bottleneck / dec4 / dec3 to dec3 / dec2 / dec1's up-convolved input."""

    def __init__(self, in_channels=1, out_channels=3, init_features=64):
        super().__init__()
        f = init_features
        widths = [f, 2 * f, 4 * f, 8 * f]
        prev = in_channels
        for lvl, w in enumerate(widths, start=1):
            setattr(self, f"encoder{lvl}", _double_conv(f"enc{lvl}", prev, w))
            setattr(self, f"pool{lvl}", nn.MaxPool3d(kernel_size=2, stride=2))
            prev = w
        for j in (1, 2, 3):
            setattr(self, f"encoder_r_{j}", _link(f"enc{j}_r", nn.Conv3d(widths[j - 1], 4 * widths[j - 1], kernel_size=3, stride=4)))
        self.bottleneck = _double_conv("bottleneck", prev, 16 * f)
        prev = 16 * f
        for lvl in (4, 3, 2, 1):
            w = widths[lvl - 1]
            setattr(self, f"upconv{lvl}", nn.ConvTranspose3d(prev, w, kernel_size=2, stride=2))
            setattr(self, f"decoder{lvl}", _double_conv(f"dec{lvl}", 2 * w, w))
            prev = w
        self.conv = nn.Conv3d(f, out_channels, kernel_size=1)
        for j, cin in ((1, 16 * f), (2, 8 * f), (3, 4 * f)):
            setattr(self, f"dncoder_r_{j}", _link(f"dnc{j}_r", nn.ConvTranspose3d(cin, cin // 4, kernel_size=4, stride=4)))

    def forward(self, x):
        e1 = self.encoder1(x)
        e2 = self.encoder2(self.pool1(e1))
        e3 = self.encoder3(self.pool2(e2)) + self.encoder_r_1(e1)
        e4 = self.encoder4(self.pool3(e3)) + self.encoder_r_2(e2)
        bn = self.bottleneck(self.pool4(e4)) + self.encoder_r_3(e3)
        d4 = self.decoder4(torch.cat((self.upconv4(bn), e4), dim=1))
        d3 = self.decoder3(torch.cat((self.upconv3(d4) + self.dncoder_r_1(bn), e3), dim=1))
        d2 = self.decoder2(torch.cat((self.upconv2(d3) + self.dncoder_r_2(d4), e2), dim=1))
        d1 = self.decoder1(torch.cat((self.upconv1(d2) + self.dncoder_r_3(d3), e1), dim=1))
        return self.conv(d1)


# --------------------------------------------------------------------------- RE_Net (residual encoders + reverse attention)
class _ResEncoder(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm3d(cout)
        self.conv2 = nn.Conv3d(cout, cout, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm3d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.conv1x1 = nn.Conv3d(cin, cout, kernel_size=1)

    def forward(self, x):
        res = self.conv1x1(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.relu(out + res)


class _PlainDecoder(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm3d(cout), nn.ReLU(inplace=True),
                                  nn.Conv3d(cout, cout, kernel_size=3, padding=1), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class RE_Net(nn.Module):
    """RE_net.py:51-161.  skip_k = enc_k * (1 - sigmoid(up(conv1x1(deeper)))) + enc_k; output = sigmoid(final)."""

    def __init__(self):
        super().__init__()
        self.encoder1, self.encoder2, self.encoder3 = _ResEncoder(1, 32), _ResEncoder(32, 64), _ResEncoder(64, 128)
        self.bridge = _ResEncoder(128, 256)
        self.conv1_1, self.conv2_2, self.conv3_3 = nn.Conv3d(256, 1, 1), nn.Conv3d(128, 1, 1), nn.Conv3d(64, 1, 1)
        self.convTrans1, self.convTrans2, self.convTrans3 = (nn.ConvTranspose3d(1, 1, kernel_size=2, stride=2) for _ in range(3))
        self.decoder3, self.decoder2, self.decoder1 = _PlainDecoder(256, 128), _PlainDecoder(128, 64), _PlainDecoder(64, 32)
        self.down = nn.MaxPool3d(kernel_size=2, stride=2)
        self.up3, self.up2, self.up1 = (nn.ConvTranspose3d(c, c // 2, kernel_size=2, stride=2) for c in (256, 128, 64))
        self.final = nn.Conv3d(32, 2, kernel_size=1, padding=0)

    @staticmethod
    def _gate(enc, deeper_map):
        rev = -1 * torch.sigmoid(deeper_map) + 1
        return rev.expand(-1, enc.shape[1], -1, -1, -1).mul(enc) + enc

    def forward(self, x):
        # op order as upstream (pool before the 1x1x1 squeeze): it fixes the order in which autograd sums the
        # gradients of the multiply-used encoder outputs, so the oracle stays bit-identical to the reference
        e1 = self.encoder1(x)
        d1 = self.down(e1)
        e2 = self.encoder2(d1)
        d2 = self.down(e2)
        s1 = self._gate(e1, self.convTrans3(self.conv3_3(e2)))
        e3 = self.encoder3(d2)
        d3 = self.down(e3)
        s2 = self._gate(e2, self.convTrans2(self.conv2_2(e3)))
        br = self.bridge(d3)
        s3 = self._gate(e3, self.convTrans1(self.conv1_1(br)))
        h = self.decoder3(torch.cat((self.up3(br), s3), dim=1))
        h = self.decoder2(torch.cat((self.up2(h), s2), dim=1))
        h = self.decoder1(torch.cat((self.up1(h), s1), dim=1))
        return torch.sigmoid(self.final(h))


# --------------------------------------------------------------------------- ER_Net (RE_Net encoder + selective-fusion decoders)
class _ResDecoder(_ResEncoder):
    def __init__(self, c):
        super().__init__(c, c)


class _SFConv(nn.Module):
    """ER_net.py:36-70: weights = softmax over the two branches of fcs[i](fc(mean_voxels(x1 + x2))); out = sum_i w_i x_i."""

    def __init__(self, features, M=2, r=4, L=32):
        super().__init__()
        d = max(int(features / r), L)
        self.fc = nn.Linear(features, d)
        self.fcs = nn.ModuleList([nn.Linear(d, features) for _ in range(M)])
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x1, x2):
        feas = torch.cat((x1.unsqueeze(dim=1), x2.unsqueeze(dim=1)), dim=1)
        fea_s = torch.sum(feas, dim=1).mean(-1).mean(-1).mean(-1)
        fea_z = self.fc(fea_s)
        att = torch.cat([fc(fea_z).unsqueeze(dim=1) for fc in self.fcs], dim=1)
        att = self.softmax(att).unsqueeze(-1).unsqueeze(-1).unsqueeze(-1)
        return (feas * att).sum(dim=1)


class _SFDecoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = _SFConv(c)
        self.bn1 = nn.BatchNorm3d(c)
        self.relu = nn.ReLU(inplace=True)
        self.ResDecoder = _ResDecoder(c)

    def forward(self, x1, x2):
        return self.ResDecoder(self.relu(self.bn1(self.conv1(x1, x2))))


class ER_Net(nn.Module):
    def __init__(self, classes, channels):
        super().__init__()
        self.encoder1, self.encoder2, self.encoder3 = _ResEncoder(channels, 32), _ResEncoder(32, 64), _ResEncoder(64, 128)
        self.bridge = _ResEncoder(128, 256)
        self.conv1_1, self.conv2_2, self.conv3_3 = nn.Conv3d(256, 1, 1), nn.Conv3d(128, 1, 1), nn.Conv3d(64, 1, 1)
        self.convTrans1, self.convTrans2, self.convTrans3 = (nn.ConvTranspose3d(1, 1, kernel_size=2, stride=2) for _ in range(3))
        self.decoder3, self.decoder2, self.decoder1 = _SFDecoder(128), _SFDecoder(64), _SFDecoder(32)
        self.down = nn.MaxPool3d(kernel_size=2, stride=2)
        self.up3, self.up2, self.up1 = (nn.ConvTranspose3d(c, c // 2, kernel_size=2, stride=2) for c in (256, 128, 64))
        self.final = nn.Conv3d(32, classes, kernel_size=1, padding=0)

    def forward(self, x):
        e1 = self.encoder1(x)
        d1 = self.down(e1)
        e2 = self.encoder2(d1)
        d2 = self.down(e2)
        s1 = RE_Net._gate(e1, self.convTrans3(self.conv3_3(e2)))
        e3 = self.encoder3(d2)
        d3 = self.down(e3)
        s2 = RE_Net._gate(e2, self.convTrans2(self.conv2_2(e3)))
        br = self.bridge(d3)
        s3 = RE_Net._gate(e3, self.convTrans1(self.conv1_1(br)))
        h = self.decoder3(self.up3(br), s3)
        h = self.decoder2(self.up2(h), s2)
        h = self.decoder1(self.up1(h), s1)
        return self.final(h)


# --------------------------------------------------------------------------- IS (three-band U-Net)
class ISUNet3D(nn.Module):
    """IS.py:10-190.  Three U-Net parameter sets ("", "_", "__"); forward runs the UNSUFFIXED encoder + bottleneck on
    the volume, its low band and its high band, with decoder set k on band k; out1 = conv(dec), out2 = conv_(sum of
    the three decoder outputs).  The suffixed encoders / bottlenecks exist only as parameters."""

    def __init__(self, in_channels=1, out_channels=3, init_features=64):
        super().__init__()
        f = init_features
        widths = [f, 2 * f, 4 * f, 8 * f]
        for tag in ("", "_", "__"):
            prev = in_channels
            for lvl, w in enumerate(widths, start=1):
                setattr(self, f"encoder{lvl}{tag}", _double_conv(f"enc{lvl}", prev, w))
                setattr(self, f"pool{lvl}{tag}", nn.MaxPool3d(kernel_size=2, stride=2))
                prev = w
            setattr(self, f"bottleneck{tag}", _double_conv("bottleneck", prev, 16 * f))
            prev = 16 * f
            for lvl in (4, 3, 2, 1):
                w = widths[lvl - 1]
                setattr(self, f"upconv{lvl}{tag}", nn.ConvTranspose3d(prev, w, kernel_size=2, stride=2))
                setattr(self, f"decoder{lvl}{tag}", _double_conv(f"dec{lvl}", 2 * w, w))
                prev = w
        self.conv = nn.Conv3d(f, out_channels, kernel_size=1)
        self.conv_ = nn.Conv3d(f, out_channels, kernel_size=1)

    def _band(self, h, tag):
        skips = []
        for lvl in (1, 2, 3, 4):
            h = getattr(self, f"encoder{lvl}")(h)
            skips.append(h)
            h = getattr(self, f"pool{lvl}")(h)          # IS.py:156 uses pool4_ for the low band: same parameter-free op
        h = self.bottleneck(h)
        for lvl in (4, 3, 2, 1):
            h = getattr(self, f"upconv{lvl}{tag}")(h)
            h = getattr(self, f"decoder{lvl}{tag}")(torch.cat((h, skips[lvl - 1]), dim=1))
        return h

    def forward(self, x, low_x, high_x):
        d0, d1, d2 = self._band(x, ""), self._band(low_x, "_"), self._band(high_x, "__")
        return self.conv(d0), self.conv_(d0 + d1 + d2)


# --------------------------------------------------------------------------- V-Net
def _act(elu, nchan):
    # vnet3d.py:14-18 -- ELU(alpha=1, inplace) at the defaults, PReLU otherwise
    return nn.ELU(inplace=True) if elu else nn.PReLU(nchan)


class LUConv(nn.Module):
    """conv k5 p2 -> BN -> act (vnet3d.py:21-31)."""

    def __init__(self, nchan, elu):
        super().__init__()
        self.relu1 = _act(elu, nchan)
        self.conv1 = nn.Conv3d(nchan, nchan, kernel_size=5, padding=2)
        self.bn1 = nn.BatchNorm3d(nchan)

    def forward(self, x):
        return self.relu1(self.bn1(self.conv1(x)))


def _n_conv(nchan, depth, elu):
    return nn.Sequential(*[LUConv(nchan, elu) for _ in range(depth)])


class InputTransition(nn.Module):
    """conv k5 -> BN -> + x repeated to 16 channels -> act (vnet3d.py:41-58)."""

    def __init__(self, in_channels, elu):
        super().__init__()
        self.num_features = 16
        self.in_channels = in_channels
        self.conv1 = nn.Conv3d(in_channels, 16, kernel_size=5, padding=2)
        self.bn1 = nn.BatchNorm3d(16)
        self.relu1 = _act(elu, 16)

    def forward(self, x):
        y = self.bn1(self.conv1(x))
        rep = int(self.num_features / self.in_channels)
        return self.relu1(y + x.repeat(1, rep, 1, 1, 1))


class DownTransition(nn.Module):
    """conv k2 s2 -> BN -> act -> nConvs x LUConv -> + down -> act (vnet3d.py:61-80)."""

    def __init__(self, inChans, nConvs, elu, dropout=False):
        super().__init__()
        out = 2 * inChans
        self.down_conv = nn.Conv3d(inChans, out, kernel_size=2, stride=2)
        self.bn1 = nn.BatchNorm3d(out)
        self.relu1 = _act(elu, out)
        self.relu2 = _act(elu, out)
        self.do1 = nn.Dropout3d() if dropout else nn.Identity()
        self.ops = _n_conv(out, nConvs, elu)

    def forward(self, x):
        down = self.relu1(self.bn1(self.down_conv(x)))
        y = self.ops(self.do1(down))
        return self.relu2(y + down)


class UpTransition(nn.Module):
    """Dropout3d(skip) -> ConvT k2 s2 -> BN -> act -> cat -> LUConvs -> + cat -> act
    (vnet3d.py:83-104).  ``do2`` (p=0.5) is always active in train mode."""

    def __init__(self, inChans, outChans, nConvs, elu, dropout=False):
        super().__init__()
        half = outChans // 2
        self.up_conv = nn.ConvTranspose3d(inChans, half, kernel_size=2, stride=2)
        self.bn1 = nn.BatchNorm3d(half)
        self.do1 = nn.Dropout3d() if dropout else nn.Identity()
        self.do2 = nn.Dropout3d()
        self.relu1 = _act(elu, half)
        self.relu2 = _act(elu, outChans)
        self.ops = _n_conv(outChans, nConvs, elu)

    def forward(self, x, skipx):
        skip = self.do2(skipx)
        up = self.relu1(self.bn1(self.up_conv(self.do1(x))))
        cat = torch.cat((up, skip), 1)
        return self.relu2(self.ops(cat) + cat)


class OutputTransition(nn.Module):
    """conv k5 -> BN -> act -> conv k1 (vnet3d.py:107-121)."""

    def __init__(self, in_channels, classes, elu):
        super().__init__()
        self.classes = classes
        self.conv1 = nn.Conv3d(in_channels, classes, kernel_size=5, padding=2)
        self.bn1 = nn.BatchNorm3d(classes)
        self.conv2 = nn.Conv3d(classes, classes, kernel_size=1)
        self.relu1 = _act(elu, classes)

    def forward(self, x):
        return self.conv2(self.relu1(self.bn1(self.conv1(x))))


class VNet(nn.Module):
    """V-Net (vnet3d.py:124-157)."""

    def __init__(self, elu=True, in_channels=1, classes=2):
        super().__init__()
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

    def forward(self, x):
        o16 = self.in_tr(x)
        o32 = self.down_tr32(o16)
        o64 = self.down_tr64(o32)
        o128 = self.down_tr128(o64)
        o256 = self.down_tr256(o128)
        h = self.up_tr256(o256, o128)
        h = self.up_tr128(h, o64)
        h = self.up_tr64(h, o32)
        h = self.up_tr32(h, o16)
        return self.out_tr(h)


# --------------------------------------------------------------------------- Residual U-Net
def _c3(cin, cout, stride=1):
    return nn.Conv3d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _c1(cin, cout):
    return nn.Conv3d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False)


class ResUNet(nn.Module):
    """Isensee-style residual U-Net; the reference class is called ``UNet``
    (residual_unet3d.py:5).  Conv3d bias=False everywhere, InstanceNorm3d without
    affine, LeakyReLU(0.01), Dropout3d(0.6), nearest x2 upsampling, deep supervision.
    ``norm_lrelu_conv_c{2..5}`` are applied twice with shared weights (:126,128 ...)."""

    def __init__(self, in_channels, n_classes, base_n_filter=8):
        super().__init__()
        self.in_channels, self.n_classes, self.base_n_filter = in_channels, n_classes, base_n_filter
        b = base_n_filter
        self.lrelu = nn.LeakyReLU()
        self.dropout3d = nn.Dropout3d(p=0.6)
        self.upsacle = nn.Upsample(scale_factor=2, mode="nearest")
        self.softmax = nn.Softmax(dim=1)

        self.conv3d_c1_1 = _c3(in_channels, b)
        self.conv3d_c1_2 = _c3(b, b)
        self.lrelu_conv_c1 = nn.Sequential(nn.LeakyReLU(), _c3(b, b))
        self.inorm3d_c1 = nn.InstanceNorm3d(b)
        for lvl, mult in ((2, 2), (3, 4), (4, 8), (5, 16)):
            w = b * mult
            setattr(self, f"conv3d_c{lvl}", _c3(w // 2, w, stride=2))
            setattr(self, f"norm_lrelu_conv_c{lvl}", self._nlc(w, w))
            if lvl < 5:
                setattr(self, f"inorm3d_c{lvl}", nn.InstanceNorm3d(w))
        self.norm_lrelu_upscale_conv_norm_lrelu_l0 = self._nlucnl(b * 16, b * 8)
        self.conv3d_l0 = _c1(b * 8, b * 8)
        self.inorm3d_l0 = nn.InstanceNorm3d(b * 8)
        for lvl, mult in ((1, 16), (2, 8), (3, 4)):
            w = b * mult
            setattr(self, f"conv_norm_lrelu_l{lvl}", self._cnl(w, w))
            setattr(self, f"conv3d_l{lvl}", _c1(w, w // 2))
            setattr(self, f"norm_lrelu_upscale_conv_norm_lrelu_l{lvl}", self._nlucnl(w // 2, w // 4))
        self.conv_norm_lrelu_l4 = self._cnl(b * 2, b * 2)
        self.conv3d_l4 = _c1(b * 2, n_classes)
        self.ds2_1x1_conv3d = _c1(b * 8, n_classes)
        self.ds3_1x1_conv3d = _c1(b * 4, n_classes)
        self.sigmoid = nn.Sigmoid()

    @staticmethod
    def _cnl(cin, cout):            # residual_unet3d.py:82-86
        return nn.Sequential(_c3(cin, cout), nn.InstanceNorm3d(cout), nn.LeakyReLU())

    @staticmethod
    def _nlc(cin, cout):            # :88-92
        return nn.Sequential(nn.InstanceNorm3d(cin), nn.LeakyReLU(), _c3(cin, cout))

    @staticmethod
    def _nlucnl(cin, cout):         # :99-107
        return nn.Sequential(nn.InstanceNorm3d(cin), nn.LeakyReLU(),
                             nn.Upsample(scale_factor=2, mode="nearest"),
                             _c3(cin, cout), nn.InstanceNorm3d(cout), nn.LeakyReLU())

    def forward(self, x):
        # level 1 context (:110-121): note inorm is applied to the *pre*-lrelu sum
        h = self.conv3d_c1_1(x)
        res = h
        h = self.conv3d_c1_2(self.lrelu(h))
        h = self.lrelu_conv_c1(self.dropout3d(h))
        h = h + res
        ctx = [self.lrelu(h)]
        h = self.lrelu(self.inorm3d_c1(h))
        # levels 2..5 (:123-168)
        for lvl in (2, 3, 4, 5):
            h = getattr(self, f"conv3d_c{lvl}")(h)
            res = h
            blk = getattr(self, f"norm_lrelu_conv_c{lvl}")
            h = blk(self.dropout3d(blk(h)))
            h = h + res
            if lvl < 5:
                h = self.lrelu(getattr(self, f"inorm3d_c{lvl}")(h))
                ctx.append(h)
        h = self.norm_lrelu_upscale_conv_norm_lrelu_l0(h)
        h = self.lrelu(self.inorm3d_l0(self.conv3d_l0(h)))
        # localisation (:174-194)
        ds = {}
        for lvl in (1, 2, 3):
            h = torch.cat([h, ctx[4 - lvl]], dim=1)
            h = getattr(self, f"conv_norm_lrelu_l{lvl}")(h)
            ds[lvl] = h
            h = getattr(self, f"conv3d_l{lvl}")(h)
            h = getattr(self, f"norm_lrelu_upscale_conv_norm_lrelu_l{lvl}")(h)
        h = torch.cat([h, ctx[0]], dim=1)
        out_pred = self.conv3d_l4(self.conv_norm_lrelu_l4(h))
        # deep supervision (:196-202)
        s = self.upsacle(self.ds2_1x1_conv3d(ds[2])) + self.ds3_1x1_conv3d(ds[3])
        return out_pred + self.upsacle(s)


# --------------------------------------------------------------------------- UNETR
class _SingleDeconv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2, padding=0, output_padding=0)

    def forward(self, x):
        return self.block(x)


class _SingleConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.block = nn.Conv3d(cin, cout, kernel_size=k, stride=1, padding=(k - 1) // 2)

    def forward(self, x):
        return self.block(x)


class _ConvBlock(nn.Module):        # unetr.py:27-37
    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.block = nn.Sequential(_SingleConv(cin, cout, k), nn.BatchNorm3d(cout), nn.ReLU(True))

    def forward(self, x):
        return self.block(x)


class _DeconvBlock(nn.Module):      # unetr.py:40-51
    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.block = nn.Sequential(_SingleDeconv(cin, cout), _SingleConv(cout, cout, k), nn.BatchNorm3d(cout), nn.ReLU(True))

    def forward(self, x):
        return self.block(x)


class _SelfAttention(nn.Module):    # unetr.py:54-99
    def __init__(self, heads, E, dropout):
        super().__init__()
        self.h, self.d = heads, int(E / heads)
        self.query, self.key, self.value = nn.Linear(E, E), nn.Linear(E, E), nn.Linear(E, E)
        self.out = nn.Linear(E, E)
        self.attn_dropout, self.proj_dropout = nn.Dropout(dropout), nn.Dropout(dropout)

    def forward(self, x):
        B, P, E = x.shape
        split = lambda t: t.view(B, P, self.h, self.d).permute(0, 2, 1, 3)
        q, k, v = split(self.query(x)), split(self.key(x)), split(self.value(x))
        s = torch.matmul(q, k.transpose(-1, -2)) / (self.d ** 0.5)
        p = self.attn_dropout(torch.softmax(s, dim=-1))
        c = torch.matmul(p, v).permute(0, 2, 1, 3).contiguous().view(B, P, E)
        return self.proj_dropout(self.out(c))


class _FFN(nn.Module):              # unetr.py:116-125 (dropout default 0.1 regardless of the model's argument)
    def __init__(self, E, dff=2048, dropout=0.1):
        super().__init__()
        self.w_1, self.w_2, self.dropout = nn.Linear(E, dff), nn.Linear(dff, E), nn.Dropout(dropout)

    def forward(self, x):
        return self.w_2(self.dropout(torch.relu(self.w_1(x))))


class _Block(nn.Module):            # unetr.py:148-168
    def __init__(self, E, heads, dropout):
        super().__init__()
        self.attention_norm, self.mlp_norm = nn.LayerNorm(E, eps=1e-6), nn.LayerNorm(E, eps=1e-6)
        self.mlp = _FFN(E, 2048)
        self.attn = _SelfAttention(heads, E, dropout)

    def forward(self, x):
        x = self.attn(self.attention_norm(x)) + x
        return self.mlp(self.mlp_norm(x)) + x


class _Embeddings(nn.Module):       # unetr.py:128-145
    def __init__(self, cin, E, cube, patch, dropout):
        super().__init__()
        self.n_patches = int(cube[0] * cube[1] * cube[2] / patch ** 3)
        self.patch_embeddings = nn.Conv3d(cin, E, kernel_size=patch, stride=patch)
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, E))
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        t = self.patch_embeddings(x).flatten(2).transpose(-1, -2)
        return self.dropout(t + self.position_embeddings)


class _Transformer(nn.Module):      # unetr.py:171-191
    def __init__(self, cin, E, cube, patch, heads, layers, dropout, taps):
        super().__init__()
        self.embeddings = _Embeddings(cin, E, cube, patch, dropout)
        self.layer = nn.ModuleList([_Block(E, heads, dropout) for _ in range(layers)])
        self.encoder_norm = nn.LayerNorm(E, eps=1e-6)      # never applied in the reference forward
        self.taps = taps

    def forward(self, x):
        out, h = [], self.embeddings(x)
        for i, blk in enumerate(self.layer):
            h = blk(h)
            if i + 1 in self.taps:
                out.append(h)
        return out


class UNETR(nn.Module):
    """UNETR (unetr.py:194-294)."""

    def __init__(self, img_shape=(128, 128, 128), input_dim=4, output_dim=3, embed_dim=768, patch_size=16, num_heads=12, dropout=0.1):
        super().__init__()
        E = self.embed_dim = embed_dim
        self.patch_dim = [int(s / patch_size) for s in img_shape]
        self.transformer = _Transformer(input_dim, E, img_shape, patch_size, num_heads, 12, dropout, [3, 6, 9, 12])
        self.decoder0 = nn.Sequential(_ConvBlock(input_dim, 32, 3), _ConvBlock(32, 64, 3))
        self.decoder3 = nn.Sequential(_DeconvBlock(E, 512), _DeconvBlock(512, 256), _DeconvBlock(256, 128))
        self.decoder6 = nn.Sequential(_DeconvBlock(E, 512), _DeconvBlock(512, 256))
        self.decoder9 = _DeconvBlock(E, 512)
        self.decoder12_upsampler = _SingleDeconv(E, 512)
        self.decoder9_upsampler = nn.Sequential(_ConvBlock(1024, 512), _ConvBlock(512, 512), _ConvBlock(512, 512), _SingleDeconv(512, 256))
        self.decoder6_upsampler = nn.Sequential(_ConvBlock(512, 256), _ConvBlock(256, 256), _SingleDeconv(256, 128))
        self.decoder3_upsampler = nn.Sequential(_ConvBlock(256, 128), _ConvBlock(128, 128), _SingleDeconv(128, 64))
        self.decoder0_header = nn.Sequential(_ConvBlock(128, 64), _ConvBlock(64, 64), _SingleConv(64, output_dim, 1))

    def forward(self, x):
        vol = lambda t: t.transpose(-1, -2).reshape(-1, self.embed_dim, *self.patch_dim)
        z3, z6, z9, z12 = [vol(t) for t in self.transformer(x)]
        z12 = self.decoder12_upsampler(z12)
        z9 = self.decoder9_upsampler(torch.cat([self.decoder9(z9), z12], dim=1))
        z6 = self.decoder6_upsampler(torch.cat([self.decoder6(z6), z9], dim=1))
        z3 = self.decoder3_upsampler(torch.cat([self.decoder3(z3), z6], dim=1))
        return self.decoder0_header(torch.cat([self.decoder0(x), z3], dim=1))
