"""Residual U-Net (Isensee et al.) on the MI355X kernels -- drop-in for the reference's
models/three_d/residual_unet3d.py (class ``UNet``).

Same constructor (``UNet(in_channels, n_classes, base_n_filter=8)``, residual_unet3d.py:11), the
same 26 conv-weight ``state_dict`` keys and forward semantics
(residual_unet3d.py:109-204): bias-free convs, InstanceNorm3d without affine fused with
LeakyReLU(0.01), Dropout3d(0.6), nearest x2 upsampling, weight-shared
``norm_lrelu_conv_c{2..5}`` applied twice per level (autograd accumulates both weight
gradients), deep-supervision sum.
"""
import torch.nn as nn

from ... import functional as F
from ...layers import Conv3d, Dropout3d, InstanceNorm3d, LeakyReLU, Upsample

_LRELU = (F.ACT_LRELU, 0.01)


class _NormLreluConv(nn.Sequential):
    """[InstanceNorm3d, LeakyReLU, Conv3d] with the norm and activation fused (keys '.2.weight')."""

    def forward(self, x, residual=None):
        norm, _act, conv = self.children()
        return conv(norm.forward_act(x, *_LRELU), residual=residual)       # residual: the block's sum rides in the convolution


class _ConvNormLrelu(nn.Sequential):
    """[Conv3d, InstanceNorm3d, LeakyReLU] (keys '.0.weight')."""

    def forward(self, x):
        conv, norm, _act = self.children()
        return F.conv_in_act(x, conv, norm, *_LRELU)       # batch of one: the statistics come out of the convolution's epilogue


class _NormLreluUpConvNormLrelu(nn.Sequential):
    """[InstanceNorm3d, LeakyReLU, Upsample, Conv3d, InstanceNorm3d, LeakyReLU] (keys '.3.weight')."""

    def forward(self, x, cat_right=None):
        """``cat_right``: the skip tensor the result is concatenated with (residual_unet3d.py:183-209) -- the result is then
        cat((block(x), cat_right), channels), written in place when cat_right was produced as the right slice of a concat buffer."""
        n0, _a0, up, conv, n1, _a1 = self.children()
        return F.conv_in_act(up(n0.forward_act(x, *_LRELU)), conv, n1, *_LRELU, cat_right=cat_right)


def _conv3(cin, cout, stride=1):
    return Conv3d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _conv1(cin, cout):
    return Conv3d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False)


class UNet(nn.Module):
    def __init__(self, in_channels, n_classes, base_n_filter=8):
        super().__init__()
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.base_n_filter = b = base_n_filter
        self.lrelu = LeakyReLU()
        self.dropout3d = Dropout3d(p=0.6)
        self.upsacle = Upsample(scale_factor=2, mode="nearest")
        self.softmax = nn.Softmax(dim=1)          # unused members kept for attribute parity
        self.sigmoid = nn.Sigmoid()

        self.conv3d_c1_1 = _conv3(in_channels, b)
        self.conv3d_c1_2 = _conv3(b, b)
        self.lrelu_conv_c1 = self.lrelu_conv(b, b)
        self.inorm3d_c1 = InstanceNorm3d(b)
        self.conv3d_c2 = _conv3(b, b * 2, 2)
        self.norm_lrelu_conv_c2 = self.norm_lrelu_conv(b * 2, b * 2)
        self.inorm3d_c2 = InstanceNorm3d(b * 2)
        self.conv3d_c3 = _conv3(b * 2, b * 4, 2)
        self.norm_lrelu_conv_c3 = self.norm_lrelu_conv(b * 4, b * 4)
        self.inorm3d_c3 = InstanceNorm3d(b * 4)
        self.conv3d_c4 = _conv3(b * 4, b * 8, 2)
        self.norm_lrelu_conv_c4 = self.norm_lrelu_conv(b * 8, b * 8)
        self.inorm3d_c4 = InstanceNorm3d(b * 8)
        self.conv3d_c5 = _conv3(b * 8, b * 16, 2)
        self.norm_lrelu_conv_c5 = self.norm_lrelu_conv(b * 16, b * 16)
        self.norm_lrelu_upscale_conv_norm_lrelu_l0 = self.norm_lrelu_upscale_conv_norm_lrelu(b * 16, b * 8)
        self.conv3d_l0 = _conv1(b * 8, b * 8)
        self.inorm3d_l0 = InstanceNorm3d(b * 8)
        self.conv_norm_lrelu_l1 = self.conv_norm_lrelu(b * 16, b * 16)
        self.conv3d_l1 = _conv1(b * 16, b * 8)
        self.norm_lrelu_upscale_conv_norm_lrelu_l1 = self.norm_lrelu_upscale_conv_norm_lrelu(b * 8, b * 4)
        self.conv_norm_lrelu_l2 = self.conv_norm_lrelu(b * 8, b * 8)
        self.conv3d_l2 = _conv1(b * 8, b * 4)
        self.norm_lrelu_upscale_conv_norm_lrelu_l2 = self.norm_lrelu_upscale_conv_norm_lrelu(b * 4, b * 2)
        self.conv_norm_lrelu_l3 = self.conv_norm_lrelu(b * 4, b * 4)
        self.conv3d_l3 = _conv1(b * 4, b * 2)
        self.norm_lrelu_upscale_conv_norm_lrelu_l3 = self.norm_lrelu_upscale_conv_norm_lrelu(b * 2, b)
        self.conv_norm_lrelu_l4 = self.conv_norm_lrelu(b * 2, b * 2)
        self.conv3d_l4 = _conv1(b * 2, n_classes)
        self.ds2_1x1_conv3d = _conv1(b * 8, n_classes)
        self.ds3_1x1_conv3d = _conv1(b * 4, n_classes)

    # the reference's factory names (residual_unet3d.py:82-107)
    def conv_norm_lrelu(self, feat_in, feat_out):
        return _ConvNormLrelu(_conv3(feat_in, feat_out), InstanceNorm3d(feat_out), LeakyReLU())

    def norm_lrelu_conv(self, feat_in, feat_out):
        return _NormLreluConv(InstanceNorm3d(feat_in), LeakyReLU(), _conv3(feat_in, feat_out))

    def lrelu_conv(self, feat_in, feat_out):
        return nn.Sequential(LeakyReLU(), _conv3(feat_in, feat_out))

    def norm_lrelu_upscale_conv_norm_lrelu(self, feat_in, feat_out):
        return _NormLreluUpConvNormLrelu(InstanceNorm3d(feat_in), LeakyReLU(), Upsample(scale_factor=2, mode="nearest"),
                                         _conv3(feat_in, feat_out), InstanceNorm3d(feat_out), LeakyReLU())

    def forward(self, x):
        h = F.to_channels_last(x)
        # level 1 context (:110-121): the sum feeds LeakyReLU (context_1) and InstanceNorm->LeakyReLU
        h = self.conv3d_c1_1(h)
        # (:110-121) the level-1 tensor feeds a LeakyReLU and, unchanged, the block sum: one backward pass forms both gradients' sum
        a, res = F.activation_fork(h, F.ACT_LRELU, self.lrelu.negative_slope)
        h = self.conv3d_c1_2(a)
        act1, conv1 = self.lrelu_conv_c1.children()
        h = conv1(act1(self.dropout3d(h)), residual=res)       # (:121) conv + residual in one launch
        # the four context tensors are born as the RIGHT channel slices of their levels' concat buffers; the localisation path's blocks
        # write their outputs into the left slices (torch.cat of :183-209 without the copies)
        left = [self.norm_lrelu_upscale_conv_norm_lrelu_l3[3].out_channels, self.norm_lrelu_upscale_conv_norm_lrelu_l2[3].out_channels,
                self.norm_lrelu_upscale_conv_norm_lrelu_l1[3].out_channels, self.conv3d_l0.out_channels]
        c1, h = F.activation_fork(h, F.ACT_LRELU, self.lrelu.negative_slope, left_pad=left[0])     # context_1 = lrelu(sum); the sum itself goes on into the norm
        ctx = [c1]
        h = self.inorm3d_c1.forward_act(h, *_LRELU)
        for lvl in (2, 3, 4, 5):                               # (:123-168)
            h = getattr(self, f"conv3d_c{lvl}")(h)
            res = h
            blk = getattr(self, f"norm_lrelu_conv_c{lvl}")     # shared weights, applied twice
            h = blk(self.dropout3d(blk(h)), residual=res)       # (:131,:141,...) the residual sum in the second convolution's epilogue
            if lvl < 5:
                h = getattr(self, f"inorm3d_c{lvl}").forward_act(h, *_LRELU, left_pad=left[lvl - 1])
                ctx.append(h)
        h = self.norm_lrelu_upscale_conv_norm_lrelu_l0(h)
        h = F.conv_in_act(h, self.conv3d_l0, self.inorm3d_l0, *_LRELU, cat_right=ctx[3])      # ... and the cat with context_4 (:174-176)
        ds = {}
        for lvl in (1, 2, 3):                                  # localisation (:174-194); h arrives concatenated with its level's context
            h = getattr(self, f"conv_norm_lrelu_l{lvl}")(h)
            ds[lvl] = h
            h = getattr(self, f"conv3d_l{lvl}")(h)
            h = getattr(self, f"norm_lrelu_upscale_conv_norm_lrelu_l{lvl}")(h, cat_right=ctx[3 - lvl])
        out_pred = self.conv3d_l4(self.conv_norm_lrelu_l4(h))
        s = F.activation(self.upsacle(self.ds2_1x1_conv3d(ds[2])), F.ACT_NONE, residual=self.ds3_1x1_conv3d(ds[3]))
        out = F.activation(out_pred, F.ACT_NONE, residual=self.upsacle(s))
        return F.to_channels_first(out)
