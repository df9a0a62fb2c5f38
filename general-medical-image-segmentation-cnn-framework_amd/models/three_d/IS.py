"""The reference's "IS" network (models/three_d/IS.py) on the MI355X kernels -- registry key ``IS`` (train.py:340-343).

What the reference builds (IS.py:10-130): three complete U-Net parameter sets, suffixed "", "_" and "__" (51 modules,
same ``_block`` child names in each), and two 1x1x1 heads ``conv`` / ``conv_``.  What its forward uses (IS.py:132-190):
ONE encoder + bottleneck (the unsuffixed set) applied in turn to the volume, its low-pass and its high-pass band, and
one decoder set per band; ``outputs1 = conv(dec1)``, ``outputs2 = conv_(dec1 + dec1_ + dec1__)``.  The suffixed encoder
/ bottleneck sets are constructed, initialised, checkpointed -- and never called.  All of that is kept: same
constructor, same state_dict keys, same call pattern (so the shared encoder's BatchNorm running statistics advance
three times per forward, as upstream), with every op a kernel of libmi355seg.so.

``frequency_bands`` is the caller-side split of train.py:76-88,198-201 (``low_pass_torch`` / ``high_pass_torch`` at
limit 0.04): an FFT mask on the last two axes.  It is data preparation in front of the hot path and runs on rocFFT
through ``torch.fft``; including the upstream quirk that the forward transform covers all five axes while the inverse
covers only the last three (exact for batch = channel = 1).
"""
import torch
import torch.nn as nn

from ... import functional as F
from ...layers import Conv3d, ConvTranspose3d, MaxPool3d
from .unet3d import UNet3D as _PlainUNet

_SETS = ("", "_", "__")
_LEVELS = (1, 2, 3, 4)


def frequency_bands(x, limit=0.04):
    """-> (low_x, high_x) as train.py:198-200 computes them for the IS network."""
    fx = torch.fft.rfftn(x)
    f_last = torch.fft.rfftfreq(x.shape[-1]).abs()
    f_prev = torch.fft.fftfreq(x.shape[-2]).abs()
    bands = []
    for keep_last, keep_prev in ((f_last < limit, f_prev < limit), (f_last > limit, f_prev > limit)):
        mask = torch.outer(keep_prev, keep_last).to(x)
        bands.append(torch.fft.irfftn(fx * mask, s=x.shape[-3:]))
    return bands[0], bands[1]


class UNet3D(nn.Module):
    def __init__(self, in_channels=1, out_channels=3, init_features=64):
        super().__init__()
        f = init_features
        widths = {1: f, 2: 2 * f, 3: 4 * f, 4: 8 * f}
        for tag in _SETS:                                   # registration order of IS.py:18-120
            prev = in_channels
            for lvl in _LEVELS:
                setattr(self, f"encoder{lvl}{tag}", _PlainUNet._block(prev, widths[lvl], name=f"enc{lvl}"))
                setattr(self, f"pool{lvl}{tag}", MaxPool3d(kernel_size=2, stride=2))
                prev = widths[lvl]
            setattr(self, f"bottleneck{tag}", _PlainUNet._block(prev, 16 * f, name="bottleneck"))
            prev = 16 * f
            for lvl in reversed(_LEVELS):
                setattr(self, f"upconv{lvl}{tag}", ConvTranspose3d(prev, widths[lvl], kernel_size=2, stride=2))
                setattr(self, f"decoder{lvl}{tag}", _PlainUNet._block(2 * widths[lvl], widths[lvl], name=f"dec{lvl}"))
                prev = widths[lvl]
        self.conv = Conv3d(in_channels=f, out_channels=out_channels, kernel_size=1)
        self.conv_ = Conv3d(in_channels=f, out_channels=out_channels, kernel_size=1)

    takes_frequency_bands = True                           # engine.train_step / predict feed (x, low_x, high_x)

    def _branch(self, volume, tag):
        """Shared encoder + bottleneck, then decoder set ``tag`` (IS.py:133-150 / 152-169 / 171-188)."""
        h = F.to_channels_last(volume)
        skips = {}
        for lvl in _LEVELS:
            up = getattr(self, f"upconv{lvl}{tag}")
            # encoder output lands in the right half of its level's concat buffer; the up-convolution fills the left half
            h, skips[lvl] = F.max_pool3d_2x_and_skip(getattr(self, f"encoder{lvl}")(h, left_pad=up.out_channels))
        h = self.bottleneck(h)
        for lvl in reversed(_LEVELS):
            up = getattr(self, f"upconv{lvl}{tag}")
            h = getattr(self, f"decoder{lvl}{tag}")(F.conv_transpose3d_k2s2_cat(h, up.weight, up.bias, skips[lvl]))
        return h

    def forward(self, x, low_x, high_x):
        dec, dec_low, dec_high = (self._branch(v, tag) for v, tag in zip((x, low_x, high_x), _SETS))
        fused = F.activation(F.activation(dec, F.ACT_NONE, residual=dec_low), F.ACT_NONE, residual=dec_high)
        return F.to_channels_first(self.conv(dec)), F.to_channels_first(self.conv_(fused))
