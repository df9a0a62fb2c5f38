"""Leaf modules of the drop-in models.  Each subclasses the torch.nn module it replaces
purely for parameter registration / initialisation / ``state_dict`` naming (so the
reference's ``weights_init_normal`` and checkpoints keep working) and routes ``forward``
to the HIP kernels on channel-last [N,D,H,W,C] tensors."""
import torch
import torch.nn as nn

from . import functional as F

_ACT_OF = {nn.ReLU: (F.ACT_RELU, 0.0), nn.ELU: (F.ACT_ELU, 1.0), nn.LeakyReLU: (F.ACT_LRELU, 0.01)}


def _iso(v, what):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError(f"{what} must be isotropic, got {v}")
        return int(v[0])
    return int(v)


class Conv3d(nn.Conv3d):
    """nn.Conv3d (cubic kernel, isotropic stride/padding, dilation 1, groups 1)."""

    def forward(self, x, residual=None):
        """residual: conv(x) + residual in one launch where the kernel allows (functional.conv3d)."""
        if self.groups != 1 or _iso(self.dilation, "dilation") != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("Conv3d: only groups=1, dilation=1, zero padding are implemented")
        return F.conv3d(x, self.weight, self.bias, _iso(self.stride, "stride"), _iso(self.padding, "padding"), residual=residual)


class ConvTranspose3d(nn.ConvTranspose3d):
    """nn.ConvTranspose3d with kernel_size = stride (non-overlapping up-sampling) and no padding: k = 2 is the U-Net
    family's form (dedicated kernels), any other k (CSRNet's 4) runs as the adjoint of the matching Conv3d."""

    def forward(self, x):
        k, s = _iso(self.kernel_size, "kernel_size"), _iso(self.stride, "stride")
        if k != s or _iso(self.padding, "padding") != 0 or _iso(self.output_padding, "output_padding") != 0 or \
                _iso(self.dilation, "dilation") != 1 or self.groups != 1:
            raise NotImplementedError("ConvTranspose3d: only kernel_size == stride, padding=0, output_padding=0 is implemented")
        if k == 2:
            return F.conv_transpose3d_k2s2(x, self.weight, self.bias)
        return F.conv_transpose3d_adjoint(x, self.weight, self.bias, k)


class BatchNorm3d(nn.BatchNorm3d):
    """nn.BatchNorm3d; ``forward_act`` fuses the following activation (and an optional
    residual add in front of it) into the normalisation kernel."""

    def forward_act(self, x, act=F.ACT_NONE, slope=0.01, residual=None):
        if self.momentum is None:
            raise NotImplementedError("BatchNorm3d: cumulative moving average (momentum=None) is not implemented")
        training = self.training or (self.running_mean is None)
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            F.bump_counter(self)
        return F.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, training,
                                self.momentum, self.eps, act, slope, residual)

    def forward(self, x):
        return self.forward_act(x)


class InstanceNorm3d(nn.InstanceNorm3d):
    def forward_act(self, x, act=F.ACT_NONE, slope=0.01, left_pad=0):
        if self.affine or self.track_running_stats:
            raise NotImplementedError("InstanceNorm3d: only affine=False, track_running_stats=False is implemented")
        return F.instance_norm_act(x, self.eps, act, slope, left_pad=left_pad)

    def forward(self, x):
        return self.forward_act(x)


class MaxPool3d(nn.MaxPool3d):
    def forward(self, x):
        if _iso(self.kernel_size, "kernel_size") != 2 or _iso(self.stride, "stride") != 2 or _iso(self.padding, "padding") != 0:
            raise NotImplementedError("MaxPool3d: only kernel_size=2, stride=2 is implemented")
        return F.max_pool3d_2x(x)


class Upsample(nn.Upsample):
    def forward(self, x):
        if self.mode != "nearest" or float(self.scale_factor) != 2.0:
            raise NotImplementedError("Upsample: only scale_factor=2, mode='nearest' is implemented")
        return F.upsample_nearest_2x(x)


class Dropout3d(nn.Dropout3d):
    """nn.Dropout3d: whole channels of a sample are zeroed with probability p and the rest scaled by
    1/(1-p).  The keep-mask comes from the device RNG; parity tests inject the oracle's mask through
    ``forced_masks`` (a list consumed front to back, each a bool/float tensor [N, C])."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__(p, inplace)
        self.forced_masks = []

    def draw_scale(self, N, C, device):
        """The per-(sample, channel) factor of this call -- keep / (1 - p) -- or None when the layer is the identity (eval, p = 0)."""
        if not self.training or self.p == 0.0:
            return None
        if self.forced_masks:
            keep = self.forced_masks.pop(0).to(device=device, dtype=torch.float32).reshape(N, C)
        else:
            return torch.nn.functional.dropout(_ones((N, C), device), self.p, True)      # keep / (1 - p) in one launch
        return keep / (1.0 - self.p)

    def forward(self, x):
        scale = self.draw_scale(x.shape[0], x.shape[-1], x.device)
        return x if scale is None else F.scale_channels(x, scale)


class ReLU(nn.ReLU):
    def forward(self, x):
        return F.activation(x, F.ACT_RELU)


class ELU(nn.ELU):
    def forward(self, x):
        if self.alpha != 1.0:
            raise NotImplementedError("ELU: only alpha=1 is implemented")
        return F.activation(x, F.ACT_ELU)


class PReLU(nn.PReLU):
    """nn.PReLU(num_parameters = channels): same parameter (``weight``), one HIP launch forward, two backward."""

    def forward(self, x, residual=None):
        return F.prelu(x, self.weight, residual=residual)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        return F.activation(x, F.ACT_LRELU, self.negative_slope)


def act_code(m):
    """(code, slope) of an activation module."""
    for cls, (code, _) in _ACT_OF.items():
        if isinstance(m, cls):
            return code, (m.negative_slope if isinstance(m, nn.LeakyReLU) else 0.0)
    raise NotImplementedError(f"no fused form for activation {type(m).__name__}")


class Linear(nn.Linear):
    """nn.Linear on the MFMA GEMM; ``forward_relu`` fuses a following ReLU into the epilogue."""

    def forward(self, x, mask=None, residual=None):
        """``mask`` (a dropout layer's keep / (1 - p) factors) and ``residual``: (x W^T + b) * mask + residual in the GEMM's epilogue."""
        return F.linear(x, self.weight, self.bias, mask=mask, residual=residual)

    def forward_relu(self, x, mask=None):
        return F.linear(x, self.weight, self.bias, relu=True, mask=mask)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        if len(self.normalized_shape) != 1 or not self.elementwise_affine:
            raise NotImplementedError("LayerNorm: only 1-D normalized_shape with affine is implemented")
        return F.layer_norm(x, self.weight, self.bias, self.eps)

    def forward_fork(self, x):
        """(LayerNorm(x), x): for x + f(LN(x)) blocks -- the backward sums the residual stream's gradient inside the norm's backward kernel."""
        if len(self.normalized_shape) != 1 or not self.elementwise_affine:
            raise NotImplementedError("LayerNorm: only 1-D normalized_shape with affine is implemented")
        return F.layer_norm_fork(x, self.weight, self.bias, self.eps)


_ONES = {}


def _ones(shape, device):
    key = (shape, str(device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(shape, device=device)
    return t


class Dropout(nn.Dropout):
    """Element-wise nn.Dropout on token tensors [..., E]; the arithmetic is the channel-scale kernel with one
    "sample" per token row.  ``forced_masks`` as in Dropout3d (mask shape = input shape)."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__(p, inplace)
        self.forced_masks = []

    def draw(self, shape, device):
        if self.forced_masks:
            return self.forced_masks.pop(0).to(device=device, dtype=torch.float32).reshape(shape) / (1.0 - self.p)
        # a slice of the step's pooled draw (functional.dropout_pool_*: one launch per training step for all layers), else one launch:
        # torch's fused dropout on a cached tensor of ones draws the keep mask AND scales it (bernoulli_ + div_ were two)
        return F.dropout_pool_take(tuple(shape), self.p, device, lambda: torch.nn.functional.dropout(_ones(tuple(shape), device), self.p, True))

    def mask_for(self, shape, device):
        """The keep / (1 - p) factors this layer would apply to a tensor of ``shape`` ([..., E]), or None when it is the identity
        (eval, p = 0): for callers that fold the product into the producing GEMM's epilogue (functional.linear(..., mask=))."""
        if not self.training or self.p == 0.0:
            return None
        rows = 1
        for e in shape[:-1]:
            rows *= int(e)
        return self.draw((rows, int(shape[-1])), device)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        E = x.shape[-1]
        x2 = x.contiguous().view(-1, 1, 1, 1, E)
        scale = self.draw((x2.shape[0], E), x.device)
        return F.scale_channels(x2, scale).view(x.shape)
