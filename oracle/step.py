"""Oracle train step: CPU restatement of the reference's per-iteration glue.

TEST INFRASTRUCTURE -- see oracle/__init__.py.
Follows train.py:187-221 (zero_grad, 2-channel gt, forward, argmax, BCE, backward,
Adam step, Dice metric) with anomaly detection off, and train.py:33-61 for the
weight-init policy.  Also the timed body of bench.py's ``cpu_baseline`` leg.
"""
import torch
import torch.nn as nn

from .losses import bce_with_logits
from .metric import metric


def weights_init_normal(init_type):
    """train.py:33-61.  Only classes whose name contains 'BatchNorm2d' get the BN
    branch, so BatchNorm3d keeps (1, 0); every Conv*/Linear weight gets ``init_type``
    and every such bias is zeroed."""
    def init_func(m):
        cls = m.__class__.__name__
        gain = 0.02
        if "BatchNorm2d" in cls:
            if getattr(m, "weight", None) is not None:
                nn.init.normal_(m.weight.data, 1.0, gain)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif hasattr(m, "weight") and ("Conv" in cls or "Linear" in cls):
            w = m.weight.data
            if init_type == "normal":
                nn.init.normal_(w, 0.0, gain)
            elif init_type == "xavier":
                nn.init.xavier_normal_(w, gain=gain)
            elif init_type == "xavier_uniform":
                nn.init.xavier_uniform_(w, gain=1.0)
            elif init_type == "kaiming":
                nn.init.kaiming_normal_(w, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                nn.init.orthogonal_(w, gain=gain)
            elif init_type == "none":
                m.reset_parameters()
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
    return init_func


def two_channel_gt(gt):
    """train.py:190-193: background channel = (gt == 0), concatenated in front."""
    back = torch.zeros_like(gt)
    back[gt == 0] = 1
    return torch.cat([back, gt], dim=1)


def train_step(model, optimizer, x, gt, criterion=bce_with_logits):
    """One iteration of train.py:187-221.  ``gt`` is the single-channel label volume
    [N,1,D,H,W].  Returns (pred, mask, loss, (jaccard, dice))."""
    optimizer.zero_grad()
    gt2 = two_channel_gt(gt).float()
    x = x.float()
    pred = model(x)
    mask = pred.argmax(dim=1, keepdim=True)
    loss = criterion(pred, gt2)
    loss.backward()
    optimizer.step()
    jd = metric(gt2.argmax(dim=1, keepdim=True), mask)
    return pred, mask, loss, jd


def frequency_bands(x, limit=0.04):
    """train.py:76-88,198-200 (``low_pass_torch`` / ``high_pass_torch``): keep |f| < limit (low) or |f| > limit (high)
    on the last two axes.  The forward rfftn runs over ALL axes of the 5-D batch while the inverse runs over the last
    three only -- upstream's behaviour, the identity on the extra axes when batch = channel = 1."""
    import torch.fft as fft
    spec = fft.rfftn(x)
    out = []
    for cmp in (torch.lt, torch.gt):
        keep_w = cmp(torch.abs(fft.rfftfreq(x.shape[-1])), limit)
        keep_h = cmp(torch.abs(fft.fftfreq(x.shape[-2])), limit)
        out.append(fft.irfftn(spec * torch.outer(keep_h, keep_w).to(x), s=x.shape[-3:]))
    return out[0], out[1]

