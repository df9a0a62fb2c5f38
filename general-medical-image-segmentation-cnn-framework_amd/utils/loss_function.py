"""Losses on the MI355X kernels -- drop-in for the reference's utils/loss_function.py
(same names, arguments and error behaviour).  The heavy reductions (BCE, Dice sums) are
HIP kernels; the scalar glue on the resulting 0-dim tensors is torch."""
import numpy as np
import torch
import torch.nn.functional as TF
from torch import nn

from .. import functional as F


class Binary_Loss(nn.Module):
    """loss_function.py:19-41 -- BCEWithLogitsLoss with mean reduction."""

    def forward(self, model_output, targets):
        return F.bce_with_logits(model_output, targets)


class BCEWithLogitsLoss(nn.Module):
    """The live criterion of the train loop (train.py:115)."""

    def forward(self, input, target):
        return F.bce_with_logits(input, target)


def cross_entropy_3D(input, target, weight=None, size_average=True):
    """loss_function.py:8-16 -- log_softmax over dim 1 + summed NLL, divided by the voxel count."""
    return F.cross_entropy_3d(input, target, weight, size_average)


def make_one_hot(input, num_classes):
    """loss_function.py:44-58 -- [N,1,*] int64 -> [N,K,*] float one-hot; like the reference the result
    lives on the CPU (it calls ``input.cpu()``)."""
    shape = np.array(input.shape)
    shape[1] = num_classes
    result = torch.zeros(tuple(shape))
    return result.scatter_(1, input.cpu(), 1)


class BinaryDiceLoss(nn.Module):
    """loss_function.py:61-99 -- per-sample 1 - (sum x*t + s) / (sum x^p + sum t^p + s), any exponent p; every sample's
    sums come from one row-segmented reduction launch."""

    def __init__(self, smooth=1, p=2, reduction="mean"):
        super().__init__()
        self.smooth, self.p, self.reduction = smooth, p, reduction

    def forward(self, predict, target):
        assert predict.shape[0] == target.shape[0], "predict & target batch size don't match"
        n = predict.shape[0]
        s = F.dice_rows_autograd(predict.contiguous().view(n, -1), target.contiguous().view(n, -1), False, self.p)
        loss = (1 - (s[:, 0] + self.smooth) / (s[:, 3] + s[:, 4] + self.smooth)).to(torch.float32)
        if self.reduction == "mean":
            return loss.mean()
        elif self.reduction == "sum":
            return loss.sum()
        elif self.reduction == "none":
            return loss
        else:
            raise Exception("Unexpected reduction {}".format(self.reduction))


class DiceLoss(nn.Module):
    """loss_function.py:102-130 -- global soft Dice on sigmoid(predict) against a one-hot target."""

    def __init__(self, weight=None, ignore_index=None, **kwargs):
        super().__init__()
        self.kwargs, self.weight, self.ignore_index = kwargs, weight, ignore_index
        self.eplison = 1e-5

    def forward(self, predict, target):
        assert predict.shape == target.shape, "predict & target shape do not match"
        s = F.dice_sums_autograd(predict, target, True)
        intersection, union = s[0], s[1] + s[2]
        return (1 - 2 * (intersection + self.eplison) / (union + self.eplison)).to(torch.float32)


class DiceLossss(nn.Module):
    """loss_function.py:148-185 -- per-class 1 - (2 sum s*t + eps) / (sum s^2 + sum t^2 + eps), weighted mean."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def _one_hot_encoder(self, input_tensor):
        return torch.stack([(input_tensor == i) for i in range(self.n_classes)], dim=1).float()

    def forward(self, inputs, target, weight=None, softmax=False):
        if softmax:
            inputs = F.softmax_channels(inputs)
        target = self._one_hot_encoder(target)
        if weight is None:
            weight = [1] * self.n_classes
        assert inputs.size() == target.size(), "predict & target shape do not match"
        smooth = 1e-5
        n = inputs.shape[0]
        # rows = (sample, class): one reduction launch for all of them, then the per-class sums over the batch
        s = F.dice_rows_autograd(inputs.contiguous().view(n * self.n_classes, -1), target.contiguous().view(n * self.n_classes, -1), False, 2.0)
        s = s.view(n, self.n_classes, 5).sum(dim=0)
        dice = 1 - (2 * s[:, 0] + smooth) / (s[:, 3] + s[:, 4] + smooth)
        w = torch.as_tensor(weight, dtype=torch.float64, device=dice.device)
        return ((dice * w).sum() / self.n_classes).to(torch.float32)
