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
