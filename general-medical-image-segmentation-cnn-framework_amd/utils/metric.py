"""Dice / Jaccard metric on the MI355X -- drop-in for ``metric(gt, pred, spacing=None)``
of the reference's utils/metric.py:20-75.  The integer counters are reduced on the device
(exact), only four int64 values cross PCIe; the final ratios use the reference's formula."""
import torch

from .. import functional as F


def metric_from_counts(counts):
    """counts = (sum gt, sum pred, nnz(gt&pred), nnz(gt|pred)) -> (jaccard, dice), metric.py:65-66."""
    gsum, psum, inter, union = [int(v) for v in counts]
    smooth = 0.001
    return inter / (union + smooth), 2 * inter / (gsum + psum + smooth)


def metric(gt, pred, spacing=None):
    if spacing:
        raise NotImplementedError("HD95 (monai) branch is out of scope (SURVEY.md section 2 row 10)")
    gt = gt.to(torch.int64) if gt.dtype != torch.int64 else gt
    pred = pred.to(torch.int64) if pred.dtype != torch.int64 else pred
    return metric_from_counts(F.dice_counts(gt, pred).cpu().tolist())
