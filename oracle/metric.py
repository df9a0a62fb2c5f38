"""Oracle metric: NumPy restatement of the reference's utils/metric.py:20-75.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  utils/metric.py itself cannot be
imported in the build container (it imports torchio and monai at module level), so
this restatement follows the source text and is pinned by hand-computed cases in
tests/test_oracle_metric.py.  Pure integer counting => exact.
"""
import numpy as np


def confusion_counts(gt, pred):
    """Integer quantities of utils/metric.py:26-43,57:
    values cast to int, squeezed; ``intersection = gdth & pred`` and
    ``union = gdth | pred`` are *bitwise* (a quirk for labels > 1);
    ``gdth_sum``/``pred_sum`` are value sums, the other two are non-zero counts."""
    g = np.asarray(gt).astype(int).squeeze()
    p = np.asarray(pred).astype(int).squeeze()
    inter = g & p
    union = g | p
    return {
        "gdth_sum": int(np.sum(g)),
        "pred_sum": int(np.sum(p)),
        "intersection_sum": int(np.count_nonzero(inter)),
        "union_sum": int(np.count_nonzero(union)),
        "tp": int(np.sum(inter)),
        "fp": int(np.sum(np.where((p - g) < 1, 0, p))),
        "fn": int(np.sum(np.where((g - p) < 1, 0, g))),
        "tn": float(np.sum(np.ones(g.shape) - union)),
    }


def metric(gt, pred, spacing=None):
    """(jaccard, dice) exactly as utils/metric.py:65-66,72-75 when ``spacing`` is
    falsy.  ``gt``/``pred`` may be torch CPU tensors or arrays.  The HD95 branch
    (monai) is out of scope."""
    if spacing:
        raise NotImplementedError("HD95 branch (monai) is out of scope for the oracle")
    to_np = lambda a: a.detach().numpy() if hasattr(a, "detach") else np.asarray(a)
    c = confusion_counts(to_np(gt), to_np(pred))
    smooth = 0.001
    jaccard = c["intersection_sum"] / (c["union_sum"] + smooth)
    dice = 2 * c["intersection_sum"] / (c["gdth_sum"] + c["pred_sum"] + smooth)
    return jaccard, dice


def rates(gt, pred):
    """precision / recall / specificity / FPR / FNR of utils/metric.py:57-63."""
    c = confusion_counts(gt, pred)
    s = 0.001
    return {
        "precision": c["tp"] / (c["pred_sum"] + s),
        "recall": c["tp"] / (c["gdth_sum"] + s),
        "specificity": c["tn"] / (c["tn"] + c["fp"] + s),
        "false_positive_rate": c["fp"] / (c["fp"] + c["tn"] + s),
        "false_negtive_rate": c["fn"] / (c["fn"] + c["tp"] + s),
    }
