"""Oracle losses: CPU restatement of the reference's utils/loss_function.py.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Plain torch on CPU tensors; each
function cites the reference lines it follows.
"""
import torch
import torch.nn.functional as F


def bce_with_logits(pred, target):
    """nn.BCEWithLogitsLoss() with mean reduction -- the live criterion
    (train.py:115,209; == Binary_Loss, utils/loss_function.py:19-41)."""
    return F.binary_cross_entropy_with_logits(pred, target)


def cross_entropy_3d(inp, target, weight=None, size_average=True):
    """utils/loss_function.py:8-16: log_softmax over dim 1, NLL summed over voxels,
    divided by the voxel count when ``size_average``."""
    n, c = inp.shape[0], inp.shape[1]
    log_p = F.log_softmax(inp, dim=1)
    log_p = log_p.permute(0, 2, 3, 4, 1).contiguous().view(-1, c)
    tgt = target.reshape(-1)
    loss = F.nll_loss(log_p, tgt, weight=weight, reduction="sum")
    if size_average:
        loss = loss / float(tgt.numel())
    return loss


def make_one_hot(labels, num_classes):
    """utils/loss_function.py:44-58: [N,1,*] int64 -> [N,K,*] float one-hot (on CPU)."""
    shape = list(labels.shape)
    shape[1] = num_classes
    return torch.zeros(shape).scatter_(1, labels.cpu(), 1)


def binary_dice_loss(predict, target, smooth=1, p=2, reduction="mean"):
    """BinaryDiceLoss.forward, utils/loss_function.py:82-99."""
    assert predict.shape[0] == target.shape[0], "predict & target batch size don't match"
    pr = predict.contiguous().view(predict.shape[0], -1)
    tg = target.contiguous().view(target.shape[0], -1)
    num = torch.sum(pr * tg, dim=1) + smooth
    den = torch.sum(pr.pow(p) + tg.pow(p), dim=1) + smooth
    loss = 1 - num / den
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":
        return loss
    raise Exception("Unexpected reduction {}".format(reduction))


def dice_loss(predict, target, eps=1e-5):
    """DiceLoss.forward, utils/loss_function.py:121-130: global soft Dice on sigmoid."""
    assert predict.shape == target.shape, "predict & target shape do not match"
    n = predict.size(0)
    pre = torch.sigmoid(predict).view(n, -1)
    tar = target.view(n, -1)
    inter = (pre * tar).sum(-1).sum()
    union = (pre + tar).sum(-1).sum()
    return 1 - 2 * (inter + eps) / (union + eps)


def dice_loss_multiclass(inputs, target, n_classes, weight=None, softmax=False):
    """DiceLossss.forward, utils/loss_function.py:148-185."""
    if softmax:
        inputs = torch.softmax(inputs, dim=1)
    onehot = torch.stack([(target == i) for i in range(n_classes)], dim=1).float()
    if weight is None:
        weight = [1] * n_classes
    assert inputs.size() == onehot.size(), "predict & target shape do not match"
    smooth = 1e-5
    loss = 0.0
    for i in range(n_classes):
        s, t = inputs[:, i], onehot[:, i]
        d = 1 - (2 * torch.sum(s * t) + smooth) / (torch.sum(s * s) + torch.sum(t * t) + smooth)
        loss = loss + d * weight[i]
    return loss / n_classes
