"""``python predict.py config=unet config.ckpt=/abs/path`` -- the reference's sliding-window inference
(predict.py:62-183) on the MI355X path: registry + ``ckpt["model"]`` load (predict.py:79-81), per-volume
ZNormalization, grid patches with overlap, eval-mode forward, argmax, crop-mode aggregation, Dice/Jaccard.

The patch grid and the 'crop' aggregation restate torchio's GridSampler / GridAggregator (third-party, pinned
torchio==0.20.3 in the reference's requirements.txt, absent here): per axis the patch origins are
``range(0, size + 1 - patch, patch - overlap)`` plus a last origin flush with the border; each predicted patch
is cropped by ``overlap // 2`` on every side that does not touch the volume border before it is pasted.
NIfTI / HD95 output (SimpleITK, monai) is out of scope; volumes are ``.npy`` and metrics go to metrics.csv.
"""
import csv
import glob
import os
import sys

import numpy as np
import torch

from . import functional as F
from .config import compose, parse_patch_size
from .engine import mixed_precision_dtype
from .registry import build_model
from .utils.metric import metric_from_counts


def grid_locations(size, patch, overlap):
    """All patch origins [(z, y, x)] covering a volume of ``size`` (torchio GridSampler._get_patches_locations).
    UNPINNED restatement (this function and the crop aggregation below): torchio's GridSampler / GridAggregator are third-party code
    that is absent here and for which the reference holds no fixtures; restated from torchio 0.20.3 and property-tested
    (tests/test_predict_grid.py).  What IS pinned on this row is the arithmetic of the patches (test_gpu_unet.py::test_inference_path_*)."""
    axes = []
    for s, p, o in zip(size, patch, overlap):
        if p > s:
            raise ValueError(f"patch size {p} larger than the volume extent {s}")
        if o >= p or o % 2:
            raise ValueError(f"patch overlap {o} must be even and smaller than the patch size {p}")
        idx = list(range(0, s + 1 - p, p - o))
        if idx[-1] != s - p:
            idx.append(s - p)
        axes.append(idx)
    return [(z, y, x) for z in axes[0] for y in axes[1] for x in axes[2]]


def crop_window(loc, patch, size, overlap):
    """(src slices into the patch, dst slices into the volume) of torchio's GridAggregator overlap_mode='crop'."""
    src, dst = [], []
    for l, p, s, o in zip(loc, patch, size, overlap):
        a = o // 2 if l != 0 else 0
        b = o // 2 if l + p != s else 0
        src.append(slice(a, p - b))
        dst.append(slice(l + a, l + p - b))
    return tuple(src), tuple(dst)


def window_table(size, patch, overlap):
    """int32 [P, 9] rows (origin z, y, x, crop lo z, y, x, crop hi z, y, x -- the crop window in PATCH coordinates, hi exclusive) for
    the device-side gather / paste kernels, patches in torchio's order.  torchio's aggregator adds the cropped patches one after the
    other, so where two crop windows overlap (only the border-flush last patch of an axis overlaps its predecessor by more than
    ``overlap``) the LATER patch wins; the table clips every window at the start of its successor's window instead, which gives
    the same volume with disjoint windows -- the paste is then one launch with no ordering between patches."""
    axes = []
    for s, p, o in zip(size, patch, overlap):
        if p > s:
            raise ValueError(f"patch size {p} larger than the volume extent {s}")
        if o >= p or o % 2:
            raise ValueError(f"patch overlap {o} must be even and smaller than the patch size {p}")
        idx = list(range(0, s + 1 - p, p - o))
        if idx[-1] != s - p:
            idx.append(s - p)
        lo = [l + (o // 2 if l != 0 else 0) for l in idx]
        hi = [l + p - (o // 2 if l + p != s else 0) for l in idx]
        for i in range(len(idx) - 1):
            hi[i] = min(hi[i], lo[i + 1])
        axes.append([(l, a - l, b - l) for l, a, b in zip(idx, lo, hi)])
    rows = [(z[0], y[0], x[0], z[1], y[1], x[1], z[2], y[2], x[2]) for z in axes[0] for y in axes[1] for x in axes[2]]
    return np.asarray(rows, dtype=np.int32)


@torch.no_grad()
def sliding_window_predict(model, volume, patch_size, overlap=(4, 4, 36), batch_size=1, dtype=None):
    """volume: float tensor [C, D, H, W] on the GPU -> int64 label volume [1, D, H, W] (argmax over classes).
    ``dtype`` = torch.bfloat16 runs the forward under mi355seg.autocast (``config.mixed_precision=bf16``).
    Device-resident: the window table is uploaded once, each batch is one gather launch (mi355seg_gather_patches_f32), the
    eval-mode forward (BatchNorm folded into the convolutions), the channel argmax and one paste launch
    (mi355seg_paste_labels_i64); no host loop over patches, no host copies."""
    from ._lib import lib
    C, D, H, W = volume.shape
    ps = (patch_size,) * 3 if isinstance(patch_size, int) else tuple(patch_size)
    table_h = window_table((D, H, W), ps, overlap)
    P = table_h.shape[0]
    dev = volume.device
    table = torch.from_numpy(table_h).to(dev)
    volume = volume.contiguous().to(torch.float32)
    out = torch.zeros((1, D, H, W), dtype=torch.int64, device=dev)
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    was_training = model.training
    model.eval()
    for i in range(0, P, batch_size):
        nb = min(batch_size, P - i)
        x = torch.empty((nb, C) + ps, dtype=torch.float32, device=dev)
        L.call("mi355seg_gather_patches_f32", volume.data_ptr(), C, D, H, W, table.data_ptr(), i, nb, ps[0], ps[1], ps[2], x.data_ptr(), st)
        with F.autocast(dtype or F.compute_dtype()):
            if getattr(model, "takes_frequency_bands", False):           # predict.py:128-131 (IS: first output only)
                from .models.three_d.IS import frequency_bands
                logits, _ = model(x, *frequency_bands(x))
            else:
                logits = model(x)
        labels = F.argmax_channels(logits)                               # predict.py:133,139
        L.call("mi355seg_paste_labels_i64", labels.data_ptr(), table.data_ptr(), i, nb, ps[0], ps[1], ps[2], out.data_ptr(), D, H, W, st)
    model.train(was_training)
    return out


def znorm(v):
    """tio.ZNormalization (predict.py:94): zero mean / unit (unbiased) std over the whole volume, no epsilon -- torchio refuses a
    constant volume (``Standard deviation is 0``), so does this.  UNPINNED restatement: torchio is absent from the build image and
    the reference holds no fixture for it; the formula is torchio 0.20's ``ZNormalization.znorm``."""
    std = v.std()
    if float(std) == 0.0:
        raise RuntimeError("znorm: standard deviation is 0 for this volume (tio.ZNormalization raises here too)")
    return (v - v.mean()) / std


def predict(config, model, log=print):
    device = torch.device("cuda", 0)
    if config.ckpt and str(config.ckpt) != "None":
        ckpt = torch.load(config.ckpt, map_location="cpu")
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ckpt["model"].items()}
        model.load_state_dict(state)
    model = model.to(device).eval()
    ps = config.patch_size
    ps = (ps,) * 3 if isinstance(ps, int) else tuple(ps)
    overlap = tuple(min(o, p - 2) // 2 * 2 for o, p in zip((4, 4, 36), ps))   # predict.py:100 overlap, kept valid for small patches
    os.makedirs(config.hydra_path, exist_ok=True)
    if str(config.pred_data_path) == "synthetic":
        g = torch.Generator().manual_seed(7)
        cases = [("synthetic_%d" % i, torch.randn((config.in_classes,) + tuple(2 * p for p in ps), generator=g),
                  (torch.rand((1,) + tuple(2 * p for p in ps), generator=g) > 0.9).float()) for i in range(2)]
    else:
        cases = []
        for f in sorted(glob.glob(os.path.join(config.pred_data_path, "*.npy"))):
            x = torch.from_numpy(np.load(f).astype(np.float32))
            y = torch.from_numpy(np.load(os.path.join(config.pred_gt_path, os.path.basename(f))).astype(np.float32))
            cases.append((os.path.splitext(os.path.basename(f))[0], x[None] if x.dim() == 3 else x, y[None] if y.dim() == 3 else y))
    rows = []
    for name, x, gt in cases:
        vol = znorm(x.to(device))
        mask = sliding_window_predict(model, vol, ps, overlap, batch_size=max(1, int(config.batch_size)), dtype=mixed_precision_dtype(config))
        counts = F.dice_counts(gt.to(device).to(torch.int64).reshape(mask.shape), mask)
        jac, dice = metric_from_counts(counts.cpu().tolist())
        np.save(os.path.join(config.hydra_path, f"{name}_pred.npy"), mask.cpu().numpy().astype(np.uint8))
        rows.append({"file": name, "jaccard": jac, "dice": dice})
        log(f"{name}: dice {dice:.4f} jaccard {jac:.4f}")
    with open(os.path.join(config.hydra_path, "metrics.csv"), "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=["file", "jaccard", "dice"])
        wr.writeheader()
        wr.writerows(rows)
    return rows


def main(argv=None, conf_dir=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    conf_dir = conf_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")
    config = compose(conf_dir, argv, job_name="predict")
    parse_patch_size(config)
    model = build_model(config)
    return config, predict(config, model)


if __name__ == "__main__":
    main()
