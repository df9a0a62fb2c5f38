"""Patch source for the train loop.  The reference feeds torchio ``Queue`` patches
(dataloader.py:52-67) as batch dicts ``{"source": {"data": x}, "gt": {"data": y}}``; torchio/NIfTI I/O is out
of scope, so this module yields the same dict shape from (a) a device-resident synthetic generator or (b) a
directory of ``.npy`` volumes with uniformly sampled patches (ZNormalization as dataloader.py:94)."""
import glob
import os

import numpy as np
import torch


class SyntheticPatches:
    """x ~ N(0,1) (mimics ZNormalization), labels from a thresholded low-frequency field; generated on the
    device so no host->device copy sits in the step."""

    def __init__(self, patch_size, in_channels, batch_size, iters, device, seed=1234, n_labels=2):
        self.ps = (patch_size,) * 3 if isinstance(patch_size, int) else tuple(patch_size)
        self.cin, self.bs, self.iters, self.device = in_channels, batch_size, iters, device
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.n_labels = n_labels

    def __len__(self):
        return self.iters

    def __iter__(self):
        for _ in range(self.iters):
            x = torch.randn((self.bs, self.cin) + self.ps, generator=self.gen, device=self.device)
            coarse = torch.rand((self.bs, 1) + tuple(max(2, p // 8) for p in self.ps), generator=self.gen, device=self.device)
            field = torch.nn.functional.interpolate(coarse, size=self.ps, mode="trilinear", align_corners=False)
            gt = (field > 0.6).to(torch.float32)
            yield {"source": {"data": x}, "gt": {"data": gt}}


class NpyPatches:
    """Uniform random patches from <data_path>/*.npy with labels <gt_path>/<same name>.npy ([C,D,H,W] or [D,H,W])."""

    def __init__(self, data_path, gt_path, patch_size, batch_size, iters, device, seed=1234):
        self.files = sorted(glob.glob(os.path.join(data_path, "*.npy")))
        if not self.files:
            raise FileNotFoundError(f"no .npy volumes under {data_path}")
        self.gt_path, self.bs, self.iters, self.device = gt_path, batch_size, iters, device
        self.ps = (patch_size,) * 3 if isinstance(patch_size, int) else tuple(patch_size)
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return self.iters

    def _load(self, f):
        x = np.load(f, mmap_mode="r")
        y = np.load(os.path.join(self.gt_path, os.path.basename(f)), mmap_mode="r")
        return (x[None] if x.ndim == 3 else x), (y[None] if y.ndim == 3 else y)

    def __iter__(self):
        for _ in range(self.iters):
            xs, ys = [], []
            for _b in range(self.bs):
                x, y = self._load(self.files[self.rng.integers(len(self.files))])
                o = [self.rng.integers(0, s - p + 1) for s, p in zip(x.shape[1:], self.ps)]
                sl = tuple(slice(a, a + p) for a, p in zip(o, self.ps))
                xv = np.asarray(x[(slice(None),) + sl], dtype=np.float32)
                xv = (xv - xv.mean()) / (xv.std() + 1e-8)
                xs.append(xv)
                ys.append(np.asarray(y[(slice(None),) + sl], dtype=np.float32))
            yield {"source": {"data": torch.from_numpy(np.stack(xs)).to(self.device)},
                   "gt": {"data": torch.from_numpy(np.stack(ys)).to(self.device)}}


def make_loader(config, device, in_channels, seed=1234):
    iters = int(getattr(config, "iters_per_epoch", 4))
    if str(config.data_path) == "synthetic":
        return SyntheticPatches(config.patch_size, in_channels, config.batch_size, iters, device, seed)
    return NpyPatches(config.data_path, config.gt_path, config.patch_size, config.batch_size, iters, device, seed)
