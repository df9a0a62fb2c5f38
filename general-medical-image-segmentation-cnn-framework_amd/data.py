"""Patch source for the train loop.  The reference feeds torchio ``Queue`` patches
(dataloader.py:52-67) as batch dicts ``{"source": {"data": x}, "gt": {"data": y}}``; torchio/NIfTI I/O is out
of scope, so this module yields the same dict shape from (a) a device-resident synthetic generator or (b) a
device-resident patch queue over a directory of ``.npy`` volumes (Queue / UniformSampler / ZNormalization semantics of
dataloader.py:52-67,94)."""
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


class DevicePatchQueue:
    """UNPINNED restatement of tio.Queue / tio.UniformSampler / tio.ZNormalization semantics (torchio is absent from the build image
    and the reference holds no fixtures for its data pipeline; tests/test_data_queue.py checks the properties stated here).
    The reference's patch pipeline (dataloader.py:52-67: ``tio.Queue(training_set, queue_length=10,
    samples_per_volume=10, UniformSampler(patch_size))`` over ``ZNormalization()``-transformed subjects), kept on the
    device.  Volumes ``<data_path>/*.npy`` (labels ``<gt_path>/<same name>.npy``; [C,D,H,W] or [D,H,W]) are uploaded
    once, z-normalised per volume over all their voxels (mean / unbiased std, tio's ZNormalization without a mask)
    and cached in HBM up to ``cache_gb`` -- 288 GB per MI355X holds whole cohorts -- so a step costs no host I/O and
    no host->device copy.  Queue semantics as torchio's: subjects are visited in a shuffled order, each contributes
    ``samples_per_volume`` uniformly placed patches, the queue is refilled to ``queue_length`` patches and shuffled
    whenever it runs dry, and a batch pops ``batch_size`` patches."""

    def __init__(self, data_path, gt_path, patch_size, batch_size, iters, device, seed=1234, queue_length=10,
                 samples_per_volume=10, cache_gb=200.0):
        self.files = sorted(glob.glob(os.path.join(data_path, "*.npy")))
        if not self.files:
            raise FileNotFoundError(f"no .npy volumes under {data_path}")
        self.gt_path, self.bs, self.iters, self.device = gt_path, batch_size, iters, device
        self.ps = (patch_size,) * 3 if isinstance(patch_size, int) else tuple(patch_size)
        self.queue_length, self.spv = int(queue_length), int(samples_per_volume)
        self.rng = np.random.default_rng(seed)
        self.cache, self.cache_bytes, self.cache_cap = {}, 0, int(cache_gb * (1 << 30))
        self._order, self._queue = [], []

    def __len__(self):
        return self.iters

    def _subject(self, idx):
        """(x, y) of subject ``idx`` on the device, x already z-normalised; cached while the budget lasts."""
        hit = self.cache.get(idx)
        if hit is not None:
            return hit
        f = self.files[idx]
        x = torch.from_numpy(np.ascontiguousarray(np.load(f), dtype=np.float32)).to(self.device)
        y = torch.from_numpy(np.ascontiguousarray(np.load(os.path.join(self.gt_path, os.path.basename(f))), dtype=np.float32)).to(self.device)
        x = x[None] if x.dim() == 3 else x
        y = y[None] if y.dim() == 3 else y
        if any(s < p for s, p in zip(x.shape[1:], self.ps)):
            raise ValueError(f"{f}: volume {tuple(x.shape[1:])} is smaller than the patch {self.ps}")
        if x.is_cuda:                                     # one fused statistics pass + one apply pass (HIP)
            from . import functional as F
            x = F.znormalize(x)
        else:                                             # CPU plumbing tests only
            x = (x - x.mean()) / x.std()
        nbytes = (x.numel() + y.numel()) * 4
        if self.cache_bytes + nbytes <= self.cache_cap:
            self.cache[idx] = (x, y)
            self.cache_bytes += nbytes
        return x, y

    def _refill(self):
        while len(self._queue) < self.queue_length:
            if not self._order:
                self._order = list(self.rng.permutation(len(self.files)))
            x, y = self._subject(int(self._order.pop()))
            for _ in range(self.spv):
                o = [int(self.rng.integers(0, s - p + 1)) for s, p in zip(x.shape[1:], self.ps)]
                sl = (slice(None),) + tuple(slice(a, a + p) for a, p in zip(o, self.ps))
                self._queue.append((x[sl], y[sl]))
        perm = self.rng.permutation(len(self._queue))
        self._queue = [self._queue[i] for i in perm]

    def __iter__(self):
        for _ in range(self.iters):
            xs, ys = [], []
            for _b in range(self.bs):
                if not self._queue:
                    self._refill()
                xp, yp = self._queue.pop()
                xs.append(xp)
                ys.append(yp)
            yield {"source": {"data": torch.stack(xs)}, "gt": {"data": torch.stack(ys)}}


NpyPatches = DevicePatchQueue


def make_loader(config, device, in_channels, seed=1234):
    iters = int(getattr(config, "iters_per_epoch", 4))
    if str(config.data_path) == "synthetic":
        return SyntheticPatches(config.patch_size, in_channels, config.batch_size, iters, device, seed)
    return DevicePatchQueue(config.data_path, config.gt_path, config.patch_size, config.batch_size, iters, device, seed,
                            queue_length=int(getattr(config, "queue_length", 10)),
                            samples_per_volume=int(getattr(config, "samples_per_volume", 10)))
