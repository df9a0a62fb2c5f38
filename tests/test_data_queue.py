"""The device patch queue against the torchio semantics the reference configures (dataloader.py:52-67,94):
per-volume ZNormalization (mean / unbiased std over all voxels), uniform patches, samples_per_volume patches per
subject per refill, labels cut with the same window."""
import numpy as np
import torch


def _write(tmp_path, n=3, shape=(12, 14, 16)):
    (tmp_path / "x").mkdir()
    (tmp_path / "y").mkdir()
    rng = np.random.default_rng(0)
    vols = []
    for i in range(n):
        x = (rng.normal(size=shape) * (i + 1) + 10 * i).astype(np.float32)
        y = (np.indices(shape).sum(0) % (i + 2) == 0).astype(np.float32)     # position-coded labels
        np.save(tmp_path / "x" / f"v{i}.npy", x)
        np.save(tmp_path / "y" / f"v{i}.npy", y)
        vols.append((x, y))
    return vols


def test_queue_semantics(tmp_path):
    import mi355seg
    from mi355seg.data import DevicePatchQueue
    vols = _write(tmp_path)
    q = DevicePatchQueue(str(tmp_path / "x"), str(tmp_path / "y"), (8, 8, 8), batch_size=2, iters=9, device="cpu", seed=7,
                         queue_length=6, samples_per_volume=3)
    batches = list(q)
    assert len(batches) == 9
    normed = [torch.from_numpy((x - x.mean()) / x.std(ddof=1)) for x, _ in vols]
    seen = {0: 0, 1: 0, 2: 0}
    for b in batches:
        x, y = b["source"]["data"], b["gt"]["data"]
        assert x.shape == (2, 1, 8, 8, 8) and y.shape == (2, 1, 8, 8, 8) and x.dtype == torch.float32
        for xp, yp in zip(x, y):
            hits = []
            for vi, (nv, (_, yv)) in enumerate(zip(normed, vols)):       # locate the patch in its (normalised) volume
                u = nv.unfold(0, 8, 1).unfold(1, 8, 1).unfold(2, 8, 1)
                m = (u - xp[0]).abs().amax(dim=(3, 4, 5)) < 1e-5
                if m.any():
                    z, yy, xx = [int(t[0]) for t in torch.nonzero(m, as_tuple=True)]
                    hits.append(vi)
                    assert np.array_equal(yp[0].numpy(), yv[z:z + 8, yy:yy + 8, xx:xx + 8])   # same window for the label
            assert len(hits) == 1
            seen[hits[0]] += 1
    assert sum(seen.values()) == 18 and min(seen.values()) >= 3          # every subject feeds whole groups of 3
    assert all(v % 3 == 0 for v in seen.values())
    assert len(q.cache) == 3                                             # volumes stay resident


def test_queue_is_seeded(tmp_path):
    import mi355seg
    from mi355seg.data import DevicePatchQueue
    _write(tmp_path)
    mk = lambda: DevicePatchQueue(str(tmp_path / "x"), str(tmp_path / "y"), 8, 1, 5, "cpu", seed=3)
    a, b = list(mk()), list(mk())
    assert all(torch.equal(p["source"]["data"], q["source"]["data"]) for p, q in zip(a, b))


import pytest


@pytest.mark.gpu
def test_device_patch_queue_on_the_gpu(tmp_path):
    """DevicePatchQueue on cuda:0 (the configuration the train loop uses): batch dicts of the reference's shape on the device,
    ZNormalization by the HIP kernels (mi355seg_znorm_f32) equal to numpy's (x - mean) / std(ddof=1), the same seeded patch
    order as a CPU-resident queue, labels cut with the same windows."""
    import mi355seg
    from mi355seg.data import DevicePatchQueue
    assert torch.cuda.is_available()
    vols = _write(tmp_path, n=3, shape=(20, 18, 24))
    mk = lambda dev: DevicePatchQueue(str(tmp_path / "x"), str(tmp_path / "y"), (8, 8, 8), batch_size=2, iters=7, device=dev, seed=11,
                                      queue_length=6, samples_per_volume=3)
    qg, qc = mk("cuda:0"), mk("cpu")
    bg, bc = list(qg), list(qc)
    assert len(bg) == 7
    for g, c in zip(bg, bc):
        xg, yg = g["source"]["data"], g["gt"]["data"]
        assert xg.is_cuda and yg.is_cuda and xg.dtype == torch.float32 and xg.shape == (2, 1, 8, 8, 8) and yg.shape == (2, 1, 8, 8, 8)
        assert torch.equal(yg.cpu(), c["gt"]["data"])                              # same windows, same order
        assert (xg.cpu() - c["source"]["data"]).abs().max() < 2e-5
    for idx, (x, _) in enumerate(vols):                                            # the cached volumes are z-normalised
        xn = qg.cache[idx][0].cpu().numpy()[0]
        ref = (x.astype(np.float64) - x.astype(np.float64).mean()) / x.astype(np.float64).std(ddof=1)
        assert np.abs(xn - ref).max() < 2e-5 and abs(float(xn.mean())) < 1e-5 and abs(float(xn.std(ddof=1)) - 1.0) < 1e-5
    # the op itself on an odd-length volume (tail elements) and a large offset (cancellation in sum x^2 - n mean^2 stays in fp64)
    v = torch.randn(100003) * 3.0 + 1000.0
    got = mi355seg.functional.znormalize(v.cuda()).cpu().double()
    want = (v.double() - v.double().mean()) / v.double().std()
    assert (got - want).abs().max() < 2e-4
