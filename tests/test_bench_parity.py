"""bench.py's `parity_vs_cpu` block (BASELINE.json metric, second half: "Dice vs CPU ref"; /root/reference/train.py:204-221) without
a GPU: the cpu_baseline leg writes its first train step, `parity_vs_cpu` compares a first-step record with it.  Here the "GPU side" is
played by a second CPU oracle run on the same seeds (must pass, every difference exactly 0) and by a perturbed copy (must fail)."""
import numpy as np
import torch

import bench


def _first_step(shape):
    from oracle.nets import UNet3D
    from oracle.metric import confusion_counts
    from oracle.step import train_step, weights_init_normal
    torch.manual_seed(0)
    m = UNet3D(1, 2, 32)
    m.apply(weights_init_normal("kaiming"))
    m.train()
    x, gt = bench.synthetic_batch(shape, 1234)
    pred, mask, loss, _ = train_step(m, torch.optim.Adam(m.parameters(), lr=1e-3), x, gt)
    c = confusion_counts(gt.numpy(), mask.numpy())
    named = dict(m.named_parameters())
    sums = bench.batch_checksums(x, gt)
    return {"pred": pred.detach().numpy().copy(), "loss": float(loss), "counts": [c["gdth_sum"], c["pred_sum"], c["intersection_sum"], c["union_sum"]],
            "grads": {k: named[k].grad.numpy().copy() for k in bench.PARITY_GRADS}}, sums


def test_synthetic_labels_are_a_low_frequency_field_of_the_input():
    x, gt = bench.synthetic_batch((2, 1, 32, 32, 32), 7)
    x2, gt2 = bench.synthetic_batch((2, 1, 32, 32, 32), 7)
    assert torch.equal(x, x2) and torch.equal(gt, gt2)                       # reproducible: parent and cpu child build the same batch
    assert gt.shape == (2, 1, 32, 32, 32) and set(gt.unique().tolist()) <= {0.0, 1.0}
    assert 0.05 < float(gt.mean()) < 0.6
    # low frequency: neighbouring voxels agree far more often than independent Bernoulli labels of the same rate would
    f = float(gt.mean())
    agree = float((gt[..., 1:] == gt[..., :-1]).float().mean())
    assert agree > f * f + (1 - f) * (1 - f) + 0.1
    # and correlated with the input's own block means
    lf = torch.nn.functional.avg_pool3d(x, 8, 8)
    g8 = torch.nn.functional.avg_pool3d(gt, 8, 8)
    assert float(torch.corrcoef(torch.stack([lf.flatten(), g8.flatten()]))[0, 1]) > 0.5
    _, lab = bench.synthetic_batch((1, 4, 16, 16, 16), 3, classes=4)
    assert lab.dtype == torch.int64 and lab.shape == (1, 1, 16, 16, 16) and int(lab.max()) <= 3 and int(lab.min()) >= 0


def test_parity_block_against_the_cpu_childs_first_step(tmp_path):
    shape = (2, 1, 16, 16, 16)
    out = str(tmp_path / "first.npz")
    threads = torch.get_num_threads()
    try:
        r = bench.cpu_baseline(shape, steps=1, parity_out=out)      # (sets the thread count to the usable cores, as the bench child does)
    finally:
        torch.set_num_threads(threads)
    assert r["value"] > 0 and r["cfg1"]["value"] > 0
    first, sums = _first_step(shape)
    rec = np.load(out)
    assert rec["pred"].shape == (2, 2, 16, 16, 16) and rec["counts"].tolist()[0] == sums[2] and rec["input_checksums"].tolist() == list(sums)
    import shutil
    shutil.copy(out, str(tmp_path / "copy.npz"))
    p = bench.parity_vs_cpu(first, out, sums)
    assert p["inputs_identical"] and p["voxels_compared"] == 4 * 16 ** 3
    assert p["dlogit_max"] < 1e-5 and p["dloss"] < 1e-6 and p["ddice"] < 1e-6 and p["masks_differ_where_decisive"] == 0 and p["pass"]
    assert 0.0 < p["dice_cpu"] < 1.0                                           # a Dice that can disagree with something
    assert set(p["grad_rel_err_of_tensor_max"]) == set(bench.PARITY_GRADS) and max(p["grad_rel_err_of_tensor_max"].values()) < 1e-3
    # a perturbed GPU side must fail: logits off by 1e-3, which also flips decisive voxels
    bad = dict(first)
    bad["pred"] = first["pred"].copy()
    bad["pred"][:, 1] += 1e-3 * np.sign(first["pred"][:, 0] - first["pred"][:, 1]) + 1e-3
    q = bench.parity_vs_cpu(bad, str(tmp_path / "copy.npz"), sums)
    assert not q["pass"] and q["dlogit_max"] > 5e-4
