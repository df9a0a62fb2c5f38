"""Hand-computed cases pinning oracle/metric.py to utils/metric.py:20-75 (the
reference file cannot be imported here: torchio/monai)."""
import numpy as np
import torch

from oracle.metric import confusion_counts, metric, rates


def test_all_zero():
    z = np.zeros((1, 1, 4, 4, 4), dtype=np.int64)
    j, d = metric(z, z)
    assert j == 0.0 and d == 0.0


def test_identical():
    a = np.zeros((2, 1, 4, 4, 4), dtype=np.int64)
    a[0, 0, :2] = 1                       # 32 voxels
    j, d = metric(torch.from_numpy(a), torch.from_numpy(a))
    assert abs(j - 32 / 32.001) < 1e-12
    assert abs(d - 64 / 64.001) < 1e-12


def test_disjoint_and_partial():
    g = np.zeros((1, 1, 2, 2, 4), dtype=np.int64)
    p = np.zeros_like(g)
    g[..., :2] = 1                        # 8 voxels
    p[..., 2:] = 1                        # 8 voxels, disjoint
    assert metric(g, p) == (0.0, 0.0)
    p[..., 1:] = 1                        # now 12 voxels, 4 overlap
    j, d = metric(g, p)
    assert abs(j - 4 / (16 + 0.001)) < 1e-12
    assert abs(d - 8 / (8 + 12 + 0.001)) < 1e-12
    c = confusion_counts(g, p)
    assert (c["tp"], c["fp"], c["fn"], c["tn"]) == (4, 8, 4, 0.0)
    r = rates(g, p)
    assert abs(r["recall"] - 4 / 8.001) < 1e-12 and abs(r["precision"] - 4 / 12.001) < 1e-12


def test_bitwise_quirk_on_multiclass_labels():
    # labels 1 and 2 share no bits: '&' gives 0 although both are foreground (metric.py:40)
    g = np.full((1, 1, 2, 2, 2), 1, dtype=np.int64)
    p = np.full((1, 1, 2, 2, 2), 2, dtype=np.int64)
    j, d = metric(g, p)
    assert j == 0.0 and d == 0.0
    # labels 3 and 1: 3&1 = 1 -> counted; sums are VALUE sums (8*3 + 8*1)
    j, d = metric(np.full_like(g, 3), g)
    assert abs(d - 16 / (24 + 8 + 0.001)) < 1e-12 and abs(j - 8 / 8.001) < 1e-12


def test_float_inputs_are_truncated_to_int():
    g = torch.tensor([[[[[0.9, 1.0], [1.7, 0.0]]]]])
    p = torch.tensor([[[[[1.0, 1.0], [1.0, 0.0]]]]])
    j, d = metric(g, p)                  # g -> [0,1,1,0], p -> [1,1,1,0]
    assert abs(j - 2 / 3.001) < 1e-12 and abs(d - 4 / 5.001) < 1e-12
