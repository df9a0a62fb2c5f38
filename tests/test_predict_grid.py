"""Host logic of the sliding-window inference (predict.py:98-147): patch grid + crop-mode aggregation restated
from torchio's GridSampler / GridAggregator (absent here).  Properties: full coverage, border-flush last patch,
every voxel written by exactly the windows torchio's crop mode assigns, identity round trip."""
import numpy as np
import pytest

import mi355seg
from mi355seg.predict import crop_window, grid_locations


def test_grid_locations_cover_and_align():
    locs = grid_locations((100, 64, 130), (64, 64, 64), (4, 4, 36))
    zs = sorted({l[0] for l in locs}); ys = sorted({l[1] for l in locs}); xs = sorted({l[2] for l in locs})
    assert zs == [0, 36] and ys == [0] and xs == [0, 28, 56, 66]      # step = patch - overlap, last flush with the border
    assert len(locs) == 2 * 1 * 4
    with pytest.raises(ValueError):
        grid_locations((32, 32, 32), (64, 64, 64), (4, 4, 4))
    with pytest.raises(ValueError):
        grid_locations((64, 64, 64), (32, 32, 32), (3, 4, 4))


@pytest.mark.parametrize("size,patch,overlap", [((100, 64, 130), (64, 64, 64), (4, 4, 36)), ((40, 50, 60), (16, 32, 24), (4, 8, 6)),
                                                ((32, 32, 32), (32, 32, 32), (4, 4, 4))])
def test_crop_aggregation_is_an_exact_identity_round_trip(size, patch, overlap):
    vol = np.arange(np.prod(size), dtype=np.int64).reshape(size)
    out = np.full(size, -1, dtype=np.int64)
    written = np.zeros(size, dtype=np.int32)
    for loc in grid_locations(size, patch, overlap):
        p = vol[loc[0]:loc[0] + patch[0], loc[1]:loc[1] + patch[1], loc[2]:loc[2] + patch[2]]
        src, dst = crop_window(loc, patch, size, overlap)
        out[dst] = p[src]
        written[dst] += 1
    assert (out == vol).all() and written.min() >= 1


@pytest.mark.parametrize("size,patch,overlap", [((100, 64, 130), (64, 64, 64), (4, 4, 36)), ((40, 50, 60), (16, 32, 24), (4, 8, 6)),
                                                ((32, 32, 32), (32, 32, 32), (4, 4, 4)), ((70, 33, 90), (32, 32, 48), (4, 4, 36))])
def test_window_table_equals_sequential_crop_aggregation(size, patch, overlap):
    """The device paste writes all patches of a batch in one launch, so its windows must be disjoint: window_table clips each
    crop window at its successor's start, which must reproduce torchio's sequential add_batch (later patch overwrites)."""
    from mi355seg.predict import window_table
    rng = np.random.default_rng(0)
    locs = grid_locations(size, patch, overlap)
    tab = window_table(size, patch, overlap)
    assert tab.shape == (len(locs), 9) and tab.dtype == np.int32 and [tuple(r[:3]) for r in tab] == locs
    labels = [rng.integers(0, 5, size=patch) for _ in locs]             # a different label map per patch: order matters
    seq = np.full(size, -1, dtype=np.int64)
    for loc, lab in zip(locs, labels):                                   # torchio: one patch after the other
        src, dst = crop_window(loc, patch, size, overlap)
        seq[dst] = lab[src]
    one = np.full(size, -1, dtype=np.int64)
    cover = np.zeros(size, dtype=np.int32)
    for r, lab in zip(tab, labels):                                      # any order: windows are disjoint
        z, y, x = r[:3]
        src = tuple(slice(int(a), int(b)) for a, b in zip(r[3:6], r[6:9]))
        dst = tuple(slice(int(o + a), int(o + b)) for o, a, b in zip(r[:3], r[3:6], r[6:9]))
        one[dst] = lab[src]
        cover[dst] += 1
    assert (cover == 1).all() and (one == seq).all()
