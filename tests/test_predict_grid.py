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
