"""functional.dropout_pool_* (r5): the element-wise dropout masks of one training step from one draw -- host logic, CPU tensors."""
import torch

import mi355seg  # noqa: F401
from mi355seg import functional as F


def test_masks_come_from_one_pooled_draw_from_the_second_step_on():
    draws = []

    def fallback(shape):
        def f():
            draws.append(shape)
            return torch.nn.functional.dropout(torch.ones(shape), 0.25, True)
        return f

    F._MASKS.pools.clear()
    shapes = [(216, 768), (216, 3072), (1, 12, 216, 216)]
    # step 1: nothing is known yet -- every layer draws for itself while the pool tallies the requests
    F.dropout_pool_begin_step()
    m1 = [F.dropout_pool_take(s, 0.25, "cpu", fallback(s)) for s in shapes]
    assert len(draws) == 3 and all(m.shape == torch.Size(s) for m, s in zip(m1, shapes))
    # step 2: one draw of the tallied size, the layers take consecutive slices of it
    F.dropout_pool_begin_step()
    st = F._MASKS.pools[(0.25, "cpu")]
    assert st[0] is not None and st[0].numel() == sum(torch.Size(s).numel() for s in shapes)
    m2 = [F.dropout_pool_take(s, 0.25, "cpu", fallback(s)) for s in shapes]
    assert len(draws) == 3                                          # no individual draw
    off = 0
    for m, s in zip(m2, shapes):
        n = torch.Size(s).numel()
        assert m.shape == torch.Size(s) and m.data_ptr() == st[0].data_ptr() + 4 * off
        assert all(v == 0.0 or abs(v - 1.0 / 0.75) < 1e-6 for v in m.unique().tolist())       # keep / (1 - p)
        off += n
    assert 0.6 < float((torch.cat([m.flatten() for m in m2]) > 0).float().mean()) < 0.9
    # a request the pool cannot serve (more than was tallied) falls back to its own draw; the next step's pool grows with it
    extra = F.dropout_pool_take((64, 64), 0.25, "cpu", fallback((64, 64)))
    assert len(draws) == 4 and extra.shape == (64, 64)
    F.dropout_pool_begin_step()
    assert F._MASKS.pools[(0.25, "cpu")][0].numel() == sum(torch.Size(s).numel() for s in shapes) + 64 * 64
    # no begin_step caller between forwards: the pool runs dry and every request draws for itself (never a mask served twice)
    for s in shapes:
        F.dropout_pool_take(s, 0.25, "cpu", fallback(s))
    F.dropout_pool_take(shapes[0], 0.25, "cpu", fallback(shapes[0]))
    n = len(draws)
    F.dropout_pool_take(shapes[1], 0.25, "cpu", fallback(shapes[1]))
    assert len(draws) == n + 1
    F._MASKS.pools.clear()
