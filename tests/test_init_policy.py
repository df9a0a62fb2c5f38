"""train.py:33-61 ``weights_init_normal``: tests/golden/init_policy.npz holds what the reference function -- lifted out of
the reference file's syntax tree and executed as it stands (tests/golden/make_golden.py: gen_init) -- wrote into a fixed
set of fresh modules under a fixed CPU seed, for every init_type.  The oracle's and the product's restatements must write
the same bits (same torch.nn.init calls in the same order on the same generator)."""
import os

import numpy as np
import pytest
import torch

from oracle.fill import init_probe_modules
from oracle.step import weights_init_normal as oracle_init

INIT_TYPES = ("normal", "xavier", "xavier_uniform", "kaiming", "orthogonal", "none")


def _apply(policy, it):
    torch.manual_seed(5)
    mods = init_probe_modules()
    torch.manual_seed(17)
    out = {}
    for name, m in mods:
        m.apply(policy(it))
        for k, v in m.state_dict().items():
            if v.is_floating_point():
                out[f"{it}/{name}/{k}"] = v.numpy().copy()
    return out


def _policies():
    import mi355seg                      # noqa: F401
    from mi355seg.engine import weights_init_normal as product_init
    return (("oracle", oracle_init), ("product", product_init))


@pytest.mark.parametrize("it", INIT_TYPES)
def test_init_policy_bit_exact_against_reference_fixture(golden_dir, it):
    g = np.load(os.path.join(golden_dir, "init_policy.npz"))
    keys = [k for k in g.files if k.startswith(it + "/")]
    assert len(keys) >= 15
    for who, pol in _policies():
        got = _apply(pol, it)
        assert sorted(got) == sorted(keys), who
        for k in keys:
            assert np.array_equal(got[k], g[k]), (who, k)
    # the branches: BatchNorm3d untouched (1, 0), BatchNorm2d ~ N(1, 0.02), conv / linear biases zeroed
    assert np.array_equal(g[f"{it}/bn3d/weight"], np.ones(6, np.float32)) and not np.array_equal(g[f"{it}/bn2d/weight"], np.ones(6, np.float32))
    assert not g[f"{it}/conv3d/bias"].any() and not g[f"{it}/linear/bias"].any() and not g[f"{it}/convT3d/bias"].any()


def test_unknown_init_type_raises_like_the_reference():
    for who, pol in _policies():
        with pytest.raises(NotImplementedError):
            torch.nn.Conv3d(1, 1, 1).apply(pol("bogus"))


def test_reference_init_runs_here_and_matches_fixture(golden_dir):
    import ast
    path = "/root/reference/train.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    ns = {"torch": torch}
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name == "weights_init_normal":
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    g = np.load(os.path.join(golden_dir, "init_policy.npz"))
    for it in INIT_TYPES:
        got = _apply(ns["weights_init_normal"], it)
        for k, v in got.items():
            assert np.array_equal(v, g[k]), k
