"""RNG-free closed-form parameters / inputs shared by the oracle, the golden-fixture
generator and the GPU parity tests ("fixture recipe").

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Nothing here depends on a torch RNG
version: every value is a float64 ``sin``/``cos`` of an index, rounded once to fp32.
"""
import zlib

import numpy as np
import torch


def _phase(name: str) -> float:
    return float(zlib.crc32(name.encode()) % 997)


@torch.no_grad()
def fill_module_(module: torch.nn.Module) -> torch.nn.Module:
    """Overwrite every parameter/buffer of ``module`` in place, keyed by its
    state_dict *name* (so registration order does not matter)."""
    for name, t in module.state_dict().items():
        if name.endswith("num_batches_tracked"):
            t.zero_()
            continue
        n = t.numel()
        idx = np.arange(n, dtype=np.float64)
        ph = _phase(name)
        if name.endswith("running_mean"):
            v = 0.05 * np.sin(0.37 * idx + ph)
        elif name.endswith("running_var"):
            v = 1.0 + 0.1 * np.cos(0.23 * idx + ph)
        elif t.dim() == 1 and ("norm" in name or "bn" in name) and name.endswith("weight"):
            v = 1.0 + 0.1 * np.sin(0.61 * idx + ph)                 # norm gamma
        elif t.dim() == 1 and "relu" in name and name.endswith("weight"):
            v = 0.25 + 0.15 * np.sin(0.53 * idx + ph)               # nn.PReLU slopes (default 0.25)
        elif t.dim() == 1:
            v = 0.05 * np.sin(0.71 * idx + ph)                      # biases / norm beta
        else:
            fan_in = n // t.shape[0] if "upconv" not in name and "up_conv" not in name else n // t.shape[1]
            v = np.sin(0.618 * idx + ph) * np.sqrt(2.0 / max(fan_in, 1))
        t.copy_(torch.from_numpy(v.astype(np.float32)).reshape(t.shape))
    return module


# the three 1x1 output heads of the Residual U-Net (residual_unet3d.py:73-79) are summed by the deep supervision: a quarter
# of the kaiming scale keeps the summed logits O(1) (abs max 3.7) in the well-conditioned 96^3 fixture
RESUNET96_HEAD_SCALE = {"ds2_1x1_conv3d": 0.25, "ds3_1x1_conv3d": 0.25, "conv3d_l4": 0.25}


def _hash_uniform(n: int, seed: float) -> np.ndarray:
    """Closed-form white-noise-like values in [-1, 1): frac(sin(12.9898 i + seed) * 43758.5453) * 2 - 1 (float64)."""
    i = np.arange(n, dtype=np.float64)
    v = np.sin(12.9898 * i + seed) * 43758.5453
    return (v - np.floor(v)) * 2.0 - 1.0


@torch.no_grad()
def fill_module_hash_(module: torch.nn.Module, scale_by_name=None) -> torch.nn.Module:
    """Like fill_module_, but the conv / linear weights are decorrelated (hash-uniform) with the variance 2 / fan_in of
    kaiming_normal_ (the reference's init policy, train.py:50-51): every layer then roughly preserves the activation
    scale, so a following InstanceNorm does not multiply rounding noise by a large 1 / std -- the WELL-CONDITIONED
    fixture of the Residual U-Net uses it.  ``scale_by_name`` = {substring: factor} rescales the matching weights
    (the three 1x1 output heads, so that the summed deep-supervision logits stay O(1))."""
    scale_by_name = scale_by_name or {}
    for name, t in module.state_dict().items():
        if name.endswith("num_batches_tracked"):
            t.zero_()
            continue
        n = t.numel()
        ph = _phase(name)
        idx = np.arange(n, dtype=np.float64)
        if name.endswith("running_mean"):
            v = 0.05 * np.sin(0.37 * idx + ph)
        elif name.endswith("running_var"):
            v = 1.0 + 0.1 * np.cos(0.23 * idx + ph)
        elif t.dim() == 1 and ("norm" in name or "bn" in name) and name.endswith("weight"):
            v = 1.0 + 0.1 * np.sin(0.61 * idx + ph)
        elif t.dim() == 1 and "relu" in name and name.endswith("weight"):
            v = 0.25 + 0.15 * np.sin(0.53 * idx + ph)
        elif t.dim() == 1:
            v = 0.05 * np.sin(0.71 * idx + ph)
        else:
            fan_in = n // t.shape[0] if "upconv" not in name and "up_conv" not in name else n // t.shape[1]
            v = _hash_uniform(n, ph) * np.sqrt(6.0 / max(fan_in, 1))        # uniform on [-a, a) has variance a^2 / 3
        for key, f in scale_by_name.items():
            if key in name:
                v = v * f
        t.copy_(torch.from_numpy(v.astype(np.float32)).reshape(t.shape))
    return module


def make_input(shape, freq=0.01, phase=0.0) -> torch.Tensor:
    """Smooth closed-form volume ``sin(freq*i + phase) + 0.5*sin(0.0037*i)``, fp32."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(freq * i + phase) + 0.5 * np.sin(0.0037 * i + 1.0)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def make_input_rough(shape, seed=0.0) -> torch.Tensor:
    """Closed-form white-noise-like volume in [-1, 1): frac(sin(12.9898 i + seed) * 43758.5453) * 2 - 1.
    Used where a smooth input would make InstanceNorm channels nearly constant (ill-conditioned parity)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(12.9898 * i + seed) * 43758.5453
    v = (v - np.floor(v)) * 2.0 - 1.0
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def make_labels(shape, thresh=0.8) -> torch.Tensor:
    """Binary labels ``cos(0.003*i) > thresh`` as float (train.py feeds float gt)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return torch.from_numpy((np.cos(0.003 * i) > thresh).astype(np.float32)).reshape(shape)


def make_class_labels(shape, n_classes) -> torch.Tensor:
    """Integer labels in [0, n_classes) from a smooth field (for CE / multi-class Dice)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    f = 0.5 * (np.sin(0.002 * i) + 1.0) * 0.999
    return torch.from_numpy(np.floor(f * n_classes).astype(np.int64)).reshape(shape)


def init_probe_modules():
    """Fresh modules the init policy is applied to (every branch of train.py:33-61)."""
    import torch.nn as nn
    return [("conv3d", nn.Conv3d(3, 5, 3)), ("convT3d", nn.ConvTranspose3d(4, 6, 2, stride=2)), ("linear", nn.Linear(7, 9)),
            ("conv3d_nobias", nn.Conv3d(2, 4, 1, bias=False)), ("bn3d", nn.BatchNorm3d(6)), ("bn2d", nn.BatchNorm2d(6)),
            ("in3d_affine", nn.InstanceNorm3d(4, affine=True)), ("ln", nn.LayerNorm(8))]
