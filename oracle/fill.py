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
        elif t.dim() == 1:
            v = 0.05 * np.sin(0.71 * idx + ph)                      # biases / norm beta
        else:
            fan_in = n // t.shape[0] if "upconv" not in name and "up_conv" not in name else n // t.shape[1]
            v = np.sin(0.618 * idx + ph) * np.sqrt(2.0 / max(fan_in, 1))
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
