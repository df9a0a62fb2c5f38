"""MI355X-native hot path for 3D medical-image segmentation training: the U-Net-family
forward/backward, losses and Dice metric as hand-written gfx950 kernels behind a C-ABI
(include/mi355seg.h), exposed with the reference framework's own module interface."""
from . import functional
from .functional import autocast
from . import custom_ops          # registers torch.ops.mi355seg.* (dispatcher view of the same C-ABI entry points)
from ._lib import LIB_PATH, Mi355SegError, lib

_MATH = {"fp32": 0, "bf16x6": 2, "f16x3": 3}
DEFAULT_CONV_MATH = "f16x3"


def set_conv_math(mode):
    """Arithmetic of the MFMA convolutions on fp32 tensors: "fp32" (exact fp32 MFMA), "bf16x6" (fp32-accurate three-piece
    split on the bf16 matrix cores) or "f16x3" (the default: fp32-accurate two-piece split on the fp16 matrix cores under
    per-tensor power-of-two scales).  See include/mi355seg.h."""
    if mode not in _MATH:
        raise ValueError(f"conv math must be one of {sorted(_MATH)}, got {mode!r}")
    lib().call("mi355seg_set_conv_math", _MATH[mode])


def get_conv_math():
    code = lib().query("mi355seg_get_conv_math")
    return next(k for k, v in _MATH.items() if v == code)


def set_x3_shape(shape):
    """MFMA shape of the bf16x6 forward / input-gradient kernels: 16 (v_mfma_f32_16x16x32_bf16, default) or 32."""
    lib().call("mi355seg_set_x3_shape", int(shape))


def get_x3_shape():
    return lib().query("mi355seg_get_x3_shape")


def set_b16_tiles(mode):
    """Tiling of the k3 / k5 convolutions on bf16 tensors: 0 automatic, 1 the 16x16x32 kernel wherever possible, 2 generic tiles only."""
    lib().call("mi355seg_set_b16_tiles", int(mode))


def get_b16_tiles():
    return lib().query("mi355seg_get_b16_tiles")


def set_wgrad_wide(mode):
    """0: never the wide f16x3 weight-gradient kernel, 1 (default): where it pays, 2: wherever the geometry allows (include/mi355seg.h)."""
    lib().call("mi355seg_set_wgrad_wide", int(mode))


def get_wgrad_wide():
    return lib().query("mi355seg_get_wgrad_wide")


__all__ = ["set_b16_tiles", "get_b16_tiles", "set_wgrad_wide", "get_wgrad_wide", "set_x3_shape", "get_x3_shape", "functional", "autocast", "lib", "LIB_PATH", "Mi355SegError", "set_conv_math", "get_conv_math"]
