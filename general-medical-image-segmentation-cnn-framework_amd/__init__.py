"""MI355X-native hot path for 3D medical-image segmentation training: the U-Net-family
forward/backward, losses and Dice metric as hand-written gfx950 kernels behind a C-ABI
(include/mi355seg.h), exposed with the reference framework's own module interface."""
from . import functional
from ._lib import LIB_PATH, Mi355SegError, lib

__all__ = ["functional", "lib", "LIB_PATH", "Mi355SegError"]
