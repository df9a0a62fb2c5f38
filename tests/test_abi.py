"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and
exports every symbol include/mi355seg.h declares (no compute calls here), and the host
mirror keeps the reference's constructor / state_dict surface."""
import ctypes
import os

import pytest
import torch

import mi355seg
from mi355seg._lib import HEADER, LIB_PATH, parse_header


def test_header_parses_and_library_exports_every_symbol():
    protos = parse_header(HEADER)
    assert len(protos) >= 30
    assert os.path.exists(LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    cdll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in mi355seg.h but not exported"
    L = mi355seg.lib()
    assert L.query("mi355seg_version") >= 100
    assert L.query("mi355seg_conv3d_ws_bytes", 2, 128, 128, 128, 32, 32, 3, 1, 1) > 0


def test_invalid_arguments_return_error_codes_not_crashes():
    L = mi355seg.lib()
    with pytest.raises(mi355seg.Mi355SegError, match="conv3d_fwd"):
        L.call("mi355seg_conv3d_fwd_f32", None, 1, None, None, None, 1, 1, 4, 4, 4, 1, 1, 3, 1, 1, None, None, None, 0, None)
    with pytest.raises(mi355seg.Mi355SegError):
        L.call("mi355seg_norm_stats_f32", None, 4, 0, 1, 4, 1e-5, None, None, None, None, 0.1, None, 0, None)


def test_cpu_tensor_is_rejected_without_fallback():
    from mi355seg.models.three_d.unet3d import UNet3D
    m = UNet3D(1, 2, 4)
    with pytest.raises(mi355seg.Mi355SegError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 16, 16, 16))


def test_unet3d_surface_matches_reference_schema():
    from mi355seg.models.three_d.unet3d import UNet3D
    from oracle.nets import UNet3D as OracleUNet
    a, b = OracleUNet(1, 2, 32), UNet3D(1, 2, 32)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and len(sa) == 136
    assert all(sa[k].shape == sb[k].shape for k in sa)
    assert sum(p.numel() for p in b.parameters()) == 22_581_250
    b.load_state_dict(sa)          # checkpoints interchange
    from oracle.step import weights_init_normal
    b.apply(weights_init_normal("kaiming"))     # the reference's init policy applies unchanged
    assert float(b.encoder1.enc1conv1.bias.abs().max()) == 0.0
    assert float(b.encoder1.enc1norm1.weight.min()) == 1.0


def test_autocast_state_is_per_thread():
    """functional.autocast keeps its stack in threading.local: a bf16 region in one thread neither leaks into nor is
    inherited by another thread's forward."""
    import threading
    import torch
    import mi355seg
    F = mi355seg.functional
    seen = {}

    def worker():
        seen["inside_other_thread"] = F.compute_dtype()
        with F.autocast(torch.bfloat16):
            seen["worker_own"] = F.compute_dtype()

    with F.autocast(torch.bfloat16):
        assert F.compute_dtype() is torch.bfloat16
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert F.compute_dtype() is torch.bfloat16
    assert F.compute_dtype() is torch.float32
    assert seen == {"inside_other_thread": torch.float32, "worker_own": torch.bfloat16}
