#!/usr/bin/env python3
"""Sliding-window inference (predict.py:98-147: grid patches with overlap (4, 4, 36), eval-mode forward, argmax, crop-mode
aggregation) of a synthetic volume: voxels of the volume per second and patches per second, fp32 (default conv math) or bf16.

usage: bench_predict.py [unet|vnet|res_unet] [--volume 256 256 256] [--patch 128] [--batch 2] [--dtype f32|bf16] [--reps 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mi355seg  # noqa: E402
from mi355seg.engine import weights_init_normal  # noqa: E402
from mi355seg.predict import grid_locations, sliding_window_predict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", nargs="?", default="unet", choices=["unet", "vnet", "res_unet"])
    ap.add_argument("--volume", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    mi355seg.lib()
    torch.manual_seed(0)
    if a.name == "unet":
        from mi355seg.models.three_d.unet3d import UNet3D
        m, cin = UNet3D(1, 2, 32), 1
    elif a.name == "vnet":
        from mi355seg.models.three_d.vnet3d import VNet
        m, cin = VNet(in_channels=1, classes=2), 1
    else:
        from mi355seg.models.three_d.residual_unet3d import UNet
        m, cin = UNet(4, 4, 32), 4
    m.apply(weights_init_normal("kaiming"))
    m = m.cuda().eval()
    vol = torch.randn((cin,) + tuple(a.volume), device="cuda")
    ps, ov = (a.patch,) * 3, (4, 4, 36)
    npatch = len(grid_locations(tuple(a.volume), ps, ov))
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    sliding_window_predict(m, vol, ps, ov, a.batch, dtype=dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = sliding_window_predict(m, vol, ps, ov, a.batch, dtype=dt)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / a.reps
    nvox = a.volume[0] * a.volume[1] * a.volume[2]
    res = {"model": a.name, "volume": a.volume, "patch": a.patch, "overlap": list(ov), "batch": a.batch, "dtype": a.dtype,
           "conv_math": "bf16" if a.dtype == "bf16" else mi355seg.get_conv_math(), "patches": npatch, "s_per_volume": sec,
           "volume_voxels_per_s": nvox / sec, "patches_per_s": npatch / sec, "labels": int(out.max().item()) + 1}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
