#!/usr/bin/env python3
"""Time the train step (train.py:187-221) of any in-scope network on synthetic patches.
usage: bench_model.py <unet|vnet|res_unet|unetr> N C D H W [steps] [classes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
from mi355seg.engine import train_step, weights_init_normal
from mi355seg import functional as F

name = sys.argv[1]
N, C, D, H, W = [int(v) for v in sys.argv[2:7]]
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
classes = int(sys.argv[8]) if len(sys.argv) > 8 else 2
torch.manual_seed(0)
if name == "unet":
    from mi355seg.models.three_d.unet3d import UNet3D
    m = UNet3D(C, classes, 32)
elif name == "vnet":
    from mi355seg.models.three_d.vnet3d import VNet
    m = VNet(in_channels=C, classes=classes)
elif name == "res_unet":
    from mi355seg.models.three_d.residual_unet3d import UNet
    m = UNet(C, classes, 32)
else:
    from mi355seg.models.three_d.unetr import UNETR
    m = UNETR(img_shape=(D, H, W), input_dim=C, output_dim=classes)
m.apply(weights_init_normal("kaiming"))
m = m.cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
x = torch.randn(N, C, D, H, W, device="cuda")
lab = torch.randint(0, classes, (N, 1, D, H, W), device="cuda")
tgt = torch.cat([(lab == i).float() for i in range(classes)], dim=1)

def step():
    opt.zero_grad(set_to_none=True)
    pred = m(x)
    loss = F.bce_with_logits(pred, tgt)
    loss.backward()
    opt.step()
    return loss

step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{name} x=[{N},{C},{D},{H},{W}] classes={classes}: {dt * 1e3:.1f} ms/step, {N * D * H * W / dt / 1e6:.1f} Mvoxel/s, loss {l.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
