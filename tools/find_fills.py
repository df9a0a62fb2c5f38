#!/usr/bin/env python3
"""Where do the torch-side fill / copy / small elementwise launches of a model's train step come from?  One profiled step, grouped by
the Python frame that called the aten op.  usage: find_fills.py MODEL N C D H W [--classes K] [--dtype bf16]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
from mi355seg import functional as F
from mi355seg.engine import make_adam, weights_init_normal
from tools.bench_model import build

ap = argparse.ArgumentParser()
ap.add_argument("model"); ap.add_argument("dims", type=int, nargs=5)
ap.add_argument("--classes", type=int, default=2); ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
N, C, D, H, W = a.dims
dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
torch.manual_seed(0)
m = build(a.model, C, a.classes, D, H, W)
m.apply(weights_init_normal("kaiming"))
m = m.cuda().train()
opt = make_adam(m.parameters(), lr=1e-3)
x = torch.randn(N, C, D, H, W, device="cuda")
lab = torch.randint(0, a.classes, (N, 1, D, H, W), device="cuda")
tgt = torch.cat([(lab == i).float() for i in range(a.classes)], dim=1)


def step():
    opt.zero_grad(set_to_none=True)
    with mi355seg.autocast(dtype):
        pred = m(x)
    loss = F.bce_with_logits(pred, tgt)
    loss.backward()
    opt.step()


for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
kc = collections.Counter()
for ev in prof.events():
    for k in getattr(ev, "kernels", []) or []:
        if any(t in k.name for t in ("Fill", "copyBuffer", "elementwise", "CatArray")):
            st = [s_ for s_ in (ev.stack or []) if "site-packages/torch" not in s_ and "find_fills" not in s_ and "<built-in" not in s_]
            kc[(k.name[:60], ev.name, st[0] if st else "?")] += 1
for (kn, name, where), n in kc.most_common(40):
    print(f"{n:5d}  {kn:60s} <- {name:28s} {where}")
print("----")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::reshape", "aten::permute",
                                                        "aten::transpose", "aten::detach", "aten::alias", "aten::select", "aten::slice", "aten::narrow", "aten::unsqueeze", "aten::squeeze",
                                                        "aten::expand", "aten::t", "aten::_unsafe_view", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::result_type", "aten::is_nonzero",
                                                        "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::resize_", "aten::unbind", "aten::flatten", "aten::view_as", "aten::chunk", "aten::split"):
        st = [s_ for s_ in (ev.stack or []) if "site-packages/torch" not in s_ and "find_fills" not in s_ and "<built-in" not in s_]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in cnt.most_common(60):
    print(f"{n:5d}  {name:22s} {where}")
