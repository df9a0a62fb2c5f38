#!/usr/bin/env python3
"""Time the train step (train.py:187-221: forward, BCE-with-logits, backward, Adam) of any in-scope network on synthetic
patches and price every kernel family against the roofline that bounds it.

usage: bench_model.py <unet|vnet|res_unet|unetr> N C D H W [--steps K] [--classes M] [--dtype f32|bf16] [--conv-math fp32|bf16x6]
                      [--json out.json]

BASELINE.json configurations:   cfg 3  vnet 2 1 128 128 128 --dtype bf16
                                cfg 4  res_unet 1 4 160 192 160 --classes 4 --dtype bf16
                                cfg 5  unetr 1 1 96 96 96 --dtype bf16
Per family (in-library HIP events on the launch stream): launches, ms, algorithmic TFLOP/s against the matrix peak of the
arithmetic in use (bf16 2500, bf16x6 416.7, fp32 157.3 TFLOP/s) and algorithmic GB/s against HBM (8 TB/s) -- the MFMA
families are compute-priced, the norm / pool / stem-head families are bandwidth-priced."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mi355seg  # noqa: E402
from mi355seg import functional as F  # noqa: E402
from mi355seg.engine import make_adam, weights_init_normal  # noqa: E402

PEAK = {"bf16": 2500.0, "bf16x6": 2500.0 / 6.0, "f16x3": 2500.0 / 3.0, "fp32": 157.3}
HBM_GBS = 8000.0
FAMILIES = ["conv_igemm_mfma", "conv_wgrad_mfma", "conv_generic", "convT_k2s2", "norm_act_stats", "pool_upsample", "loss_metric", "conv_direct_stem_head"]
BOUND = {"conv_igemm_mfma": "mfma", "conv_wgrad_mfma": "mfma", "conv_generic": "valu", "convT_k2s2": "hbm", "norm_act_stats": "hbm",
         "pool_upsample": "hbm", "loss_metric": "hbm", "conv_direct_stem_head": "hbm"}


def build(name, C, classes, D, H, W):
    if name == "unet":
        from mi355seg.models.three_d.unet3d import UNet3D
        return UNet3D(C, classes, 32)
    if name == "vnet":
        from mi355seg.models.three_d.vnet3d import VNet
        return VNet(in_channels=C, classes=classes)
    if name == "res_unet":
        from mi355seg.models.three_d.residual_unet3d import UNet
        return UNet(C, classes, 32)
    from mi355seg.models.three_d.unetr import UNETR
    return UNETR(img_shape=(D, H, W), input_dim=C, output_dim=classes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=["unet", "vnet", "res_unet", "unetr"])
    ap.add_argument("shape", type=int, nargs=5, metavar=("N", "C", "D", "H", "W"))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--conv-math", default=None, choices=["fp32", "bf16x6"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write (family, ms, GFLOP, TFLOP/s, GB, GB/s) of every profiled launch of the last step")
    ap.add_argument("--graph", action="store_true", help="capture the whole step (fwd, loss, bwd, Adam) into one HIP graph and time its replays")
    a = ap.parse_args()
    N, C, D, H, W = a.shape
    L = mi355seg.lib()
    if a.conv_math:
        mi355seg.set_conv_math(a.conv_math)
    if os.environ.get("MI355SEG_WGRAD_WIDE"):          # A/B knob: 0 never, 1 where it pays (default), 2 wherever the geometry allows
        mi355seg.set_wgrad_wide(int(os.environ["MI355SEG_WGRAD_WIDE"]))
    if os.environ.get("MI355SEG_B16_TILES"):           # A/B knob: 0 auto, 1 the 16x16x32 bf16 tiles wherever the geometry allows, 2 never
        mi355seg.set_b16_tiles(int(os.environ["MI355SEG_B16_TILES"]))
    math = "bf16" if a.dtype == "bf16" else mi355seg.get_conv_math()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    m = build(a.name, C, a.classes, D, H, W)
    m.apply(weights_init_normal("kaiming"))
    m = m.cuda().train()
    opt = make_adam(m.parameters(), lr=1e-3, **({"capturable": True} if a.graph else {}))    # what train.py builds (torch's fused Adam on the GPU)
    x = torch.randn(N, C, D, H, W, device="cuda")
    lab = torch.randint(0, a.classes, (N, 1, D, H, W), device="cuda")
    tgt = torch.cat([(lab == i).float() for i in range(a.classes)], dim=1)

    bns = [b for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm) and b.training and b.num_batches_tracked is not None]

    def step():
        opt.zero_grad(set_to_none=True)
        F.dropout_pool_begin_step()                  # (as engine.train_step: the step's element-wise dropout masks from one draw)
        with F.prepacked_weights(m, (a.dtype, F.conv_math_signature(), tuple(x.shape))):     # (every weight packing of the step by one launch, as engine.train_step)
            with mi355seg.autocast(dtype), F.counters_batched(bns):      # (the BatchNorm counters by one multi-tensor launch, as engine.train_step)
                pred = m(x)
            if bns:
                torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
            loss = F.bce_with_logits(pred, tgt)
            loss.backward()
        opt.step()
        return loss

    if a.graph:                                      # stream capture: eager warm-up on a side stream, then one capture, then replays
        a.no_prof = True
        L.call("mi355seg_prof_enable", 0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = step()
        eager_step = step

        def step():                                   # noqa: F811
            graph.replay()
            return static_loss
    step()
    step()
    torch.cuda.synchronize()
    # timed leg: the in-library profiler off (its event pairs around ~all launches cost 1-2 ms per step on the many-kernel
    # models); then the same number of steps again with every family bracketed, for the per-family table only
    L.call("mi355seg_prof_enable", 0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    dt_prof = None
    if not a.no_prof:
        L.call("mi355seg_prof_reset")
        L.call("mi355seg_prof_enable", 1)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.cuda.synchronize()
        dt_prof = (time.perf_counter() - t1) / a.steps
        L.call("mi355seg_prof_enable", 0)
    res = {"model": a.name, "x": [N, C, D, H, W], "classes": a.classes, "dtype": a.dtype, "conv_math": math, "steps": a.steps, "hip_graph": bool(a.graph), "optimizer": "torch.optim.Adam" + ("(fused=True)" if opt.defaults.get("fused") else ""),
           "ms_per_step": dt * 1e3, "ms_per_step_with_profiler": None if dt_prof is None else dt_prof * 1e3, "voxels_per_s": N * D * H * W / dt, "loss": float(loss.item()),
           "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30, "families": {}}
    if not a.no_prof:
        buf = (ctypes.c_double * 32)()
        L.call("mi355seg_prof_read", buf, 32)
        covered = 0.0
        for f, nm in enumerate(FAMILIES):
            n, tms, fl, by = buf[4 * f], buf[4 * f + 1], buf[4 * f + 2], buf[4 * f + 3]
            if n <= 0:
                continue
            tf, gb = fl / (tms * 1e-3) / 1e12, by / (tms * 1e-3) / 1e9
            covered += tms / a.steps
            ent = {"launches_per_step": n / a.steps, "ms_per_step": tms / a.steps, "algorithmic_tflops": tf, "algorithmic_gbs": gb, "bound": BOUND[nm]}
            if BOUND[nm] == "mfma":
                ent["roofline"] = {"peak_tflops": PEAK[math], "frac": tf / PEAK[math], "hbm_frac": gb / HBM_GBS}
            elif BOUND[nm] == "hbm":
                ent["roofline"] = {"peak_gbs": HBM_GBS, "frac": gb / HBM_GBS}
            res["families"][nm] = ent
        res["ms_in_profiled_families"] = covered
        if a.dump_launches:
            nmax = 65536
            rec = (ctypes.c_double * (4 * nmax))()
            nrec = ctypes.c_int(0)
            L.call("mi355seg_prof_records", rec, nmax, ctypes.byref(nrec))
            per = nrec.value // a.steps
            with open(a.dump_launches, "w") as fh:
                fh.write("family,ms,gflop,tflops,alg_gbytes,alg_gbs\n")
                for r in range(nrec.value - per, nrec.value):
                    f_, ms_, fl_, by_ = int(rec[4 * r]), rec[4 * r + 1], rec[4 * r + 2], rec[4 * r + 3]
                    fh.write(f"{FAMILIES[f_]},{ms_:.4f},{fl_ / 1e9:.2f},{fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:.2f},{by_ / 1e9:.4f},{by_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0:.1f}\n")
    print(f"{a.name} x=[{N},{C},{D},{H},{W}] classes={a.classes} {a.dtype} (conv math {math}): {dt * 1e3:.1f} ms/step, "
          f"{N * D * H * W / dt / 1e6:.1f} Mvoxel/s, loss {loss.item():.4f}, peak mem {res['peak_mem_gib']:.1f} GiB")
    for nm, e in res["families"].items():
        r = e.get("roofline", {})
        print(f"  {nm:24s} {e['launches_per_step']:6.1f} launches {e['ms_per_step']:8.3f} ms  {e['algorithmic_tflops']:8.1f} TFLOP/s  {e['algorithmic_gbs']:8.1f} GB/s  "
              f"[{e['bound']}-bound: {r.get('frac', float('nan')):.3f} of peak]")
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
