#!/usr/bin/env python3
"""bench.py -- train voxels/sec of the 3D U-Net hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one full iteration of the reference's train loop (train.py:187-221) on one
synthetic batch already resident in HBM: zero_grad, 2-channel gt, forward, argmax,
BCE-with-logits, backward, (gradient all-reduce), Adam step, Dice counters.
Workload = BASELINE.json configs[1]: UNet3D(1,2,32), x = fp32 [2,1,128,128,128] per GPU.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel family = the fp32-MFMA
implicit-GEMM conv kernel, timed live with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle == the reference's PyTorch-CPU arithmetic, timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_TBS = 8.0               # HBM3E spec peak

WORKLOADS = {
    # name: (in_ch, classes, width, batch, D, H, W, conv fwd+bwd FLOPs per voxel, conv algorithmic bytes per voxel)
    "unet3d_f32_2x128": (1, 2, 32, 2, 128, 128, 128, 1359168.0, 4407.0),     # BASELINE configs[1]
    "unet3d_f32_1x64": (1, 2, 32, 1, 64, 64, 64, 1359168.0, 5376.0),         # BASELINE configs[0] size
}


# MI355SEG_CONV_MATH=bf16 is an opt-in reduced-precision experiment (bf16 MFMA operands in the k3 conv fwd/dgrad); the graded
# configuration is the default: exact fp32 everywhere
_MATH = os.environ.get("MI355SEG_CONV_MATH", "")
DTYPE = "f32" if not _MATH.startswith("b") else (
    "f32 via bf16x6 split MFMA in conv fwd/dgrad (three bf16 parts per operand, six products, fp32 accumulate; opt-in experiment, NOT the graded configuration)"
    if "x6" in _MATH else
    "f32 tensors + fp32 accumulate, bf16 MFMA operands in conv fwd/dgrad (opt-in experiment, NOT the graded configuration)")


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box
    exposes all 256 hardware threads but grants a share of them; oversubscribing 256 threads on that share is
    several times slower than matching it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return min(n, 64)           # ATen CPU conv scaling is flat beyond a few dozen threads


def cpu_baseline(sample_shape, steps=1):
    """Oracle (== reference arithmetic on ATen CPU) train step timed on the host cores."""
    from oracle.nets import UNet3D as OracleUNet
    from oracle.step import train_step as oracle_step, weights_init_normal
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = OracleUNet(1, 2, 32)
    m.apply(weights_init_normal("kaiming"))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(sample_shape, generator=g)
    gt = (torch.rand(sample_shape, generator=g) > 0.9).float()
    oracle_step(m, opt, x, gt)                     # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    for _ in range(steps):
        oracle_step(m, opt, x, gt)
    dt = (time.perf_counter() - t0) / steps
    vox = x.numel()
    return {"value": vox / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"1 warm-up + {steps} timed train step(s) of the CPU oracle (reference arithmetic, anomaly mode off) "
                      f"on x={list(sample_shape)} fp32, {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unet3d_f32_2x128", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1,1,128,128,128")
    ap.add_argument("--no-prof", action="store_true", help="skip the in-library HIP-event kernel timing")
    ap.add_argument("--prof-all", action="store_true", help="HIP-event timing of every kernel family (adds ~1 %% to the step)")
    ap.add_argument("--hip-graph", action="store_true", help="experiment: replay the whole train step as one captured HIP graph (N=1, implies --no-prof)")
    ap.add_argument("--dump-launches", default=None, help="write per-launch (family, ms, GFLOP, TFLOP/s) of the LAST timed step to this file")
    args = ap.parse_args()

    import mi355seg
    from mi355seg import distributed as D
    from mi355seg.engine import train_step, weights_init_normal
    from mi355seg.models.three_d.unet3d import UNet3D

    # MI355SEG_DIST_BACKEND=gloo lets the N > 1 path be rehearsed on a one-GPU box (ranks share cuda:0); the
    # driver's real runs use the default: nccl == RCCL over xGMI, one rank per GPU
    rank, world, local = D.init_from_env(backend=os.environ.get("MI355SEG_DIST_BACKEND"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = mi355seg.lib()

    cin, ncls, width, B, Dd, Hh, Ww, flop_per_vox, bytes_per_vox = WORKLOADS[args.workload]
    torch.manual_seed(0)
    model = UNet3D(in_channels=cin, out_channels=ncls, init_features=width)
    model.apply(weights_init_normal("kaiming"))
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=bool(args.hip_graph))
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.randn((B, cin, Dd, Hh, Ww), generator=g).to(dev)
    gt = (torch.rand((B, 1, Dd, Hh, Ww), generator=g) > 0.9).float().to(dev)
    reducer = D.GradAllReducer(model) if world > 1 else None
    if world > 1:
        D.flatten_buffers(model)                    # buffer broadcast = two collectives, no copies

    graphed = None
    if args.hip_graph:                  # experiment: the whole step as one hipGraphLaunch (single process, no kernel timing)
        assert world == 1, "--hip-graph is a single-process experiment"
        from mi355seg.engine import GraphedTrainStep
        args.no_prof = True
        graphed = GraphedTrainStep(model, opt, x, gt, warmup=3)

    def step():
        if graphed is not None:
            return graphed(x, gt, sync_metric=False)
        if world > 1:
            D.broadcast_buffers(model)
        return train_step(model, opt, x, gt, sync_metric=False, grad_hook=reducer)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not args.no_prof:
        L.call("mi355seg_prof_reset")
        # default: bracket only the two MFMA conv families (51 launches per step); --prof-all brackets all ~400
        L.call("mi355seg_prof_enable", 1 if args.prof_all else 2 * 0b11)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    L.call("mi355seg_prof_enable", 0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    counts, loss = D.all_reduce_metric(out["counts"], out["loss"])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    from mi355seg.utils.metric import metric_from_counts
    jac, dice = metric_from_counts(counts.cpu().tolist())
    vox_per_step = world * B * Dd * Hh * Ww
    ms = dt / args.steps * 1e3
    res = {
        "metric": "train voxels/sec (128^3 patches) 3D U-Net", "value": vox_per_step * args.steps / dt, "unit": "voxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"{args.workload}: UNet3D(1,2,32) fwd+BCE+bwd+Adam+Dice, x=[{B},{cin},{Dd},{Hh},{Ww}] fp32 per GPU, "
                               "random-init (kaiming) weights, data-parallel replicas with RCCL gradient all-reduce",
                   "global_batch": B * world, "patch": [Dd, Hh, Ww], "parallelism": f"dp{world}"},
        "loss": float(loss.item()), "dice": dice,
    }
    if not args.no_prof:
        import ctypes
        buf = (ctypes.c_double * 32)()
        L.call("mi355seg_prof_read", buf, 32)
        names = ["conv_igemm_mfma", "conv_wgrad_mfma", "conv_generic", "convT_k2s2", "norm_act", "pool_upsample", "loss_metric", "conv_direct"]
        fam = {}
        for f, nm in enumerate(names):
            n, tms, fl, by = buf[4 * f], buf[4 * f + 1], buf[4 * f + 2], buf[4 * f + 3]
            if n > 0:
                fam[nm] = {"launches_per_step": n / args.steps, "ms_per_step": tms / args.steps,
                           "tflops": fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0, "gbs": by / (tms * 1e-3) / 1e9 if tms > 0 else 0.0}
        res["kernel_families"] = fam
        if args.dump_launches:
            nmax = 65536
            rec = (ctypes.c_double * (4 * nmax))()
            nrec = ctypes.c_int(0)
            L.call("mi355seg_prof_records", rec, nmax, ctypes.byref(nrec))
            per = nrec.value // args.steps
            with open(args.dump_launches, "w") as fh:
                fh.write("family,ms,gflop,tflops,alg_gbytes,alg_gbs\n")
                for r in range(nrec.value - per, nrec.value):
                    f_, ms_, fl_, by_ = int(rec[4 * r]), rec[4 * r + 1], rec[4 * r + 2], rec[4 * r + 3]
                    fh.write(f"{names[f_]},{ms_:.4f},{fl_ / 1e9:.2f},{fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:.2f},{by_ / 1e9:.4f},{by_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0:.1f}\n")
        n, tms, fl, by = buf[0], buf[1], buf[2], buf[3]
        if n > 0 and tms > 0:
            ach = fl / (tms * 1e-3) / 1e12
            traffic = None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get("conv_igemm_bytes_per_launch")
                except Exception:
                    traffic = None
            res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                               "kernel": "conv_igemm_kernel (Conv3d k3 fwd+dgrad, fp32 MFMA 32x32x2)",
                               "launches": int(n), "avg_launch_ms": tms / n,
                               "algorithmic_gflop_per_launch": fl / n / 1e9, "hbm_algorithmic_gbs": by / (tms * 1e-3) / 1e9}
    t_mfma = flop_per_vox * B * Dd * Hh * Ww / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e3
    t_hbm = bytes_per_vox * B * Dd * Hh * Ww / (PEAK_HBM_TBS * 1e12) * 1e3
    res["step_roofline"] = {"conv_t_mfma_ms": t_mfma, "conv_t_hbm_ms": t_hbm, "frac_of_mfma_bound": t_mfma / ms, "frac_of_hbm_bound": t_hbm / ms}
    if world == 1 and not args.no_cpu_baseline:
        shape = tuple(int(v) for v in args.cpu_sample.split(","))
        res["cpu_baseline"] = cpu_baseline(shape, steps=3)
        res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
