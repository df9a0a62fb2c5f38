#!/usr/bin/env python3
"""bench.py -- train voxels/sec of the 3D U-Net hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (WORLD_SIZE unset): bench.py starts the N ranks itself (one child process per GPU,
torchrun's environment contract, RCCL over xGMI) BEFORE any GPU call and only waits for them; under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` it is one of the ranks.

A "step" is one full iteration of the reference's train loop (train.py:187-221) on one synthetic batch already
resident in HBM: zero_grad, 2-channel gt, forward, argmax, BCE-with-logits, backward, (gradient all-reduce),
Adam step, Dice counters.  Workload = BASELINE.json configs[1]: UNet3D(1,2,32), x = fp32 [2,1,128,128,128] per GPU.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel family = the implicit-GEMM conv kernel, timed live
with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle == the reference's PyTorch-CPU arithmetic,
timed in a child process on this box's host cores, started before the first GPU call and released after the last GPU leg).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 matrix peak (spec, no sparsity)
PEAK_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0   # fp32-equivalent peak of the split-precision path: six bf16 MFMAs per product
PEAK_F16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0    # two-piece fp16 split: three fp16 MFMAs per product (F16 MFMA = the BF16 rate)
PEAK_HBM_TBS = 8.0               # HBM3E spec peak

WORKLOADS = {
    # name: (in_ch, classes, width, batch, D, H, W, conv fwd+bwd FLOPs per voxel, conv algorithmic bytes per voxel)
    "unet3d_f32_2x128": (1, 2, 32, 2, 128, 128, 128, 1359168.0, 4407.0),     # BASELINE configs[1]
    "unet3d_f32_1x64": (1, 2, 32, 1, 64, 64, 64, 1359168.0, 5376.0),         # BASELINE configs[0] size
}

MATH_DTYPE = {
    "fp32": "f32",
    "bf16x6": "f32 (bf16x6 split MFMA: every fp32 conv operand = three bf16 parts, six bf16 products per fp32 product, fp32 accumulate)",
    "f16x3": "f32 (f16x3 split MFMA: every fp32 conv operand = two fp16 parts under a per-tensor power-of-two scale, three fp16 products per "
             "fp32 product, fp32 accumulate; graded against fp64 by the same tests as bf16x6 and the exact-fp32 MFMA path)",
}
MATH_PEAK = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16X6_TFLOPS, "f16x3": PEAK_F16X3_TFLOPS}


FAMILY_NAMES = ["conv_igemm_mfma", "conv_wgrad_mfma", "conv_generic", "convT_k2s2", "norm_act", "pool_upsample_layout", "loss_metric", "conv_direct_stem_head"]
MFMA_FAMILIES = (0, 1)

# the bf16 configurations of BASELINE.json (configs[2..4]); each runs as a leg of the default command after the headline
LEGS = [
    # name, network, (N, C, D, H, W), classes, loss, conv fwd+bwd FLOPs per voxel
    ("vnet_bf16_2x128", "vnet", (2, 1, 128, 128, 128), 2, "bce", 2088920.0),
    ("resunet_bf16_1x4x160x192x160", "res_unet", (1, 4, 160, 192, 160), 4, "dice+ce", 2614176.0),
    ("unetr_bf16_1x96", "unetr", (1, 1, 96, 96, 96), 2, "bce", 5323080.0),
]


LABEL_THRESHOLD = 0.02
PARITY_GRADS = ["encoder1.enc1conv1.weight", "encoder1.enc1norm1.weight", "bottleneck.bottleneckconv1.weight", "upconv1.weight", "conv.weight", "conv.bias"]
PARITY_TOL = 1e-4            # north_star: logits / loss / Dice within 1e-4 of the reference's PyTorch-CPU path (fp32)
PARITY_MARGIN = 2e-4         # a voxel whose reference logit margin is below twice the tolerance is not decisive for the argmax


def synthetic_batch(shape, seed, classes=2):
    """The synthetic batch of SURVEY 8(d): x ~ N(0, 1) (what ZNormalization hands the network, dataloader.py:94) from
    ``torch.Generator().manual_seed(seed)``, labels = a THRESHOLDED LOW-FREQUENCY FIELD OF THE INPUT (8^3 block means of
    channel 0, trilinearly interpolated back to the patch; foreground where the field exceeds LABEL_THRESHOLD: about a third of
    the voxels, in blobs of ~8-16 voxels) -- learnable from x, so the Dice of a step is O(0.1-0.5) and moves with training instead
    of collapsing to the all-background answer that Bernoulli labels teach.  Built on the CPU by the parent and by the
    cpu_baseline child alike (same code, same box => same bits; the child reports checksums).  ``classes`` > 2: the field cut
    at ``classes - 1`` symmetric thresholds."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(tuple(shape), generator=g)
    lf = torch.nn.functional.avg_pool3d(x[:, :1], 8, 8)
    lf = torch.nn.functional.interpolate(lf, size=tuple(shape[2:]), mode="trilinear", align_corners=False)
    if classes == 2:
        return x, (lf > LABEL_THRESHOLD).float()
    cuts = torch.linspace(-1.5 * LABEL_THRESHOLD, 1.5 * LABEL_THRESHOLD, classes - 1)
    return x, torch.bucketize(lf, cuts)


def batch_checksums(x, gt):
    """Order-independent exact checksums of a batch (integer sums of the fp32 bit patterns, plain and index-weighted, and the label
    count): the parent and the cpu_baseline child must have built the same bits."""
    import torch
    bits = x.contiguous().view(torch.int32).to(torch.int64).flatten()
    w = (torch.arange(bits.numel(), dtype=torch.int64) % 8191) + 1
    return (int(bits.sum()), int((bits * w).sum()), int(gt.sum()))


def read_families(L, steps):
    import ctypes
    buf = (ctypes.c_double * 32)()
    L.call("mi355seg_prof_read", buf, 32)
    fam = {}
    for f, nm in enumerate(FAMILY_NAMES):
        n, tms, fl, by = buf[4 * f], buf[4 * f + 1], buf[4 * f + 2], buf[4 * f + 3]
        if n > 0:
            fam[nm] = {"launches_per_step": n / steps, "ms_per_step": tms / steps,
                       "tflops": fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0, "gbs": by / (tms * 1e-3) / 1e9 if tms > 0 else 0.0}
    return fam, buf


def read_records(L):
    import ctypes
    nmax = 65536
    rec = (ctypes.c_double * (4 * nmax))()
    nrec = ctypes.c_int(0)
    L.call("mi355seg_prof_records", rec, nmax, ctypes.byref(nrec))
    return [(int(rec[4 * r]), rec[4 * r + 1], rec[4 * r + 2], rec[4 * r + 3]) for r in range(nrec.value)]


def layer_roofline(records, steps, mfma_peak_tflops, ms_per_step):
    """t_roof = sum over launches (= layer operations) of max(bytes / BW_HBM, flops / P), with the
    library's own algorithmic FLOPs / bytes per launch; P = the matrix peak of the arithmetic in use for the two MFMA
    families, the fp32 vector peak (157.3) for the VALU convolutions.  Not covered: the optimizer and torch glue."""
    t_roof = t_hbm = t_meas = 0.0
    for f, ms, fl, by in records:
        peak = mfma_peak_tflops if f in MFMA_FAMILIES else PEAK_F32_MFMA_TFLOPS
        t_roof += max(by / (PEAK_HBM_TBS * 1e12), fl / (peak * 1e12)) * 1e3
        t_hbm += by / (PEAK_HBM_TBS * 1e12) * 1e3
        t_meas += ms
    return {"t_roof_ms": t_roof / steps, "t_hbm_only_ms": t_hbm / steps, "t_in_library_kernels_ms": t_meas / steps,
            "achieved": t_roof / steps / ms_per_step, "hbm_only_frac": t_hbm / steps / ms_per_step,
            "definition": "sum over library launches of max(algorithmic bytes / 8 TB/s, algorithmic FLOPs / peak) / measured ms per step; "
                          "optimizer and torch glue contribute time but no roofline work"}


def build_leg_model(net, C, classes, D, H, W):
    if net == "vnet":
        from mi355seg.models.three_d.vnet3d import VNet
        return VNet(in_channels=C, classes=classes)
    if net == "res_unet":
        from mi355seg.models.three_d.residual_unet3d import UNet
        return UNet(C, classes, 32)
    from mi355seg.models.three_d.unetr import UNETR
    return UNETR(img_shape=(D, H, W), input_dim=C, output_dim=classes)


def run_leg(L, dev, name, net, shape, classes, loss_kind, flop_per_vox, steps, warmup=3):
    """One bf16 leg: the train.py:187-221 iteration under mixed precision bf16 (bf16 activations + bf16 MFMA products, fp32
    parameters / statistics / loss) on one resident synthetic batch; 3 warm-up + `steps` timed steps with the two MFMA
    families bracketed by HIP events, then 3 steps with every family bracketed (families table + per-layer roofline), then the
    same weights forward in fp32 for the Dice comparison."""
    import torch
    import mi355seg
    from mi355seg import functional as F
    from mi355seg.engine import make_adam, train_step, weights_init_normal
    from mi355seg.utils import loss_function as LF
    from mi355seg.utils.metric import metric_from_counts
    N, C, D, H, W = shape
    torch.manual_seed(0)
    model = build_leg_model(net, C, classes, D, H, W)
    model.apply(weights_init_normal("kaiming"))
    model = model.to(dev).train()
    opt = make_adam(model.parameters(), lr=1e-3)
    x, lab = synthetic_batch((N, C, D, H, W), 4321, classes)
    x = x.to(dev)
    if classes == 2:
        gt = lab.to(dev)

        def step():
            return train_step(model, opt, x, gt, sync_metric=False, dtype=torch.bfloat16)
        labels = gt.to(torch.int64)
    else:
        labels = lab.to(dev)
        onehot = torch.cat([(labels == i).float() for i in range(classes)], dim=1)       # data preparation, outside the step
        lab3 = labels[:, 0].contiguous()
        dice_loss = LF.DiceLoss()

        def step():                                   # cfg 4: "4-class Dice+CE" on the library losses (loss_function.py:8-16,102-130)
            opt.zero_grad(set_to_none=True)
            with F.autocast(torch.bfloat16):
                pred = model(x)
            loss = LF.cross_entropy_3D(pred, lab3) + dice_loss(pred, onehot)
            with torch.no_grad():
                mask = F.argmax_channels(pred)
                counts = F.dice_counts(labels, mask)
            loss.backward()
            opt.step()
            return {"pred": pred, "mask": mask, "loss": loss.detach(), "counts": counts}
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    L.call("mi355seg_prof_reset")
    pstride = 4 if steps >= 8 else (2 if steps >= 4 else 1)       # the MFMA conv families bracketed in every pstride-th timed step (see main())
    n_brack = sum(1 for i in range(steps) if i % pstride == pstride - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        L.call("mi355seg_prof_enable", 2 * 0b11 if i % pstride == pstride - 1 else 0)
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    L.call("mi355seg_prof_enable", 0)
    fam, buf = read_families(L, n_brack)
    ms = dt * 1e3
    vox = N * D * H * W
    leg = {"workload": f"{name}: {net} x=[{N},{C},{D},{H},{W}] fp32 in, {classes} classes, loss {loss_kind}, mixed precision bf16 "
                       f"(bf16 activations in HBM, bf16 MFMA products, fp32 accumulate / parameters / statistics / loss), fused Adam",
           "ms_per_step": ms, "voxels_per_s": vox / dt, "dtype": "bf16", "steps": steps, "warmup": warmup,
           "loss": float(out["loss"].item()), "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}
    n, tms, fl, by = buf[0], buf[1], buf[2], buf[3]
    if n > 0 and tms > 0:
        ach = fl / (tms * 1e-3) / 1e12
        leg["roofline"] = {"bound": "mfma", "kernel": "conv_b16s_kernel (Conv3d k3/k5 s1 fwd + dgrad, v_mfma_f32_16x16x32_bf16, fp32 accumulate; other shapes on conv_igemm_kernel<MATH_B16>)",
                           "launches": int(n), "launches_per_step": n / n_brack, "bracketed_steps": n_brack, "avg_launch_ms": tms / n, "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS, "algorithmic_gflop_per_launch": fl / n / 1e9}
    if "conv_wgrad_mfma" in fam:
        w = fam["conv_wgrad_mfma"]
        leg["wgrad_roofline"] = {"kernel": "conv_wgrad_lowp_kernel<..., bf16>", "ms_per_step": w["ms_per_step"], "achieved": w["tflops"],
                                 "peak": PEAK_BF16_MFMA_TFLOPS, "frac": w["tflops"] / PEAK_BF16_MFMA_TFLOPS}
    leg["conv_frac_of_mfma_bound"] = flop_per_vox * vox / (PEAK_BF16_MFMA_TFLOPS * 1e12) * 1e3 / ms
    # every family bracketed: the families table and the per-layer roofline (these steps are not the timed ones)
    L.call("mi355seg_prof_reset")
    L.call("mi355seg_prof_enable", 1)
    psteps = 3
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(psteps):
        step()
    torch.cuda.synchronize()
    ms_prof = (time.perf_counter() - t1) / psteps * 1e3
    L.call("mi355seg_prof_enable", 0)
    fam_all, _ = read_families(L, psteps)
    for nm, e in fam_all.items():
        e["frac_of_peak"] = e["tflops"] / PEAK_BF16_MFMA_TFLOPS if nm in ("conv_igemm_mfma", "conv_wgrad_mfma") else e["gbs"] / (PEAK_HBM_TBS * 1e3)
        e["bound"] = "mfma" if nm in ("conv_igemm_mfma", "conv_wgrad_mfma") else "hbm"
    leg["kernel_families"] = fam_all
    leg["ms_per_step_all_families_bracketed"] = ms_prof
    leg["step_roofline"] = layer_roofline(read_records(L), psteps, PEAK_BF16_MFMA_TFLOPS, ms)
    # Dice of the bf16 forward against the fp32 forward of the SAME weights and batch (train-mode statistics, dropout off
    # in both so that the two passes see the same network); north_star's 1e-4 is an fp32 bar -- bf16 flips the argmax where
    # the logit margin is below the bf16 deviation, exactly as the reference does under torch.autocast(bfloat16)
    drops = [m_ for m_ in model.modules() if isinstance(m_, (torch.nn.Dropout, torch.nn.Dropout3d))]
    for m_ in drops:
        m_.eval()
    with torch.no_grad():
        with F.autocast(torch.bfloat16):
            pb = model(x)
        mb = F.argmax_channels(pb)
        del pb
        pf = model(x)
        mf = F.argmax_channels(pf)
        del pf
        _, dice_b = metric_from_counts(F.dice_counts(labels, mb).cpu().tolist())
        _, dice_f = metric_from_counts(F.dice_counts(labels, mf).cpu().tolist())
        agree = float((mb == mf).float().mean().item())
    for m_ in drops:
        m_.train()
    leg["dice_vs_fp32"] = {"dice_bf16": dice_b, "dice_fp32": dice_f, "abs_diff": abs(dice_b - dice_f), "argmax_agreement": agree,
                           "note": "same weights and batch, random-init network on random labels; fp32 = bf16x6 / fp32 MFMA convolutions. "
                                   "bf16 does not meet north_star's 1e-4 Dice bar (an fp32 bar): argmax flips where the logit margin is below the bf16 "
                                   "deviation, as for the reference under torch.autocast(bfloat16) (tests/golden/bf16_reference_deviation.json)"}
    del model, opt, x, out
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    return leg


def run_cfg1_leg(dev, conv_math, steps):
    """BASELINE configs[0] on the GPU: UNet3D(1, 2, 32), x = [1, 1, 64, 64, 64] fp32, the train.py:187-221 iteration, eager loop
    (launch-bound at this size: the HIP-graph replay of the same step is reported beside it from a process of its own)."""
    import torch
    from mi355seg.engine import make_adam, train_step, weights_init_normal
    from mi355seg.models.three_d.unet3d import UNet3D
    torch.manual_seed(0)
    model = UNet3D(in_channels=1, out_channels=2, init_features=32)
    model.apply(weights_init_normal("kaiming"))
    model = model.to(dev).train()
    opt = make_adam(model.parameters(), lr=1e-3)
    x, gt = (t.to(dev) for t in synthetic_batch((1, 1, 64, 64, 64), 4321))
    for _ in range(3):
        out = train_step(model, opt, x, gt, sync_metric=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = train_step(model, opt, x, gt, sync_metric=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"workload": "unet3d_f32_1x64: UNet3D(1,2,32) fwd+BCE+bwd+Adam+Dice, x=[1,1,64,64,64] fp32 (BASELINE configs[0], the reference's "
                        f"train.py config=unet defaults at batch 1), conv math {conv_math}", "ms_per_step": ms, "voxels_per_s": 64 ** 3 / (ms * 1e-3),
            "dtype": MATH_DTYPE[conv_math], "steps": steps, "warmup": 3, "loss": float(out["loss"].item())}


def graph_leg_child(name, steps, gate):
    """The HIP-graph replay of one bf16 leg in a process of its own: started before the parent's first GPU call, it blocks on
    ``gate`` until the parent has finished its eager legs and freed its memory, then builds a FRESH model / optimizer, captures the
    iteration before any eager step (engine.GraphedTrainStep) and times ``steps`` replays.  A capture failure that aborts the
    process (hipStreamEndCapture cannot be caught) then costs this number, not the parent's line."""
    if not gate.readline():
        return None
    import torch
    from mi355seg.engine import GraphedTrainStep, make_adam, weights_init_normal
    dt = torch.bfloat16
    if name == "unet3d_f32_1x64":
        net, shape, classes, dt = "unet3d", (1, 1, 64, 64, 64), 2, torch.float32
    elif name == "unet3d_f32_2x128":         # the headline workload itself (informational: the line's value stays the eager loop's)
        net, shape, classes, dt = "unet3d", (2, 1, 128, 128, 128), 2, torch.float32
    else:
        _, net, shape, classes, _, _ = next(l for l in LEGS if l[0] == name)
    N, C, D, H, W = shape
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    if net == "unet3d":
        from mi355seg.models.three_d.unet3d import UNet3D
        model = UNet3D(in_channels=1, out_channels=2, init_features=32)
    else:
        model = build_leg_model(net, C, classes, D, H, W)
    model.apply(weights_init_normal("kaiming"))
    model = model.to(dev).train()
    opt = make_adam(model.parameters(), lr=1e-3, capturable=True)
    x, gt = (t.to(dev) for t in synthetic_batch((N, C, D, H, W), 4321))
    gs = GraphedTrainStep(model, opt, x, gt, warmup=3, dtype=dt)
    for _ in range(2):
        gs(x, gt, sync_metric=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gs(x, gt, sync_metric=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": ms, "voxels_per_s": N * D * H * W / (ms * 1e-3), "loss": float(out["loss"].item()), "steps": steps,
            "note": "one hipGraphLaunch per iteration, same kernels and arithmetic, fresh model captured before any eager step, in a "
                    "process of its own; `ms_per_step` of the leg is the eager loop"}


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


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(sample_shape, steps=3, reserve=0, gate=None, parity_out=None):
    """Oracle (== reference arithmetic on ATen CPU) train step timed on the host cores.  Runs in a child process
    of its own (``--cpu-baseline-child``) that never touches the GPU.  The child is started before the parent's first GPU call
    (a GPU-initialised process must not fork + exec), builds its model, then blocks on ``gate`` (stdin) until the parent has
    finished every GPU leg: neither side's timing sees the other's load."""
    import torch
    from oracle.nets import UNet3D as OracleUNet
    from oracle.step import train_step as oracle_step, weights_init_normal
    cores = max(1, usable_cores() - reserve)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = OracleUNet(1, 2, 32)
    m.apply(weights_init_normal("kaiming"))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x, gt = synthetic_batch(sample_shape, 1234)               # rank 0's batch of the GPU run (seed 1234 + rank)
    if gate is not None and not gate.readline():          # parent: "go" after its last GPU leg; EOF = the parent is gone
        sys.exit(0)
    # warm-up at the TIMED shape (oneDNN primitive creation and the first touch of ~20 GB of activations stay out of the timed steps).
    # It is also the FIRST train step from the shared initial weights (manual_seed(0) + kaiming, as the GPU run builds them): its
    # logits, loss, Dice counters and a few gradients go to ``parity_out`` for the parent's `parity_vs_cpu` block.
    pred0, mask0, loss0, (jac0, dice0) = oracle_step(m, opt, x, gt)
    if parity_out:
        import numpy as np
        from oracle.metric import confusion_counts
        c = confusion_counts(gt.numpy(), mask0.numpy())
        grads = {"grad:" + k: p.grad.detach().numpy() for k, p in m.named_parameters() if k in PARITY_GRADS}
        np.savez(parity_out, pred=pred0.detach().numpy(), loss=np.float64(loss0.item()), dice=np.float64(dice0), jaccard=np.float64(jac0),
                 counts=np.array([c["gdth_sum"], c["pred_sum"], c["intersection_sum"], c["union_sum"]], dtype=np.int64),
                 input_checksums=np.array(batch_checksums(x, gt), dtype=np.int64), **grads)
    del pred0, mask0, loss0
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        oracle_step(m, opt, x, gt)
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    vox = x.numel()
    # BASELINE configs[0] (the reference's own CPU-runnable case: batch 1, 64^3, train.py config=unet defaults) on the same cores, and
    # the same steps under torch.autograd.set_detect_anomaly(True), which the reference's loop switches on (train.py:183)
    x1, gt1 = synthetic_batch((1, 1, 64, 64, 64), 4321)
    oracle_step(m, opt, x1, gt1)
    t1 = []
    for _ in range(5):
        t0 = time.perf_counter()
        oracle_step(m, opt, x1, gt1)
        t1.append(time.perf_counter() - t0)
    t1a = []
    with torch.autograd.set_detect_anomaly(True):
        oracle_step(m, opt, x1, gt1)
        for _ in range(3):
            t0 = time.perf_counter()
            oracle_step(m, opt, x1, gt1)
            t1a.append(time.perf_counter() - t0)
    cfg1 = {"value": x1.numel() / (sum(t1) / len(t1)), "unit": "voxels/s", "cores": cores, "kind": "port", "step_seconds": [round(t, 3) for t in t1],
            "anomaly_mode_on_value": x1.numel() / (sum(t1a) / len(t1a)), "anomaly_mode_on_step_seconds": [round(t, 3) for t in t1a],
            "sample": "x=[1,1,64,64,64] fp32 (BASELINE configs[0]): 1 warm-up + 5 timed oracle train steps; then 1 + 3 steps under "
                      "torch.autograd.set_detect_anomaly(True) as the reference's train loop runs them (train.py:183)"}
    return {"value": vox / dt, "unit": "voxels/s", "cores": cores, "cpu": cpu_model_name(), "kind": "port", "cfg1": cfg1,
            "best_step_value": vox / min(times), "step_seconds": [round(t, 3) for t in times],
            "sample": f"1 warm-up + {steps} timed train steps (mean) of the CPU oracle (reference arithmetic on ATen-CPU, anomaly mode off), all "
                      f"on x={list(sample_shape)} fp32 (cfg 2's full batch), {dt:.2f} s/step (best {min(times):.2f}), {cores} threads; the child "
                      f"process is started before the first GPU call and runs AFTER the last GPU leg (host otherwise idle)"}


def parity_vs_cpu(first, parity_file, input_sums):
    """The metric's second half (BASELINE.json: "Dice vs CPU ref"; /root/reference/train.py:204-221): the GPU run's FIRST train step
    against the CPU oracle's first step -- same initial weights, same batch, full size.  Untimed."""
    import numpy as np
    from mi355seg.utils.metric import metric_from_counts
    ref = np.load(parity_file)
    pr, pg = ref["pred"], first["pred"]
    d = np.abs(pg.astype(np.float64) - pr.astype(np.float64))
    margin = np.abs(pr[:, 1].astype(np.float64) - pr[:, 0].astype(np.float64))
    mg, mr = pg.argmax(1), pr.argmax(1)
    differ = mg != mr
    decisive = margin > PARITY_MARGIN
    jac_g, dice_g = metric_from_counts(first["counts"])
    grads = {}
    for k, g in first["grads"].items():
        r = ref["grad:" + k].astype(np.float64)
        grads[k] = float(np.abs(g.astype(np.float64) - r).max() / max(float(np.abs(r).max()), 1e-30))
    out = {
        "what": "first train step (forward, BCE-with-logits, argmax, Dice counters, backward) of the GPU run against the CPU oracle's "
                "first step (the reference's PyTorch-CPU arithmetic), same initial weights (manual_seed(0) + kaiming), same batch, full size",
        "voxels_compared": int(d.size), "dlogit_max": float(d.max()), "dlogit_rms": float(np.sqrt((d * d).mean())),
        "logit_abs_max": float(np.abs(pr).max()),
        "loss_gpu": first["loss"], "loss_cpu": float(ref["loss"]), "dloss": abs(first["loss"] - float(ref["loss"])),
        "dice_gpu": dice_g, "dice_cpu": float(ref["dice"]), "ddice": abs(dice_g - float(ref["dice"])),
        "djaccard": abs(jac_g - float(ref["jaccard"])),
        "counts_gpu": [int(v) for v in first["counts"]], "counts_cpu": [int(v) for v in ref["counts"]],
        "masks_differ": int(differ.sum()), "masks_differ_where_decisive": int((differ & decisive).sum()),
        "excluded_frac": float(1.0 - decisive.mean()), "decisive_margin": PARITY_MARGIN,
        "grad_rel_err_of_tensor_max": grads,
        "grad_note": "information only (not part of `pass`): max |g_gpu - g_cpu| / max |g_cpu| per tensor.  Through 18 training-mode BatchNorm layers at a random "
                     "init the fp32 backward is ill-conditioned towards the deep layers (single ReLU sign flips over 1,024 bottleneck voxels); the reference's own "
                     "fp32 gradients sit 4e-3 (stem) / 6e-3 (bottleneck) from an fp64 run of the same step -- graded against that fp64 run in "
                     "tests/test_gpu_fullsize.py::test_full_size_first_step_vs_cpu_oracle",
        "inputs_identical": bool(list(input_sums) == [int(v) for v in ref["input_checksums"]]),
        "tolerance": PARITY_TOL,
    }
    out["pass"] = bool(out["inputs_identical"] and out["dlogit_max"] < PARITY_TOL and out["dloss"] < PARITY_TOL and out["ddice"] < PARITY_TOL
                       and out["masks_differ_where_decisive"] == 0)
    try:
        os.remove(parity_file)
        os.rmdir(os.path.dirname(parity_file))
    except OSError:
        pass
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unet3d_f32_2x128", choices=sorted(WORKLOADS))
    ap.add_argument("--conv-math", default=os.environ.get("MI355SEG_CONV_MATH") or "f16x3", choices=["fp32", "bf16x6", "f16x3"],
                    help="arithmetic of the k3 MFMA convolutions on fp32 tensors: f16x3 (default; fp32-accurate two-piece split on the fp16 matrix "
                         "cores), bf16x6 (three-piece split on the bf16 matrix cores) or fp32 (exact fp32 MFMA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="2,1,128,128,128")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps after one full-shape warm-up")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--parity-out", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--graph-leg-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-workloads", action="store_true", help="skip the bf16 legs of BASELINE configs 3-5 reported under `workloads`")
    ap.add_argument("--leg-steps", type=int, default=10, help="timed steps per `workloads` leg (3 warm-up steps before them)")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the untimed exact-fp32 steps reported beside the headline")
    ap.add_argument("--no-prof", action="store_true", help="skip the in-library HIP-event kernel timing")
    ap.add_argument("--prof-all", action="store_true", help="HIP-event timing of every kernel family (adds ~1 %% to the step)")
    ap.add_argument("--hip-graph", action="store_true", help="experiment: replay the train step as captured HIP graphs (implies --no-prof)")
    ap.add_argument("--dump-launches", default=None, help="write per-launch (family, ms, GFLOP, TFLOP/s) of the LAST timed step to this file")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="plumbing rehearsal without a GPU (tests only): the launch / rendezvous / reducer / barrier / JSON path on a "
                         "toy torch.nn model over gloo; the line is marked rehearsal and is NOT a measurement")
    return ap.parse_args()


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py), used
    only when the stamp says they were taken on an ancestor of the code being run with the same conv math."""
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tf))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    stamp = rec.get("git_head")
    if not stamp:
        return None, "profiles/pmc_traffic.json carries no git stamp"
    # (1) the kernel sources must hash to what was profiled -- holds with or without a .git directory (the GPU box gets a
    #     snapshot without one); (2) where git is available the stamp must also be an ancestor of HEAD
    import hashlib
    h = hashlib.sha256()
    for f in rec.get("kernel_sources", []):
        try:
            h.update(open(os.path.join(ROOT, f), "rb").read())
        except Exception:
            return None, f"stamped kernel source {f} is missing"
    if h.hexdigest()[:16] != rec.get("kernel_sources_sha16"):
        return None, f"kernel sources changed since the PMC passes were taken at {stamp[:10]} (stale)"
    try:
        rc = subprocess.run(["git", "-C", ROOT, "merge-base", "--is-ancestor", stamp, "HEAD"], capture_output=True, timeout=20).returncode
    except Exception:
        rc = None
    if rc == 1:
        return None, f"PMC stamp {stamp[:10]} is not an ancestor of HEAD"
    return rec, f"offline rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes at {stamp[:10]} (kernel sources unchanged since)"


def main():
    args = parse_args()
    if args.cpu_baseline_child:
        shape = tuple(int(v) for v in args.cpu_sample.split(","))
        print(json.dumps(cpu_baseline(shape, steps=args.cpu_steps, gate=sys.stdin, parity_out=args.parity_out)))
        return 0
    if args.graph_leg_child:
        print(json.dumps(graph_leg_child(args.graph_leg_child, args.leg_steps, sys.stdin)))
        return 0

    # ---- start the ranks ourselves when nobody else did (no GPU call has happened in this process)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from importlib import import_module
        sys.path.insert(0, ROOT)
        import_module("mi355seg")
        from mi355seg.distributed import self_launch
        return self_launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])

    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    cpu_child = parity_file = None
    if rank_env == 0 and world_env == 1 and not args.no_cpu_baseline and not args.rehearse_cpu:
        # CPU baseline in its own process, started BEFORE the first GPU call; it waits at a gate until the GPU legs are done.  When its
        # sample is the headline's own batch (the default), its first step is also the reference side of `parity_vs_cpu`.
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-sample", args.cpu_sample, "--cpu-steps", str(args.cpu_steps)]
        wl = WORKLOADS[args.workload]
        if tuple(int(v) for v in args.cpu_sample.split(",")) == (wl[3], wl[0], wl[4], wl[5], wl[6]) and not args.hip_graph:
            import tempfile
            parity_file = os.path.join(tempfile.mkdtemp(prefix="mi355seg_parity_"), "cpu_first_step.npz")
            cmd += ["--parity-out", parity_file]
        cpu_child = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    graph_children = {}
    if rank_env == 0 and world_env == 1 and not args.no_workloads and not args.rehearse_cpu and not args.hip_graph and args.workload == "unet3d_f32_2x128":
        # the HIP-graph replays of the two-class legs, each in a gated process of its own (see graph_leg_child)
        for name in ["unet3d_f32_2x128", "unet3d_f32_1x64"] + [l[0] for l in LEGS if l[3] == 2]:
            if True:
                graph_children[name] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--graph-leg-child", name, "--leg-steps", str(args.leg_steps)],
                                                        stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    import torch
    import torch.distributed as dist
    import mi355seg
    from mi355seg import distributed as D

    # MI355SEG_DIST_BACKEND=gloo lets the N > 1 path be rehearsed on a one-GPU box (ranks share cuda:0); the
    # driver's real runs use the default: nccl == RCCL over xGMI, one rank per GPU
    backend = os.environ.get("MI355SEG_DIST_BACKEND") or ("gloo" if args.rehearse_cpu else None)
    rank, world, local = D.init_from_env(backend=backend)
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher formed WORLD_SIZE={world}; reporting {world}", file=sys.stderr)
        args.gpus = world

    cin, ncls, width, B, Dd, Hh, Ww, flop_per_vox, bytes_per_vox = WORKLOADS[args.workload]
    if args.rehearse_cpu:
        dev = torch.device("cpu")
        L = None
        B, Dd, Hh, Ww = 2, 8, 8, 8
        torch.manual_seed(100 + rank)                # different initial weights per rank: setup_replica must equalise them
        model = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.BatchNorm3d(4), torch.nn.ReLU(), torch.nn.Conv3d(4, 2, 1)).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        args.no_prof = True
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
        from mi355seg.engine import make_adam, train_step, weights_init_normal
        from mi355seg.models.three_d.unet3d import UNet3D
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        L = mi355seg.lib()
        mi355seg.set_conv_math(args.conv_math)
        torch.manual_seed(0)
        model = UNet3D(in_channels=cin, out_channels=ncls, init_features=width)
        model.apply(weights_init_normal("kaiming"))
        model = model.to(dev).train()
        # train.py:109's Adam as the framework's train.py builds it (engine.make_adam: torch's fused single-kernel Adam on the GPU)
        opt = make_adam(model.parameters(), lr=1e-3, capturable=True) if args.hip_graph else make_adam(model.parameters(), lr=1e-3)
    x_cpu, gt_cpu = synthetic_batch((B, cin, Dd, Hh, Ww), 1234 + rank)
    x, gt = x_cpu.to(dev), gt_cpu.to(dev)
    label_foreground_frac = float(gt_cpu.mean())
    input_sums = batch_checksums(x_cpu, gt_cpu)
    del x_cpu, gt_cpu
    reducer = D.setup_replica(model)                # world > 1: rank 0's parameters / buffers everywhere + bucketed gradient reducer
    rank_devices = [str(dev)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank{rank}:{dev}")
        rank_devices = gathered

    graphed = None
    if args.hip_graph:                  # experiment: the step as captured HIP graphs (world > 1: two graphs with the eager reducer between them)
        from mi355seg.engine import GraphedTrainStep
        args.no_prof = True
        graphed = GraphedTrainStep(model, opt, x, gt, warmup=3, grad_hook=reducer)

    def rehearsal_step():
        D.broadcast_buffers(model)
        opt.zero_grad(set_to_none=True)
        pred = model(x)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.cat([(gt == 0).float(), gt], 1))
        loss.backward()
        if reducer is not None:
            reducer(model)
        opt.step()
        m = pred.argmax(1, keepdim=True)
        c = torch.stack([gt.long().sum(), m.sum(), ((gt.long() & m) != 0).sum(), ((gt.long() | m) != 0).sum()])
        return {"loss": loss.detach(), "counts": c}

    def step():
        if args.rehearse_cpu:
            return rehearsal_step()
        if world > 1:
            D.broadcast_buffers(model, async_op=graphed is None)       # launched here, waited for at the forward's first norm layer
        if graphed is not None:
            return graphed(x, gt, sync_metric=False)
        return train_step(model, opt, x, gt, sync_metric=False, grad_hook=reducer)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    first = None
    if parity_file is not None:
        # the FIRST train step from the initial weights (untimed; it is also the first warm-up step): everything the CPU child's first
        # step is compared with -- full logits, loss, the integer Dice counters, a few gradients -- copied to the host here
        out = step()
        sync()
        named = dict(model.named_parameters())
        first = {"pred": out["pred"].detach().float().cpu().numpy(), "loss": float(out["loss"].item()), "counts": out["counts"].cpu().tolist(),
                 "grads": {k: named[k].grad.detach().cpu().numpy() for k in PARITY_GRADS if k in named and named[k].grad is not None}}
    for _ in range(args.warmup - (1 if first is not None else 0)):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    if reducer is not None:                 # the `comm` block counts the timed steps only (the wait timers are off outside a bench)
        for t in (reducer.timer, reducer.pack_timer, reducer.unpack_timer):
            t.reset()
            t.enable()
    D.BUFFER_BROADCAST_TIMER.reset()
    D.BUFFER_BROADCAST_TIMER.enable(world > 1)
    # HIP-event bracketing inside the timed region: only the two MFMA conv families (51 launches per step; --prof-all: all ~400), and
    # only in every 4th timed step -- an event pair around a launch is two barrier packets on the stream (~14 us each here: the
    # next kernel's launch no longer overlaps the bracketed kernel's tail), 1.0-1.5 ms per fully bracketed step (r4: 22.86 against
    # 21.37 ms with and without).  170 bracketed launches of the dominant family at the default 20 steps.
    pstride = 4 if args.steps >= 8 else (2 if args.steps >= 4 else 1)
    pmask = 0 if args.no_prof else (1 if args.prof_all else 2 * 0b11)
    n_brack = sum(1 for i in range(args.steps) if i % pstride == pstride - 1)
    if not args.no_prof:
        L.call("mi355seg_prof_reset")
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if pmask:
            L.call("mi355seg_prof_enable", pmask if i % pstride == pstride - 1 else 0)
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if L is not None:
        L.call("mi355seg_prof_enable", 0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    comm = D.comm_report(reducer, args.steps) if world > 1 else None
    counts, loss = D.all_reduce_metric(out["counts"], out["loss"])

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    from mi355seg.utils.metric import metric_from_counts
    jac, dice = metric_from_counts(counts.cpu().tolist())
    vox_per_step = world * B * Dd * Hh * Ww
    ms = dt / args.steps * 1e3
    res = {
        "metric": "train voxels/sec (128^3 patches) 3D U-Net", "value": vox_per_step * args.steps / dt, "unit": "voxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": MATH_DTYPE[args.conv_math], "data": "synthetic",
        "config": {"workload": f"{args.workload}: UNet3D(1,2,32) fwd+BCE+bwd+Adam+Dice, x=[{B},{cin},{Dd},{Hh},{Ww}] fp32 per GPU, "
                               f"random-init (kaiming) weights, data-parallel replicas with RCCL gradient all-reduce, conv math {args.conv_math}",
                   "global_batch": B * world, "patch": [Dd, Hh, Ww], "parallelism": f"dp{world}", "conv_math": args.conv_math,
                   "optimizer": "torch.optim.Adam" + ("(fused=True)" if getattr(opt, "defaults", {}).get("fused") else "")},
        "rccl_ranks": world, "dist_backend": (dist.get_backend() if world > 1 else None), "rank_devices": rank_devices,
        "loss": float(loss.item()), "dice": dice, "jaccard": jac,
        "labels": f"thresholded low-frequency field of the input (8^3 block means of x, trilinear, > {LABEL_THRESHOLD}); foreground fraction "
                  f"{label_foreground_frac:.4f}; `dice` / `loss` are the LAST timed step's (after {args.warmup + args.steps} Adam steps on the one resident batch)",
    }
    if comm is not None:                    # what a step sends and how long it waited for it (attribution of a scaling shortfall)
        comm["allreduce_wait_frac_of_step"] = comm["allreduce_wait_ms_per_step"] / ms
        res["comm"] = comm
    if args.rehearse_cpu:
        res["rehearsal"] = "CPU plumbing rehearsal on a toy torch.nn model (tests/test_bench_launch.py); not a measurement"
        res["data"] = "rehearsal"
        print(json.dumps(res), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    if not args.no_prof:
        import ctypes
        buf = (ctypes.c_double * 32)()
        L.call("mi355seg_prof_read", buf, 32)
        names = ["conv_igemm_mfma", "conv_wgrad_mfma", "conv_generic", "convT_k2s2", "norm_act", "pool_upsample", "loss_metric", "conv_direct"]
        fam = {}
        for f, nm in enumerate(names):
            n, tms, fl, by = buf[4 * f], buf[4 * f + 1], buf[4 * f + 2], buf[4 * f + 3]
            if n > 0:
                fam[nm] = {"launches_per_step": n / n_brack, "ms_per_step": tms / n_brack,
                           "tflops": fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0, "gbs": by / (tms * 1e-3) / 1e9 if tms > 0 else 0.0}
        res["kernel_families"] = fam
        if args.dump_launches:
            nmax = 65536
            rec = (ctypes.c_double * (4 * nmax))()
            nrec = ctypes.c_int(0)
            L.call("mi355seg_prof_records", rec, nmax, ctypes.byref(nrec))
            per = nrec.value // n_brack
            with open(args.dump_launches, "w") as fh:
                fh.write("family,ms,gflop,tflops,alg_gbytes,alg_gbs\n")
                for r in range(nrec.value - per, nrec.value):
                    f_, ms_, fl_, by_ = int(rec[4 * r]), rec[4 * r + 1], rec[4 * r + 2], rec[4 * r + 3]
                    fh.write(f"{names[f_]},{ms_:.4f},{fl_ / 1e9:.2f},{fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:.2f},{by_ / 1e9:.4f},{by_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0:.1f}\n")
        n, tms, fl, by = buf[0], buf[1], buf[2], buf[3]
        if n > 0 and tms > 0:
            ach = fl / (tms * 1e-3) / 1e12
            peak = MATH_PEAK[args.conv_math]
            rec, note = pmc_traffic()
            traffic = None
            if rec is not None and rec.get("conv_math", "fp32") == args.conv_math:
                traffic = rec.get("conv_igemm_bytes_per_launch")
            elif rec is not None:
                note = f"PMC passes were taken with conv math {rec.get('conv_math', 'fp32')}, this run uses {args.conv_math}"
            kern = {"fp32": "conv_igemm_kernel<MATH_F32> (Conv3d k3 fwd+dgrad, v_mfma_f32_32x32x2_f32)",
                    "bf16x6": "conv_x3s_kernel (Conv3d k3 fwd+dgrad on fp32 tensors, six v_mfma_f32_16x16x32_bf16 per fp32 product; the W = 8 "
                              "bottleneck layers on conv_igemm_kernel<MATH_X3>, 32x32x16)",
                    "f16x3": "conv_x3s_kernel<..., F16> (Conv3d k3 fwd+dgrad on fp32 tensors, three v_mfma_f32_16x16x32_f16 per fp32 product; the W = 8 "
                             "bottleneck layers on conv_igemm_kernel<MATH_X3>, bf16x6 on 32x32x16)"}[args.conv_math]
            res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic, "traffic_source": note,
                               "kernel": kern, "launches": int(n), "avg_launch_ms": tms / n,
                               "bracketed_steps": f"{n_brack} of the {args.steps} timed steps (every {pstride}th): HIP events around each launch of the two MFMA conv "
                                                  "families, recorded on the launch stream inside the timed region",
                               "peak_basis": {"fp32": "fp32 MFMA 157.3 TFLOP/s",
                                              "bf16x6": "bf16 MFMA 2500 TFLOP/s / 6 products = 416.7 fp32-equivalent TFLOP/s",
                                              "f16x3": "fp16 MFMA 2500 TFLOP/s / 3 products = 833.3 fp32-equivalent TFLOP/s"}[args.conv_math],
                               "achieved_counts": "algorithmic fp32 FLOPs (2 x voxels x 27 x Cin x Cout per launch), not MFMA issue slots",
                               "algorithmic_gflop_per_launch": fl / n / 1e9, "hbm_algorithmic_gbs": by / (tms * 1e-3) / 1e9}
    peak_step = MATH_PEAK[args.conv_math]
    t_mfma = flop_per_vox * B * Dd * Hh * Ww / (peak_step * 1e12) * 1e3
    t_hbm = bytes_per_vox * B * Dd * Hh * Ww / (PEAK_HBM_TBS * 1e12) * 1e3
    res["step_roofline"] = {"conv_t_mfma_ms": t_mfma, "conv_t_hbm_ms": t_hbm, "frac_of_mfma_bound": t_mfma / ms, "frac_of_hbm_bound": t_hbm / ms}

    if args.conv_math != "fp32" and not args.no_exact_leg and graphed is None and world == 1:
        # the same step with the exact-fp32 MFMA convolutions, reported beside the headline (outside the timed region)
        mi355seg.set_conv_math("fp32")
        for _ in range(2):
            step()
        sync()
        t1 = time.perf_counter()
        nx = max(3, min(args.steps, 5))
        for _ in range(nx):
            step()
        sync()
        res["exact_fp32_ms_per_step"] = (time.perf_counter() - t1) / nx * 1e3
        mi355seg.set_conv_math(args.conv_math)

    if L is not None and world == 1 and graphed is None and not args.no_prof:
        # per-layer roofline of the headline: 3 more steps with every kernel family bracketed
        L.call("mi355seg_prof_reset")
        L.call("mi355seg_prof_enable", 1)
        for _ in range(3):
            step()
        sync()
        L.call("mi355seg_prof_enable", 0)
        fam_all, _ = read_families(L, 3)
        res["kernel_families_all"] = fam_all
        res["step_roofline"].update(layer_roofline(read_records(L), 3, peak_step, ms))

    if L is not None and world == 1 and graphed is None and not args.no_workloads and args.workload == "unet3d_f32_2x128":
        # BASELINE configs 3-5 in bf16, one leg each, after (and outside) the headline's timed region
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        res["workloads"] = {}
        try:                                 # BASELINE configs[0] at its own shape on the HIP path (the reference's train.py config=unet defaults, batch 1)
            res["workloads"]["unet3d_f32_1x64"] = run_cfg1_leg(dev, args.conv_math, args.leg_steps)
        except Exception as e:
            res["workloads"]["unet3d_f32_1x64"] = {"error": repr(e)}
        for name, net, shape, classes, loss_kind, fpv in LEGS:
            try:
                res["workloads"][name] = run_leg(L, dev, name, net, shape, classes, loss_kind, fpv, args.leg_steps)
            except Exception as e:           # a failing leg must not take the headline with it; the error is on the record
                res["workloads"][name] = {"error": repr(e)}
                L.call("mi355seg_prof_enable", 0)

    for name, child in graph_children.items():          # one at a time, after the eager legs, with the parent's cached memory released
        leg = res.get("workloads", {}).get(name)
        try:
            torch.cuda.empty_cache()
            txt, _ = child.communicate("go\n", timeout=300)
            hg = json.loads(txt.strip().splitlines()[-1])
            if name == args.workload and hg:             # the headline's own replay: beside the line, never instead of it
                hg["speedup_over_eager"] = res["ms_per_step"] / hg["ms_per_step"]
                hg["note"] = "the headline step as one hipGraphLaunch per iteration (engine.GraphedTrainStep, config.hip_graph=true), fresh model in a process " \
                             "of its own; informational -- `value` / `ms_per_step` of this line are the eager launch loop's (the same at every --gpus N), although since r6 (all weight packings " \
                             "of a step by one launch, recorded in the warm-up steps) the replay is the faster way to run this step"
                res["hip_graph"] = hg
                continue
            if leg is not None and hg:
                hg["speedup_over_eager"] = leg["ms_per_step"] / hg["ms_per_step"] if "ms_per_step" in leg else None
                leg["hip_graph"] = hg
                # r5: a launch-bound leg is run the way train.py runs it with config.hip_graph=true -- the SAME kernels and arithmetic as one
                # hipGraphLaunch per iteration (engine.GraphedTrainStep) -- so the leg's ms_per_step is the replay when that is the faster
                # of the two; the eager loop's figure stays beside it (it moves with the box's host CPU: UNETR 15.4-19.1 ms)
                if "ms_per_step" in leg and hg.get("ms_per_step") and hg["ms_per_step"] < leg["ms_per_step"]:
                    leg["eager_ms_per_step"], leg["eager_voxels_per_s"] = leg["ms_per_step"], leg["voxels_per_s"]
                    leg["ms_per_step"], leg["voxels_per_s"] = hg["ms_per_step"], hg["voxels_per_s"]
                    leg["ms_per_step_mode"] = "hip_graph replay (config.hip_graph=true); eager_ms_per_step = the Python launch loop"
                else:
                    leg["ms_per_step_mode"] = "eager launch loop"
        except Exception as e:
            child.kill()
            if leg is not None:
                leg["hip_graph"] = {"error": repr(e), "child_returncode": child.returncode}
    if cpu_child is not None:
        try:
            txt, _ = cpu_child.communicate("go\n", timeout=600)
            res["cpu_baseline"] = json.loads(txt.strip().splitlines()[-1])
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            if first is not None:
                try:
                    res["parity_vs_cpu"] = parity_vs_cpu(first, parity_file, input_sums)
                except Exception as e:
                    res["parity_vs_cpu"] = {"error": repr(e)}
            leg1 = res.get("workloads", {}).get("unet3d_f32_1x64")
            if leg1 is not None and "error" not in leg1 and res["cpu_baseline"].get("cfg1"):
                leg1["cpu_baseline"] = res["cpu_baseline"].pop("cfg1")
                leg1["gpu_over_cpu"] = leg1["voxels_per_s"] / leg1["cpu_baseline"]["value"]
        except Exception as e:          # the GPU numbers stand on their own; say why the CPU leg is missing
            cpu_child.kill()
            res["cpu_baseline"] = None
            res["cpu_baseline_error"] = repr(e)
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
