#!/usr/bin/env python3
"""How far does the REFERENCE itself move when it runs in bf16?  Each reference module (vnet3d.py, residual_unet3d.py,
unetr.py) is run once in fp32 and once under ``torch.autocast("cpu", dtype=torch.bfloat16)`` -- what
``Accelerator(mixed_precision="bf16")`` does for it -- on the closed-form fixture inputs / weights, train mode, same
dropout seed, BCE-with-logits loss, backward.  The relative L2 and max deviation of the logits, the loss difference and the
global relative L2 deviation of all parameter gradients are written to tests/golden/bf16_reference_deviation.json.

These numbers are the yardstick of tests/test_gpu_bf16.py: the MI355X bf16 path (bf16 activations in HBM, bf16 MFMA, fp32
accumulate) must stay within 1.5x of the reference's own bf16 deviation from its fp32 result.  Build container only
(needs /root/reference); the .json is data, no reference source is stored.

    python tests/golden/make_bf16_deviation.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle.fill import (RESUNET96_HEAD_SCALE, fill_module_, fill_module_hash_, make_class_labels, make_input,  # noqa: E402
                         make_input_rough, make_labels)
from oracle.step import two_channel_gt  # noqa: E402

torch.set_num_threads(8)


SAMPLES = {}


def _sample(t, k=32768):
    """Evenly strided sample of the logits (the GPU test samples its own logits the same way)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // k)
    return f[::step][:k].float().numpy().copy()


def deviation(build, x, target, seed=11):
    runs = []
    for autocast in (False, True):
        m = build().train()
        torch.manual_seed(seed)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            pred = m(x)
        pred = pred.float()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
        loss.backward()
        runs.append((pred.detach().double(), float(loss), {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}))
    (p32, l32, g32), (p16, l16, g16) = runs
    SAMPLES[len(SAMPLES)] = (_sample(p32), _sample(p16))
    num = sum(float((g16[k] - g32[k]).norm() ** 2) for k in g32)
    den = sum(float(g32[k].norm() ** 2) for k in g32)
    return {"logits_rel_l2": float((p16 - p32).norm() / p32.norm()), "logits_max_abs": float((p16 - p32).abs().max()),
            "logits_absmax_fp32": float(p32.abs().max()), "loss_fp32": l32, "loss_bf16": l16, "grad_global_rel_l2": (num / den) ** 0.5}


def main():
    from models.three_d.residual_unet3d import UNet
    from models.three_d.unetr import UNETR
    from models.three_d.vnet3d import VNet
    out = {"_what": "reference modules: torch.autocast(cpu, bfloat16) vs fp32, train mode, closed-form fixtures (see make_bf16_deviation.py)",
           "_torch": torch.__version__}
    out["vnet_32"] = deviation(lambda: fill_module_(VNet(in_channels=1, classes=2)), make_input((2, 1, 32, 32, 32)),
                               two_channel_gt(make_labels((2, 1, 32, 32, 32))).float())
    labels = make_class_labels((1, 96, 96, 96), 4)
    onehot = torch.stack([(labels == i) for i in range(4)], dim=1).float()
    out["resunet_f4_96"] = deviation(lambda: fill_module_hash_(UNet(in_channels=4, n_classes=4, base_n_filter=4), RESUNET96_HEAD_SCALE),
                                     make_input_rough((1, 4, 96, 96, 96), seed=2.0), onehot)
    kw = dict(img_shape=(32, 32, 32), input_dim=1, output_dim=2, embed_dim=96, patch_size=16, num_heads=4, dropout=0.0)

    def unetr():
        m = fill_module_(UNETR(**kw))
        with torch.no_grad():
            m.transformer.embeddings.position_embeddings.copy_(0.1 * make_input_rough((1, 8, 96), seed=3.0))
        return m
    out["unetr_small"] = deviation(unetr, make_input_rough((2, 1, 32, 32, 32)), two_channel_gt(make_labels((2, 1, 32, 32, 32))).float())
    from models.three_d.unet3d import UNet3D
    out["unet3d_f8_32"] = deviation(lambda: fill_module_(UNet3D(in_channels=1, out_channels=2, init_features=8)), make_input((2, 1, 32, 32, 32)),
                                    two_channel_gt(make_labels((2, 1, 32, 32, 32))).float())
    json.dump(out, open(os.path.join(HERE, "bf16_reference_deviation.json"), "w"), indent=1)
    # the reference's OWN logits, fp32 and autocast-bf16, on a strided sample per fixture: the GPU bf16 path is placed next to
    # the reference's bf16 result directly (tests/test_gpu_bf16.py), not only next to its own fp32 result
    import numpy as np
    names = ["vnet_32", "resunet_f4_96", "unetr_small", "unet3d_f8_32"]
    np.savez_compressed(os.path.join(HERE, "bf16_reference_logits.npz"),
                        **{f"{n}/fp32": SAMPLES[i][0] for i, n in enumerate(names)}, **{f"{n}/bf16": SAMPLES[i][1] for i, n in enumerate(names)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
