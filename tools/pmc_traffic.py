#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950) of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof` into profiles/pmc_traffic.json: L2-to-fabric bytes per
launch of the dominant kernel family (bf16x6 k3 forward + input gradient: conv_x3s_kernel and conv_igemm_kernel<MATH_X3, 3, ...>), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes.

Collect on the GPU box (program directly after `--`, counters in their own runs):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof
then (here, in the git checkout, with NO source change since the snapshot that was profiled):
  python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/pmc_traffic.json [conv_math] [kernel-name substring]

The result is stamped with the git HEAD, the conv math of the run and a hash of the kernel sources; bench.py only pastes it
into `roofline.traffic` when the stamp is an ancestor of the code being run (or, on a box without .git, when the kernel
sources hash the same) and the conv math matches -- otherwise `traffic` is null and `traffic_source` says why.
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["general-medical-image-segmentation-cnn-framework_amd/csrc/igemm_kernel.h",
                  "general-medical-image-segmentation-cnn-framework_amd/csrc/conv_mfma.hip",
                  "general-medical-image-segmentation-cnn-framework_amd/csrc/conv_igemm_lowp.hip",
                  "general-medical-image-segmentation-cnn-framework_amd/csrc/conv_x3s.hip",
                  "general-medical-image-segmentation-cnn-framework_amd/csrc/conv_b16s.hip"]


def per_launch(directory, counter, match):
    files = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    tot, n = 0.0, 0
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter and any(m in row["Kernel_Name"] for m in match.split("|")):
            tot += float(row["Counter_Value"])
            n += 1
    return tot, n


def main():
    fdir, wdir, out = sys.argv[1:4]
    conv_math = sys.argv[4] if len(sys.argv) > 4 else "bf16x6"
    # the bf16x6 k3 forward / input-gradient family of cfg 2: conv_x3s_kernel (16-wide tiles) + the generic kernel's MATH_X3 tiles (W = 8 layers)
    match = sys.argv[5] if len(sys.argv) > 5 else "conv_x3s_kernel|conv_igemm_kernel<1, 3"
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--"] + KERNEL_SOURCES, capture_output=True, text=True).stdout.strip()
    assert not dirty, f"kernel sources differ from HEAD, the stamp would lie: {dirty}"
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    f, nf = per_launch(fdir, "FETCH_SIZE", match)
    w, nw = per_launch(wdir, "WRITE_SIZE", match)
    assert nf and nf == nw, (nf, nw)
    fetch = 2.0 * f * 1024.0 / nf          # KB -> bytes, doubled (gfx950 tallies 128-byte requests at 64 bytes)
    write = w * 1024.0 / nw
    res = {
        "git_head": head, "conv_math": conv_math, "kernel_sources": KERNEL_SOURCES, "kernel_sources_sha16": h.hexdigest()[:16],
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof",
        "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of a wide coalesced read stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; unit KB*1024; "
                      "L2-to-fabric requests: Infinity-Cache hits are included, so this is an upper bound on HBM bytes",
        "kernel": "kernel names containing one of '" + match + "' (all k3 fwd + dgrad launches)",
        "launches_counted": nf,
        "conv_igemm_bytes_per_launch": fetch + write,
        "conv_igemm_fetch_bytes_per_launch_corrected": fetch,
        "conv_igemm_write_bytes_per_launch": write,
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
