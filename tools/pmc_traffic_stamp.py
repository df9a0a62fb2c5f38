#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel traffic table of `bash tools/profile_round.sh traffic` (gpurun_out/r03prof/
traffic_cfg2.csv): launch-weighted L2-to-fabric bytes per launch of the bf16x6 k3 forward / input-gradient family of cfg 2,
stamped with git HEAD and the hash of the kernel sources (run in the checkout the snapshot was taken from, sources unchanged).
usage: pmc_traffic_stamp.py [traffic_cfg2.csv] [out.json]"""
import csv, hashlib, importlib.util, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pt", os.path.join(ROOT, "tools", "pmc_traffic.py"))
pt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pt)
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06prof", "traffic_cfg2.csv")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
srcs = pt.KERNEL_SOURCES
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--"] + srcs, capture_output=True, text=True).stdout.strip()
assert not dirty, f"kernel sources differ from HEAD, the stamp would lie: {dirty}"
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
h = hashlib.sha256()
for f in srcs:
    h.update(open(os.path.join(ROOT, f), "rb").read())
rows = [r for r in csv.DictReader(open(src)) if "conv_x3s_kernel" in r["kernel"] or "conv_igemm_kernel<1, 3" in r["kernel"]]
n = sum(int(r["launches"]) for r in rows)
fetch = sum(float(r["fetch_bytes_per_launch_corrected"]) * int(r["launches"]) for r in rows) / n
write = sum(float(r["write_bytes_per_launch"]) * int(r["launches"]) for r in rows) / n
res = {"git_head": head, "conv_math": sys.argv[3] if len(sys.argv) > 3 else "f16x3", "kernel_sources": srcs, "kernel_sources_sha16": h.hexdigest()[:16],
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh traffic): python3 bench.py --steps 3 --warmup 2 "
                 "--no-cpu-baseline --no-exact-leg --no-workloads --no-prof; per-kernel table profiles/r06_traffic_cfg2.csv",
       "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of a wide coalesced read stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; "
                     "unit KB*1024; L2-to-fabric requests: Infinity-Cache hits are included, so this is an upper bound on HBM bytes.  The halo loads "
                     "of these kernels are 16 bytes per lane in 64-byte runs (one voxel's 16-channel chunk), a width the guide calls uncalibrated: the "
                     "TCC miss count of the same launches (profiles/r06_tcc_cfg2.csv, 64-byte requests) is the cross-check",
       "kernel": "split-precision k3 forward + input-gradient family of cfg 2: " + ", ".join(f"{r['kernel']} x{r['launches']}" for r in rows),
       "launches_counted": n, "conv_igemm_bytes_per_launch": fetch + write, "conv_igemm_fetch_bytes_per_launch_corrected": fetch,
       "conv_igemm_write_bytes_per_launch": write}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
