#!/bin/bash
# quick A/B of the four workloads: per-family ms of the step (tools/bench_model.py)
set -e
O=gpurun_out/${1:-quick}
mkdir -p $O
python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --json $O/unet.json > $O/unet.log 2>&1
python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 --json $O/vnet.json > $O/vnet.log 2>&1
if [ "$2" = all ]; then
python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --json $O/resunet.json > $O/resunet.log 2>&1
python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 --json $O/unetr.json > $O/unetr.log 2>&1
fi
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    r = json.load(open(f))
    fam = r.get("families") or r.get("kernel_families") or {}
    print(os.path.basename(f), "ms/step", round(r.get("ms_per_step", 0), 3), {k: round(v.get("ms_per_step", v.get("ms", 0)), 3) for k, v in fam.items()} if isinstance(fam, dict) else fam)
PY
