#!/bin/bash
# grid / duration of every weight-gradient launch of one cfg-2 step (rocprofv3 kernel trace): which kernel each layer runs on
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_wgrad
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof > $O/bench.log 2>&1
F=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python3 - $F <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = [r for r in rows if "wgrad_lowp_kernel" in r["Kernel_Name"] or "wgrad_f16w" in r["Kernel_Name"]]
n = len(w) // 3
for r in w[-n:]:
    print(r["Kernel_Name"][:70], "grid", r["Grid_Size_X"], "wg", r["Workgroup_Size_X"], "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
rm -rf $O/raw
