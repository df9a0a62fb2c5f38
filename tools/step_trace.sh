#!/bin/bash
# per-launch timeline of ONE cfg-2 step (the last of a short run): tools/step_trace.sh <tag>  ->  gpurun_out/trace_<tag>/last_step.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof > $O/bench.log 2>&1
cd $R
python3 tools/step_trace.py $(find $O/raw -name "*kernel_trace.csv" | head -1) 6 > $O/last_step.txt
rm -rf $O/raw
tail -5 $O/last_step.txt
