#!/bin/bash
# per-launch timeline of ONE step of a bf16 leg: tools/step_trace_leg.sh <tag> <marker kernel> <bench_model args...>  ->  gpurun_out/trace_<tag>/last_step.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$1
MARK=$2
shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/tools/bench_model.py "$@" --steps 4 --no-prof > $O/bench.log 2>&1
cd $R
python3 tools/step_trace.py $(find $O/raw -name "*kernel_trace.csv" | head -1) 4 $MARK > $O/last_step.txt
rm -rf $O/raw
tail -5 $O/last_step.txt
