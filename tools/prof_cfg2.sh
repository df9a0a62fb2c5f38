#!/bin/bash
# rocprofv3 kernel stats of the cfg-2 headline (no legs, no CPU baseline): tools/prof_cfg2.sh <tag>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof > $O/bench.log 2>&1
cp $(find $O/raw -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/raw
cd $R
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms over 13 steps", tot / 1e6, "per step", tot / 1e6 / 13)
for r in rows[:45]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):5d} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} avg_us {float(r["AverageNs"])/1e3:8.1f} {float(r["Percentage"]):5.2f}%')
PY
