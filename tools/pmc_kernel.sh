#!/bin/bash
# SQ counter passes for one layer's kernels (tools/bench_layer.py): tools/pmc_kernel.sh <tag> "<counters pass 1>" "<counters pass 2>" ... -- N D H W Cin Cout
R=$GRAFT_REPO_ROOT
tag=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
O=$R/gpurun_out/pmck_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for p in "${passes[@]}"; do
  rocprofv3 --pmc $p GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/bench_layer.py $@ 3 5 --conv-math f16x3 > $O/p$i.log 2>&1
  i=$((i+1))
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set)); ns = collections.defaultdict(float)
for d in sorted(glob.glob(sys.argv[1] + "/p*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "seg::" not in k or "conv_" not in k: continue
        k = k.replace("void ", "").replace("seg::", "").replace("(anonymous namespace)::", "")[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k, c in acc.items():
    print("==", k)
    for cn, v in sorted(c.items()):
        print(f"   {cn:32s} {v / len(n[k][cn]):16.0f} per launch")
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
