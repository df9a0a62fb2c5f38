#!/bin/bash
# FETCH_SIZE / TCC hit-miss / SQ counters of one layer's three kernels: tools/pmc_layer.sh <tag> N D H W Cin Cout
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/pmcl_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/tools/bench_layer.py $@ 3 5 --conv-math f16x3 > $O/f.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/h -- python3 $R/tools/bench_layer.py $@ 3 5 --conv-math f16x3 > $O/h.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
for sub in ("f", "h"):
    fs = glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, c in acc.items():
        if "seg::" in k and ("conv_" in k):
            print(sub, k, "launches", len(n[k]), {cn: round(v / len(n[k]) / 1e6, 2) for cn, v in c.items()}, "(per launch, 1e6)")
PY
rm -rf $O/f $O/h
