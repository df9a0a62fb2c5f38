#!/bin/bash
# LDS bank-conflict share / MFMA busy of one layer's kernels: tools/pmc_lds.sh <tag> N D H W Cin Cout   (env passes through, e.g. MI355SEG_X3Q=0)
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/pmclds_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/s -- python3 $R/tools/bench_layer.py $@ 3 5 --conv-math f16x3 > $O/s.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
fs = glob.glob(f"{sys.argv[1]}/s/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    if "seg::" in k and "conv_" in k:
        print(k, "launches", len(n[k]), "lds conflict/active", round(c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"]), 3),
              "mfma busy", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1, c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3), "lds insts/launch (1e6)", round(c["SQ_INSTS_LDS"] / len(n[k]) / 1e6, 2))
PY
rm -rf $O/s
