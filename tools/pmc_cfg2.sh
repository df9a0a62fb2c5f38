#!/bin/bash
# SQ counter passes of the cfg-2 headline only: tools/pmc_cfg2.sh <tag>   -> gpurun_out/pmc_<tag>/{sqa,sqb}.csv
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqa -- python3 $CMD > $O/sqa.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqb -- python3 $CMD > $O/sqb.log 2>&1
cd $R
python tools/pmc_reduce.py $O/sqa $O/sqa.csv > /dev/null
python tools/pmc_reduce.py $O/sqb $O/sqb.csv > /dev/null
rm -rf $O/sqa $O/sqb
python3 - $O <<'PY'
import csv, sys
for f in ("sqa", "sqb"):
    rows = list(csv.DictReader(open(f"{sys.argv[1]}/{f}.csv")))
    keys = [k for k in rows[0].keys() if k in ("kernel","launches","total_ms","eff_clock_GHz","mfma_busy","wait_any_share","wait_inst_share","active_inst_share","lds_conflict_share","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_LDS_IDX_ACTIVE","SQ_WAIT_INST_LDS","SQ_WAVE_CYCLES")]
    print(",".join(keys))
    for r in rows[:8]:
        print(",".join(r[k][:46] for k in keys))
PY
