#!/bin/bash
# Profiling recipe of the round (r3 onwards; output names carry the round: ROUND=r04) (run on the GPU box through gpurun; every rocprofv3 call has the program itself after `--`, counters
# in their own passes, no trace domains beside --kernel-trace):  bash tools/profile_round.sh <part>
#   part stats   : rocprofv3 --kernel-trace --stats of bench.py (cfg 2 headline only) and of the three bf16 legs (tools/bench_model.py)
#   part sq      : SQ counter passes (MFMA busy, effective clock, LDS conflicts, wait / active shares) for the same four workloads
#   part traffic : FETCH_SIZE and WRITE_SIZE passes (separate) for the same four workloads
set -e
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
O=$R/gpurun_out/${ROUND}prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFG2="$R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof"
declare -A LEG
LEG[vnet]="$R/tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 3 --no-prof"
LEG[resunet]="$R/tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 3 --no-prof"
LEG[unetr]="$R/tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 3 --no-prof"
case "$1" in
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-leg --no-workloads > $O/stats_cfg2.log 2>&1
  for n in vnet resunet unetr; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -- python3 ${LEG[$n]} > $O/stats_$n.log 2>&1
  done ;;
sq)
  for n in cfg2 vnet resunet unetr; do
    if [ $n = cfg2 ]; then CMD="$CFG2"; else CMD="${LEG[$n]}"; fi
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqa_$n -- python3 $CMD > $O/sqa_$n.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqb_$n -- python3 $CMD > $O/sqb_$n.log 2>&1
  done ;;
traffic)
  for n in cfg2 vnet resunet unetr; do
    if [ $n = cfg2 ]; then CMD="$CFG2"; else CMD="${LEG[$n]}"; fi
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$n -- python3 $CMD > $O/fetch_$n.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$n -- python3 $CMD > $O/write_$n.log 2>&1
    # L2 hit / miss requests (r3 verdict item 6): what share of the halo and weight re-reads the L2 absorbs
    if [ $n = cfg2 ]; then rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc_$n -- python3 $CMD > $O/tcc_$n.log 2>&1; fi
  done ;;
esac
cd $R
# keep only the summaries (the per-dispatch traces are large): reduce here, copy the small CSVs back
for d in $O/sqa_* $O/sqb_*; do [ -d $d ] && python tools/pmc_reduce.py $d $d.csv > /dev/null; done
for d in $O/stats_*; do [ -d $d ] && cp $(find $d -name "*kernel_stats.csv" | head -1) $d.csv; done
if [ "$1" = traffic ]; then
  python tools/pmc_traffic_all.py $O > $O/traffic_summary.txt
  for d in $O/tcc_*; do [ -d $d ] && python tools/pmc_reduce.py $d $d.csv > /dev/null; done
fi
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $O
