set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ctpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="$R/tools/bench_convt.py 2 64 64 64 64 32 10 64"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqa -- python3 $CMD > $O/sqa.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqb -- python3 $CMD > $O/sqb.log 2>&1
cd $R
python tools/pmc_reduce.py $O/sqa $O/sqa.csv > /dev/null; python tools/pmc_reduce.py $O/sqb $O/sqb.csv > /dev/null
rm -rf $O/sqa $O/sqb
grep -i "convt_\|kernel" $O/sqa.csv | cut -c1-400; grep -i "convt_\|kernel" $O/sqb.csv | cut -c1-400
