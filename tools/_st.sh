set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/st; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vnet -- python3 $R/tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 5 --no-prof > $O/vnet.log 2>&1
cp $(find $O/vnet -name "*kernel_stats.csv" | head -1) $O/vnet_stats.csv; rm -rf $O/vnet
cd $R; head -45 $O/vnet_stats.csv | cut -c1-150
