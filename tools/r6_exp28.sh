#!/bin/bash
# r6 experiment 28 (verdict r5 item 6): conv_x3s staging probes on the Cout = 32 layers, loads and staging separately (TUNE build, MI355SEG_DBG):
# 0 plain; 16 = the later chunks' halo loads NOT requested (their split + LDS writes still run, on stale registers); 4 = loads requested, split + LDS writes skipped after the
# first chunk; 1 = neither (the r5 probe)
O=gpurun_out/r6_exp28.log
: > $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32"; do
  for d in 0 16 4 1 0 16 4 1; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
awk '/^--/{h=$0; next} /^fwd/{printf "%s  fwd %s", h, $2} /^dgrad/{printf "  dgrad %s\n", $2}' $O
