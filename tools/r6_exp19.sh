#!/bin/bash
# r6 experiment 19: conv_b16s<3, 4> -- a tile's prologue + epilogue without its K loop (TUNE build, MI355SEG_DBG 256), with / without halo loads (64) and stores (32)
O=gpurun_out/r6_exp19.log
: > $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
for shp in "1 160 192 160 32 32" "1 160 192 160 64 32"; do
  for d in 0 256 288 320 352 0; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --dtype bf16 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
cat $O
