#!/bin/bash
# r6 experiment 23: conv_x3s -- a tile's fixed part (TUNE build; MI355SEG_DBG 512 no K loop, 1024 no first halo request, 2048 no stores) on the Cout = 32 / 64 layers of the 128^3 level
O=gpurun_out/r6_exp23.log
: > $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 128 128 128 32 64"; do
  for d in 0 512 1536 2560 3584 0; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
awk '/^--/{h=$0; next} /^fwd/{printf "%s  fwd %s", h, $2} /^dgrad/{printf "  dgrad %s\n", $2}' $O
