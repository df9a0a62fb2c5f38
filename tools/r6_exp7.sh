#!/bin/bash
O=gpurun_out/r6_exp7.log
: > $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
MI355SEG_DBG=256 python tools/presplit_debug.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== pre-split operands, real bits (TUNE build, MI355SEG_DBG=256) vs the plain kernels" >> $O
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128"; do
  for rep in 1 2; do
    MI355SEG_DBG=0 python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad\|^wgrad\|rror" | sed 's/^/plain    /' >> $O
    MI355SEG_DBG=256 python tools/bench_layer.py $shp 3 30 --conv-math f16x3 --presplit 2>&1 | grep "^fwd\|^dgrad\|^wgrad\|rror\|presplit" | sed 's/^/presplit /' >> $O
  done
done
cat $O
