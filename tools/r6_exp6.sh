#!/bin/bash
O=gpurun_out/r6_exp6.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q -k "split_k" 2>&1 | tail -30 >> $O
python -m pytest tests/test_gpu_ops.py -q 2>&1 | tail -8 >> $O
echo "== pre-split operands, real bits (TUNE build, MI355SEG_DBG=256) vs the plain kernels" >> $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 128 64"; do
  for rep in 1 2; do
    MI355SEG_DBG=0 python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad\|^wgrad\|rror" | sed 's/^/plain    /' >> $O
    MI355SEG_DBG=256 python tools/bench_layer.py $shp 3 30 --conv-math f16x3 --presplit 2>&1 | grep "^fwd\|^dgrad\|^wgrad\|rror\|presplit" | sed 's/^/presplit /' >> $O
  done
done
cat $O
