#!/bin/bash
# r6 experiment 5: what would operands handed over ALREADY SPLIT be worth?  TUNE build, MI355SEG_DBG=256: the f16x3 staging of conv_x3s and of
# both weight-gradient kernels is a copy (same loads, same LDS writes, same MFMAs on garbage operands, no conversion VALU)
O=gpurun_out/r6_exp5.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q -k "split_k or stem_weight" 2>&1 | tail -3 >> $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128"; do
  for d in 0 256 0 256; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad\|^dgbn\|^wgrad\|rror" >> $O
  done
done
cat $O
