#!/bin/bash
# r6 experiment 1: (a) changed tests, (b) BN-sum loads ahead of the stores A/B, (c) stagger probe
O=gpurun_out/r6_exp1.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q -k "stem_weight_gradient or bnsums or bn_sums or dgrad" 2>&1 | tail -5 >> $O
echo "== A/B BNX_EARLY (old = ab/base.so)" >> $O
python tools/_ab.py $PWD/ab/base.so --what dgbn,dgrad --math f16x3 -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 64 64 64 64 64 3" "2 64 64 64 128 64 3" >> $O 2>&1
echo "== stagger probe (tune.so)" >> $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64"; do
  for d in 0 $((64+(1<<16))) $((64+(2<<16))) $((64+(4<<16))) $((128+(1<<16))) $((128+(2<<16))) $((128+(4<<16))) 0; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad\|^dgbn\|rror" >> $O
  done
done
cat $O
