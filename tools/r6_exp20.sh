#!/bin/bash
# r6 experiment 20: conv_b16s store loop in two instantiations (activation switch out of the training path's instruction stream): probes, then layers vs ab/base.so
O=gpurun_out/r6_exp20.log
: > $O
for shp in "1 160 192 160 32 32"; do
  for d in 0 256 352 0; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_LIB_PATH=$PWD/ab/tune.so MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --dtype bf16 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
SH=("1 160 192 160 32 32 3" "1 160 192 160 64 32 3" "1 80 96 80 64 64 3" "1 80 96 80 128 64 3" "1 40 48 40 128 128 3" "2 128 128 128 32 32 5" "2 64 64 64 64 64 5" "1 96 96 96 32 32 3" "1 48 48 48 128 64 3")
echo "== in-tree ('new') vs ab/base.so ('old')" >> $O
python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
cat $O
