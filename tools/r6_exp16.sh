#!/bin/bash
# r6 experiment 16: where a conv_b16s<3, 4> launch's time goes (Res-U-Net full-resolution Cout = 32 layers, MFMA busy 0.35): TUNE-build probes on one box
O=gpurun_out/r6_exp16.log
: > $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
echo "== conv_b16s probes (TUNE build; MI355SEG_DBG: 0 plain, 32 no stores, 1 halo for the first chunk only, 64 no halo loads, 128 no MFMAs, combinations)" >> $O
for shp in "1 160 192 160 32 32" "1 160 192 160 64 32" "1 80 96 80 64 64"; do
  for d in 0 32 1 33 64 96 128 160 224 0; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --dtype bf16 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
cat $O
