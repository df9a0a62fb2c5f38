#!/bin/bash
# the dominant cfg-2 layers under the two split maths (tools/bench_layer.py; fwd / dgrad / wgrad per layer)
O=${1:-gpurun_out/f16x3_layers.log}
: > $O
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128" "2 16 16 16 256 256"; do
  for m in bf16x6 f16x3; do
    python tools/bench_layer.py $shp 3 20 --conv-math $m >> $O 2>&1
  done
done
cat $O
