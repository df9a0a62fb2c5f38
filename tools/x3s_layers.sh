#!/bin/bash
# fwd / dgrad / wgrad of the dominant cfg-2 layers under the default math
O=${1:-gpurun_out/x3s_layers.log}
: > $O
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128" "2 32 32 32 256 128" "2 16 16 16 256 256"; do
  python tools/bench_layer.py $shp 3 20 --conv-math f16x3 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
