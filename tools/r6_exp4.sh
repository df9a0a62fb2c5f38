#!/bin/bash
O=gpurun_out/r6_exp4.log
: > $O
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 128 128 128 32 64" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128" "2 32 32 32 256 128" "2 16 16 16 256 256"; do
  python tools/overlap_probe.py $shp 20 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
