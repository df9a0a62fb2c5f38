#!/bin/bash
# bf16 weight gradient per layer: the one-block kernel (default) against the two-block LDS-DMA kernel (MI355SEG_B16_TILES=3)
for shp in "2 128 128 128 32 64 5" "2 64 64 64 64 64 5" "2 32 32 32 128 128 5" "1 160 192 160 32 64 3" "1 80 96 80 64 64 3" "1 40 48 40 128 128 3" "2 128 128 128 64 64 3"; do
  for m in 0 3; do
    echo "== $shp  b16_tiles=$m"
    MI355SEG_B16_TILES=$m python tools/bench_layer.py $shp 20 --dtype bf16 2>&1 | grep "wgrad"
  done
done
