#!/bin/bash
# timing probes of conv_x3s (TUNE build at ab/tune.so): which of staging / weight loads / voxel-fragment reads bounds the loop
# MI355SEG_DBG bits: 1 no re-staging at all, 2 weight fragments of the first unit only, 8 voxel fragments of the first region only,
# 16 halo pieces not requested (stale registers are converted and written), 4 no conversion / LDS writes (barriers kept)
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=${NO_X3W:-1}
for shp in "${SHAPE1:-2 128 128 128 64 32}" "${SHAPE2:-2 64 64 64 128 64}"; do
  for d in ${PROBES:-0 1 2 8 3 11}; do
    echo "== $shp  MI355SEG_DBG=$d"
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 20 --conv-math f16x3 2>&1 | grep "fwd\|dgrad\|rror"
  done
done
