#!/bin/bash
# timing probes of the f16x3 weight-gradient kernel (TUNE build at ab/tune.so); MI355SEG_DBG bits: see LWgradArgs::dbg
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
for shp in "${SHAPE1:-2 128 128 128 64 64}" "${SHAPE2:-2 64 64 64 128 64}"; do
  for d in ${PROBES:-0 8 16 24 64}; do
    echo "== $shp  MI355SEG_DBG=$d"
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 20 --conv-math f16x3 2>&1 | grep "wgrad"
  done
done
