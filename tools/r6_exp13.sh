#!/bin/bash
# r6 experiment 13: the phases of a strided input gradient in one launch (bf16): tests, layers (in-tree vs ab/base.so = HEAD before), legs
O=gpurun_out/r6_exp13.log
: > $O
python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -4 >> $O || { cat $O; exit 1; }
python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3 >> $O
for shp in "1 160 192 160 32 64" "1 80 96 80 64 128" "1 40 48 40 128 256" "1 20 24 20 256 512"; do
  for lib in "$PWD/ab/base.so" ""; do
    echo "-- $shp k3 s2 lib=${lib:-in-tree}" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_layer.py $shp 3 30 2 1 --dtype bf16 2>&1 | grep "^fwd\|^dgrad\|^wgrad" >> $O
  done
done
for rep in 1 2; do
  for lib in "$PWD/ab/base.so" ""; do
    echo "== lib=${lib:-in-tree}" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  done
done
cat $O
