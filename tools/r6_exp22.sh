#!/bin/bash
# r6 experiment 22: the generic kernel's store loop in two instantiations (k1 convolutions, strided convolutions, ConvT on the generic tiles, the bf16x6 bottleneck): tests + layers + legs
O=gpurun_out/r6_exp22.log
: > $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_models.py -x -q 2>&1 | tail -2 >> $O
echo "== bf16 k3 s2 / k1 layers: in-tree ('new') vs ab/base.so ('old')" >> $O
for shp in "1 160 192 160 32 64 3 30 2 1" "1 80 96 80 64 128 3 30 2 1" "1 40 48 40 128 256 3 30 2 1" "1 20 24 20 256 512 3 30 2 1" "1 80 96 80 128 64 1 30" "1 40 48 40 256 128 1 30"; do
  for lib in "$PWD/ab/base.so" ""; do
    echo "-- $shp lib=${lib:-in-tree}" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_layer.py $shp --dtype bf16 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
for rep in 1 2; do
  for lib in "$PWD/ab/base.so" ""; do
    echo "== lib=${lib:-in-tree}" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  done
done
cat $O
