#!/bin/bash
# r6 experiment 12: (a) conv_x3s early half-tile store (ab/early.so = -DX3S_EARLY_STORE=1) against the in-tree build: op tests, then layers;
# (b) prepack with grouped waits (16 jobs per wait) with / without on two workloads
O=gpurun_out/r6_exp12.log
: > $O
MI355SEG_LIB_PATH=$PWD/ab/early.so python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3 >> $O
echo "== A/B: 'old' = ab/early.so (early store), 'new' = in-tree (x < 1: early store is faster)" >> $O
python tools/_ab.py $PWD/ab/early.so --math f16x3 --what fwd,dgrad -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 128 128 128 32 64 3" "2 64 64 64 64 64 3" "2 64 64 64 128 64 3" "2 32 32 32 128 128 3" >> $O 2>&1
for rep in 1 2; do
for np in 0 1; do
  echo "== MI355SEG_NO_PREPACK=$np" >> $O
  export MI355SEG_NO_PREPACK=$np; [ $np = 0 ] && unset MI355SEG_NO_PREPACK
  python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
done
done
echo "== whole step with ab/early.so (prepack off)" >> $O
MI355SEG_NO_PREPACK=1 MI355SEG_LIB_PATH=$PWD/ab/early.so python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
MI355SEG_NO_PREPACK=1 python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
MI355SEG_NO_PREPACK=1 MI355SEG_LIB_PATH=$PWD/ab/early.so python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
MI355SEG_NO_PREPACK=1 python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
cat $O
