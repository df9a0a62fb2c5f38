#!/bin/bash
# r6 experiment 14: every weight packing of a step in one launch (prepack.hip, same stream): tests, then the four workloads with / without
O=gpurun_out/r6_exp14.log
: > $O
python -m pytest tests/test_gpu_prepack.py -x -q 2>&1 | tail -15 >> $O || { cat $O; exit 1; }
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_unet.py -x -q 2>&1 | tail -3 >> $O
for rep in 1 2; do
for np in 0 1; do
  echo "== MI355SEG_NO_PREPACK=$np" >> $O
  export MI355SEG_NO_PREPACK=$np; [ $np = 0 ] && unset MI355SEG_NO_PREPACK
  python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
done
done
unset MI355SEG_NO_PREPACK
python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --graph 2>&1 | grep "ms/step" >> $O
cat $O
