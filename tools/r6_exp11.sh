#!/bin/bash
# r6 experiment 11: weight packings ahead of their use (prepack.hip) -- tests, then the four workloads with and without
O=gpurun_out/r6_exp11.log
: > $O
python -m pytest tests/test_gpu_prepack.py -x -q 2>&1 | tail -15 >> $O || exit 1
python -m pytest tests/test_gpu_unet.py -x -q -k "graph or fixture" 2>&1 | tail -3 >> $O
for rep in 1 2; do
for np in 0 1; do
  echo "== MI355SEG_NO_PREPACK=$np" >> $O
  export MI355SEG_NO_PREPACK=$np; [ $np = 0 ] && unset MI355SEG_NO_PREPACK
  python tools/bench_model.py unet 2 1 128 128 128 --steps 10 --no-prof 2>&1 | grep -i "ms/step\|ms_per_step\|step " | head -3 >> $O
  python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --no-prof 2>&1 | grep -i "ms/step\|ms_per_step\|step " | head -3 >> $O
  python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 --no-prof 2>&1 | grep -i "ms/step\|ms_per_step\|step " | head -3 >> $O
  python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 --no-prof 2>&1 | grep -i "ms/step\|ms_per_step\|step " | head -3 >> $O
  python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --graph 2>&1 | grep -i "ms/step\|ms_per_step\|step " | head -3 >> $O
done
done
cat $O
