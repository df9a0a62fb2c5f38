#!/bin/bash
# r6 experiment 29: the streaming norm / activation kernels instantiated per activation (no per-element run-time switch): tests, then the four workloads, in-tree vs ab/base.so (HEAD)
O=gpurun_out/r6_exp29.log
: > $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_models.py tests/test_gpu_unet.py -x -q 2>&1 | tail -2 >> $O
for rep in 1 2; do
  for lib in "$PWD/ab/base.so" ""; do
    echo "== lib=${lib:-in-tree}" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 2>&1 | grep "ms/step\|norm_act" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 2>&1 | grep "ms/step\|norm_act" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 2>&1 | grep "ms/step\|norm_act" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py unet 2 1 128 128 128 --steps 10 2>&1 | grep "ms/step\|norm_act" >> $O
  done
done
cat $O
